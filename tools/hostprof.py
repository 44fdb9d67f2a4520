import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gtsam-vslam_amd"))
import numpy as np, torch, synth, vslam_capi as vc
rig = synth.RIGS["euroc"]; w,h = rig["w"], rig["h"]
dev = torch.device("cuda",0)
frames=[]; poses=[]
for i in range(6):
    L,R,T = synth.stereo_frame(i); frames.append((torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev))); poses.append((T, synth.pose_at(i-0.3)))
ba = synth.make_ba_problem("euroc", 10, 4, 3000)
fe = vc.Extractor(w,h,1500,batch=2); fm = vc.Matcher(rig, fe,0,fe,1)
acc = {}
def tm(name, f):
    t=time.perf_counter(); r=f(); acc[name]=acc.get(name,0)+time.perf_counter()-t; return r
N=60
for n in range(N+6):
    if n==6: acc={}
    i = n % 6
    dL,dR = frames[i]
    tm("set_image", lambda: (fe.set_image_device(0,dL.data_ptr(),w), fe.set_image_device(1,dR.data_ptr(),w)))
    tm("extract", fe.run)
    tm("stereo", fm.stereo_match)
    if i>0: tm("track", lambda: vc.tracker_track(fm, poses[i][1], 5))
    tm("init_map", lambda: vc.tracker_init_map(fm, poses[i][0]))
    if n%5==4: tm("ba", lambda: vc.local_ba(rig, fe.sigmaFactor, fe.InvSigmaFactor, ba))
for k,v in acc.items(): print("%-10s %.3f ms/frame" % (k, 1e3*v/N))
print("ba per call %.3f ms" % (1e3*acc["ba"]/(N/5)))
