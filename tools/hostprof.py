"""Host wall-clock per C-ABI call of the C2 bench loop (tracking thread), with and without the local-BA thread."""
import sys, os, time, threading, queue
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "gtsam-vslam_amd")); sys.path.insert(0, root)
import numpy as np, torch, synth, vslam_capi as vc, bench
rig = synth.RIGS["euroc"]; w, h = rig["w"], rig["h"]
dev = torch.device("cuda", 0)
NF = 8
frames, poses, imus = [], [], []
for f in range(NF):
    L, R, T = synth.stereo_frame(f, "euroc")
    frames.append((torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)))
    poses.append((T, synth.pose_at(f - 0.3, rig["fps"])))
    S, dts, _ = synth.imu_samples(f - 1, f, rig["fps"], noise_seed=0x1A00 + f)
    hh = 1e-4
    v_prev = (synth.pose_at(f - 1 + hh * rig["fps"], rig["fps"])[:3, 3] - synth.pose_at(f - 1 - hh * rig["fps"], rig["fps"])[:3, 3]) / (2 * hh)
    imus.append((S, dts, np.arange(len(dts)) * 5e6, v_prev))
ba = synth.make_ba_problem("euroc", 10, 4, 3000)
fe = vc.Extractor(w, h, 1500, batch=2); fm = vc.Matcher(rig, fe, 0, fe, 1)
for with_ba in (False, True):
    acc = {}
    def tm(name, f):
        t = time.perf_counter(); r = f(); acc[name] = acc.get(name, 0) + time.perf_counter() - t; return r
    q = queue.Queue(maxsize=1)
    def worker():
        while True:
            j = q.get()
            if j is None: q.task_done(); return
            vc.local_ba(rig, fe.sigmaFactor, fe.InvSigmaFactor, ba); q.task_done()
    th = threading.Thread(target=worker, daemon=True); th.start()
    N = 200
    for n in range(N + 16):
        if n == 16: acc = {}; t0 = time.perf_counter()
        i = n % NF
        dL, dR = frames[i]
        tm("set_image", lambda: (fe.set_image_device(0, dL.data_ptr(), w), fe.set_image_device(1, dR.data_ptr(), w)))
        tm("extract", fe.run)
        tm("stereo", fm.stereo_match)
        if i > 0:
            S, dts, ts, v_prev = imus[i]
            tm("track_imu", lambda: vc.tracker_track_imu(fm, poses[i][1], 5, bench.GRAVITY, bench.IMU_NOISE, synth.T_BC1, poses[i - 1][0], v_prev, np.zeros(6), S[:, :3], S[:, 3:], ts, 200))
        tm("init_map", lambda: vc.tracker_init_map(fm, poses[i][0]))
        if with_ba and n % 5 == 4: tm("ba_put", lambda: q.put(1))
    q.join(); tot = time.perf_counter() - t0
    q.put(None)
    print("with_ba=%s: %.3f ms/frame" % (with_ba, 1e3 * tot / N))
    for k, v in acc.items(): print("   %-10s %.3f ms/frame" % (k, 1e3 * v / N))
    fe.timings(); tt = fm.timings()
