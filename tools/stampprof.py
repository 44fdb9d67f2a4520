"""Run a few C2 tracking frames against the stamped diagnostic library (tools/stampbuild.sh)."""
import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "gtsam-vslam_amd"))
import numpy as np, torch, synth, vslam_capi as vc
vc.LIB_PATH = os.path.join(root, "tools/_stamp/libvslam_stamp.so")
sys.path.insert(0, root)
import bench
rig = synth.RIGS["euroc"]; w, h = rig["w"], rig["h"]
dev = torch.device("cuda", 0)
fe = vc.Extractor(w, h, 1500, batch=2); fm = vc.Matcher(rig, fe, 0, fe, 1)
for f in range(4):
    L, R, T = synth.stereo_frame(f, "euroc")
    dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    fe.set_image_device(0, dL.data_ptr(), w); fe.set_image_device(1, dR.data_ptr(), w)
    fe.run(); fm.stereo_match()
    if f > 0:
        S, dts, _ = synth.imu_samples(f - 1, f, rig["fps"], noise_seed=0x1A00 + f)
        hh = 1e-4
        v_prev = (synth.pose_at(f - 1 + hh * rig["fps"], rig["fps"])[:3, 3] - synth.pose_at(f - 1 - hh * rig["fps"], rig["fps"])[:3, 3]) / (2 * hh)
        print("frame", f, file=sys.stderr)
        vc.tracker_track_imu(fm, synth.pose_at(f - 0.3, rig["fps"]), 5, bench.GRAVITY, bench.IMU_NOISE, synth.T_BC1, Tprev, v_prev, np.zeros(6),
                             S[:, :3], S[:, 3:], np.arange(len(dts)) * 5e6, 200)
    vc.tracker_init_map(fm, T); Tprev = T
