"""One session through the whole corridor sequence of bench.py (device images): where does tracking degrade?
usage: python tools/corridor_check.py [n_frames] [speed] [use_imu] [first]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gtsam-vslam_amd"), ROOT]
import numpy as np, torch, synth, vslam_capi as vc, bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
speed = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
use_imu = int(sys.argv[3]) if len(sys.argv) > 3 else 1
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cfg = bench.CONFIGS["c2" if use_imu else "c1"]
rig = synth.RIGS["euroc"]
dev = torch.device("cuda", 0)
Ls, Rs, P, idx = synth.corridor_sequence("euroc", n, dev, speed=speed, first=first)
vel, fwd, bwd = bench.motion_data(cfg, idx, 1, lambda f, fp: synth.corridor_pose(f, fp, speed))
imu = dict(gravity=bench.GRAVITY, noise=bench.IMU_NOISE, T_bs=synth.T_BC1, hz=200, velocity=vel[0]) if use_imu else None
S = vc.System(rig, 1500, T0=P[0], imu=imu, local_mapping=2, mapping_delay=4)
for j in range(n):
    T, rep = S.track(Ls[j].data_ptr(), Rs[j].data_ptr(), j, imu_bucket=fwd[j] if (use_imu and j > 0) else None, on_device=True, stride=rig["w"])
    err = float(np.abs(T[:3, 3] - P[j][:3, 3]).max())
    if j % 20 == 0 or rep["n_inliers"] < 80 and j > 0 or err > 0.1:
        yaw = 0.70 * np.sin(0.5 * idx[j] / rig["fps"])
        print(j, "kf" if rep["keyframe_inserted"] else "  ", rep["n_inliers"], rep["n_stereo"], rep["n_active"], rep["rounds"], "err %.3f" % err, "yaw %.2f z %.2f" % (yaw, P[j][2, 3]), flush=True)
    if err > 5:
        break
