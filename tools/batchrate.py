"""Throughput of the batched fleet for a few (sessions, lanes) shapes + device time per stage and host phases of a step.
usage: python tools/batchrate.py [config] [frames] [frame_step] [mapping] [shapes e.g. 16x16,32x16,32x32]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gtsam-vslam_amd"))
import numpy as np
import bench
cfgname = sys.argv[1] if len(sys.argv) > 1 else "c2"
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fstep = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mapping = int(sys.argv[4]) if len(sys.argv) > 4 else 2
shapes = sys.argv[5] if len(sys.argv) > 5 else "1x0,1x1,8x8,16x16,32x32,32x16,64x32"
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 100
cfg = bench.CONFIGS[cfgname]
rig, frames, poses, vel, fwd, bwd = bench.make_sequence(cfg, nfr, fstep, 0)
import torch, synth
import vslam_capi as vc
dev = torch.device("cuda", 0)
bufs = [(torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)) for (L, R, _) in frames]
lp = [b[0].data_ptr() for b in bufs]; rp = [b[1].data_ptr() for b in bufs]
imu = dict(gravity=bench.GRAVITY, noise=bench.IMU_NOISE, T_bs=synth.T_BC1, hz=200) if cfg["imu"] else None
scfg = vc.system_config(rig, cfg["nfeat"], imu=imu, local_mapping=mapping, device=0)
for sh in shapes.split(","):
    S, lanes = (int(v) for v in sh.split("x"))
    fl = vc.Fleet(scfg, S, lp, rp, rig["w"], True, poses=poses, velocities=vel, imu_forward=fwd if cfg["imu"] else None,
                  imu_backward=bwd if cfg["imu"] else None, lanes=lanes)
    fl.run(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = fl.run(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    fl.set_sampling(4)
    r2 = fl.run(24)
    tm, cnt = fl.timings()
    fl.set_sampling(0)
    print(json.dumps({"sessions": S, "lanes": lanes, "frames_per_s": round(S * steps / el, 1), "ms_per_step": round(1e3 * el / steps, 3),
                      "keyframes": rep["keyframes"], "local_bas": rep["mappings"], "lost": rep["lost_frames"], "min_inliers": rep["min_inliers"],
                      "rms_pos_err": float(np.sqrt(rep["sum_sq_position_error"] / max(rep["frames"], 1))),
                      "stage_ms_per_sampled_step": {k: round(v / max(cnt["frames"] / max(lanes, 1), 1), 4) for k, v in tm.items()}, "sampled": cnt}), flush=True)
    fl.close()
