#!/usr/bin/env python3
"""Benchmark of the tracking + local-BA hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched through
torch.distributed.run, one rank per GPU.  W untimed warm-up steps, then exactly K timed steps bracketed by barrier +
device synchronize; MAX over ranks; rank 0 prints ONE JSON line.

A "step" = one stereo frame of every session: S independent SLAM sessions (sequences) share a GPU, each driven by its own
host thread INSIDE the library (vslam_fleet; Python only starts the run and waits).  Every session runs the CLOSED LOOP
(vslam_system): extraction L+R, stereo match, tracking against ITS OWN map (removeOutOfFrameMPs, {projection match, pose
solve} rounds, refinement), the keyframe rule, insertKeyFrame, and - on the optimizer thread - covisibility window,
findNewPoints, local BA on the tracker's window, write-back; nothing is re-seeded from ground truth.  The frames of a short
rendered sequence are resident in HBM before the timed region (or, with --host-images, in pinned host memory: then every
frame's host-to-device copy is inside its step) and replayed as a ping-pong, i.e. a continuous camera motion.

  --config c1|c2|c3   C1 EuRoC stereo, C2 EuRoC stereo + IMU (the headline, default), C3 KITTI-like 1241x376 / 2000 features
  --config c5         the 64-keyframe / 100 000-landmark global BA, landmarks sharded over the N ranks (RCCL all-reduce of the
                      reduced camera system per trial round); a step = one full BA (two LM passes + chi2 re-check)
PyTorch is used for device buffers and torch.distributed only.  Multi-GPU, c1-c3: replicas (no data-path collective, weak
scaling); with N > 1 the line also carries the C5 sharded BA over the same ranks ("c5_sharded_ba"), the one collective of
this path.
"""
import argparse
import json
import os
import sys
import time

# Hardware queues: the runtime's default is 4, and every stream of the process is mapped onto them - the small launches of
# the mapping threads then sit behind a lockstep group's wide kernels in the same queue.  8 queues: +4-6 % frames/s and
# shorter mapping passes (7.8 -> 7.4 ms) at 128 sessions; 16: no further gain; 32: slower launches (22-38 us per launch at
# 8 launching threads, tools/launchrate.hip).  Stated here so that the environment cannot silently change it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gtsam-vslam_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6  # fp64 vector / matrix peak (MI355X_MICROARCH.md)
GRAVITY = (0.0, 9.81, 0.0)
IMU_NOISE = (1.6968e-04, 1.9393e-05, 2.0e-3, 3.0e-3)   # gyro density / walk, accel density / walk (config_V1_02.yaml)
CONFIGS = {
    "c1": dict(rig="euroc", nfeat=1500, imu=False, name="C1-class: EuRoC-like stereo (slamMode 1) 752x480, 1500 features/image"),
    "c2": dict(rig="euroc", nfeat=1500, imu=True, name="C2-class: EuRoC-like stereo+IMU (slamMode 0) 752x480, 1500 features/image, "
                                                       "200 Hz IMU pre-integration factor in every pose solve"),
    "c3": dict(rig="kitti", nfeat=2000, imu=False, name="C3-class: KITTI-00-like stereo 1241x376, 2000 features/image"),
}
PMC_FILE = os.path.join(ROOT, "profiles", "r02_i_pmc_summary.json")


def level_pixels(w, h, nlevels=8, scale=1.2):
    """level pixel counts from the reference constructor formulas (float arithmetic)."""
    px, s = [], np.float32(1.0)
    for _ in range(nlevels):
        inv = np.float32(1.0) / s
        px.append(int(np.rint(np.float32(w) * inv)) * int(np.rint(np.float32(h) * inv)))
        s = np.float32(s * np.float32(scale))
    return px


def _render_one(a):
    import synth
    return synth.stereo_frame(a[0], a[1])


def render_frames(idx, rig_name):
    """the rendered stereo pairs (about two seconds per pair on one core; numpy releases the GIL): spread over threads.
    Threads, not processes: under a profiler the GPU is initialised before main() and a fork would inherit it."""
    from concurrent.futures import ThreadPoolExecutor
    n = min(len(idx), max(1, (os.cpu_count() or 2) // max(int(os.environ.get("WORLD_SIZE", "1")), 1)), 16)
    if n <= 1:
        return [_render_one((f, rig_name)) for f in idx]
    with ThreadPoolExecutor(n) as pool:
        return list(pool.map(_render_one, [(f, rig_name) for f in idx]))


def make_sequence(cfg, n_frames, step, rank):
    """n_frames rendered stereo pairs `step` source frames apart, ground truth, velocities, IMU buckets of both directions."""
    import synth
    rig = synth.RIGS[cfg["rig"]]
    fps = rig["fps"]
    f0 = 7 * rank
    idx = [f0 + step * j for j in range(n_frames)]
    frames = render_frames(idx, cfg["rig"])
    poses = np.stack([fr[2] for fr in frames])
    h = 1e-4
    vel = np.stack([(synth.pose_at(f + h * fps, fps)[:3, 3] - synth.pose_at(f - h * fps, fps)[:3, 3]) / (2 * h) for f in idx])
    fwd, bwd = [None] * n_frames, [None] * n_frames
    if cfg["imu"]:
        for j in range(1, n_frames):
            S, dts, _ = synth.imu_samples(idx[j - 1], idx[j], fps, noise_seed=0x1A00 + idx[j])
            fwd[j] = (S[:, :3].copy(), S[:, 3:].copy(), np.arange(len(dts)) * 5e6)
        for j in range(n_frames - 1):
            a = idx[j + 1]
            S, dts, _ = synth.imu_samples(a, a + step, fps, noise_seed=0x2B00 + idx[j], pose_fn=lambda f, fp, a=a: synth.pose_at(2 * a - f, fp))
            bwd[j] = (S[:, :3].copy(), S[:, 3:].copy(), np.arange(len(dts)) * 5e6)
    return rig, frames, poses, vel, fwd, bwd


def cpu_baseline(cfg, frames, poses, vel, fwd, budget_s=20.0, warm=3):
    """The oracle's closed loop (oracle/vo_system.py on the stage functions of liboracle, built -O3 -march=native on this
    host) on a bounded sample of the same frames: frames/s, median and p95 of the per-frame time, one core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    po.use_native_build()
    import vo_system
    import synth
    rig = synth.RIGS[cfg["rig"]]
    imu = None
    if cfg["imu"]:
        imu = dict(prm=po.imu_params(GRAVITY, IMU_NOISE[0], IMU_NOISE[2], IMU_NOISE[1], IMU_NOISE[3], synth.T_BC1))
    S = vo_system.System(rig, cfg["nfeat"], T0=poses[0], imu=imu)
    if cfg["imu"]:
        S.velocity = vel[0].copy()
    per, t_all = [], time.perf_counter()
    for n in range(len(frames)):
        L, R, _ = frames[n]
        b = None
        if cfg["imu"] and n > 0:
            b = (np.concatenate([fwd[n][0], fwd[n][1]], 1), np.full(len(fwd[n][2]), 1.0 / 200))
        t0 = time.perf_counter()
        S.track(L, R, n, imu_bucket=b)
        if n >= warm:
            per.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s and len(per) >= 5:
            break
    per = np.array(per)
    return {"value": float(len(per) / per.sum()), "unit": "frames/s", "cores": 1, "kind": "port",
            "median_ms": float(np.median(per) * 1e3), "p95_ms": float(np.percentile(per, 95) * 1e3),
            "sample": "%d frames of the same sequence after %d warm-up frames through the oracle's closed loop (extract L+R, stereo, "
                      "tracking, keyframe insertion, new points + local BA on the tracker's windows; %d keyframes, %d local BAs), "
                      "liboracle built -O3 -march=native -ffp-contract=off on this host, single thread; the reference itself cannot "
                      "be built here (OpenCV / GTSAM absent)" % (len(per), warm, len(S.keyFrames), sum(1 for l in S.log if "mapping" in l))}


def run_c5(args, rank, world, local, dist, torch, backend):
    """C5: 64-keyframe / 100 000-landmark BA, landmarks sharded lm % world over the ranks, RCCL all-reduce of the NB reduced
    camera systems per trial round.  A step = one vslam_local_ba call."""
    import synth
    import vslam_capi as vc
    rig = synth.RIGS["synthetic"]
    prob = synth.make_ba_problem_c5(n_lm=args.c5_landmarks)
    fe = vc.Extractor(752, 480, 1500, device=local)
    sig, isig = fe.sigmaFactor, fe.InvSigmaFactor
    comm = None
    if world > 1:
        def bcast(b):
            obj = [b if rank == 0 else None]
            dist.broadcast_object_list(obj, src=0)
            return obj[0]
        comm = vc.comm_create_rccl(rank, world, local, bcast)
    steps, warm = max(1, args.c5_steps), max(1, args.c5_warmup)
    for _ in range(warm):
        r = vc.local_ba(rig, sig, isig, prob, device=local, comm=comm)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    vc.local_ba_set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        r = vc.local_ba(rig, sig, isig, prob, device=local, comm=comm)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=torch.device("cuda", local) if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    tm = vc.local_ba_timings()
    trials = r["reports"][0]["inner"] + r["reports"][1]["inner"]
    iters = r["reports"][0]["iterations"] + r["reports"][1]["iterations"]
    n6 = 6 * r["free_kf"]
    out = {"workload": "C5: 64-keyframe (62 free + 2 fixed) / %d-landmark global BA, %d (keyframe, landmark) pairs, %d residual blocks, "
                       "landmarks sharded lm %% %d over %d rank(s)" % (args.c5_landmarks, len(prob["pair_kf"]), r["residuals"], world, world),
           "ranks": world, "ms_per_ba": 1e3 * el / steps, "lm_trials_per_ba": trials, "lm_iterations_per_ba": iters,
           "lm_trials_per_s": trials * steps / el,
           "allreduce_bytes_per_trial_round": (n6 * n6 + n6) * 8 * 4 + 8 * 8 if world > 1 else 0,
           "allreduce_note": "fp64 sum over xGMI of the 4 look-ahead candidates' [S | rhs] (contiguous, one call) + the cost scalars",
           "device_ms_last_ba": {k: round(v, 3) for k, v in tm.items()},
           "final_error": r["reports"][1]["finalError"], "transport": "rccl" if world > 1 else "none (single GPU)"}
    if comm is not None:
        comm.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--sessions", type=int, default=192, help="independent SLAM sessions (sequences) sharing each GPU")
    ap.add_argument("--lanes", type=int, default=96, help="sessions per lockstep group (vslam_batch: one launch per stage for all lanes of a group); "
                                                          "0 = one host thread and one set of launches per session")
    ap.add_argument("--frames", type=int, default=0, help="distinct rendered stereo frames of the replayed sequence (0: 100 for c1 / c2, 60 for c3)")
    ap.add_argument("--frame-step", type=int, default=0, help="source frames between two sequence frames (0: 2 for c1 / c2, 1 for c3: the "
                                                              "synthetic scene's extent in units of the rig's baseline bounds the camera speed)")
    ap.add_argument("--host-images", action="store_true", help="frames in pinned host memory: every frame pays its H2D copy inside the step")
    ap.add_argument("--mapping", type=int, default=2, help="local mapping: 2 = optimizer thread per session (reference), 1 = synchronous, 0 = off")
    ap.add_argument("--mapping-delay", type=int, default=-1, help="(default: 4 frames, 2 for c3) --mapping 2: the fixed schedule of the optimizer "
                    "thread's hand-over (vslam_system_config::mapping_delay): new points arrive with the frame after the keyframe, the "
                    "local BA's write-back + changePosesLCA k frames after it.  Keyframes are at least 5 frames apart, so with k <= 5 every "
                    "keyframe gets its pass - the reference's steady state at camera rate")
    ap.add_argument("--prime", type=int, default=60, help="untimed frames every session tracks BEFORE the warm-up steps, so that the timed "
                    "steps see sessions in their steady state (a map with more than three keyframes, the local mapper running) "
                    "whatever --warmup / --steps are; part of the set-up like rendering the frames")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-line", action="store_true", help="skip the extra single-session run")
    ap.add_argument("--sweep", default="", help="comma-separated SESSIONSxLANES shapes (e.g. 1x0,1x1,8x8,32x32,64x32): throughput of each in the "
                                                "\"sweep\" field (one process, same frames)")
    ap.add_argument("--c5-landmarks", type=int, default=100000)
    ap.add_argument("--c5-steps", type=int, default=5)
    ap.add_argument("--c5-warmup", type=int, default=1)
    ap.add_argument("--c5-timeout", type=int, default=180, help="N > 1: seconds the sharded-BA section may take before the line is printed without it")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.frames <= 0:
        args.frames = 60 if args.config == "c3" else 100
    if args.frame_step <= 0:
        args.frame_step = 1 if args.config == "c3" else 2
    seq = None
    if args.config != "c5":      # rendered before torch / HIP start (the renderer forks worker processes)
        seq = make_sequence(CONFIGS[args.config], args.frames, args.frame_step, rank)

    import torch
    import torch.distributed as dist
    import vslam_capi as vc

    # VSLAM_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow with several ranks on ONE GPU (RCCL refuses two
    # ranks on the same device); the driver's runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("VSLAM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.config == "c5":
        c5 = run_c5(args, rank, world, local, dist, torch, backend)
        if rank == 0:
            print(json.dumps({"metric": "LM trial rounds/s of the 64-KF / 100k-landmark global BA (landmark-sharded, RCCL all-reduce)",
                              "value": c5["lm_trials_per_s"], "unit": "trials/s", "n_gpus": world, "steps": args.c5_steps, "warmup": args.c5_warmup,
                              "ms_per_step": c5["ms_per_ba"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                              "dtype": "f64", "data": "synthetic", "config": {"workload": c5["workload"], "parallelism": "landmark shards x%d" % world},
                              "c5": c5}))
        if world > 1:
            dist.destroy_process_group()
        return

    cfg = CONFIGS[args.config]
    rig, frames, poses, vel, fwd, bwd = seq
    w, h = rig["w"], rig["h"]
    if args.host_images:
        bufs = [(torch.from_numpy(L).pin_memory(), torch.from_numpy(R).pin_memory()) for (L, R, _) in frames]
    else:
        bufs = [(torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)) for (L, R, _) in frames]
    lp = [b[0].data_ptr() for b in bufs]
    rp = [b[1].data_ptr() for b in bufs]
    imu = dict(gravity=GRAVITY, noise=IMU_NOISE, T_bs=__import__("synth").T_BC1, hz=200) if cfg["imu"] else None
    if args.mapping_delay < 0:
        # The reference's optimizer thread ends a pass within a frame or two of camera time: k = 2 for the KITTI-like sequence
        # (10 fps, ~10x the motion per frame), k = 4 for the EuRoC-like ones (20 fps).  The mode is parity-tested frame by frame
        # against the oracle's restatement of the same schedule (tests/test_gpu_system.py).
        args.mapping_delay = 2 if args.config == "c3" else 4
    scfg = vc.system_config(rig, cfg["nfeat"], imu=imu, local_mapping=args.mapping, device=local, mapping_delay=args.mapping_delay)

    def make_fleet(S, lanes):
        return vc.Fleet(scfg, S, lp, rp, w, not args.host_images, poses=poses, velocities=vel,
                        imu_forward=fwd if cfg["imu"] else None, imu_backward=bwd if cfg["imu"] else None, lanes=lanes)

    lanes = min(args.lanes, args.sessions)
    fleet = make_fleet(args.sessions, lanes)
    if args.prime > 0:
        fleet.run(args.prime)        # set-up: the sessions' first frames (map initialisation, the first keyframes)
    # per-stage HIP events on every 3rd step of group / session 0 (every step of a short timed region).  The warm-up steps are
    # sampled too, so that the first timed sample does not carry the timers' one-time set-up; their readings are discarded.
    sample_every = 1 if args.steps < 60 else 3
    fleet.set_sampling(sample_every)
    fleet.run(args.warmup)
    fleet.timings()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = fleet.run(args.steps)      # every frame and every local BA of the timed steps completes inside
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    stage_ms, cnt = fleet.timings()
    fleet.set_sampling(0)

    latency = None
    if rank == 0 and not args.no_latency_line and args.sessions > 1:
        fleet.close()
        f1 = make_fleet(1, 0)
        f1.run(max(args.prime, 1) + min(args.warmup, 20))
        torch.cuda.synchronize()
        n1 = min(args.steps, 200)
        t1 = time.perf_counter()
        r1 = f1.run(n1)
        torch.cuda.synchronize()
        e1 = time.perf_counter() - t1
        latency = {"sessions": 1, "frames_per_s": n1 / e1, "ms_per_frame": 1e3 * e1 / n1, "keyframes": r1["keyframes"], "local_bas": r1["mappings"],
                   "note": "one session alone on the GPU: the per-frame latency of the closed loop (same workload, same threads)"}
        f1.close()

    sweep = []
    if rank == 0 and args.sweep:
        try:
            fleet.close()
        except Exception:      # noqa: BLE001
            pass
        for shape in [v for v in args.sweep.split(",") if v]:
            Sx, Lx = (int(v) for v in shape.split("x"))
            fx = make_fleet(Sx, Lx)
            fx.run(max(args.prime, 5))
            torch.cuda.synchronize()
            nx = max(20, min(args.steps, 8000 // Sx))
            tx = time.perf_counter()
            rx = fx.run(nx)
            torch.cuda.synchronize()
            ex = time.perf_counter() - tx
            sweep.append({"sessions": Sx, "lanes_per_group": Lx, "frames_per_s": Sx * nx / ex, "ms_per_step": 1e3 * ex / nx, "keyframes": rx["keyframes"],
                          "local_bas": rx["mappings"], "lost_frames": rx["lost_frames"]})
            fx.close()

    fleet.close()      # (idempotent) library threads joined before the process winds down
    # N > 1: the landmark-sharded 64-keyframe / 100 k-landmark BA over RCCL as an extra section of the line.  It is the
    # only collective of the path and cannot be rehearsed on the one-GPU development box, so it runs under a watchdog: if
    # it has not finished in time (or raises), the closed-loop line is still printed, with the failure recorded.
    c5 = None
    c5_done = [False]
    emit_box = {}
    if world > 1 and not os.environ.get("VSLAM_BENCH_SKIP_C5"):
        import threading

        def _watchdog():
            if c5_done[0]:
                return
            if rank == 0 and "emit" in emit_box:
                emit_box["emit"]({"error": "the sharded-BA section did not finish within %d s" % args.c5_timeout})
            os._exit(0)
        wd = threading.Timer(args.c5_timeout, _watchdog)
        wd.daemon = True
        emit_box["timer"] = wd

    if rank == 0:
        S = args.sessions
        px = level_pixels(w, h)
        sumP, nimg, nfeat = sum(px), 2, cfg["nfeat"]
        nBA = max(cnt["ba"], 1)
        nBAall = max(rep["mappings"], 1)
        R_, L_, k2, Fk = rep["ba_residuals"] / nBAall, rep["ba_landmarks"] / nBAall, rep["ba_sum_k2"] / nBAall, rep["ba_free_kf"] / nBAall
        trialsPerBA = rep["ba_trials"] / nBAall
        linPerBA = rep["ba_iterations"] / nBAall + 2
        nBA6 = 6 * Fk
        nk = float(nfeat)
        Mact = 600.0    # active map points per tracked frame (order of magnitude; the byte formulas are linear in it)
        frames_per_ba = rep["frames"] / nBAall
        # One launch of a tracking / extraction stage serves `lb` sessions (the lanes of a lockstep group; 1 without batching).
        lb = max(lanes, 1)
        nStep = max(cnt["frames"] / lb, 1)         # sampled steps
        # kernel group -> (launches over the sampled region, algorithmic bytes per launch, algorithmic flops per launch)
        b_cand, b_res = Mact * 60 + 2 * nk * 60 + Mact * 128, Mact * (128 + 8 + 8) + 2 * nk * 8
        b_pose = Mact * (24 + 8 + 4) + 2 * nk * 28 + (8 * 514 if cfg["imu"] else 0)
        groups = {
            "pyramid": (7 * nStep, lb * nimg * (sum(px[:-1]) + sum(px[1:])) / 7.0, 0),     # read level l-1, write level l
            "fast": (nStep, lb * nimg * (sumP + 4 * 3.3 * nfeat), 0),                      # every level read once + packed candidates
            "gather": (nStep, lb * nimg * (8 * 3.3 * nfeat), 0),
            "ssc": (nStep, lb * nimg * (8 * 3.3 * nfeat + 4 * nk), 0),                      # candidates in, picks out
            "blur": (nStep, lb * nimg * (2 * sumP), 0),                                    # read + write every level
            "orient_desc": (nStep, lb * nimg * nk * (28 + 32 + 709 + 512), 0),             # keypoint + descriptor + disc + BRIEF taps
            "stereo_rows": (nStep, lb * nk * 28, 0),
            "stereo_match": (nStep, lb * (nk * (28 + 32) * 2 + nk * 16), 0),
            "stereo_finalize": (nStep, lb * nk * 24, 0),
            "track_predict": (nStep, lb * (nk * (24 + 32 + 5) + Mact * (60 + 24 + 12)), 0),
            "track_repredict": (nStep, lb * Mact * (24 + 60 + 12), 0),
            "pack": (nStep, lb * (Mact * 30 + nk * 8), 0),
            "imu_preintegrate": (2 * nStep, lb * (10 * 56 + 8 * (289 + 225)), lb * (10 * 2 * 2 * 15 ** 3 + 15 ** 3)),
            "proj_cells": (nStep, lb * 2 * nk * (28 + 2 + 2), 0),                             # keypoints in, bucketed indices out
            "proj_candidates": (2 * nStep, lb * b_cand, 0),                                # two passes per frame
            "proj_resolve": (2 * nStep, lb * b_res, 0),
            "pose_imu_lm": (2 * nStep, lb * b_pose, 0),                                    # two solves per frame
            "pose_lm": (2 * nStep, lb * b_pose, 0),
            "ba_linearize": (nBA * linPerBA, R_ * (8 + 16 + 8 + 96 + 24 + 160), 0),          # idx, uv, sigma, pose, point, stored J
            "ba_schur": (nBA * trialsPerBA, R_ * 160 + L_ * 24, 2 * 36 * k2),                 # stored J read once (S stays in LDS)
            "ba_solve": (nBA * trialsPerBA, 8 * (nBA6 * nBA6 + nBA6) * 2, nBA6 ** 3 / 3.0),   # reduced system in, delta out
            "ba_back": (nBA * trialsPerBA, R_ * 160 + L_ * 48, 0),
            "ba_eval": (nBA * trialsPerBA, R_ * (160 + 16 + 8 + 96 + 24), 0),
            "ba_chi2": (2 * nBA, R_ * (16 + 8 + 96 + 24), 0),
        }
        nS = nStep * lb
        per_frame = {}
        for k, v in stage_ms.items():
            per_frame[k] = v / (nBA * frames_per_ba) if k.startswith("ba_") else v / nS      # device ms per tracked frame
        out = {
            "metric": "frames/sec (extract+match+localBA), %d feat stereo %dx%d" % (nfeat, w, h),
            "value": world * S * args.steps / el, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i32 (extract, match) + f64 (pose LM, local BA)", "data": "synthetic",
            "config": {"workload": cfg["name"] + "; closed loop per session (keyframe rule, insertKeyFrame, new points + local BA on the tracker's "
                                   "own covisibility window on the optimizer thread, write-back, changePosesLCA); %d rendered frames %s, replayed as a "
                                   "ping-pong; measured per run: one keyframe every %.1f frames, one local BA every %.1f frames (%.0f landmarks, %.0f "
                                   "residual blocks, %.1f free keyframes on average)"
                                   % (args.frames, "in pinned host memory (H2D inside the step)" if args.host_images else "resident in HBM",
                                      rep["frames"] / max(rep["keyframes"], 1), frames_per_ba, L_, R_, Fk),
                       "sessions_per_gpu": S, "lanes_per_group": lanes, "prime_frames": args.prime,
                       "local_mapping": {0: "off", 1: "inside the frame (synchronous)", 2: "device work beside tracking, fixed schedule"}.get(args.mapping, "?") +
                                        ("; mapping_delay = %d: new points with the next frame, the BA's write-back + changePosesLCA %d frames after "
                                         "the hand-over (the mode of tests/test_gpu_system.py::test_closed_loop_parity_async_*)" % (args.mapping_delay, args.mapping_delay)
                                         if args.mapping == 2 else ""),
                       "step": "one stereo frame of each of the %d sessions" % S,
                       "threads": ("%d lockstep groups of %d sessions (vslam_batch: one launch per stage for all lanes), one driver thread + a "
                                   "host-phase pool + 3 mapping threads per group" % ((S + lanes - 1) // lanes, lanes)) if lanes > 0 else
                                  "one host thread per session inside the library (vslam_fleet) + one optimizer thread per session",
                       "parallelism": "replicas x%d, %d sessions per GPU" % (world, S)},
            "tracking": {"mean_inliers": rep["sum_inliers"] / max(rep["frames"] - 0, 1), "min_inliers": rep["min_inliers"],
                         "lost_frames": rep["lost_frames"], "mean_rounds": rep["sum_rounds"] / max(rep["frames"], 1),
                         "keyframes": rep["keyframes"], "local_bas": rep["mappings"], "new_points": rep["new_points"],
                         "rms_position_error_m": float(np.sqrt(rep["sum_sq_position_error"] / max(rep["frames"], 1))),
                         "max_position_error_m": rep["max_position_error"]},
            "stage_ms_per_frame": {k: v for k, v in sorted(per_frame.items())},
            "stage_sampling": "HIP events on every %s step of group 0 (all its lanes per launch) and on the local BAs of its first sessions that "
                              "complete in those steps (%d frames, %d BAs); per tracked frame; BA stages amortised over %.1f frames per BA"
                              % ("3rd" if sample_every == 3 else "single", cnt["frames"], cnt["ba"], frames_per_ba),
        }
        if per_frame:
            # Which kernel dominates a GPU that runs several streams at once?  Elapsed time alone over-counts narrow launches: a
            # one-wave Cholesky (k_ba_solve_mfma64) that waits 150 us for a free CU occupies 4 of the GPU's 8192 wave slots.  A
            # group's weight is therefore its device time per tracked frame x the share of the wave slots one launch can fill
            # (SQ_WAVES per launch from the committed PMC pass, same launch shape, against 256 CUs x 32 waves; x the kernel's
            # resident-wave limit where registers or LDS cap it).  `roofline` prices that
            # kernel; `roofline_top5` lists the five largest groups by the same weight.
            bsfx = "_b" if lanes > 0 else ""
            KNAME = {"proj_resolve": "k_proj_resolve" + bsfx, "pose_imu_lm": "k_pose_imu_lm" + bsfx, "pose_lm": "k_pose_lm" + bsfx,
                     "stereo_match": "k_stereo_match" + bsfx, "stereo_finalize": "k_stereo_finalize" + bsfx, "stereo_rows": "k_stereo_rows" + bsfx,
                     "proj_candidates": "k_proj_candidates" + bsfx, "proj_cells": "k_proj_cells" + bsfx, "imu_preintegrate": "k_imu_preintegrate" + bsfx,
                     "track_predict": "k_track_predict" + bsfx, "track_repredict": "k_track_repredict" + bsfx, "pack": "k_track_pack" + bsfx,
                     "pyramid": "k_resize", "ba_solve": "k_ba_solve_mfma64", "ssc": "k_ssc<2>",
                     "ba_schur": "k_ba_schur", "ba_back": "k_ba_back", "ba_eval": "k_ba_factors<1>", "ba_linearize": "k_ba_factors<0>", "ba_chi2": "k_ba_chi2",
                     "fast": "k_fast", "blur": "k_blur", "gather": "k_gather", "orient_desc": "k_orient_desc"}
            try:
                pmc_all = json.load(open(PMC_FILE))
            except Exception:      # noqa: BLE001
                pmc_all = {}
            WAVE_SLOTS = 256 * 32

            # resident-wave limit of a kernel, as a fraction of a CU's 32 wave slots (-Rpass-analysis=kernel-resource-usage /
            # LDS per workgroup): k_ssc<2> = two 8-wave tasks per CU (67 KB of LDS each), k_fast = 7 waves per SIMD (71 VGPRs)
            OCC_LIMIT = {"ssc": 16 / 32.0, "fast": 28 / 32.0}

            def slot_share(k):
                w = pmc_all.get(KNAME.get(k, ""), {}).get("SQ_WAVES_avg")
                return (min(1.0, w / WAVE_SLOTS) if w else 1.0) * OCC_LIMIT.get(k, 1.0)
            dom = max(per_frame, key=lambda k: per_frame[k] * slot_share(k))
            dom_elapsed = max(per_frame, key=lambda k: per_frame[k])
            n_launch, alg_bytes, alg_flops = groups.get(dom, (nS, 0, 0))
            dom_ms = stage_ms[dom] / max(n_launch, 1)
            achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
            # with S sessions sharing the chip the kernel's launches overlap: aggregate rate of this kernel over the run =
            # bytes of all its launches / wall time of the timed region
            launches_per_frame = n_launch / nS if not dom.startswith("ba_") else n_launch / (nBA * frames_per_ba)
            agg = alg_bytes * launches_per_frame * S * args.steps / el / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launch_ms": dom_ms, "algorithmic_bytes": alg_bytes,
                    "launches_timed": n_launch,
                    "aggregate_GBps_all_sessions": agg, "aggregate_frac": agg / HBM_PEAK_GBS,
                    "lanes_per_launch": lb if not dom.startswith("ba_") else 1,
                    "wave_slot_share": slot_share(dom), "largest_by_elapsed_time": dom_elapsed,
                    "note": "per-launch figure of the dominant kernel group = the largest (device time per tracked frame x share of the GPU's "
                            "8192 wave slots one launch fills, SQ_WAVES of the PMC pass); one launch serves all lanes of a lockstep group, local-BA "
                            "kernels serve one session; aggregate_* = the same group's algorithmic bytes over all sessions / wall time; "
                            "roofline_top5 = the five largest groups by that weight, largest_by_elapsed_time = by elapsed time alone (a narrow local-BA launch)"}
            kname = KNAME.get(dom)
            if kname in pmc_all:
                roof["traffic"] = (2.0 * pmc_all[kname]["FETCH_SIZE_avg"] + pmc_all[kname]["WRITE_SIZE_avg"]) * 1024.0
                roof["traffic_source"] = os.path.relpath(PMC_FILE, ROOT) + " (2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, per launch)"
            if alg_flops:
                roof["achieved_gflops"] = alg_flops / (dom_ms * 1e-3) / 1e9
                roof["fp64_frac"] = roof["achieved_gflops"] / (FP64_PEAK_TFLOPS * 1e3)
            if dom == "ba_solve":
                # the one MFMA-shaped kernel of the path (v_mfma_f64 Cholesky of the reduced camera system): priced against the
                # dense fp64 matrix peak; the HBM view of the same launch stays in hbm_*
                roof.update({"bound": "mfma", "hbm_achieved_GBps": achieved, "hbm_frac": achieved / HBM_PEAK_GBS,
                             "achieved": roof["achieved_gflops"] / 1e3, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": roof["achieved_gflops"] / 1e3 / FP64_PEAK_TFLOPS,
                             "why_small": "a %.0f-unknown system (%.1f free keyframes) is %.0f flops per candidate: one workgroup, latency-bound; "
                                          "launch_ms is measured with the other sessions' kernels sharing the GPU" % (nBA6, Fk, alg_flops)})
            out["roofline"] = roof
            # the same per-launch pricing for the five largest groups by the same weight (context for `roofline`: which kernels move
            # bytes and which are latency- / instruction-bound chains); the largest by elapsed time alone is named in `roofline`
            top = []
            for k in sorted(per_frame, key=lambda q: -per_frame[q] * slot_share(q))[:5]:
                n_l, ab, _ = groups.get(k, (nS, 0, 0))
                ms_l = stage_ms[k] / max(n_l, 1)
                gbs = ab / (ms_l * 1e-3) / 1e9 if ms_l > 0 else 0.0
                top.append({"kernel": k, "ms_per_frame": per_frame[k], "launch_ms": ms_l, "algorithmic_bytes": ab,
                            "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS, "wave_slot_share": slot_share(k)})
            out["roofline_top5"] = top
            if "ba_solve" not in stage_ms:   # no local BA of the sampled sessions fell into this (short) timed region
                out["ba_solve_mfma"] = {"launch_ms": None, "unknowns": nBA6, "flops": groups["ba_solve"][2], "achieved_gflops": None, "fp64_frac": None,
                                        "note": "no sampled local BA in the timed region; profiles/r02_i_c2_kernel_stats.csv has the kernel "
                                                "(k_ba_solve_mfma64, ~35-40 us per launch): (6F)^3/3 flops per candidate - << 1 % of the fp64 MFMA peak"}
            if "ba_solve" in stage_ms:       # the only MFMA-eligible term of the path: always reported
                n_l, _, fl = groups["ba_solve"]
                ms_l = stage_ms["ba_solve"] / max(n_l, 1)
                out["ba_solve_mfma"] = {"launch_ms": ms_l, "unknowns": nBA6, "flops": fl, "achieved_gflops": fl / (ms_l * 1e-3) / 1e9 if ms_l > 0 else 0.0,
                                        "fp64_frac": (fl / (ms_l * 1e-3) / 1e9) / (FP64_PEAK_TFLOPS * 1e3) if ms_l > 0 else 0.0,
                                        "note": "reduce + dense reduced-camera Cholesky + substitutions of one trial round (4 lambda candidates); "
                                                "(6F)^3/3 flops per candidate - << 1 % of the fp64 MFMA peak by construction"}
        if latency:
            out["latency_single_session"] = latency
        if sweep:
            out["sweep"] = sweep
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, frames, poses, vel, fwd)

        def emit(c5res):
            if c5res is not None:
                out["c5_sharded_ba"] = c5res
            print(json.dumps(out), flush=True)
        emit_box["emit"] = emit
    if "timer" in emit_box:
        emit_box["timer"].start()
        try:
            c5 = run_c5(args, rank, world, local, dist, torch, backend)
        except Exception as e:      # noqa: BLE001
            c5 = {"error": "%s: %s" % (type(e).__name__, e)}
        c5_done[0] = True
        emit_box["timer"].cancel()
    if rank == 0:
        emit_box["emit"](c5)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
