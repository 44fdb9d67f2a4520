#!/usr/bin/env python3
"""Benchmark of the tracking + local-BA hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched through
torch.distributed.run, one rank per GPU.  W untimed warm-up steps, then exactly K timed steps bracketed by barrier +
device synchronize; MAX over ranks; rank 0 prints ONE JSON line.

A "step" = one stereo frame of every session: S independent SLAM sessions (sequences) share a GPU, each driven by its own
host thread INSIDE the library (vslam_fleet; Python only starts the run and waits).  Every session runs the CLOSED LOOP
(vslam_system): extraction L+R, stereo match, tracking against ITS OWN map (removeOutOfFrameMPs, {projection match, pose
solve} rounds, refinement), the keyframe rule, insertKeyFrame, and - on the optimizer thread - covisibility window,
findNewPoints, local BA on the tracker's window, write-back; nothing is re-seeded from ground truth.  The frames are resident
in HBM before the timed region (or, with --host-images, in pinned host memory: then every frame's host-to-device copy is
inside its step).  Default scene: a long corridor (synth.corridor_sequence, rendered on the GPU as part of the set-up):
every session walks forward through DISTINCT poses and never turns around inside a run, so its map, keyframe rate and BA
windows are those of a camera that keeps exploring (--scene room: the small test scene replayed as a ping-pong, round 2's
workload).

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
# PMC passes of the default command's launch shape (tools/measure_set.sh); "_meta" names the shape they were taken on
PMC_FILE = os.path.join(ROOT, "profiles", os.environ.get("VSLAM_BENCH_PMC", "r03_d_pmc_summary.json"))


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
    """--scene room: n_frames rendered stereo pairs `step` source frames apart (numpy renderer), ground truth, velocities, IMU
    buckets of both directions."""
    import synth
    rig = synth.RIGS[cfg["rig"]]
    fps = rig["fps"]
    f0 = 7 * rank
    idx = [f0 + step * j for j in range(n_frames)]
    frames = render_frames(idx, cfg["rig"])
    poses = np.stack([fr[2] for fr in frames])
    vel, fwd, bwd = motion_data(cfg, idx, step, synth.pose_at)
    return rig, frames, poses, vel, fwd, bwd


def motion_data(cfg, idx, step, pose_fn):
    """velocities and the IMU buckets of both directions for the frames `idx` of the trajectory pose_fn(frame, fps)"""
    import synth
    rig = synth.RIGS[cfg["rig"]]
    fps = rig["fps"]
    n_frames = len(idx)
    h = 1e-4
    vel = np.stack([(pose_fn(f + h * fps, fps)[:3, 3] - pose_fn(f - h * fps, fps)[:3, 3]) / (2 * h) for f in idx])
    fwd, bwd = [None] * n_frames, [None] * n_frames
    if cfg["imu"]:
        for j in range(1, n_frames):
            S, dts, _ = synth.imu_samples(idx[j - 1], idx[j], fps, noise_seed=0x1A00 + idx[j], pose_fn=pose_fn)
            fwd[j] = (S[:, :3].copy(), S[:, 3:].copy(), np.arange(len(dts)) * 5e6)
        for j in range(n_frames - 1):
            a = idx[j + 1]
            S, dts, _ = synth.imu_samples(a, a + step, fps, noise_seed=0x2B00 + idx[j], pose_fn=lambda f, fp, a=a: pose_fn(2 * a - f, fp))
            bwd[j] = (S[:, :3].copy(), S[:, 3:].copy(), np.arange(len(dts)) * 5e6)
    return vel, fwd, bwd


def make_corridor_sequence(cfg, n_frames, step, speed, rank, dev):
    """--scene corridor (default): n_frames distinct stereo pairs along the corridor, rendered on the GPU into HBM."""
    import synth
    rig = synth.RIGS[cfg["rig"]]
    Ls, Rs, poses, idx = synth.corridor_sequence(cfg["rig"], n_frames, dev, frame_step=step, speed=speed, first=3 * rank)
    vel, fwd, bwd = motion_data(cfg, idx, step, lambda f, fp: synth.corridor_pose(f, fp, speed))
    return rig, Ls, Rs, poses, vel, fwd, bwd


def cpu_baseline(cfg, frames, poses, vel, fwd, delay, threads, budget_s=16.0, warm=20, timed=200):
    """The oracle's closed loop (oracle/vo_system.py on the stage functions of liboracle, built -O3 -march=native on this
    host) on a bounded sample of the same frames, SURVEY section 8(d)'s protocol: 20 warm-up + up to 200 timed frames
    (bounded by budget_s), median and p95 of the per-frame time.  threads = True: the reference's threading - left || right
    extraction on two threads (src/FeatureTracker.cpp:58-61), the local BA on the optimizer thread (src/System.cpp:19) on the
    same hand-over schedule as the GPU run -> 3 cores; False: one core."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    po.use_native_build()
    import vo_system
    import synth
    rig = synth.RIGS[cfg["rig"]]
    imu = None
    if cfg["imu"]:
        imu = dict(prm=po.imu_params(GRAVITY, IMU_NOISE[0], IMU_NOISE[2], IMU_NOISE[1], IMU_NOISE[3], synth.T_BC1))
    S = vo_system.System(rig, cfg["nfeat"], T0=poses[0], imu=imu, mapping_delay=delay[0], mapping_np_delay=delay[1], threads=threads)
    if cfg["imu"]:
        S.velocity = vel[0].copy()
    per, t_all = [], None
    for n in range(min(len(frames), warm + timed)):
        L, R = frames[n][0], frames[n][1]
        b = None
        if cfg["imu"] and n > 0:
            b = (np.concatenate([fwd[n][0], fwd[n][1]], 1), np.full(len(fwd[n][2]), 1.0 / 200))
        if n == warm:
            t_all = time.perf_counter()
        t0 = time.perf_counter()
        S.track(L, R, n, imu_bucket=b)
        if n >= warm:
            per.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > budget_s and len(per) >= 20:
                break
    per = np.array(per)
    return {"value": float(len(per) / per.sum()), "unit": "frames/s", "cores": 3 if threads else 1, "kind": "port",
            "median_ms": float(np.median(per) * 1e3), "p95_ms": float(np.percentile(per, 95) * 1e3),
            "sample": "%d timed frames of the same sequence after %d warm-up frames through the oracle's closed loop (extract L+R, stereo, "
                      "tracking, keyframe insertion, new points + local BA on the tracker's windows, mapping_delay %d / np %d; %d keyframes, %d local BAs), "
                      "%s; liboracle built -O3 -march=native -ffp-contract=off on this host; the reference itself cannot be built here "
                      "(OpenCV / GTSAM absent)" % (len(per), warm, delay[0], delay[1], len(S.keyFrames), sum(1 for l in S.log if "mapping" in l),
                                                   "the reference's threading: left || right extraction threads + the local BA's solve on the optimizer thread (3 cores)"
                                                   if threads else "single thread")}


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
    if world > 1 and backend != "nccl":
        # rehearsal transport (VSLAM_BENCH_BACKEND=gloo): the same sharded path, the reduced camera systems all-reduced by the
        # process group on host memory (vslam_comm_create_callback) - ranks may share one GPU
        def allreduce(a):
            dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM)
        comm = vc.comm_create_callback(rank, world, local, allreduce)
    elif world > 1:
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
           "final_error": r["reports"][1]["finalError"],
           "transport": ("rccl" if backend == "nccl" else "caller-supplied all-reduce (%s, host-staged)" % backend) if world > 1 else "none (single GPU)"}
    if comm is not None:
        comm.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--sessions", type=int, default=384, help="independent SLAM sessions (sequences) sharing each GPU")
    ap.add_argument("--lanes", type=int, default=128, help="sessions per lockstep group (vslam_batch: one launch per stage for all lanes of a group); "
                                                          "0 = one host thread and one set of launches per session")
    ap.add_argument("--scene", default="corridor", choices=["corridor", "room"], help="corridor: distinct poses along a long corridor, rendered on "
                    "the GPU, no session turns around inside a run; room: the parity tests' small scene replayed as a ping-pong (round 2)")
    ap.add_argument("--frames", type=int, default=0, help="distinct stereo frames of the sequence (0: corridor 640 / 400 for c3; room 100 / 60)")
    ap.add_argument("--frame-step", type=int, default=0, help="source frames between two sequence frames (0: corridor 1; room 2, 1 for c3)")
    ap.add_argument("--speed", type=float, default=0.0, help="corridor: forward speed in m/s (0: 0.5 for c1 / c2 as SURVEY section 8d, 1.0 for c3)")
    ap.add_argument("--host-images", action="store_true", help="frames in pinned host memory: every frame pays its H2D copy inside the step")
    ap.add_argument("--mapping", type=int, default=2, help="local mapping: 2 = optimizer thread per session (reference), 1 = synchronous, 0 = off")
    ap.add_argument("--mapping-delay", type=int, default=-1, help="(default: 4 frames, 2 for c3) --mapping 2: the fixed schedule of the optimizer "
                    "thread's hand-over (vslam_system_config::mapping_delay): new points arrive with the frame after the keyframe, the "
                    "local BA's write-back + changePosesLCA k frames after it.  Keyframes are at least 5 frames apart, so with k <= 5 every "
                    "keyframe gets its pass - the reference's steady state at camera rate")
    ap.add_argument("--mapping-np-delay", type=int, default=-1, help="(default: 2 frames, 1 for c3) --mapping 2: the new points of a pass are written a "
                    "frames after the hand-over (vslam_system_config::mapping_np_delay); the local BA collects its window at that moment")
    ap.add_argument("--prime", type=int, default=-1, help="(default 160 corridor / 60 room) ""untimed frames every session tracks BEFORE the warm-up steps, so that the timed "
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
    corridor = args.scene == "corridor"
    if os.environ.get("VSLAM_DUMP_MAPS"):
        # diagnostics (tools/resolve_stack.py): keep a fresh copy of this process's address map, so that the raw PCs and the
        # fault address of a crash trace can be resolved against the mappings of the process that crashed
        import threading

        def _dump_maps(path=os.environ["VSLAM_DUMP_MAPS"]):
            while True:
                try:
                    data = open("/proc/self/maps").read()
                    with open(path + ".tmp", "w") as f:
                        f.write(data)
                    os.replace(path + ".tmp", path)
                except Exception:      # noqa: BLE001
                    pass
                time.sleep(0.25)
        threading.Thread(target=_dump_maps, daemon=True).start()
    if args.frames <= 0:
        args.frames = (400 if args.config == "c3" else 640) if corridor else (60 if args.config == "c3" else 100)
    if args.frame_step <= 0:
        args.frame_step = 1 if (corridor or args.config == "c3") else 2
    if args.speed <= 0:
        args.speed = 1.0 if args.config == "c3" else 0.5
    if args.prime < 0:
        args.prime = 160 if corridor else 60
    seq = None
    if args.config != "c5" and not corridor:      # rendered before torch / HIP start
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
    if corridor:
        rig, Ls, Rs, poses, vel, fwd, bwd = make_corridor_sequence(cfg, args.frames, args.frame_step, args.speed, rank, dev)
        torch.cuda.synchronize()
        if args.host_images:
            Ls, Rs = Ls.cpu().pin_memory(), Rs.cpu().pin_memory()
        bufs = [(Ls[j], Rs[j]) for j in range(args.frames)]
        frames = None
    else:
        rig, frames, poses, vel, fwd, bwd = seq
        if args.host_images:
            bufs = [(torch.from_numpy(L).pin_memory(), torch.from_numpy(R).pin_memory()) for (L, R, _) in frames]
        else:
            bufs = [(torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)) for (L, R, _) in frames]
    w, h = rig["w"], rig["h"]
    lp = [b[0].data_ptr() for b in bufs]
    rp = [b[1].data_ptr() for b in bufs]
    # no session turns around inside the run: start offsets leave room for every frame the run tracks (else: ping-pong)
    need = args.prime + args.warmup + args.steps + 2
    start_span = max(args.frames - need, 0) if corridor else 0
    turns = corridor and start_span < 8
    if turns:
        start_span = 0
    imu = dict(gravity=GRAVITY, noise=IMU_NOISE, T_bs=__import__("synth").T_BC1, hz=200) if cfg["imu"] else None
    if args.mapping_delay < 0:
        # The reference's optimizer thread ends a pass within a frame or two of camera time: k = 2 for the KITTI-like sequence
        # (10 fps, ~10x the motion per frame), k = 4 for the EuRoC-like ones (20 fps).  The mode is parity-tested frame by frame
        # against the oracle's restatement of the same schedule (tests/test_gpu_system.py).
        args.mapping_delay = 2 if args.config == "c3" else 4
    if args.mapping_np_delay < 0:
        args.mapping_np_delay = 1 if args.config == "c3" else 2
    args.mapping_np_delay = max(1, min(args.mapping_np_delay, args.mapping_delay))
    scfg = vc.system_config(rig, cfg["nfeat"], imu=imu, local_mapping=args.mapping, device=local, mapping_delay=args.mapping_delay,
                            mapping_np_delay=args.mapping_np_delay)

    def make_fleet(S, lanes):
        return vc.Fleet(scfg, S, lp, rp, w, not args.host_images, poses=poses, velocities=vel,
                        imu_forward=fwd if cfg["imu"] else None, imu_backward=bwd if cfg["imu"] else None, lanes=lanes, start_span=start_span)

    lanes = min(args.lanes, args.sessions)
    fleet = make_fleet(args.sessions, lanes)
    if args.prime > 0:
        fleet.run(args.prime)        # set-up: the sessions' first frames (map initialisation, the first keyframes)
    # per-stage HIP events on every 3rd step of group / session 0 (every step of a short timed region).  The warm-up steps are
    # sampled too, so that the first timed sample does not carry the timers' one-time set-up; their readings are discarded.
    sample_every = 1 if args.steps < 60 else 3
    if os.environ.get("VSLAM_BENCH_NO_SAMPLING"):
        sample_every = 0
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

    # workload probe (after the timed region, its own extractor / matcher): the figures the byte formulas need and the fleet
    # report does not carry - keypoints per image, FAST candidates per image, the stereo kernel's own counters
    probe = None
    if rank == 0:
        probe = workload_probe(vc, rig, cfg, bufs, w, h, args.host_images, local)

    # N > 1: the landmark-sharded 64-keyframe / 100 k-landmark BA over RCCL as an extra section of the line.  It is the
    # only collective of the path, so it runs under a watchdog: if it has not finished in time the closed-loop line is still
    # printed, with the failure recorded, and the process ends with a NON-ZERO status.
    c5 = None
    emit_lock = __import__("threading").Lock()
    emitted = [False]
    emit_box = {}
    if world > 1 and not os.environ.get("VSLAM_BENCH_SKIP_C5"):
        import threading

        def _watchdog():
            if rank == 0 and "emit" in emit_box:
                emit_box["emit"]({"error": "the sharded-BA section did not finish within %d s" % args.c5_timeout})
            os._exit(3)
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
        roundsPerBA = max(rep["ba_rounds"] / nBAall, 1.0)      # trial rounds per local BA (one round = up to 4 lambda candidates at once)
        launch_n = {k[:-2]: v for k, v in stage_ms.items() if k.endswith("#n")}      # measured launches behind the local-BA sums
        stage_ms = {k: v for k, v in stage_ms.items() if not k.endswith("#n")}

        def ba_group(name, per_ba_launches, lane_bytes, lane_flops=0.0):
            """one launch of a local-BA stage serves the lanes of a cohort that are still iterating: launches as measured, bytes per
            launch = per-lane bytes x (lane-launches of the sampled BAs / launches)"""
            n_l = max(launch_n.get(name, nBA * per_ba_launches), 1.0)
            lanes_eff = max(nBA * per_ba_launches / n_l, 1.0)
            return (n_l, lane_bytes * lanes_eff, lane_flops * lanes_eff)
        nBA6 = 6 * Fk
        nk = probe["keys_per_image"]                 # measured: kept keypoints per image
        ncand = probe["fast_candidates_per_image"]   # measured: FAST corners per image before the suppression
        Mact = rep["sum_active"] / max(rep["frames"], 1)      # measured: active map points per tracked frame
        frames_per_ba = rep["frames"] / nBAall
        # One launch of a tracking / extraction stage serves `lb` sessions (the lanes of a lockstep group; 1 without batching).
        lb = max(lanes, 1)
        nStep = max(cnt["frames"] / lb, 1)         # sampled steps
        # kernel group -> (launches over the sampled region, ALGORITHMIC bytes per launch, algorithmic flops per launch); SURVEY 8(d)
        b_cand, b_res = Mact * 60 + 2 * nk * 60 + Mact * 128, Mact * (128 + 8 + 8) + 2 * nk * 8
        b_pose = Mact * (24 + 8 + 4) + 2 * nk * 28 + (8 * 514 if cfg["imu"] else 0)
        b_stereo = 32.0 * (nk + probe["stereo_hamming_tests"]) + 1452.0 * probe["stereo_sad_refinements"]     # B_st
        groups = {
            "pyramid": (7 * nStep, lb * nimg * (sum(px[:-1]) + sum(px[1:])) / 7.0, 0),     # read level l-1, write level l
            "fast": (nStep, lb * nimg * (sumP + 4 * ncand), 0),                            # every level read once + packed candidates
            "gather": (nStep, lb * nimg * (8 * ncand), 0),
            "ssc": (nStep, lb * nimg * (8 * ncand + 4 * nk), 0),                            # candidates in, picks out
            "blur": (nStep, lb * nimg * (2 * sumP), 0),                                    # read + write every level
            "orient_desc": (nStep, lb * nimg * nk * (28 + 32 + 709 + 512), 0),             # keypoint + descriptor + disc + BRIEF taps
            "stereo_rows": (nStep, lb * nk * 28, 0),
            "stereo_match": (nStep, lb * b_stereo, 0),
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
            "ba_linearize": ba_group("ba_linearize", 2, R_ * (8 + 16 + 8 + 96 + 24 + 160)),      # idx, uv, sigma, pose, point, stored J
            "ba_schur": ba_group("ba_schur", roundsPerBA, R_ * 160 + L_ * 24, 2 * 36 * k2),       # stored J read once (S stays in LDS)
            "ba_solve": ba_group("ba_solve", roundsPerBA, 8 * (nBA6 * nBA6 + nBA6) * 2, nBA6 ** 3 / 3.0),   # reduced system in, delta out
            "ba_back": ba_group("ba_back", roundsPerBA, R_ * 160 + L_ * 48),
            "ba_eval": ba_group("ba_eval", roundsPerBA, R_ * (160 + 16 + 8 + 96 + 24)),
            "ba_chi2": ba_group("ba_chi2", 2, R_ * (16 + 8 + 96 + 24)),
        }
        ba_lanes = nBA / max(cnt.get("ba_cohorts", nBA), 1)      # sessions served per batched local-BA call
        nS = nStep * lb
        per_frame = {}
        for k, v in stage_ms.items():
            per_frame[k] = v / (nBA * frames_per_ba) if k.startswith("ba_") else v / nS      # device ms per tracked frame
        # algorithmic bytes of one tracked stereo frame over the whole path = sum over the groups of bytes per launch x launches per frame
        path_bytes = 0.0
        for k, (n_l, ab, _) in groups.items():
            if k in ("pose_lm", "pose_imu_lm") and k != ("pose_imu_lm" if cfg["imu"] else "pose_lm"):
                continue
            if k == "imu_preintegrate" and not cfg["imu"]:
                continue
            path_bytes += ab * (n_l / (nBA * frames_per_ba) if k.startswith("ba_") else n_l / nS)
        value = world * S * args.steps / el
        out = {
            "metric": "frames/sec (extract+match+localBA), %d feat stereo %dx%d" % (nfeat, w, h),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i32 (extract, match) + f64 (pose LM, local BA)", "data": "synthetic",
            "config": {"workload": cfg["name"] + "; closed loop per session (keyframe rule, insertKeyFrame, new points + local BA on the tracker's "
                                   "own covisibility window, write-back, changePosesLCA); %s; measured per run: one keyframe every %.1f frames, one local BA "
                                   "every %.1f frames (%.0f landmarks, %.0f residual blocks, %.1f free keyframes on average), %.0f keypoints per image, "
                                   "%.0f active map points per frame"
                                   % (("corridor scene: %d distinct stereo frames %s, %.2f m/s, every session walks forward from its own start frame%s"
                                       % (args.frames, "in pinned host memory (H2D inside the step)" if args.host_images else "rendered into HBM",
                                          args.speed, " and turns around at the end (the run is longer than the sequence)" if turns else
                                          " and never revisits a place inside the run"))
                                      if corridor else "room scene: %d rendered frames %s, replayed as a ping-pong"
                                      % (args.frames, "in pinned host memory (H2D inside the step)" if args.host_images else "resident in HBM"),
                                      rep["frames"] / max(rep["keyframes"], 1), frames_per_ba, L_, R_, Fk, nk, Mact),
                       "sessions_per_gpu": S, "lanes_per_group": lanes, "prime_frames": args.prime,
                       "local_mapping": {0: "off", 1: "inside the frame (synchronous)", 2: "device work beside tracking, fixed schedule"}.get(args.mapping, "?") +
                                        ("; new points %d frame(s) after the hand-over (mapping_np_delay), the BA's write-back + changePosesLCA %d frames "
                                         "after it (mapping_delay): the schedule of tests/test_gpu_system.py::test_closed_loop_parity_async_*"
                                         % (args.mapping_np_delay, args.mapping_delay) if args.mapping == 2 else ""),
                       "step": "one stereo frame of each of the %d sessions" % S,
                       "threads": ("%d lockstep groups of %d sessions (vslam_batch: one launch per stage for all lanes), one driver thread + a "
                                   "host-phase pool + the mapping engine per group" % ((S + lanes - 1) // lanes, lanes)) if lanes > 0 else
                                  "one host thread per session inside the library (vslam_fleet) + one mapping thread per session",
                       "parallelism": "replicas x%d, %d sessions per GPU" % (world, S)},
            "tracking": {"mean_inliers": rep["sum_inliers"] / max(rep["frames"] - 0, 1), "min_inliers": rep["min_inliers"],
                         "lost_frames": rep["lost_frames"], "mean_rounds": rep["sum_rounds"] / max(rep["frames"], 1),
                         "keyframes": rep["keyframes"], "local_bas": rep["mappings"], "new_points": rep["new_points"],
                         "rms_position_error_m": float(np.sqrt(rep["sum_sq_position_error"] / max(rep["frames"], 1))),
                         "max_position_error_m": rep["max_position_error"]},
            "workload_probe": probe,
            "stage_ms_per_frame": {k: v for k, v in sorted(per_frame.items())},
            "stage_sampling": "HIP events on every %s step of group 0 (all its lanes per launch) and on the local BAs of its first sessions that "
                              "complete in those steps (%d frames, %d BAs); per tracked frame; BA stages amortised over %.1f frames per BA"
                              % ("3rd" if sample_every == 3 else "single", cnt["frames"], cnt["ba"], frames_per_ba),
            "path": {"algorithmic_bytes_per_frame": path_bytes, "achieved_GBps": path_bytes * value / world / 1e9, "peak_GBps": HBM_PEAK_GBS,
                     "frac": path_bytes * value / world / 1e9 / HBM_PEAK_GBS,
                     "note": "whole path: sum over the kernel groups of (algorithmic bytes per launch x launches per tracked frame, SURVEY 8(d) formulas "
                             "with the measured keypoint / candidate / active-point / stereo-test counts) x frames/s of one GPU / HBM peak"},
        }
        if per_frame:
            # `roofline` prices the kernel group with the LARGEST ELAPSED DEVICE TIME per tracked frame in this run (HIP events on the
            # launching stream, summed over the sampled launches) - no weighting.  roofline_top5: the five largest by the same measure.
            bsfx = "_b" if lanes > 0 else ""
            KNAME = {"proj_resolve": "k_proj_resolve" + bsfx, "pose_imu_lm": "k_pose_imu_lm" + bsfx, "pose_lm": "k_pose_lm" + bsfx,
                     "stereo_match": "k_stereo_match" + bsfx, "stereo_finalize": "k_stereo_finalize" + bsfx, "stereo_rows": "k_stereo_rows" + bsfx,
                     "proj_candidates": "k_proj_candidates" + bsfx, "proj_cells": "k_proj_cells" + bsfx, "imu_preintegrate": "k_imu_preintegrate" + bsfx,
                     "track_predict": "k_track_predict" + bsfx, "track_repredict": "k_track_repredict" + bsfx, "pack": "k_track_pack" + bsfx,
                     "pyramid": "k_resize", "ba_solve": "k_ba_solve_mfma64", "ssc": "k_ssc<2>",
                     "ba_schur": "k_ba_schur2", "ba_back": "k_ba_back2", "ba_eval": "k_ba_factors<1>", "ba_linearize": "k_ba_factors<0>", "ba_chi2": "k_ba_chi2",
                     "fast": "k_fast", "blur": "k_blur", "gather": "k_gather", "orient_desc": "k_orient_desc"}
            try:
                pmc_all = json.load(open(PMC_FILE))
            except Exception:      # noqa: BLE001
                pmc_all = {}
            # the committed counters apply only to the launch shape they were collected on
            meta = pmc_all.get("_meta", {})
            pmc_ok = bool(meta) and meta.get("config") == args.config and meta.get("lanes") == lanes and meta.get("scene", "room") == args.scene

            def price(k):
                n_l, ab, fl = groups.get(k, (nS, 0, 0))
                ms_l = stage_ms[k] / max(n_l, 1)
                gbs = ab / (ms_l * 1e-3) / 1e9 if ms_l > 0 else 0.0
                return n_l, ab, fl, ms_l, gbs
            dom = max(per_frame, key=lambda k: per_frame[k])
            n_launch, alg_bytes, alg_flops, dom_ms, achieved = price(dom)
            roof = {"bound": "hbm", "kernel": KNAME.get(dom, dom), "group": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launch_ms": dom_ms, "algorithmic_bytes": alg_bytes,
                    "launches_timed": n_launch, "device_ms_per_frame": per_frame[dom],
                    "sessions_per_launch": (ba_lanes if dom.startswith("ba_") else lb),
                    "local_ba_lanes_per_cohort": ba_lanes,
                    "note": "the kernel group with the largest elapsed device time per tracked frame in THIS run (HIP events on the launching "
                            "stream); achieved = algorithmic bytes of one launch / its average duration"}
            kname = KNAME.get(dom)
            if pmc_ok and kname in pmc_all:
                roof["traffic"] = (2.0 * pmc_all[kname]["FETCH_SIZE_avg"] + pmc_all[kname]["WRITE_SIZE_avg"]) * 1024.0
                roof["traffic_source"] = ("from " + os.path.relpath(PMC_FILE, ROOT) + ", same launch shape (%s): 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, per launch"
                                          % json.dumps(meta, sort_keys=True))
            else:
                roof["traffic_source"] = "no committed PMC pass for this config / launch shape"
            if alg_flops:
                roof["achieved_gflops"] = alg_flops / (dom_ms * 1e-3) / 1e9
                roof["fp64_frac"] = roof["achieved_gflops"] / (FP64_PEAK_TFLOPS * 1e3)
            out["roofline"] = roof
            top = []
            for k in sorted(per_frame, key=lambda q: -per_frame[q])[:5]:
                n_l, ab, _, ms_l, gbs = price(k)
                e = {"group": k, "kernel": KNAME.get(k, k), "device_ms_per_frame": per_frame[k], "launch_ms": ms_l, "algorithmic_bytes": ab,
                     "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS}
                if pmc_ok and KNAME.get(k) in pmc_all:
                    e["traffic"] = (2.0 * pmc_all[KNAME[k]]["FETCH_SIZE_avg"] + pmc_all[KNAME[k]]["WRITE_SIZE_avg"]) * 1024.0
                top.append(e)
            out["roofline_top5"] = top
            if "ba_solve" in stage_ms:       # the only MFMA-eligible term of the path: always reported
                n_l, _, fl, ms_l, _ = price("ba_solve")
                out["ba_solve_mfma"] = {"launch_ms": ms_l, "unknowns": nBA6, "flops": fl, "achieved_gflops": fl / (ms_l * 1e-3) / 1e9 if ms_l > 0 else 0.0,
                                        "fp64_frac": (fl / (ms_l * 1e-3) / 1e9) / (FP64_PEAK_TFLOPS * 1e3) if ms_l > 0 else 0.0,
                                        "note": "reduce + dense reduced-camera Cholesky + substitutions of one trial round (4 lambda candidates); "
                                                "(6F)^3/3 flops per candidate - << 1 % of the fp64 MFMA peak by construction"}
            else:
                out["ba_solve_mfma"] = {"launch_ms": None, "unknowns": nBA6, "flops": groups["ba_solve"][2], "achieved_gflops": None, "fp64_frac": None,
                                        "note": "no sampled local BA in the timed region"}
        if latency:
            out["latency_single_session"] = latency
        if sweep:
            out["sweep"] = sweep
        if not args.no_cpu_baseline and world == 1:
            # the same frames on the host's cores (rank 0, N = 1 only): the first frames of the sequence
            nb = 224
            if corridor:
                hostf = [(Ls[j].cpu().numpy(), Rs[j].cpu().numpy()) for j in range(min(nb, args.frames))]
            else:
                hostf = [(f[0], f[1]) for f in frames[:nb]]
            dl = (args.mapping_delay, args.mapping_np_delay) if args.mapping == 2 else (0, 1)
            out["cpu_baseline"] = cpu_baseline(cfg, hostf, poses, vel, fwd, dl, True)
            out["cpu_baseline_single_thread"] = cpu_baseline(cfg, hostf, poses, vel, fwd, dl, False)

        def emit(c5res):
            with emit_lock:
                if emitted[0]:
                    return
                emitted[0] = True
                if c5res is not None:
                    out["c5_sharded_ba"] = c5res
                print(json.dumps(out), flush=True)
        emit_box["emit"] = emit
    failed = False
    if "timer" in emit_box:
        emit_box["timer"].start()
        try:
            c5 = run_c5(args, rank, world, local, dist, torch, backend)
        except Exception as e:      # noqa: BLE001
            c5 = {"error": "%s: %s" % (type(e).__name__, e)}
            failed = True
        emit_box["timer"].cancel()
    if rank == 0:
        emit_box["emit"](c5)
    if failed:
        # a rank whose sharded section raised leaves the others inside a collective: no destroy_process_group (it would block),
        # no in-process restart - the line (rank 0) carries the error, the status is non-zero
        sys.stdout.flush()
        os._exit(3)
    if world > 1:
        dist.destroy_process_group()


def workload_probe(vc, rig, cfg, bufs, w, h, host_images, device):
    """Keypoints / FAST candidates per image and the stereo kernel's counters (Hamming tests, SAD refinements), averaged over a
    few frames of the sequence: one extractor + matcher of their own, after the timed region."""
    idxs = sorted(set(int(v) for v in np.linspace(0, len(bufs) - 1, 6)))
    fe = vc.Extractor(w, h, cfg["nfeat"], batch=2, device=device)
    m = vc.Matcher(rig, fe, 0, fe, 1)
    keys = cands = tests = sads = 0.0
    for j in idxs:
        L, R = bufs[j]
        if host_images:
            fe.set_image(0, L.numpy()); fe.set_image(1, R.numpy())
        else:
            fe.set_image_device(0, L.data_ptr(), w); fe.set_image_device(1, R.data_ptr(), w)
        fe.run()
        kL, _ = fe.fetch(0); kR, _ = fe.fetch(1)
        keys += 0.5 * (len(kL) + len(kR))
        cands += 0.5 * sum(len(fe.candidates(i, l)) for i in (0, 1) for l in range(8))
        m.use_extractor_keys()
        m.stereo_match()
        st = m.stereo_fetch(len(kL), len(kR))
        tests += st["candidates"]; sads += st["sad"]
    n = float(len(idxs))
    m.close(); fe.close()
    return {"frames_probed": len(idxs), "keys_per_image": keys / n, "fast_candidates_per_image": cands / n,
            "stereo_hamming_tests": tests / n, "stereo_sad_refinements": sads / n}


if __name__ == "__main__":
    main()
