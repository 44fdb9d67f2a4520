#!/usr/bin/env python3
"""Benchmark of the tracking + local-BA hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched
through torch.distributed.run, one rank per GPU.  W untimed warm-up steps, then exactly K timed
steps bracketed by barrier + device synchronize; MAX over ranks; rank 0 prints ONE JSON line.

A "step" is one stereo frame through the hot path, inputs (the rendered stereo images) already
resident in HBM:
    extract L+R (pyramid, FAST, SSC, orientation, blur, BRIEF)  ->  stereo match  ->
    tracking loop against the map points of the previous frame
        (removeOutOfFrameMPs, {projection match, pose-only LM} rounds, PredictMPsPosition, refine)  ->
    initializeMap-style map refresh  ->  every KF_PERIOD-th frame one local BA (amortised).
PyTorch is used for device buffers and torch.distributed only; all compute goes through the C ABI
of gtsam-vslam_amd/libvslam_hip.so.  Multi-GPU: replicas (each rank tracks its own sequence and
runs its own local BAs; the path has no cross-frame exchange — DESIGN.md "multi-GPU").
"""
import argparse
import json
import os
import sys
import time

# The path uses 5 HIP streams (2 extractor pairs, matcher/tracker, local BA, torch); the HIP runtime multiplexes
# streams onto 4 hardware queues by default, which serialises independent streams behind each other's long
# single-workgroup kernels.  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8" if "--sessions" not in sys.argv else "24")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gtsam-vslam_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GRAVITY = (0.0, 9.81, 0.0)
IMU_NOISE = (1.6968e-04, 1.9393e-05, 2.0e-3, 3.0e-3)   # gyro density / walk, accel density / walk (config_V1_02.yaml)
KF_PERIOD = 5           # keyFrameCountEnd (include/FeatureTracker.h): a keyframe, hence a local BA, every 5 frames


def level_pixels(w, h, nlevels=8, scale=1.2):
    """level pixel counts from the reference constructor formulas (float arithmetic)."""
    px, s = [], np.float32(1.0)
    for _ in range(nlevels):
        inv = np.float32(1.0) / s
        px.append(int(np.rint(np.float32(w) * inv)) * int(np.rint(np.float32(h) * inv)))
        s = np.float32(s * np.float32(scale))
    return px


def cpu_baseline(frames, poses, imus, rig, nfeat, ba_prob, budget_s=12.0, threaded=True):
    """Oracle (CPU restatement of the reference) on a bounded sample of the same per-frame workload.
    threaded = the reference's own threading: left || right extraction on two threads (src/FeatureTracker.cpp:58-61),
    local BA on the optimizer thread (src/System.cpp:19), everything else on the tracking thread -> 3 cores."""
    import threading
    import queue
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pyoracle as po
    from test_gpu_track import oracle_init_map, oracle_track
    import synth
    eL, eR = po.Extractor(nfeat), po.Extractor(nfeat)
    prm = po.imu_params(GRAVITY, IMU_NOISE[0], IMU_NOISE[2], IMU_NOISE[1], IMU_NOISE[3], synth.T_BC1)
    pool = ThreadPoolExecutor(1) if threaded else None
    ba_q = queue.Queue(maxsize=1)

    def ba_worker():
        while True:
            if ba_q.get() is None:
                ba_q.task_done()
                return
            po.local_ba(rig, eL.sigmaFactor, eL.InvSigmaFactor, ba_prob)
            ba_q.task_done()

    if threaded:
        threading.Thread(target=ba_worker, daemon=True).start()
    n, t0, mp = 0, time.perf_counter(), None
    while True:
        i = n % len(frames)
        L, R = frames[i]
        if threaded:
            fut = pool.submit(eR.extract, R)
            kL, dL = eL.extract(L)
            kR, dR = fut.result()
        else:
            kL, dL = eL.extract(L)
            kR, dR = eR.extract(R)
        st = po.stereo_match(eL, eR, rig, kL, dL, kR, dR)
        if mp is not None and i > 0:
            oracle_track(po, rig, eL, (kL, dL, kR, dR), st, mp, poses[i][1], 5,
                         imu=(prm, poses[i - 1][0], imus[i][3], np.zeros(6), imus[i][0], imus[i][1]))
        mp = oracle_init_map(rig, eL, kL, dL, st, poses[i][0])
        if n % KF_PERIOD == KF_PERIOD - 1:
            if threaded:
                ba_q.put(1)
            else:
                po.local_ba(rig, eL.sigmaFactor, eL.InvSigmaFactor, ba_prob)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 200:
            break
    if threaded:
        ba_q.join()
        el = time.perf_counter() - t0
        ba_q.put(None)
        pool.shutdown()
    return {"value": n / el, "unit": "frames/s", "cores": 3 if threaded else 1, "kind": "port",
            "sample": "%d stereo frames of the same workload (extract L+R, stereo, tracking loop, local BA every %d "
                      "frames), oracle/ built -O2, %s" % (n, KF_PERIOD, "reference-like threading: L || R extraction threads + "
                      "optimizer thread" if threaded else "single thread")}


STAGE_SAMPLE = 3        # per-kernel HIP-event timing on every 3rd frame (coprime to the 8-frame replay cycle)
BA_SAMPLE = 2           # ... and on every 2nd local BA: two event records per launch are a real cost on this launch-bound path

# fp64 vector peak used for the latency-bound solver kernels (MI355X_MICROARCH.md: 78.6 TFLOP/s fp64 vector/matrix)
FP64_PEAK_TFLOPS = 78.6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic stereo frames kept in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sessions", type=int, default=1,
                    help="independent SLAM sessions (sequences) sharing each GPU; the headline number uses 1")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="extract and track on one host thread (frame n+1 is not extracted while frame n is tracked)")
    args = ap.parse_args()

    import threading
    import queue
    import torch
    import torch.distributed as dist
    import synth
    import vslam_capi as vc

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    rig_name, nfeat = "euroc", 1500
    rig = synth.RIGS[rig_name]
    w, h = rig["w"], rig["h"]
    # a short rendered sequence (consecutive frames, so that tracking has real inter-frame motion);
    # it is replayed cyclically, frame 0 of each cycle re-initialises the map
    frames, poses, imus = [], [], []
    for i in range(args.frames):
        f = i + 11 * rank
        L, R, T = synth.stereo_frame(f, rig_name)
        frames.append((L, R))
        poses.append((T, synth.pose_at(f - 0.3, rig["fps"])))   # (ground truth, constant-velocity style prediction)
        # IMU bucket between frame f-1 and f (200 Hz, reference noise densities) + velocity at frame f-1
        S, dts, _ = synth.imu_samples(f - 1, f, rig["fps"], noise_seed=0x1A00 + f)
        hh = 1e-4
        v_prev = (synth.pose_at(f - 1 + hh * rig["fps"], rig["fps"])[:3, 3] - synth.pose_at(f - 1 - hh * rig["fps"], rig["fps"])[:3, 3]) / (2 * hh)
        imus.append((S, dts, np.arange(len(dts)) * 5e6, v_prev))
    d_frames = [(torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)) for (L, R) in frames]
    ba_prob = synth.make_ba_problem(rig_name, n_local=10, n_fixed=4, n_lm=3000, seed=0xBA5E + rank)

    pipelined = not args.no_pipeline
    stage_ms = {}
    counters = {"track_inliers": 0, "track_frames": 0, "ba_calls": 0, "ba_residuals": 0, "ba_landmarks": 0, "ba_sum_k2": 0,
                "ba_trials": 0, "ba_iters": 0, "ba_free_kf": 0, "sampled_frames": 0, "sampled_solves": 0, "sampled_tracked": 0}
    lock = threading.Lock()

    def add(d):
        with lock:
            for k, v in d.items():
                stage_ms[k] = stage_ms.get(k, 0.0) + v

    class Session:
        """One SLAM session (one camera rig / sequence): its extractor pair(s), matcher + tracker state, local-BA thread.
        Session 0 carries the stage timers and counters; further sessions (--sessions) are independent replicas that share
        the GPU (every kernel of this path occupies a few CUs at most, so sequences are the unit that fills the chip)."""

        def __init__(self, sid):
            self.sid = sid
            self.timed = sid == 0
            self.fes = [vc.Extractor(w, h, nfeat, batch=2, device=local) for _ in range(2 if pipelined else 1)]
            self.fm = vc.Matcher(rig, self.fes[0], 0, self.fes[0], 1)
            self.sigmaF, self.invSigmaF = self.fes[0].sigmaFactor, self.fes[0].InvSigmaFactor
            # Local BA runs on its own host thread / HIP stream, concurrently with tracking - the reference's optimizer
            # thread (src/System.cpp:19, LocalMapper::beginLocalMapping).  At most one BA is in flight (the reference's
            # keyFrameAdded / LBADone handshake: the tracker blocks on the hand-over while the previous one still runs);
            # all of them finish inside the timed region.
            self.ba_q = queue.Queue(maxsize=1)
            self.ba_state = {"record": False, "n": 0}
            self.ba_thread = threading.Thread(target=self.ba_worker, daemon=True)
            self.ba_thread.start()

        def ba_worker(self):
            torch.cuda.set_device(local)
            while True:
                job = self.ba_q.get()
                if job is None:
                    self.ba_q.task_done()
                    return
                sample = self.timed and self.ba_state["record"] and (self.ba_state["n"] % BA_SAMPLE == 0)
                self.ba_state["n"] += 1
                vc.local_ba_set_timing(sample)
                r = vc.local_ba(rig, self.sigmaF, self.invSigmaF, ba_prob, device=local)
                if sample:
                    add(vc.local_ba_timings())
                    counters["ba_calls"] += 1
                    counters["ba_residuals"], counters["ba_landmarks"], counters["ba_sum_k2"] = r["residuals"], r["landmarks"], r["sum_k2"]
                    counters["ba_free_kf"] = r["free_kf"]
                    counters["ba_trials"] += r["reports"][0]["inner"] + r["reports"][1]["inner"]
                    counters["ba_iters"] += r["reports"][0]["iterations"] + r["reports"][1]["iterations"]
                self.ba_q.task_done()

        def sampled(self, n, record):
            return self.timed and record and (n % STAGE_SAMPLE == 0)

        def extract(self, n, record):
            fe = self.fes[n % len(self.fes)]
            dL, dR = d_frames[n % len(d_frames)]
            rec = self.sampled(n, record)
            fe.set_timing(rec)
            fe.set_image_device(0, dL.data_ptr(), w)
            fe.set_image_device(1, dR.data_ptr(), w)
            fe.run()
            if rec:
                add(fe.timings())

        def track(self, n, record):
            i = n % len(d_frames)
            fe, fm = self.fes[n % len(self.fes)], self.fm
            rec = self.sampled(n, record)
            fm.set_timing(rec)
            if len(self.fes) > 1:
                fm.bind_extractors(fe, 0, fe, 1)
            fm.stereo_match()
            if i > 0:
                S, dts, ts, v_prev = imus[i]
                T_cw, rep, vel, bias = vc.tracker_track_imu(fm, poses[i][1], 5, GRAVITY, IMU_NOISE, synth.T_BC1, poses[i - 1][0],
                                                            v_prev, np.zeros(6), S[:, :3], S[:, 3:], ts, 200)
                if record and self.timed:
                    counters["track_inliers"] += rep["n_inliers"]; counters["track_frames"] += 1
                if rec:
                    counters["sampled_solves"] += rep["rounds"] + 1; counters["sampled_tracked"] += 1
            vc.tracker_init_map(fm, poses[i][0])
            if rec:
                counters["sampled_frames"] += 1
                add(fm.timings())
            if n % KF_PERIOD == KF_PERIOD - 1 and not os.environ.get("VSLAM_BENCH_SKIP_BA"):     # (diagnostic switch only)
                self.ba_state["record"] = record
                self.ba_q.put(1)          # blocks while the previous local BA is still running

        def run_frames(self, first, count, record):
            """`count` frames through the path; returns when every one of them (and every local BA) has completed."""
            torch.cuda.set_device(local)
            if not pipelined:
                for n in range(first, first + count):
                    self.extract(n, record)
                    self.track(n, record)
            else:
                # frame-level pipeline: the extraction thread works on frame n+1 (own extractor pair, own stream)
                # while this thread matches / tracks frame n; two extractor buffers, so it is at most one frame ahead
                free = threading.Semaphore(len(self.fes))
                ready = queue.Queue()
                err = []

                def extract_worker():
                    try:
                        torch.cuda.set_device(local)
                        for n in range(first, first + count):
                            free.acquire()
                            self.extract(n, record)
                            ready.put(n)
                    except Exception as e:      # noqa: BLE001
                        err.append(e)
                        ready.put(-1)

                th = threading.Thread(target=extract_worker, daemon=True)
                th.start()
                for _ in range(count):
                    n = ready.get()
                    if n < 0:
                        raise err[0]
                    self.track(n, record)
                    free.release()
                th.join()
            self.ba_q.join()

    sessions = [Session(i) for i in range(args.sessions)]
    fes = sessions[0].fes

    def run_frames(first, count, record):
        if len(sessions) == 1:
            sessions[0].run_frames(first, count, record)
            return
        errs = []

        def go(s):
            try:
                s.run_frames(first, count, record)
            except Exception as e:      # noqa: BLE001
                errs.append(e)

        ths = [threading.Thread(target=go, args=(s,), daemon=True) for s in sessions]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]

    run_frames(0, args.warmup, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_frames(args.warmup, args.steps, True)     # every frame and every local BA of the timed steps completes inside
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        px = level_pixels(w, h)
        sumP, nimg = sum(px), 2
        nk = sum(len(fes[0].fetch(i)[0]) for i in range(2)) / 2.0
        R_, L_, k2, Fk = counters["ba_residuals"], counters["ba_landmarks"], counters["ba_sum_k2"], counters["ba_free_kf"]
        nS, nT = max(counters["sampled_frames"], 1), max(counters["sampled_tracked"], 1)
        nBA, nTrial = max(counters["ba_calls"], 1), max(counters["ba_trials"], 1)
        nLin = max(counters["ba_iters"] + 2 * counters["ba_calls"], 1)      # one linearisation per iteration + the initial one of each pass
        nSolve = max(counters["sampled_solves"], 1)
        nBA6 = 6 * Fk
        Mact = 620.0    # active map points per tracked frame of this sequence (track report n_active)
        # kernel group -> (launches over the sampled region, algorithmic bytes per launch, algorithmic flops per launch)
        groups = {
            "pyramid": (7 * nS, nimg * (sum(px[:-1]) + sum(px[1:])) / 7.0, 0),     # read level l-1, write level l
            "fast": (nS, nimg * (sumP + 4 * 3.3 * nfeat), 0),                      # every level read once + packed candidates
            "gather": (nS, nimg * (8 * 3.3 * nfeat), 0),
            "ssc": (nS, nimg * (8 * 3.3 * nfeat + 4 * nk), 0),                      # candidates in, picks out
            "blur": (nS, nimg * (2 * sumP), 0),                                    # read + write every level
            "orient_desc": (nS, nimg * nk * (28 + 32 + 709 + 512), 0),             # keypoint + descriptor + disc + BRIEF taps
            "stereo_match": (nS, nk * (28 + 32) * 2 + nk * 16, 0),
            "stereo_finalize": (nS, nk * 24, 0),
            "track_predict": (nT, nk * (24 + 32 + 5) + Mact * (60 + 24 + 12), 0),
            "track_init_map": (nS, nk * (28 + 32 + 4 + 24 + 32 + 5), 0),
            "track_repredict": (nT, Mact * (24 + 60 + 12), 0),
            "imu_preintegrate": (nT, 10 * 56 + 8 * (289 + 225), 10 * 2 * 2 * 15 ** 3 + 15 ** 3),
            "proj_candidates": (nSolve, Mact * 60 + 2 * nk * 60 + Mact * 128, 0),
            "proj_resolve": (nSolve, Mact * (128 + 8 + 8) + 2 * nk * 8, 0),
            "pose_imu_lm": (nSolve, Mact * (24 + 8 + 4) + 2 * nk * 28 + 8 * 514, 0),
            "pose_lm": (nSolve, Mact * (24 + 8 + 4) + 2 * nk * 28, 0),
            "ba_linearize": (nLin, R_ * (8 + 16 + 8 + 96 + 24 + 160), 0),          # idx, uv, sigma, pose, point, stored J
            "ba_schur": (nTrial, R_ * 160 + L_ * 24, 2 * 36 * k2),                 # stored J read once (S stays in LDS)
            "ba_solve": (nTrial, 8 * (nBA6 * nBA6 + nBA6) * 2, nBA6 ** 3 / 3.0),   # reduced system in, delta out
            "ba_back": (nTrial, R_ * 160 + L_ * 48, 0),
            "ba_eval": (nTrial, R_ * (160 + 16 + 8 + 96 + 24), 0),
            "ba_chi2": (2 * nBA, R_ * (16 + 8 + 96 + 24), 0),
        }
        per_frame = {}
        for k, v in stage_ms.items():
            per_frame[k] = v / (nBA * KF_PERIOD) if k.startswith("ba_") else v / nS
        dom = max(per_frame, key=lambda k: per_frame[k])
        n_launch, alg_bytes, alg_flops = groups.get(dom, (nS, 0, 0))
        dom_ms = stage_ms[dom] / n_launch
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launch_ms": dom_ms, "algorithmic_bytes": alg_bytes,
                "launches_timed": n_launch,
                "note": "latency-bound kernel, one workgroup per problem instance (serial dependency chain of the reference algorithm): "
                        "the fraction is reported against the HBM roofline as the contract asks, the kernel is bound by "
                        "instruction latency, not by bytes or flops (DESIGN.md section 4)"}
        # HBM-side traffic of the same kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are
        # collected in separate runs of this bench, profiles/r01_h_pmc_summary.json; KB per launch).  Correction per
        # MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads -> doubled.
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_h_pmc_summary.json")))
            kname = {"proj_resolve": "k_proj_resolve", "pose_imu_lm": "k_pose_imu_lm", "ba_solve": "k_ba_solve_mfma64", "ssc": "k_ssc",
                     "ba_schur": "k_ba_schur<true>", "ba_back": "k_ba_back", "ba_eval": "k_ba_factors<1>",
                     "ba_linearize": "k_ba_factors<0>", "fast": "k_fast", "blur": "k_blur", "gather": "k_gather",
                     "stereo_match": "k_stereo_match", "orient_desc": "k_orient_desc"}.get(dom)
            if kname in pmc:
                roof["traffic"] = (2.0 * pmc[kname]["FETCH_SIZE_avg"] + pmc[kname]["WRITE_SIZE_avg"]) * 1024.0
                roof["traffic_source"] = "profiles/r01_h_pmc_summary.json (2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, per launch)"
        except Exception:      # noqa: BLE001
            pass
        if alg_flops:
            roof["achieved_gflops"] = alg_flops / (dom_ms * 1e-3) / 1e9
            roof["fp64_frac"] = roof["achieved_gflops"] / (FP64_PEAK_TFLOPS * 1e3)
        out = {
            "metric": "frames/sec (extract+match+localBA), 1500 feat stereo 752x480",
            "value": world * args.sessions * args.steps / el, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i32 (extract, match) + f64 (pose LM, local BA)", "data": "synthetic",
            "config": {"workload": "C2-class: EuRoC-like stereo+IMU (slamMode 0) 752x480, 1500 features/image, 200 Hz IMU "
                                   "pre-integration factor in every pose solve, %d rendered consecutive frames resident in "
                                   "HBM replayed cyclically, 10-KF local BA (3000 landmarks, ~%d residuals) every %d frames"
                                   % (args.frames, counters["ba_residuals"], KF_PERIOD),
                       "stages": ["extract L+R", "stereo match", "tracking loop (projection match + IMU pre-integration + 15-dof pose/velocity/bias LM)",
                                  "map refresh", "local BA (amortised)"],
                       "threads": ("frame-level pipeline: extraction of frame n+1 on one host thread / HIP stream while frame n is "
                                   "matched and tracked on another; " if pipelined else "extraction and tracking on one host thread; ") +
                                  "local BA on its own host thread / HIP stream (the reference's optimizer thread)",
                       "parallelism": "replicas x%d" % world + (", %d sessions per GPU" % args.sessions if args.sessions > 1 else "")},
            "stage_ms_per_step": {k: v for k, v in sorted(per_frame.items())},
            "stage_sampling": "HIP events on every %d-th frame / %d-th local BA; BA stages amortised over %d frames" % (STAGE_SAMPLE, BA_SAMPLE, KF_PERIOD),
            "mean_track_inliers": counters["track_inliers"] / max(counters["track_frames"], 1),
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, poses, imus, rig, nfeat, ba_prob, threaded=True)
            out["cpu_baseline_single_thread"] = cpu_baseline(frames, poses, imus, rig, nfeat, ba_prob, threaded=False)
        print(json.dumps(out))
    for sess in sessions:
        sess.ba_q.put(None)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
