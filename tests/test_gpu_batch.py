"""vslam_batch (B sequences in lockstep, one launch per stage for all lanes) against vslam_system (one sequence, its own
launches) on the same inputs: every lane must reproduce the single-session run of its sequence - same keyframe decisions,
inlier counts, match tables, map sizes, local-BA statistics, poses.  The batched kernels are the one-session kernels'
bodies fed per-lane argument tables, so the only tolerated difference is the summation order of the local BA's LDS
atomics (poses compared to 1e-9).  Lanes start at different source frames, one lane joins late (idle lane + a map
initialisation in the middle of a run), and the schedule makes keyframes / local BAs fall on different steps per lane."""
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu

G = (0.0, 9.81, 0.0)
NOISE = (1.6968e-4, 1.9393e-5, 2.0e-3, 3.0e-3)


def _velocity(f, fps):
    h = 1e-4
    return (synth.pose_at(f + h * fps, fps)[:3, 3] - synth.pose_at(f - h * fps, fps)[:3, 3]) / (2 * h)


def _bucket(f0, f1, fps):
    S, dts, _ = synth.imu_samples(f0, f1, fps, noise_seed=0x1A00 + f1)
    return (S[:, :3], S[:, 3:], np.arange(len(dts)) * 5e6)


def _single(capi, rig_name, nfeat, frames, use_imu, mapping, delay=0, np_delay=1):
    rig = synth.RIGS[rig_name]
    imu = dict(gravity=G, noise=NOISE, T_bs=synth.T_BC1, hz=200, velocity=_velocity(frames[0], rig["fps"])) if use_imu else None
    s = capi.System(rig, nfeat, T0=synth.pose_at(frames[0], rig["fps"]), imu=imu, local_mapping=mapping, mapping_delay=delay, mapping_np_delay=np_delay)
    out = []
    for n, f in enumerate(frames):
        L, R, _ = synth.stereo_frame(f, rig_name)
        b = _bucket(frames[n - 1], f, rig["fps"]) if (use_imu and n > 0) else None
        P, rep = s.track(L, R, n, imu_bucket=b)
        out.append((P, rep, s.last_frame() if n > 0 else None))
    res = (out, s.counts(), s.keyframes())
    s.close()
    return res


def _batched(capi, rig_name, nfeat, schedules, starts, use_imu, mapping, delay=0, np_delay=1):
    """schedules[b]: source frames of lane b; starts[b]: the step at which lane b begins"""
    rig = synth.RIGS[rig_name]
    B = len(schedules)
    imu = dict(gravity=G, noise=NOISE, T_bs=synth.T_BC1, hz=200) if use_imu else None
    bt = capi.Batch(rig, nfeat, B, T0s=[synth.pose_at(sc[0], rig["fps"]) for sc in schedules], imu=imu,
                    velocities=[_velocity(sc[0], rig["fps"]) for sc in schedules] if use_imu else None, local_mapping=mapping,
                    host_threads=3, mapping_delay=delay, mapping_np_delay=np_delay)
    nSteps = max(starts[b] + len(schedules[b]) for b in range(B))
    out = [[] for _ in range(B)]
    for step in range(nSteps):
        Ls, Rs, fn, bk, mask = [None] * B, [None] * B, [0] * B, [None] * B, [0] * B
        for b in range(B):
            n = step - starts[b]
            if n < 0 or n >= len(schedules[b]):
                continue
            f = schedules[b][n]
            Ls[b], Rs[b], _ = synth.stereo_frame(f, rig_name)
            fn[b] = n; mask[b] = 1
            if use_imu and n > 0:
                bk[b] = _bucket(schedules[b][n - 1], f, rig["fps"])
        T, reps = bt.track(Ls, Rs, fn, imu_buckets=bk if use_imu else None, mask=mask)
        for b in range(B):
            if mask[b]:
                out[b].append((T[b].copy(), reps[b], bt.system(b).last_frame() if fn[b] > 0 else None))
    res = [(out[b], bt.system(b).counts(), bt.system(b).keyframes()) for b in range(B)]
    bt.close()
    return res


INT_KEYS = ("keyframe_inserted", "n_active", "n_inliers", "n_stereo", "rounds", "lm_iterations", "n_keyframes", "n_map_points",
            "n_active_after", "mapping_ran", "new_points", "ba_keyframes", "ba_local", "ba_landmarks", "ba_pairs", "ba_wrong", "ba_outliers")


def _same(one, lane, tol=1e-9):
    (o1, c1, k1), (o2, c2, k2) = one, lane
    assert len(o1) == len(o2)
    nKF = nBA = 0
    for n, ((P1, r1, l1), (P2, r2, l2)) in enumerate(zip(o1, o2)):
        for k in INT_KEYS:
            assert r1[k] == r2[k], (n, k, r1[k], r2[k])
        assert np.abs(P1 - P2).max() <= tol, (n, np.abs(P1 - P2).max())
        if l1 is not None:
            assert np.array_equal(l1[0], l2[0]) and np.array_equal(l1[1], l2[1]), n
        nKF += r1["keyframe_inserted"]; nBA += r1["mapping_ran"]
        for s in range(2):
            assert r1["ba_report"][s]["iterations"] == r2["ba_report"][s]["iterations"], n
    assert c1 == c2
    assert list(k1[0]) == list(k2[0]) and np.abs(k1[1] - k2[1]).max() <= tol
    return nKF, nBA


@pytest.mark.parametrize("use_imu,mapping,delay,np_delay", [(False, 1, 0, 1), (True, 1, 0, 1), (False, 2, 3, 1), (True, 2, 4, 2), (False, 2, 1, 1)])
def test_batch_lanes_equal_single_sessions(capi, use_imu, mapping, delay, np_delay):
    """mapping 1: the pass inside the step; mapping 2: the lanes' passes run on the batch's mapping engine beside the steps, on
    the fixed schedule (new points at the next frame, write-back `delay` frames after the hand-over) - in both modes a lane
    equals the single session of its sequence."""
    schedules = [list(range(0, 60, 2)), list(range(6, 66, 2)), list(range(12, 64, 2))]
    starts = [0, 0, 3]
    lanes = _batched(capi, "euroc", 1500, schedules, starts, use_imu, mapping, delay, np_delay)
    tot = [0, 0]
    for b, sc in enumerate(schedules):
        nKF, nBA = _same(_single(capi, "euroc", 1500, sc, use_imu, mapping, delay, np_delay), lanes[b])
        tot[0] += nKF; tot[1] += nBA
    assert tot[0] >= 4 and tot[1] >= 2      # the comparison covered keyframe insertions and local BAs


def test_fleet_batched_with_prefetch_equals_one_thread_per_session(capi):
    """vslam_fleet on device-resident frames: the lockstep groups (whose driver prefetches the next frames' extraction into the
    extractor's second output set while a step's host phases run) must add up to the same run as one thread + one set of
    launches per session: same keyframes, inlier sums, lost frames; position error sums equal to round-off of the local BAs."""
    rig = synth.RIGS["euroc"]
    n = 24
    frames = [synth.stereo_frame(2 * i, "euroc") for i in range(n)]
    imgs = [(capi.DeviceImage(f[0]), capi.DeviceImage(f[1])) for f in frames]
    poses = np.stack([f[2] for f in frames])
    cfg = capi.system_config(rig, 1500, local_mapping=1)
    out = []
    for lanes in (0, 3, 5):            # 5 sessions: one group of 5; two groups (3 + 2); unbatched
        fl = capi.Fleet(cfg, 5, [a.ptr for a, _ in imgs], [b.ptr for _, b in imgs], rig["w"], True, poses=poses, lanes=lanes)
        r1 = fl.run(30)
        r2 = fl.run(17)                # a second job continues the sequences (and the prefetch chain)
        fl.close()
        out.append((r1, r2))
    for k in ("frames", "keyframes", "mappings", "new_points", "sum_inliers", "min_inliers", "lost_frames", "sum_rounds", "ba_landmarks", "ba_pairs"):
        for j in (1, 2):
            assert out[j][0][k] == out[0][0][k] and out[j][1][k] == out[0][1][k], (k, j, out[j][0][k], out[0][0][k])
    for j in (1, 2):
        assert abs(out[j][1]["sum_sq_position_error"] - out[0][1]["sum_sq_position_error"]) < 1e-9
    assert out[0][0]["keyframes"] >= 10 and out[0][1]["mappings"] + out[0][0]["mappings"] >= 3
    for a, b in imgs:
        a.free(); b.free()


def test_async_mapping_is_reproducible_and_complete(capi):
    """local_mapping = 2 on the fixed schedule (mapping_delay = 4, bench.py's setting): two runs of the same fleet are identical
    (counts equal, summed position error equal to the round-off of the local BA's LDS atomics: results do not depend on when the
    mapping threads finish), every keyframe
    after the third gets its pass unless one was still in flight when it was inserted, and the grouping (one group of 6, two
    of 3) does not change the result."""
    rig = synth.RIGS["euroc"]
    n = 30
    frames = [synth.stereo_frame(2 * i, "euroc") for i in range(n)]
    imgs = [(capi.DeviceImage(f[0]), capi.DeviceImage(f[1])) for f in frames]
    poses = np.stack([f[2] for f in frames])
    cfg = capi.system_config(rig, 1500, local_mapping=2, mapping_delay=4, mapping_np_delay=2)
    res = []
    for lanes in (3, 3, 6):
        fl = capi.Fleet(cfg, 6, [a.ptr for a, _ in imgs], [b.ptr for _, b in imgs], rig["w"], True, poses=poses, lanes=lanes)
        res.append(fl.run(90))
        fl.close()
    for k in ("frames", "keyframes", "mappings", "new_points", "sum_inliers", "min_inliers", "lost_frames", "sum_rounds", "ba_landmarks", "ba_pairs",
              "ba_residuals", "ba_trials"):
        assert res[1][k] == res[0][k], (k, res[1][k], res[0][k])
        assert res[2][k] == res[0][k], (k, res[2][k], res[0][k])
    assert abs(res[1]["sum_sq_position_error"] - res[0]["sum_sq_position_error"]) < 1e-9     # (round-off of the BA's LDS atomics only)
    assert abs(res[2]["sum_sq_position_error"] - res[0]["sum_sq_position_error"]) < 1e-9
    assert res[0]["mappings"] >= 3 and res[0]["lost_frames"] == 0
    assert res[0]["mappings"] >= res[0]["keyframes"] - 3 * 6 - 6 - 6       # (first three keyframes of a session; one pass pending at the end)
    for a, b in imgs:
        a.free(); b.free()


def test_thread_release_and_reuse(capi, oracle):
    """vslam_thread_release() frees the calling thread's cached device resources (scratch pool, local-BA workspace); the
    thread can keep using the library afterwards, and results do not change."""
    import threading
    prob = synth.make_ba_problem(n_local=5, n_fixed=2, n_lm=600, seed=3)
    ex = oracle.Extractor(1500)
    out = {}

    def work():
        L = capi.lib()
        L.vslam_thread_release.restype = None
        a = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
        d1 = capi.calc_descriptors([np.arange(96, dtype=np.uint8).reshape(3, 32)])
        L.vslam_thread_release()
        L.vslam_thread_release()          # (idempotent)
        b = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
        d2 = capi.calc_descriptors([np.arange(96, dtype=np.uint8).reshape(3, 32)])
        L.vslam_thread_release()
        out["ok"] = np.abs(a["kf_pose"] - b["kf_pose"]).max() < 1e-10 and list(d1) == list(d2) and a["reports"][0]["iterations"] == b["reports"][0]["iterations"]

    th = threading.Thread(target=work)
    th.start(); th.join(120)
    assert out.get("ok") is True
