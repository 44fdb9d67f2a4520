"""N > 1 path of the landmark-sharded bundle adjustment.

CPU (gloo, world_size 2): the decomposition the multi-GPU path relies on — the partial reduced camera
systems [S | rhs | cost] of the landmark shards, all-reduced (sum), equal the single-rank system — checked
with the oracle's shard function over a real torch.distributed all-reduce.
GPU: the sharded HIP path with the in-process `local` transport (2 and 3 ranks = host threads sharing
the one GPU of the test box), with a 1-rank RCCL communicator, and ACROSS PROCESSES (two spawned ranks sharing the GPU, the
reduced camera systems all-reduced by torch.distributed / gloo through the caller-supplied transport) must reproduce the
single-GPU result."""
import os
import sys
import threading
import numpy as np
import pytest
import synth
# One RCCL per process: PyTorch-ROCm brings its own librccl; if this library's communicator (dlopen of librccl.so.1) comes first and
# torch is imported afterwards, the process ends up with two copies whose exit handlers collide (glibc: "double free or corruption" at
# interpreter exit).  Importing torch first makes the loader hand the already-loaded copy to both - bench.py does the same.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gloo_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "gtsam-vslam_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import pyoracle as po
    import synth as sy
    import vslam_capi as vc
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ex = po.Extractor(1500)
    prob = sy.make_ba_problem(n_local=5, n_fixed=2, n_lm=300, seed=4)
    part, F = po.ba_reduced_system_shard(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob, rank, world)
    t = torch.from_numpy(part.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)          # the exchange step of the sharded BA
    full, F1 = po.ba_reduced_system_shard(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob, 0, 1)
    owned = sum(1 for l in range(len(prob["lm"])) if vc.landmark_owner(l, world) == rank)
    cnt = torch.tensor([owned], dtype=torch.int64)
    dist.all_reduce(cnt)
    ok = bool(F == F1 and np.abs(t.numpy() - full).max() <= 1e-12 * np.abs(full).max() and int(cnt.item()) == len(prob["lm"]))
    # the 128-byte id broadcast used to bootstrap RCCL (payload is opaque to the transport)
    obj = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(obj, src=0)
    ok = ok and obj[0] == bytes(range(128))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_shard_decomposition_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ba_local_transport_matches_single_gpu(oracle, capi, world):
    ex = oracle.Extractor(1500)
    prob = synth.make_ba_problem(n_local=10, n_fixed=4, n_lm=1500, seed=31)
    single = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    comms = capi.comm_create_local(world)
    out = [None] * world

    def run(r):
        out[r] = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob, comm=comms[r])

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    for r in range(world):
        g = out[r]
        assert g is not None
        for s in range(2):
            assert (g["reports"][s]["iterations"], g["reports"][s]["inner"]) == \
                   (single["reports"][s]["iterations"], single["reports"][s]["inner"])
            assert abs(g["reports"][s]["finalError"] - single["reports"][s]["finalError"]) <= 1e-8 * single["reports"][s]["finalError"]
        assert np.abs(g["kf_pose"] - single["kf_pose"]).max() < 1e-8       # S is summed in a different order
        assert np.array_equal(g["pair_wrong"], single["pair_wrong"])
        assert np.median(np.linalg.norm(g["lm"] - single["lm"], axis=1)) < 1e-6
        assert (g["residuals"], g["landmarks"], g["sum_k2"]) == (single["residuals"], single["landmarks"], single["sum_k2"])
        # every rank returns the full, identical result
        assert np.array_equal(g["kf_pose"], out[0]["kf_pose"]) and np.array_equal(g["lm"], out[0]["lm"])
    for c in comms:
        c.close()


@pytest.mark.gpu
def test_sharded_ba_rccl_single_rank(oracle, capi):
    """World-size-1 RCCL communicator: exercises the RCCL bootstrap + all-reduce plumbing on one GPU."""
    ex = oracle.Extractor(1500)
    prob = synth.make_ba_problem(n_local=6, n_fixed=2, n_lm=400, seed=12)
    single = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob)
    comm = capi.comm_create_rccl(0, 1, 0, lambda b: b)
    got = capi.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob, comm=comm)
    assert np.abs(got["kf_pose"] - single["kf_pose"]).max() < 1e-12
    comm.close()


def _xproc_worker(rank, world, port, q, window):
    sys.path.insert(0, os.path.join(ROOT, "gtsam-vslam_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import pyoracle as po
    import synth as sy
    import vslam_capi as vc
    try:
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        ex = po.Extractor(1500)
        if window == "c5":      # the C5 window shape at a size the test finishes in seconds: 62 free keyframes, windowed Schur + block-column Cholesky
            prob = sy.make_ba_problem_c5(n_lm=4000)
        else:
            prob = sy.make_ba_problem(n_local=10, n_fixed=4, n_lm=1500, seed=31)

        def allreduce(a):
            t = torch.from_numpy(a)                      # (shares the buffer: summed in place)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        comm = vc.comm_create_callback(rank, world, 0, allreduce)
        got = vc.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob, comm=comm)
        comm.close()
        single = vc.local_ba(prob["rig"], ex.sigmaFactor, ex.InvSigmaFactor, prob) if rank == 0 else None
        q.put((rank, "ok", got["kf_pose"], got["pair_wrong"], [(r["iterations"], r["inner"], r["finalError"]) for r in got["reports"]],
               None if single is None else (single["kf_pose"], single["pair_wrong"], [(r["iterations"], r["inner"], r["finalError"]) for r in single["reports"]])))
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        q.put((rank, "error: %s: %s" % (type(e).__name__, e), None, None, None, None))


@pytest.mark.gpu
@pytest.mark.parametrize("window", ["tracker", "c5"])
def test_sharded_ba_across_processes_gloo_callback_transport(window):
    """Two PROCESSES, one GPU: each rank runs the HIP landmark-sharded BA (lm % 2 == rank) and the partial reduced camera systems of
    every trial round travel through a real process-group all-reduce (gloo, host-staged by vslam_comm_create_callback) - the
    cross-process execution of the path RCCL serves on a multi-GPU node (RCCL refuses two ranks on one device).  Both ranks must
    return the identical result, equal to the single-GPU one."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_xproc_worker, args=(r, 2, port, q, window)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda v: v[0])
    for p in procs:
        p.join(timeout=120)
    assert [v[1] for v in res] == ["ok", "ok"], [v[1] for v in res]
    (_, _, pose0, wrong0, rep0, single), (_, _, pose1, wrong1, rep1, _) = res
    assert np.array_equal(pose0, pose1) and np.array_equal(wrong0, wrong1)          # every rank returns the identical result
    spose, swrong, srep = single
    assert [(a, b) for a, b, _ in rep0] == [(a, b) for a, b, _ in srep]
    for (_, _, e), (_, _, es) in zip(rep0, srep):
        assert abs(e - es) <= 1e-8 * max(es, 1.0)
    assert np.abs(pose0 - spose).max() < 1e-8 and np.array_equal(wrong0, swrong)
