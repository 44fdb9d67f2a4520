"""A cohort of tracker-window local BAs alone on the GPU (vslam_local_ba_batch): wall time per cohort and device time per stage.
usage: bacohort.py [lanes ...]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gtsam-vslam_amd"))
import numpy as np, synth, vslam_capi as vc
rig = synth.RIGS["euroc"]
fe = vc.Extractor(752, 480, 1500)
sizes = [int(v) for v in sys.argv[1:]] or [1, 10, 20, 40]
allp = [synth.make_ba_problem(n_local=10, n_fixed=4, n_lm=1000 + 15 * (s % 11), seed=100 + s) for s in range(max(sizes))]
for n in sizes:
    probs = allp[:n]
    vc.local_ba_set_timing(False)
    for it in range(2):
        r = vc.local_ba_batch(rig, fe.sigmaFactor, fe.InvSigmaFactor, probs)
    t = time.perf_counter()
    for it in range(5):
        r = vc.local_ba_batch(rig, fe.sigmaFactor, fe.InvSigmaFactor, probs)
    wall = (time.perf_counter() - t) / 5
    vc.local_ba_set_timing(True)
    r = vc.local_ba_batch(rig, fe.sigmaFactor, fe.InvSigmaFactor, probs)
    tm = vc.local_ba_timings()
    rounds = [q["rounds"] for q in r]
    it = [q["reports"][0]["iterations"] + q["reports"][1]["iterations"] for q in r]
    print("%3d lanes: wall %.2f ms (incl. the ctypes marshalling of the problems) | device %.2f ms | residual blocks %d, landmarks %d | rounds per lane %.1f (max %d), "
          "iterations %.1f | stages ms: %s" % (n, 1e3 * wall, sum(v for k, v in tm.items() if not k.endswith("#n")), r[0]["residuals"], r[0]["landmarks"], np.mean(rounds), max(rounds), np.mean(it),
                                               {k: round(v, 3) for k, v in tm.items()}))
