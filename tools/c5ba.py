"""C5-class BA (64 KF, many landmarks) on one GPU: wall time, LM report, reprojection RMS."""
import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gtsam-vslam_amd"))
import numpy as np, synth, vslam_capi as vc
nlm = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nloc = int(sys.argv[2]) if len(sys.argv) > 2 else 62
t = time.perf_counter()
prob = synth.make_ba_problem_c5(n_lm=nlm, n_local=nloc, n_fixed=2)
print("problem built in %.1f s: pairs %d" % (time.perf_counter() - t, len(prob["pair_kf"])))
fe = vc.Extractor(752, 480, 1500)
rig = synth.RIGS["synthetic"]
for it in range(2):
    t = time.perf_counter()
    r = vc.local_ba(rig, fe.sigmaFactor, fe.InvSigmaFactor, prob)
    print("BA wall %.1f ms  residuals %d landmarks %d freeKF %d reports %s" % (1e3 * (time.perf_counter() - t), r["residuals"], r["landmarks"], r["free_kf"], r["reports"]))
print({k: round(v, 2) for k, v in vc.local_ba_timings().items()})
