"""Local BA alone: wall time per call with event timing off, then the per-stage device times of one call."""
import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gtsam-vslam_amd"))
import numpy as np, synth, vslam_capi as vc
rig = synth.RIGS["euroc"]
ba = synth.make_ba_problem("euroc", 10, 4, 3000)
fe = vc.Extractor(752, 480, 1500)
vc.local_ba_set_timing(False)
for it in range(3):
    vc.local_ba(rig, fe.sigmaFactor, fe.InvSigmaFactor, ba)
t = time.perf_counter()
for it in range(10):
    r = vc.local_ba(rig, fe.sigmaFactor, fe.InvSigmaFactor, ba)
print("wall (timing off) %.3f ms / call" % (1e2 * (time.perf_counter() - t)))
vc.local_ba_set_timing(True)
t = time.perf_counter(); r = vc.local_ba(rig, fe.sigmaFactor, fe.InvSigmaFactor, ba); el = time.perf_counter() - t
tm = vc.local_ba_timings()
print("wall %.2f ms  device %.2f ms  trials %d iters %d" % (1e3 * el, sum(tm.values()), r["reports"][0]["inner"] + r["reports"][1]["inner"],
      r["reports"][0]["iterations"] + r["reports"][1]["iterations"]), {k: round(v, 3) for k, v in tm.items()})
