"""Host-side stamps of one local BA (library built with -DVSLAM_HOST_STAMPS: tools/stampbuild.sh)."""
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gtsam-vslam_amd"))
import numpy as np, synth, vslam_capi as vc
vc.LIB_PATH = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/_stamp/libvslam_stamp.so")
c5 = len(sys.argv) > 1 and sys.argv[1] == "c5"
rig = synth.RIGS["synthetic" if c5 else "euroc"]
ba = synth.make_ba_problem_c5() if c5 else synth.make_ba_problem("euroc", 10, 4, 3000)
fe = vc.Extractor(752, 480, 1500)
vc.local_ba_set_timing(False)
for it in range(4):
    sys.stderr.write("---- call %d\n" % it)
    vc.local_ba(rig, fe.sigmaFactor, fe.InvSigmaFactor, ba)
