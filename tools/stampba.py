import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "gtsam-vslam_amd"))
import numpy as np, synth, vslam_capi as vc
vc.LIB_PATH = os.path.join(root, "tools/_stamp/libvslam_stamp.so")
rig = synth.RIGS["euroc"]
ba = synth.make_ba_problem("euroc", 10, 4, 3000)
fe = vc.Extractor(752, 480, 1500)
vc.local_ba(rig, fe.sigmaFactor, fe.InvSigmaFactor, ba)
print("---- second call", file=sys.stderr)
vc.local_ba(rig, fe.sigmaFactor, fe.InvSigmaFactor, ba)
print("---- third call", file=sys.stderr)
vc.local_ba(rig, fe.sigmaFactor, fe.InvSigmaFactor, ba)
