import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "gtsam-vslam_amd"))
import numpy as np, synth, vslam_capi as vc
vc.LIB_PATH = os.path.join(root, "tools/_stamp/libvslam_stamp.so")
L, R, _ = synth.stereo_frame(2)
ge = vc.Extractor(752, 480, 1500, batch=2)
ge.extract([L, R])
print("----", file=sys.stderr)
ge.extract([L, R])
print(ge.timings())
