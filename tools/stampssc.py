"""Per-task cycle stamps of the suppression kernel (tools/stampbuild.sh -DVSLAM_SSC_STAMPS) on one stereo pair: room scene
(default) or `corridor` (the bench's default scene)."""
import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "gtsam-vslam_amd"))
import numpy as np, synth, vslam_capi as vc
vc.LIB_PATH = os.path.join(root, "tools/_stamp/libvslam_stamp.so")
if len(sys.argv) > 1 and sys.argv[1] == "corridor":
    import torch
    Ls, Rs, _, _ = synth.corridor_sequence("euroc", 1, torch.device("cuda"), first=200)
    L, R = Ls[0].cpu().numpy(), Rs[0].cpu().numpy()
else:
    L, R, _ = synth.stereo_frame(2)
ge = vc.Extractor(752, 480, 1500, batch=2)
ge.extract([L, R])
print("----", file=sys.stderr)
ge.extract([L, R])
print(ge.timings())
