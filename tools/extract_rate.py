"""Per-kernel device time of the extractor for a batch of images (HIP events, average over repeated runs).
usage: python tools/extract_rate.py [nimg] [repeats] [room|corridor]   (corridor: frames of the bench's default scene, rendered on the GPU)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gtsam-vslam_amd'))
import numpy as np, synth, vslam_capi as vc
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scene = sys.argv[3] if len(sys.argv) > 3 else "room"
if scene == "corridor":
    import torch
    Ls, Rs, _, _ = synth.corridor_sequence("euroc", 4, torch.device("cuda"), frame_step=40, first=100)
    frames = [(Ls[i].cpu().numpy(), Rs[i].cpu().numpy()) for i in range(4)]
else:
    frames = [synth.stereo_frame(2 * i, 'euroc') for i in range(4)]
fe = vc.Extractor(752, 480, 1500, batch=nimg)      # (after torch has initialised the GPU: the other order leaves torch without a device)
for i in range(nimg):
    fe.set_image(i, frames[(i // 2) % 4][i % 2])
fe.run(); fe.fetch(0)
tot = {}
for _ in range(rep):
    fe.run()
    fe.fetch(0)
    for k, v in fe.timings().items():
        tot[k] = tot.get(k, 0.0) + v
print({k: round(1e3 * v / rep, 1) for k, v in tot.items()}, "us per run of", nimg, scene, "images; total", round(1e3 * sum(tot.values()) / rep, 1))
