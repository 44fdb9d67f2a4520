"""Per-kernel device time of the extractor for a batch of images (HIP events, average over repeated runs).
usage: python tools/extract_rate.py [nimg] [repeats]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gtsam-vslam_amd'))
import numpy as np, synth, vslam_capi as vc
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fe = vc.Extractor(752, 480, 1500, batch=nimg)
frames = [synth.stereo_frame(2 * i, 'euroc') for i in range(4)]
for i in range(nimg):
    fe.set_image(i, frames[(i // 2) % 4][i % 2])
fe.run(); fe.fetch(0)
tot = {}
for _ in range(rep):
    fe.run()
    fe.fetch(0)
    for k, v in fe.timings().items():
        tot[k] = tot.get(k, 0.0) + v
print({k: round(1e3 * v / rep, 1) for k, v in tot.items()}, "us per run of", nimg, "images; total", round(1e3 * sum(tot.values()) / rep, 1))
