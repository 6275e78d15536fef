#!/usr/bin/env python3
"""Side measurement: BASELINE config 4 (4 x 3840x2160, cylindrical, exposure gain maps, 7 bands) on one GPU.
Frames beyond 2048 x 2048 use the projecting warp kernel (no remap table)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import c4_rig, synth_frame
pano = importlib.import_module("img-stitching_amd")
g = c4_rig()
F = int(os.environ.get("F", "4"))
gains = int(os.environ.get("GAINS", "1"))
ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], projector=1, num_bands=7, device=0)
for i in range(4):
    ctx.set_camera(i, g["K"][i], g["R"][i])
ctx.prepare(); ctx.build_masks_voronoi()
if gains:
    rng = np.random.default_rng(3)
    for i in range(4):
        r = ctx.roi(i)
        gw, gh = (r[2] + 31) // 32, (r[3] + 31) // 32
        ctx.set_gain_map(i, (0.9 + 0.2 * rng.random((gh, gw))).astype(np.float32))
ctx.set_frame_slots(F)
frames = [torch.from_numpy(synth_frame(g["w"], g["h"], 900 + i)).cuda() for i in range(4)]
ow, oh = ctx.output_size()
outs = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(F)]
streams = [torch.cuda.Stream() for _ in range(F)]
fp = [t.data_ptr() for t in frames]
def step(k):
    f = k % F
    ctx.select_frame_slot(f)
    ctx.compose(fp, [g["w"] * 3] * 4, outs[f].data_ptr(), ow * 3, streams[f].cuda_stream)
for k in range(12): step(k)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for k in range(N): step(k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(json.dumps({"config": "C4: 4 x 3840x2160 cylindrical, 7 bands, gains=%d, pano %dx%d" % (gains, ow, oh), "frames_in_flight": F,
                  "ms_per_pano": round(dt * 1e3, 4), "panoramas_per_s": round(1 / dt, 1)}))
