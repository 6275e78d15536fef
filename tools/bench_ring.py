#!/usr/bin/env python3
"""Side measurement: the 8 x 1080p rig as ONE 360-degree ring in one context instead of the reference's
2 groups x 4.  FIRST=180 puts a camera on the +-pi seam (its tile is the full width, live at both ends);
PANO_FULL_TILES=1 shows what the dead-span skipping is worth."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import ry, synth_frame
pano = importlib.import_module("img-stitching_amd")
w, h, f = 1920, 1080, 1002.416
first = float(os.environ.get("FIRST", "180"))
F = int(os.environ.get("F", "4"))
K = [f, 0.0, w / 2.0, 0.0, f, h / 2.0, 0.0, 0.0, 1.0]
ctx = pano.Context(8, w, h, scale=f, num_bands=5, device=0)
for i in range(8):
    ctx.set_camera(i, K, ry(first - 45.0 * i))
ctx.prepare(); ctx.build_masks_voronoi()
ctx.set_frame_slots(F)
frames = [torch.from_numpy(synth_frame(w, h, 900 + i)).cuda() for i in range(8)]
ow, oh = ctx.output_size()
outs = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(F)]
streams = [torch.cuda.Stream() for _ in range(F)]
fp = [t.data_ptr() for t in frames]
def step(k):
    s = k % F
    ctx.select_frame_slot(s)
    ctx.compose(fp, [w * 3] * 8, outs[s].data_ptr(), ow * 3, streams[s].cuda_stream)
for k in range(12): step(k)
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for k in range(N): step(k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
st = ctx.stage_stats() if hasattr(ctx, "stage_stats") else None
print(json.dumps({"config": "ring: 8 x 1920x1080 in one context, first yaw %g, 5 bands, pano %dx%d" % (first, ow, oh),
                  "full_tiles": os.environ.get("PANO_FULL_TILES", "0"), "frames_in_flight": F,
                  "roi_widths": [ctx.roi(i)[2] for i in range(8)], "gaps_level0": [ctx.live_gap(i, 0) for i in range(8)],
                  "ms_per_pano": round(dt * 1e3, 4), "panoramas_per_s": round(1 / dt, 1)}))
