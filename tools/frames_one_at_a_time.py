#!/usr/bin/env python3
"""110 config-2 frames (8 x 1080p, both stitchers per launch sequence: pano_compose_pair) composed one at a time - the
workload the rocprofv3 --pmc passes of tools/pmc_*.sh run over (10 warm-up + 100 counted launches of every kernel)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
g = c2_group()
ctxs = []
for k in range(2):
    ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
    for i in range(4):
        ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi(); ctxs.append(ctx)
frames = [[torch.from_numpy(synth_frame(g["w"], g["h"], 42 + 4 * k + i)).cuda() for i in range(4)] for k in range(2)]
ow, oh = ctxs[0].output_size()
outs = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
st = torch.cuda.current_stream().cuda_stream
ctxs[0].set_profiling(True)
def step():
    ctxs[0].compose_pair(ctxs[1], [t.data_ptr() for t in frames[0]], [g["w"] * 3] * 4, outs[0].data_ptr(), ow * 3,
                         [t.data_ptr() for t in frames[1]], [g["w"] * 3] * 4, outs[1].data_ptr(), ow * 3, st)
for _ in range(10): step()
ctxs[0].stage_stats(True)
for _ in range(100): step()
torch.cuda.synchronize()
ms, n = ctxs[0].stage_stats(True)
print(json.dumps({"warp8_us": round(ms[0] / n[0] * 1e3, 2), "pyr_us": round(ms[1] / n[1] * 1e3, 2), "blend_us": round(ms[2] / n[2] * 1e3, 2),
                  "blend_level0_us": round(ms[3] / max(n[3], 1) * 1e3, 2)}))
