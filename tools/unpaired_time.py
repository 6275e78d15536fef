#!/usr/bin/env python3
"""The cost of the per-launch overhead, from the other side: config 2 with the two stitchers composed by SEPARATE launch sequences
(pano_compose twice per frame, 18 launches of half the work) against pano_compose_pair (9 launches), four frames in flight."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
g = c2_group(); F = 4
ctxs = []
for k in range(2):
    ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
    for i in range(4): ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi(); ctxs.append(ctx)
fr = [[torch.from_numpy(synth_frame(g["w"], g["h"], 42 + 4 * k + i)).cuda() for i in range(4)] for k in range(2)]
ptr = [[t.data_ptr() for t in f] for f in fr]
ow, oh = ctxs[0].output_size()
for c in ctxs: c.set_frame_slots(F)
fs, _ = ctxs[0].frame_streams(F)
outs = [[torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in range(F)]
st = [g["w"] * 3] * 4
def step(k, paired):
    f = k % F
    ctxs[0].select_frame_slot(f); ctxs[1].select_frame_slot(f)
    if paired:
        ctxs[0].compose_pair(ctxs[1], ptr[0], st, outs[f][0].data_ptr(), ow * 3, ptr[1], st, outs[f][1].data_ptr(), ow * 3, fs[f])
    else:
        for q in range(2): ctxs[q].compose(ptr[q], st, outs[f][q].data_ptr(), ow * 3, fs[f])
res = {}
for paired in (True, False, True, False):
    for k in range(300): step(k, paired)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(2000): step(k, paired)
    torch.cuda.synchronize()
    res.setdefault("paired" if paired else "separate", []).append(round((time.perf_counter() - t0) / 2000 * 1e6, 2))
print(json.dumps(res))
