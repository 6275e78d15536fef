#!/usr/bin/env python3
"""How long does the HOST need to enqueue a config-2 frame (select slot + pano_compose_pair through ctypes: 9 launches)?  If that is
close to what the GPU needs per frame the timed loop is host-bound.  Prints us per step: loop return (all enqueued) and synchronize."""
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
def step(k):
    f = k % F
    ctxs[0].select_frame_slot(f); ctxs[1].select_frame_slot(f)
    ctxs[0].compose_pair(ctxs[1], ptr[0], st, outs[f][0].data_ptr(), ow * 3, ptr[1], st, outs[f][1].data_ptr(), ow * 3, fs[f])
for k in range(200): step(k)
torch.cuda.synchronize()
res = []
for n in (20, 20, 100, 100, 1000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n): step(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res.append({"steps": n, "enqueue_us_per_step": round((t1 - t0) / n * 1e6, 1), "done_us_per_step": round((t2 - t0) / n * 1e6, 1)})
print(json.dumps(res))
