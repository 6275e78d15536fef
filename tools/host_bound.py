#!/usr/bin/env python3
"""Diagnostic: is the frames-in-flight loop bound by the host (enqueue time) or by the GPU?"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
g = c2_group()
F = int(os.environ.get("F", "4"))
ctxs = []
for k in range(2):
    ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
    for i in range(4):
        ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi(); ctx.set_frame_slots(F); ctxs.append(ctx)
frames = [[torch.from_numpy(synth_frame(g["w"], g["h"], 42 + 4 * k + i)).cuda() for i in range(4)] for k in range(2)]
ow, oh = ctxs[0].output_size()
outs = [[torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in range(F)]
streams = [torch.cuda.Stream() for _ in range(F)]
fp = [[t.data_ptr() for t in fr] for fr in frames]
st = [g["w"] * 3] * 4
def step(k):
    f = k % F
    ctxs[0].select_frame_slot(f); ctxs[1].select_frame_slot(f)
    ctxs[0].compose_pair(ctxs[1], fp[0], st, outs[f][0].data_ptr(), ow * 3, fp[1], st, outs[f][1].data_ptr(), ow * 3, streams[f].cuda_stream)
for k in range(40): step(k)
torch.cuda.synchronize()
N = 400
t0 = time.perf_counter()
for k in range(N): step(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps({"F": F, "enqueue_us_per_step": round((t1 - t0) / N * 1e6, 1), "total_us_per_step": round((t2 - t0) / N * 1e6, 1),
                  "drain_us": round((t2 - t1) * 1e6, 1)}))
# pure host cost: a short burst into empty queues
best = 1e9
for rep in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(8): step(k)
    best = min(best, (time.perf_counter() - t0) / 8)
torch.cuda.synchronize()
print(json.dumps({"host_us_per_step_burst_of_8": round(best * 1e6, 1)}))
