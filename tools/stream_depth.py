#!/usr/bin/env python3
"""panoramas/s of the streaming entry (pano_stream_*: H2D of the 8 frames, compose, D2H of the panorama) with D panoramas in flight.
The product has PANO_STREAM_SLOTS = 2; a build with more slots (PANO_LIB) takes D up to its count.
    python3 tools/stream_depth.py [D] [steps]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
g = c2_group()
ctxs = []
for k in range(2):
    ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
    for i in range(4):
        ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi(); ctxs.append(ctx)
for k in range(2):
    for s in range(D):
        for i in range(4):
            ctxs[k].stream_input(s, i)[:] = synth_frame(g["w"], g["h"], 42 + 4 * k + i)
def run(n):
    t0 = time.perf_counter()
    for k in range(n + D - 1):
        if k < n:
            for c in ctxs: c.stream_submit(k % D)
        if k >= D - 1:
            for c in ctxs: c.stream_wait((k - (D - 1)) % D)
    return n / (time.perf_counter() - t0)
run(60)
rates = [round(run(steps), 1) for _ in range(3)]
print(json.dumps({"lib": os.path.basename(os.environ.get("PANO_LIB", "product")), "panoramas_in_flight": D, "panoramas_per_s": rates}))
