#!/usr/bin/env python3
"""Diagnostic: two independent compose_pair pipelines (2 x 2 stitchers) on one stream vs on two streams."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
g = c2_group()
def make_pair(seed):
    ctxs = []
    for k in range(2):
        ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
        for i in range(4):
            ctx.set_camera(i, g["K"][i], g["R"][i])
        ctx.prepare(); ctx.build_masks_voronoi(); ctxs.append(ctx)
    frames = [[torch.from_numpy(synth_frame(g["w"], g["h"], seed + 4 * k + i)).cuda() for i in range(4)] for k in range(2)]
    ow, oh = ctxs[0].output_size()
    outs = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    return ctxs, frames, outs, ow
NP = int(os.environ.get("PANO_PIPES", "4"))
pairs = [make_pair(42 + 57 * i) for i in range(NP)]
def step(p, st):
    ctxs, frames, outs, ow = p
    ctxs[0].compose_pair(ctxs[1], [t.data_ptr() for t in frames[0]], [g["w"] * 3] * 4, outs[0].data_ptr(), ow * 3,
                         [t.data_ptr() for t in frames[1]], [g["w"] * 3] * 4, outs[1].data_ptr(), ow * 3, st)
s0 = torch.cuda.current_stream().cuda_stream
streams = [torch.cuda.Stream() for _ in range(NP)]
res = {}
for ns in [1] + list(range(2, NP + 1)) + [1]:
    sl = [s0] * NP if ns == 1 else [streams[i % ns].cuda_stream for i in range(NP)]
    use = NP if ns == 1 else ns  # ns pipelines, each on its own stream
    for _ in range(20):
        for i in range(use): step(pairs[i], sl[i])
    torch.cuda.synchronize()
    N = 200
    t0 = time.perf_counter()
    for _ in range(N):
        for i in range(use): step(pairs[i], sl[i])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (N * use)
    res["streams_%d%s" % (ns, "" if ns > 1 or "streams_1" not in res else "_again")] = {"us_per_pair": round(dt * 1e6, 1), "c2_pano_per_s": round(1 / dt, 1)}
print(json.dumps(res))
