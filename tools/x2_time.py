#!/usr/bin/env python3
"""us per config-2 frame with TWO frame sets per launch sequence (pano_compose_pair_x2) and S such sequences in flight, against F
single frames in flight.   python3 tools/x2_time.py [steps]   env: SEQ=2 (sequences in flight, x2 mode), F=4 (plain mode)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
g = c2_group()
ctxs = []
for k in range(2):
    ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
    for i in range(4): ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi(); ctxs.append(ctx)
nsets = 4
sets = [[[torch.from_numpy(synth_frame(g["w"], g["h"], 42 + 100 * s + 4 * k + i)).cuda() for i in range(4)] for k in range(2)] for s in range(nsets)]
ptr = [[[t.data_ptr() for t in fr] for fr in s] for s in sets]
ow, oh = ctxs[0].output_size()
strides = [g["w"] * 3] * 4
NS = int(os.environ.get("SLOTS", "4"))   # 8 needs a build with PANO_MAX_FRAME_SLOTS=8 (PANO_LIB)
for c in ctxs: c.set_frame_slots(NS)
fs, distinct = ctxs[0].frame_streams(4)
outs = [[torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)] for _ in range(NS)]
out = {"distinct_hw_queues": distinct}
def run(step, label):
    for k in range(300): step(k)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(steps): step(k)
        torch.cuda.synchronize()
        res.append(round((time.perf_counter() - t0) / steps * 1e6, 2))
    out[label] = res
for F in (4, 3):
    def step_plain(k, F=F):
        f = k % F
        ctxs[0].select_frame_slot(f); ctxs[1].select_frame_slot(f)
        p = ptr[k % nsets]
        ctxs[0].compose_pair(ctxs[1], p[0], strides, outs[f][0].data_ptr(), ow * 3, p[1], strides, outs[f][1].data_ptr(), ow * 3, fs[f])
    run(step_plain, "plain_F%d_us_per_frame" % F)
for SEQ in ((4, 3, 2, 1) if NS >= 8 else (2, 1)):
    def step_x2(k, SEQ=SEQ):      # one call = two frames: slots (2q, 2q + 1) on stream q
        q = k % SEQ
        a, b = ptr[(2 * k) % nsets], ptr[(2 * k + 1) % nsets]
        ctxs[0].compose_pair_x2(ctxs[1], a[0], a[1], outs[2 * q][0].data_ptr(), outs[2 * q][1].data_ptr(), b[0], b[1], outs[2 * q + 1][0].data_ptr(),
                                outs[2 * q + 1][1].data_ptr(), strides, strides, ow * 3, ow * 3, 2 * q, 2 * q + 1, fs[q])
    for k in range(300): step_x2(k)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(steps // 2): step_x2(k)
        torch.cuda.synchronize()
        res.append(round((time.perf_counter() - t0) / (steps // 2 * 2) * 1e6, 2))
    out["x2_SEQ%d_us_per_frame" % SEQ] = res
print(json.dumps(out))
