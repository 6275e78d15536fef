#!/usr/bin/env python3
"""DIAGNOSTIC (round 5): per-wave phase times of the streaming K1 experiment (experiments/k1_streaming_persistent.patch + stamps).
PANO_LIB=experiments/_build/libpano_streamdiag.so PANO_K1_STREAM_WGS=7 python tools/stream_timeline.py"""
import ctypes as C, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
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
def step():
    ctxs[0].compose_pair(ctxs[1], [t.data_ptr() for t in frames[0]], [g["w"] * 3] * 4, outs[0].data_ptr(), ow * 3,
                         [t.data_ptr() for t in frames[1]], [g["w"] * 3] * 4, outs[1].data_ptr(), ow * 3, st)
for _ in range(10): step()
torch.cuda.synchronize()
nw = 8 * 256 * 4
dbg = torch.zeros(nw * 64, dtype=torch.int64, device="cuda")
lib = pano.load_library()
lib.pano_debug_set_k1_dbg.argtypes = [C.c_void_p]
assert lib.pano_debug_set_k1_dbg(C.c_void_p(dbg.data_ptr())) == 0
step(); torch.cuda.synchronize()
lib.pano_debug_set_k1_dbg(C.c_void_p(0))
d = dbg.cpu().numpy().reshape(nw, 8, 8)
n_it = d[:, 7, 7]
used = n_it > 0
t0 = d[used, 0, 0].min()
ph = {"describe next (scalar)": [], "wait + barrier": [], "issue next copies": [], "compute + stores issued": [], "not beside: barrier, describe, issue": [], "period": []}
beside = []
for w in np.nonzero(used)[0]:
    for j in range(min(int(n_it[w]), 8)):
        s = d[w, j]
        if s[4] == 0: continue
        ph["describe next (scalar)"].append(s[1] - s[0]); ph["wait + barrier"].append(s[2] - s[1]); ph["issue next copies"].append(s[3] - s[2])
        ph["compute + stores issued"].append(s[4] - s[3])
        if s[5]: ph["not beside: barrier, describe, issue"].append(s[5] - s[4])
        if j + 1 < int(n_it[w]) and j + 1 < 8 and d[w, j + 1, 0]: ph["period"].append(d[w, j + 1, 0] - s[0])
        beside.append(int(s[6] & 1))
out = {"waves": int(used.sum()), "patches_per_wave_mean": float(n_it[used].mean()), "beside_fraction": float(np.mean(beside)),
       "phase_us_mean": {k: round(float(np.mean(v)) / 100.0, 3) for k, v in ph.items() if v},
       "phase_us_p90": {k: round(float(np.percentile(v, 90)) / 100.0, 3) for k, v in ph.items() if v},
       "first_start_to_last_end_us": round(float(d[used][:, :, :6].max() - t0) / 100.0, 2)}
print(json.dumps(out, indent=1))
