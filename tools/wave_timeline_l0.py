#!/usr/bin/env python3
"""Per-wave timeline of ONE level-0 blend launch (config 2) from the diagnostic build of experiments/wave_timeline.patch:
entry / prologue done / end stamps (s_memrealtime, 100 MHz), the wave's owner hint and its XCC.
    PANO_LIB=experiments/_build/libpano_trace.so python tools/wave_timeline_l0.py"""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
lib = C.CDLL(pano.LIB_PATH)
lib.pano_debug_l0_trace.argtypes = [C.c_void_p, C.c_uint, C.c_int]
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
for _ in range(200): step()
torch.cuda.synchronize()
res = []
for rep in range(3):
    lib.pano_debug_l0_trace(None, 0, 1)
    step(); torch.cuda.synchronize()
    buf = np.zeros((1 << 16, 4), dtype=np.uint64)
    n = lib.pano_debug_l0_trace(buf.ctypes.data, 1 << 16, 0)
    tr = buf[:n]; tr = tr[tr[:, 0] != 0]
    t = tr[:, :3].astype(np.int64); t0 = t[:, 0].min(); us = (t - t0) / 100.0
    hint = (tr[:, 3] & np.uint64(0xff)).astype(np.int64); xcc = ((tr[:, 3] >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
    life = us[:, 2] - us[:, 0]; end = us[:, 2].max()
    seam = hint == 15
    bins = np.arange(0, end + 1.0, 1.0)
    res.append({"waves": int(len(tr)), "seam_waves": int(seam.sum()), "launch_us": round(float(end), 2),
                "start_us_percentiles": {p: round(float(np.percentile(us[:, 0], p)), 2) for p in (1, 10, 25, 50, 75, 90, 99, 100)},
                "prologue_us_mean": round(float((us[:, 1] - us[:, 0]).mean()), 3),
                "lifetime_us_single_owner": {"mean": round(float(life[~seam].mean()), 2), "p90": round(float(np.percentile(life[~seam], 90)), 2)},
                "lifetime_us_seam": {"mean": round(float(life[seam].mean()), 2), "p90": round(float(np.percentile(life[seam], 90)), 2), "last_end_us": round(float(us[seam, 2].max()), 2)} if seam.any() else None,
                "waves_resident_every_1us": [int(((us[:, 0] <= b) & (us[:, 2] > b)).sum()) for b in bins],
                "waves_per_xcc": np.bincount(xcc, minlength=8).tolist(),
                "seam_waves_per_xcc": np.bincount(xcc[seam], minlength=8).tolist(),
                "last_wave_end_us_per_xcc": [round(float(us[xcc == k, 2].max()), 2) if (xcc == k).any() else None for k in range(8)]})
print(json.dumps({"kernel": "blend_level_ordered_kernel<true,3>, config 2, one launch at a time, instrumented build", "launches": res}, indent=1))
