#!/usr/bin/env python3
"""Per-wave timeline of ONE K1 launch (config 2, 8 cameras) from the diagnostic build of experiments/wave_timeline.patch: every wave
stamps s_memrealtime (100 MHz, one clock for all XCDs) at nine points; printed: when waves start (the dispatch ramp), how long each
phase of a wave takes, how many waves are resident over the launch.

    tools/build_variant.sh trace experiments/wave_timeline.patch
    PANO_LIB=experiments/_build/libpano_trace.so python tools/wave_timeline.py
"""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
lib = C.CDLL(pano.LIB_PATH)
lib.pano_debug_k1_trace.argtypes = [C.c_void_p, C.c_uint, C.c_int]
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
W = 10
res = []
for rep in range(3):
    lib.pano_debug_k1_trace(None, 0, 1)
    step(); torch.cuda.synchronize()
    buf = np.zeros((1 << 16, W), dtype=np.uint64)
    n = lib.pano_debug_k1_trace(buf.ctypes.data, 1 << 16, 0)
    tr = buf[:n]
    tr = tr[tr[:, 0] != 0]                      # slots of workgroups that left early were not written
    n = len(tr)
    lds = tr[:, 3] != 0                         # waves whose patch has an LDS box (stamps 3 and 4 exist)
    t = tr[:, :9].astype(np.int64)
    t[~lds, 3] = t[~lds, 2]; t[~lds, 4] = t[~lds, 2]
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0                       # 100 MHz -> us
    xcc = (tr[:, 9] >> np.uint64(32)).astype(np.int64) & 15
    names = ["hot scalar load", "box / live scalar loads", "copies issued + decode", "own copies landed", "barrier", "LDS taps read", "bilinear + pack", "stores done"]
    ph = np.diff(us, axis=1)
    end = us[:, 8].max()
    start_pct = {p: round(float(np.percentile(us[:, 0], p)), 2) for p in (1, 10, 25, 50, 75, 90, 99, 100)}
    life = us[:, 8] - us[:, 0]
    # waves resident over time (0.25 us bins)
    bins = np.arange(0, end + 0.25, 0.25)
    resident = [int(((us[:, 0] <= b) & (us[:, 8] > b)).sum()) for b in bins]
    res.append({"waves": int(n), "launch_us_first_start_to_last_end": round(float(end), 2),
                "wave_start_us_percentiles": start_pct,
                "wave_lifetime_us": {"mean": round(float(life.mean()), 2), "p10": round(float(np.percentile(life, 10)), 2), "p50": round(float(np.percentile(life, 50)), 2), "p90": round(float(np.percentile(life, 90)), 2)},
                "phase_us_mean": {nm: round(float(ph[:, i].mean()), 3) for i, nm in enumerate(names)},
                "phase_us_p90": {nm: round(float(np.percentile(ph[:, i], 90)), 3) for i, nm in enumerate(names)},
                "waves_resident_every_1us": resident[::4],
                "waves_per_xcc": np.bincount(xcc, minlength=8).tolist(),
                "wave_lifetime_us_mean_per_xcc": [round(float(life[xcc == k].mean()), 2) if (xcc == k).any() else None for k in range(8)],
                "phase_us_mean_per_xcc": {nm: [round(float(ph[xcc == k, i].mean()), 3) if (xcc == k).any() else None for k in range(8)] for i, nm in enumerate(names)},
                "last_wave_end_us_per_xcc": [round(float(us[xcc == k, 8].max()), 2) if (xcc == k).any() else None for k in range(8)]})
print(json.dumps({"kernel": "warp_tiles_lut_kernel<false>, config 2, one launch at a time, instrumented build (nine s_memrealtime + waits per wave: the launch takes longer than the product's)", "launches": res}, indent=1))
