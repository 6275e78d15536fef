#!/usr/bin/env python3
"""DIAGNOSTIC: per-wave timelines of the blend levels in seam-first order (blend_level_ordered_kernel: level 0 = <true,3>, levels 1 and 2
= <false,1>) from the build of experiments/blend_wave_timeline.patch - entry / prologue done / stores issued / stores done stamps
(s_memrealtime, 100 MHz), the wave's owner hint, level and XCC - config 2, one frame at a time.
    PANO_LIB=experiments/_build/libpano_blendtrace.so python tools/blend_timeline.py"""
import ctypes as C, importlib, json, os, sys
import numpy as np
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
def step():
    ctxs[0].compose_pair(ctxs[1], [t.data_ptr() for t in frames[0]], [g["w"] * 3] * 4, outs[0].data_ptr(), ow * 3,
                         [t.data_ptr() for t in frames[1]], [g["w"] * 3] * 4, outs[1].data_ptr(), ow * 3, st)
for _ in range(100): step()
torch.cuda.synchronize()
lib = pano.load_library()
lib.pano_debug_set_blend_dbg.argtypes = [C.c_void_p]
cap = 1 << 17
res = {}
for rep in range(3):
    dbg = torch.zeros(cap * 4, dtype=torch.int64, device="cuda")
    assert lib.pano_debug_set_blend_dbg(C.c_void_p(dbg.data_ptr())) == 0
    step(); torch.cuda.synchronize()
    lib.pano_debug_set_blend_dbg(C.c_void_p(0))
    tr = dbg.cpu().numpy().view(np.uint64).reshape(cap, 4)   # a slot per (level, workgroup, wave): no atomics in the kernel
    tr = tr[tr[:, 0] != 0]
    tag = tr[:, 3]
    hint = (tag & np.uint64(0xff)).astype(np.int64); lvl = ((tag >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
    xcc = ((tag >> np.uint64(16)) & np.uint64(15)).astype(np.int64); done = (tag >> np.uint64(32)).astype(np.int64) / 100.0
    for l in (2, 1, 0):
        m = lvl == l
        if not m.any(): continue
        t = tr[m, :3].astype(np.int64)
        hwid = (t[:, 2] >> 32) & 0xffff                      # HW_REG_HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
        lo = t[:, 2] & 0xffffffff                              # stores issued: low 32 bits of the clock, the rest from the entry stamp
        t[:, 2] = (t[:, 0] & ~0xffffffff) | lo
        t[t[:, 2] < t[:, 0], 2] += 1 << 32
        t0 = t[:, 0].min(); us = (t - t0) / 100.0
        life = done[m]; end = us[:, 0] + life; seam = hint[m] == 15; launch = float(end.max())
        bins = np.arange(0, launch + 0.5, 0.5)
        resident = [int(((us[:, 0] <= b) & (end > b)).sum()) for b in bins]
        # per SIMD (XCC, SE, SH, CU, SIMD): how many waves it holds over time, and how long after a wave's end the next one starts there
        simd = xcc[m] * 65536 + (hwid & 0xfff0)
        conc_time = {}; refill = []
        for key in np.unique(simd):
            k = simd == key
            ev = sorted([(a, 1) for a in us[k, 0]] + [(b, -1) for b in end[k]])
            c = 0; prev = ev[0][0]; ends_waiting = []
            for tt, d in ev:
                conc_time[c] = conc_time.get(c, 0.0) + (tt - prev); prev = tt
                if d < 0: ends_waiting.append(tt)
                elif ends_waiting: refill.append(tt - ends_waiting.pop(0))
                c += d
        tot = sum(v for kk, v in conc_time.items() if kk > 0)
        simd_stats = {"simds_seen": int(len(np.unique(simd))), "waves_per_simd_max": int(max(conc_time)),
                      "share_of_busy_time_at_n_waves": {str(kk): round(v / tot, 3) for kk, v in sorted(conc_time.items()) if kk > 0},
                      "end_to_next_entry_on_the_same_simd_us": {"mean": round(float(np.mean(refill)), 2), "p50": round(float(np.percentile(refill, 50)), 2),
                                                                 "p90": round(float(np.percentile(refill, 90)), 2)} if refill else None}
        res.setdefault("level %d" % l, []).append({"per_simd": simd_stats,
            "waves": int(m.sum()), "seam_waves": int(seam.sum()), "first_entry_to_last_store_done_us": round(launch, 2),
            "entry_us_percentiles": {p: round(float(np.percentile(us[:, 0], p)), 2) for p in (50, 90, 99, 100)},
            "prologue_us_mean": round(float((us[:, 1] - us[:, 0]).mean()), 3),
            "entry_to_stores_issued_us": {"single owner": round(float((us[:, 2] - us[:, 0])[~seam].mean()), 2), "seam": round(float((us[:, 2] - us[:, 0])[seam].mean()), 2) if seam.any() else None},
            "lifetime_us_single_owner": {"mean": round(float(life[~seam].mean()), 2), "p90": round(float(np.percentile(life[~seam], 90)), 2)},
            "lifetime_us_seam": {"mean": round(float(life[seam].mean()), 2), "p90": round(float(np.percentile(life[seam], 90)), 2), "last_done_us": round(float(end[seam].max()), 2)} if seam.any() else None,
            "wave_us_total": round(float(life.sum()), 1),
            "waves_resident_mean_while_any": round(float(life.sum() / launch), 0),
            "waves_resident_every_half_us": resident,
            "last_wave_done_us_per_xcc": [round(float(end[xcc[m] == k].max()), 2) if (xcc[m] == k).any() else None for k in range(8)]})
print(json.dumps({"what": "blend levels in seam-first order, config 2 (both canvases per launch), one frame at a time, instrumented build (stamps + a wait for "
                          "the wave's stores at its end: lifetimes include the stores' completion)", "launches": res}, indent=1))
