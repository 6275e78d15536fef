#!/usr/bin/env python3
"""Per-wave timeline of ONE launch of the FUSED K1 (experiments/k1_fused_level1.patch + the stamps described in docs/EXPERIMENTS.md,
round 4; config 2, 8 cameras): every wave stamps s_memrealtime (100 MHz) at entry, after its scalar prologue, in front of and behind
the two barriers and at its end.  Printed per wave kind (patch waves 0-2 = with the level-1 epilogue, patch wave 3, ring wave 4):
medians of the phases in us, and the launch's span.
    PANO_LIB=experiments/_build/libpano_fused_trace.so python3 tools/fused_wave_timeline.py"""
import ctypes as C, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
lib = C.CDLL(pano.LIB_PATH)
lib.pano_debug_k1f_trace.argtypes = [C.c_void_p, C.c_uint, C.c_int]
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
reps = []
for rep in range(3):
    lib.pano_debug_k1f_trace(None, 0, 1)
    step(); torch.cuda.synchronize()
    buf = np.zeros((1 << 16, 8), dtype=np.uint64)
    n = lib.pano_debug_k1f_trace(buf.ctypes.data, 1 << 16, 0)
    assert n == 1 << 16
    t = buf[buf[:, 0] != 0].astype(np.int64)
    wv = t[:, 7] & 0xff
    t0 = t[:, 0].min()
    us = lambda a: round(float(np.median(a)) / 100.0, 2)
    out = {"waves": int(len(t)), "launch_span_us": round(float(t[:, 6].max() - t0) / 100.0, 2),
           "last_wave_starts_us": round(float(t[:, 0].max() - t0) / 100.0, 2)}
    for name, sel in (("patch_waves_0_2_with_epilogue", wv < 3), ("patch_wave_3", wv == 3), ("ring_wave_4", wv == 4)):
        w = t[sel]
        if not len(w): continue
        out[name] = {"n": int(len(w)), "prologue": us(w[:, 1] - w[:, 0]), "to_first_barrier": us(w[:, 2] - w[:, 1]),
                     "parked_at_first_barrier": us(w[:, 3] - w[:, 2]), "body_to_second_barrier": us(w[:, 4] - w[:, 3]),
                     "parked_at_second_barrier": us(w[:, 5] - w[:, 4]), "epilogue_and_drain": us(w[:, 6] - w[:, 5]),
                     "lifetime": us(w[:, 6] - w[:, 0]), "lifetime_p90": round(float(np.percentile(w[:, 6] - w[:, 0], 90)) / 100.0, 2)}
    # waves resident over the launch (sampled every 0.5 us)
    grid = np.arange(0, int(t[:, 6].max() - t0), 50)
    res = [(int(((t[:, 0] - t0) <= x).sum() - ((t[:, 6] - t0) <= x).sum())) for x in grid]
    out["waves_resident_every_half_us"] = res
    reps.append(out)
print(json.dumps({"lib": os.path.basename(pano.LIB_PATH), "unit": "us (medians over the waves of one launch; the clock ticks at 100 MHz)", "launches": reps}))
