#!/usr/bin/env python3
"""Device-copy ceilings measured by the library (pano_probe_copy): the float4 grid-stride copy the guide quotes
6.29 TB/s for, and a copy in K1's traffic shape (direct-to-LDS 16-byte loads + an 8-byte table entry per lane in, one dword per lane
and plane out) - at the size of config 2's K1 launch (warm / cold) and at a size where launch ramp no longer counts.
    python tools/copy_ceiling.py [--reps 100]      -> one JSON object on stdout"""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=100)
    a = ap.parse_args()
    pano = importlib.import_module("img-stitching_amd")
    ctx = pano.Context(1, 64, 64, scale=50.0, num_bands=0, device=0)   # a context names the device and owns a stream; nothing is composed
    ctx.set_camera(0, [50.0, 0, 32, 0, 50.0, 32, 0, 0, 1], [1.0, 0, 0, 0, 1, 0, 0, 0, 1])
    ctx.prepare()
    out = {"kernel_source_id": pano.kernel_source_id(), "reps": a.reps}
    C = pano.Context
    r = lambda d: {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d.items()}
    out["f4_copy_512MiB"] = r(ctx.probe_copy(C.PROBE_COPY_F4, 512 << 20, sets=1, reps=max(10, a.reps // 5)))
    out["f4_flat_copy_512MiB"] = r(ctx.probe_copy(C.PROBE_COPY_F4_FLAT, 512 << 20, sets=1, reps=max(10, a.reps // 5)))
    out["f4_copy_128MiB_warm"] = r(ctx.probe_copy(C.PROBE_COPY_F4, 128 << 20, sets=1, reps=max(10, a.reps // 2)))
    out["f4_copy_44MiB_k1_sized_warm"] = r(ctx.probe_copy(C.PROBE_COPY_F4, 44 << 20, sets=1, reps=a.reps))
    out["f4_copy_44MiB_k1_sized_cold"] = r(ctx.probe_copy(C.PROBE_COPY_F4, 44 << 20, sets=8, reps=a.reps))
    for wg, name in ((9300, "k1_shape_9300_workgroups"), (18600, "k1_shape_18600_workgroups"), (150000, "k1_shape_150000_workgroups")):
        per = wg * (4608 + 2048 + 3072)
        cold_sets = max(2, -(-(320 << 20) // per))
        out[name + "_warm"] = r(ctx.probe_copy(C.PROBE_COPY_K1_SHAPE, wg, sets=1, reps=a.reps))
        if cold_sets <= 64:
            out[name + "_cold"] = dict(r(ctx.probe_copy(C.PROBE_COPY_K1_SHAPE, wg, sets=cold_sets, reps=a.reps)), sets=cold_sets)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
