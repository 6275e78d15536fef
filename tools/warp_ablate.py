#!/usr/bin/env python3
"""Diagnostic: time the warp kernel (K1) and its ablation variants on the C2 workload.
Uses libpano_hip_diag.so (make -C img-stitching_amd/csrc diag); not part of the product."""
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child():
    import torch
    from helpers import c2_group, synth_frame
    pano = importlib.import_module("img-stitching_amd")
    g = c2_group()
    ctx = pano.Context(4, g["w"], g["h"], scale=g["scale"], num_bands=5, device=0)
    for i in range(4):
        ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare()
    ctx.build_masks_voronoi()
    frames = [torch.from_numpy(synth_frame(g["w"], g["h"], 42 + i)).cuda() for i in range(4)]
    ptrs = [t.data_ptr() for t in frames]
    st = torch.cuda.current_stream().cuda_stream
    ctx.set_profiling(True)
    for _ in range(10):
        ctx.feed_cameras(0xF, ptrs, [g["w"] * 3] * 4, st)
    ctx.stage_stats(True)
    for _ in range(100):
        ctx.feed_cameras(0xF, ptrs, [g["w"] * 3] * 4, st)
    torch.cuda.synchronize()
    ms, n = ctx.stage_stats(True)
    print(json.dumps({"abl": os.environ.get("PANO_LUT_ABL", os.environ.get("PANO_WARP_ABL", "0")), "warp_us": round(ms[0] / n[0] * 1e3, 2),
                      "pyr_us": round(ms[1] / n[1] * 1e3, 2)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        names = {0: "full", 1: "no tap loads", 4: "no stores"}
        var = "PANO_LUT_ABL" if not os.environ.get("PANO_WARP_ON_THE_FLY") else "PANO_WARP_ABL"
        for abl in (0, 1, 4):
            env = dict(os.environ, PANO_LIB=os.path.join(ROOT, "img-stitching_amd", "libpano_hip_diag.so"))
            env[var] = str(abl)
            out = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print(names[abl], out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
