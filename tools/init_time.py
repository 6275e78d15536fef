#!/usr/bin/env python3
"""Side measurement: the init / mask-refresh entry points on the C2 group (4 x 1080p) and on rig R (2 x 960x540):
wall time of pano_build_masks_voronoi, pano_build_masks_graphcut (host frames in, includes the H2D copies and the host
max-flow) and pano_estimate_gains."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
def best(fn, reps=5):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return round(min(t) * 1e3, 2)
out = {}
for name, d in (("c2_group_4x1080p", c2_group()), ("4x960x540", c2_group(w=960, h=540, f=501.2))):
    ctx = pano.Context(4, d["w"], d["h"], scale=d["scale"], num_bands=5, device=0)
    for i in range(4): ctx.set_camera(i, d["K"][i], d["R"][i])
    ctx.prepare()
    frames = [synth_frame(d["w"], d["h"], 42 + i) for i in range(4)]
    out[name] = {"build_masks_voronoi_ms": best(ctx.build_masks_voronoi),
                 "build_masks_graphcut_ms": best(lambda: ctx.build_masks_graphcut(frames)),
                 "estimate_gains_ms": best(lambda: ctx.estimate_gains(frames))}
print(json.dumps(out))
