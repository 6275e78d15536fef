#!/usr/bin/env python3
"""The mask refresh of the WHOLE rig (both stitchers, 8 x 1080p) beside the frame loop: the two contexts are independent, each runs
its graph cuts on a thread of the library - started together a rig refresh costs one stitcher's max-flow time, not two
(VERDICT r03 #9).  Prints one JSON object: ms for begin (upload + seam-scale warps, on the caller's thread), for the cuts
(begin -> masks ready) one stitcher after the other and both at once, and for the install (poll that finds them ready)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import c2_group, synth_frame
pano = importlib.import_module("img-stitching_amd")
w, h, n = 1920, 1080, 4
g = c2_group(w=w, h=h, f=1002.416)
ctxs, frames = [], []
for k in range(2):
    ctx = pano.Context(n, w, h, scale=g["scale"], num_bands=5, device=0)
    for i in range(n): ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi()
    fr = [synth_frame(w, h, 5 + 4 * k + i) for i in range(n)]
    ctx.compose_host(fr); ctx.build_masks_graphcut(fr); ctx.compose_host(fr)   # pools warm, like after calibration
    ctxs.append(ctx); frames.append(fr)
out = {"config": "2 stitchers x 4 x 1920x1080, GraphCutSeamFinder(COST_COLOR) at the seam scale, host max-flow on one library thread per stitcher", "runs": []}
for rep in range(3):
    r = {}
    t0 = time.perf_counter()
    for k in range(2):
        ctxs[k].refresh_masks_begin(frames[k]); ctxs[k].refresh_masks_wait()
    r["one_after_the_other_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    t0 = time.perf_counter()
    for k in range(2): ctxs[k].refresh_masks_begin(frames[k])
    t1 = time.perf_counter()
    for k in range(2): ctxs[k].refresh_masks_wait()
    t2 = time.perf_counter()
    r["both_at_once_ms"] = round(1e3 * (t2 - t0), 2)
    r["begin_both_ms"] = round(1e3 * (t1 - t0), 2)
    for k in range(2): ctxs[k].refresh_masks_begin(frames[k])
    time.sleep(0.6)
    t0 = time.perf_counter()
    done = [ctxs[k].refresh_masks_poll() for k in range(2)]
    r["install_both_ms"] = round(1e3 * (time.perf_counter() - t0), 2); r["installed"] = done
    for k in range(2): ctxs[k].compose_host(frames[k])
    out["runs"].append(r)
print(json.dumps(out))
