#!/usr/bin/env python3
"""debug aid: where does a config-1 panorama differ from the oracle (bounding box and count of differing pixels per band count)"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pano_oracle as po
from conftest import load_png_bgr, GOLDEN
pano = importlib.import_module("img-stitching_amd")
d1 = json.load(open(os.path.join(GOLDEN, "c1_cams.json")))
frames = [load_png_bgr(os.path.join(GOLDEN, f"c1_cam{i}.png")) for i in range(4)]
K = [d1["K"]] * 4
masks = po.prepare_masks_voronoi(0, 480, 270, K, d1["R"], d1["scale"])
for bands in (2, 3, 4, 5):
    ctx = pano.Context(4, 480, 270, scale=d1["scale"], num_bands=bands, device=0)
    for i in range(4):
        ctx.set_camera(i, K[i], d1["R"][i])
    ctx.prepare()
    for i in range(4):
        ctx.set_mask(i, masks[i])
    got = ctx.compose_host(frames)
    want, _ = po.compose(frames, K, d1["R"], d1["scale"], masks, bands)
    bad = np.argwhere((got != want).any(axis=2))
    if len(bad):
        print("bands", bands, "differing pixels", len(bad), "rows", bad[:, 0].min(), bad[:, 0].max(), "cols", bad[:, 1].min(), bad[:, 1].max(),
              "max abs diff", int(np.abs(got.astype(int) - want.astype(int)).max()))
        cols = np.unique(bad[:, 1] // 64)
        print("  64-px column bins:", cols.tolist())
    else:
        print("bands", bands, "equal")
