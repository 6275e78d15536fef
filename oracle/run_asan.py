#!/usr/bin/env python3
"""Run the config-1 oracle paths under AddressSanitizer + UBSan (TEST INFRASTRUCTURE).
  make -C oracle asan && LD_PRELOAD=$(gcc -print-file-name=libasan.so) python oracle/run_asan.py"""
import ctypes, json, os, sys
import numpy as np
from PIL import Image
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pano_oracle as po
po._SO = os.path.join(HERE, "libpano_oracle_asan.so")
po.build = lambda force=False: po._SO
G = os.path.join(HERE, "..", "tests", "golden")
d = json.load(open(os.path.join(G, "c1_cams.json")))
frames = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(G, f"c1_cam{i}.png")).convert("RGB"))[:, :, ::-1]) for i in range(4)]
K = [d["K"]] * 4
for kind in (0, 1):
    masks = po.prepare_masks_voronoi(kind, 480, 270, K, d["R"], d["scale"])
    for nb in (-1, 0, 3):
        out, _ = po.compose(frames, K, d["R"], d["scale"], masks, nb, kind=kind)
        out2, _ = po.compose(frames, K, d["R"], d["scale"], masks, nb, kind=kind, cut=(5, 7, 300, 100))
print("oracle ran clean under ASan/UBSan")
