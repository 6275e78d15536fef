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
# gain estimation (BlocksGainCompensator::feed from frames) and the caller-side / front-end helpers
for kind in (0, 1):
    for bl in ((32, 32), (16, 24), (500, 500)):
        po.estimate_gains(frames, K, d["R"], d["scale"], kind, *bl)
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, (37, 53, 3), dtype=np.uint8), rng.integers(0, 256, (41, 47, 3), dtype=np.uint8)]
po.gain_blocks_feed([(0, 0), (30, 5)], imgs, [np.full(i.shape[:2], 255, np.uint8) for i in imgs], 8, 8)
po.gain_feed([(0, 0), (1000, 5)], imgs, [np.full(i.shape[:2], 255, np.uint8) for i in imgs])
# graph-cut seam finder: the whole updateMask restatement, and flat frames (ties: free vertices)
for kind in (0, 1):
    po.prepare_masks_graphcut(frames, K, d["R"], d["scale"], kind)
po.prepare_masks_graphcut([np.full_like(f, 77) for f in frames], K, d["R"], d["scale"])
print("oracle ran clean under ASan/UBSan")
