#!/usr/bin/env python3
"""Generate the committed OUTPUT fixtures of config 1 by running the CPU oracle (oracle/pano_oracle.c) on the
committed input fixtures.  PARITY UNPINNED: these vectors pin the oracle against regressions and give the GPU
tests a reference that does not depend on rebuilding the oracle; they are NOT outputs of OpenCV or of the
reference (which ships none and cannot be built here).

  c1_golden.json   sha256 of: per-camera warped image and Voronoi blend mask, panorama for bands 0/2/4 and
                   Blender::NO, cut panorama; plus ROI / size integers
                   graph-cut blend masks (the reference's seam finder) and the panorama under them; block gain maps
                   (BlocksGainCompensator::feed as initSeam runs it, raw f32 bytes) and the panorama with them applied
  c1_pano_b4.png   the 1333x257 4-band panorama (RGB PNG) for eyeballing and byte comparison
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pano_oracle as po  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    d = json.load(open(os.path.join(HERE, "c1_cams.json")))
    frames = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(HERE, f"c1_cam{i}.png")).convert("RGB"))[:, :, ::-1])
              for i in range(4)]
    K = [d["K"]] * 4
    out = {"rois": [], "warp_sha256": [], "mask_sha256": [], "pano_sha256": {}}
    masks = po.prepare_masks_voronoi(po.SPHERICAL, 480, 270, K, d["R"], d["scale"])
    for i in range(4):
        p = po.projector(po.SPHERICAL, d["scale"], K[i], d["R"][i])
        out["rois"].append(list(po.warp_roi(p, 480, 270)))
        out["warp_sha256"].append(sha(po.warp(p, frames[i])[1]))
        out["mask_sha256"].append(sha(masks[i]))
    for nb in (-1, 0, 2, 4):
        pano, _ = po.compose(frames, K, d["R"], d["scale"], masks, nb)
        out["pano_sha256"][str(nb)] = sha(pano)
        if nb == 4:
            Image.fromarray(np.ascontiguousarray(pano[:, :, ::-1])).save(os.path.join(HERE, "c1_pano_b4.png"), optimize=True)
    cut = (100, 20, 1000, 200)
    pano, _ = po.compose(frames, K, d["R"], d["scale"], masks, 2, cut=cut)
    out["cut"] = list(cut)
    out["pano_cut_sha256"] = sha(pano)
    out["pano_size"] = [1333, 257]
    # the reference's own seam finder and its exposure compensator, fed the way initSeam feeds them
    gc = po.prepare_masks_graphcut(frames, K, d["R"], d["scale"])
    out["graphcut_mask_sha256"] = [sha(m) for m in gc]
    out["graphcut_pano_b4_sha256"] = sha(po.compose(frames, K, d["R"], d["scale"], gc, 4)[0])
    gains, sizes = po.estimate_gains(frames, K, d["R"], d["scale"])
    out["gain_map_shape"] = [list(g.shape) for g in gains]
    out["gain_map_sha256"] = [sha(g) for g in gains]
    full = [po.resize_linear_32f(g, r[2], r[3]) for g, r in zip(gains, out["rois"])]
    out["gain_pano_b4_sha256"] = sha(po.compose(frames, K, d["R"], d["scale"], masks, 4, gain_maps=full)[0])
    json.dump(out, open(os.path.join(HERE, "c1_golden.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
