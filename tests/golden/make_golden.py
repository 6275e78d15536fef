#!/usr/bin/env python3
"""Generate the committed OUTPUT fixtures by running the CPU oracle (oracle/pano_oracle.c) on the committed input
fixtures.  PARITY UNPINNED: these vectors pin the oracle against regressions and give the GPU tests a reference that
does not depend on rebuilding the oracle; they are NOT outputs of OpenCV or of the reference (which ships none and
cannot be built here).

  c1_golden.json   config 1 (2222/1..4.png, cameraparaout_1.txt): sha256 of per-camera warped image and Voronoi blend
                   mask, panorama for bands 0/2/4 and Blender::NO, cut panorama; plus ROI / size integers;
                   graph-cut blend masks (the reference's seam finder) and the panorama under them; block gain maps
                   (BlocksGainCompensator::feed as initSeam runs it, raw f32 bytes) and the panorama with them applied
  c1_pano_b4.png   the 1333x257 4-band panorama (RGB PNG) for eyeballing and byte comparison
  c1b_golden.json  the other half of the bundled set (2222/5..8.png, cameraparaout_2.txt), same keys (bands 4 only)
  r_golden.json    rig R (cfg/cameras.yaml 4cam-black/960) on the REAL frames 2222/4cam/0..3.png: per stitcher the
                   ROIs, graph-cut masks, band count from strength 1, the cut panorama (what ocvStitcher::process returns),
                   and master.cpp's stacked output of the two halves
  st258_golden.json the 2222/258st frames (see make_inputs.py) under the scaled config-1 / 1b parameters: masks and 4-band panoramas
  s_golden.json    rig S (cfg/cameras.yaml 4cam-silver/640, :212-228) on ITS frames 2222/4cam/1/0..3.png: the same keys
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


def load_bgr(name):
    return np.ascontiguousarray(np.asarray(Image.open(os.path.join(HERE, name)).convert("RGB"))[:, :, ::-1])


def group_480(prefix, all_bands):
    d = json.load(open(os.path.join(HERE, f"{prefix}_cams.json")))
    frames = [load_bgr(f"{prefix}_cam{i}.png") for i in range(4)]
    K = [d["K"]] * 4
    out = {"rois": [], "warp_sha256": [], "mask_sha256": [], "pano_sha256": {}}
    masks = po.prepare_masks_voronoi(po.SPHERICAL, 480, 270, K, d["R"], d["scale"])
    for i in range(4):
        p = po.projector(po.SPHERICAL, d["scale"], K[i], d["R"][i])
        out["rois"].append(list(po.warp_roi(p, 480, 270)))
        out["warp_sha256"].append(sha(po.warp(p, frames[i])[1]))
        out["mask_sha256"].append(sha(masks[i]))
    for nb in ((-1, 0, 2, 4) if all_bands else (4,)):
        pano, _ = po.compose(frames, K, d["R"], d["scale"], masks, nb)
        out["pano_sha256"][str(nb)] = sha(pano)
        if nb == 4:
            out["pano_size"] = [pano.shape[1], pano.shape[0]]
            if all_bands:
                Image.fromarray(np.ascontiguousarray(pano[:, :, ::-1])).save(os.path.join(HERE, f"{prefix}_pano_b4.png"), optimize=True)
    if all_bands:
        cut = (100, 20, 1000, 200)
        pano, _ = po.compose(frames, K, d["R"], d["scale"], masks, 2, cut=cut)
        out["cut"] = list(cut)
        out["pano_cut_sha256"] = sha(pano)
    # the reference's own seam finder and its exposure compensator, fed the way initSeam feeds them
    gc = po.prepare_masks_graphcut(frames, K, d["R"], d["scale"])
    out["graphcut_mask_sha256"] = [sha(m) for m in gc]
    out["graphcut_pano_b4_sha256"] = sha(po.compose(frames, K, d["R"], d["scale"], gc, 4)[0])
    gains, sizes = po.estimate_gains(frames, K, d["R"], d["scale"])
    out["gain_map_shape"] = [list(g.shape) for g in gains]
    out["gain_map_sha256"] = [sha(g) for g in gains]
    full = [po.resize_linear_32f(g, r[2], r[3]) for g, r in zip(gains, out["rois"])]
    out["gain_pano_b4_sha256"] = sha(po.compose(frames, K, d["R"], d["scale"], masks, 4, gain_maps=full)[0])
    json.dump(out, open(os.path.join(HERE, f"{prefix}_golden.json"), "w"), indent=1)


def rig(prefix="r"):
    """what replay.cpp does with 2222/4cam: frames 0,1 -> stitcher 0 ("up"), 2,3 -> stitcher 1 ("down"), each
    init(yaml) + calibration (graph-cut masks from those frames, bands from strength 1) + process; then master.cpp's
    resize + vconcat + divider.  prefix "r": rig R (4cam-black/960, 2222/4cam/0..3.png); "s": rig S (4cam-silver/640,
    2222/4cam/1/0..3.png)"""
    r = json.load(open(os.path.join(HERE, f"{prefix}_cams.json")))
    W, H = r["width"], r["height"]
    frames = [load_bgr(f"{prefix}_cam{i}.png") for i in range(4)]
    out = {"stitchers": []}
    halves = []
    for s, st in enumerate(r["stitchers"]):
        v = st["cams"]
        K = [v[0:9], v[18:27]]
        R = [v[9:18], v[27:36]]
        scale = v[-1]
        fr = frames[2 * s:2 * s + 2]
        rois = [list(po.warp_roi(po.projector(po.SPHERICAL, scale, K[i], R[i]), W, H)) for i in range(2)]
        full = po.result_roi([q[:2] for q in rois], [q[2:] for q in rois])
        bands = po.bands_from_strength(full[2], full[3], 1.0)
        gc = po.prepare_masks_graphcut(fr, K, R, scale)
        pano, _ = po.compose(fr, K, R, scale, gc, bands, cut=tuple(st["cut"]))
        halves.append(pano)
        out["stitchers"].append({"rois": rois, "pano_roi": list(full), "bands": bands, "cut": st["cut"],
                                 "graphcut_mask_sha256": [sha(m) for m in gc], "pano_cut_sha256": sha(pano),
                                 "pano_cut_size": [pano.shape[1], pano.shape[0]]})
    stacked = po.stack_master(halves[0], halves[1])
    out["stack_master_sha256"] = sha(stacked)
    out["stack_master_size"] = [stacked.shape[1], stacked.shape[0]]
    Image.fromarray(np.ascontiguousarray(stacked[:, :, ::-1])).save(os.path.join(HERE, f"{prefix}_stacked.png"), optimize=True)
    json.dump(out, open(os.path.join(HERE, f"{prefix}_golden.json"), "w"), indent=1)


def st258():
    """the 2222/258st frames (320x180 after the 2x2 box) under the config-1 / config-1b parameters scaled by 2/3: Voronoi masks,
    4-band panorama of each group of four"""
    out = {"groups": []}
    for g, prefix in enumerate(("c1", "c1b")):
        d = json.load(open(os.path.join(HERE, f"{prefix}_cams.json")))
        K = [v * (2.0 / 3.0) if i in (0, 2, 4, 5) else v for i, v in enumerate(d["K"])]
        scale = float(np.float32(d["scale"]) * np.float32(2.0 / 3.0))
        frames = [load_bgr(f"st258_cam{4 * g + i}.png") for i in range(4)]
        masks = po.prepare_masks_voronoi(po.SPHERICAL, 320, 180, [K] * 4, d["R"], scale)
        pano, _ = po.compose(frames, [K] * 4, d["R"], scale, masks, 4)
        out["groups"].append({"K": K, "scale": scale, "pano_size": [pano.shape[1], pano.shape[0]], "pano_b4_sha256": sha(pano),
                              "mask_sha256": [sha(m) for m in masks]})
    json.dump(out, open(os.path.join(HERE, "st258_golden.json"), "w"), indent=1)


def main():
    st258()
    group_480("c1", True)
    group_480("c1b", False)
    rig("r")
    rig("s")


if __name__ == "__main__":
    main()
