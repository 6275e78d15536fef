#!/usr/bin/env python3
"""Generate the committed INPUT fixtures from the reference's data files.

Run once in the build container (needs /root/reference, PIL).  Nothing here is
executed on the GPU box; the outputs are committed:

  c1_cam{0..3}.png   4 x 480x270 RGB8, box-downsampled (4x4 mean, round-half-up)
                     from /root/reference/2222/{1..4}.png (1920x1080).  480x270 is
                     the frame size the K in 2222/cameraparaout_1.txt was
                     calibrated at (cx=240, cy=135).
  c1_cams.json       last record of 2222/cameraparaout_1.txt (old 7-line format:
                     one shared K, 4 R, scale), parsed verbatim as decimal strings
                     -> floats.
  r_cams.json        cfg/cameras.yaml `4cam-black / inputsz 960` structure
                     (2 stitchers x 2 cams, 18*N+1 floats + cut) - data only.
  c1b_cam{0..3}.png  the other half of the bundled set: 2222/{5..8}.png, same 4x4 box
  c1b_cams.json      last record of 2222/cameraparaout_2.txt (f=535.415, scale 591.584)
  r_cam{0..3}.png    /root/reference/2222/4cam/{0..3}.png (960x540, pixels unchanged,
                     re-encoded): the real rig-R frames; replay.cpp:211-215 gives 0,1 to the
                     "up" stitcher and 2,3 to the "down" one.
  st258_cam{0..7}.png /root/reference/2222/258st/{1..8}.png (640x360: another scene from an 8-camera rig of the same kind; the tree
                     holds no parameters for it), 2x2 box-downsampled to 320x180.  The tests pair them with the config-1 /
                     config-1b parameters scaled by 2/3 - a pairing of this repo's own, used for real image content at a third
                     frame size, with no claim that those parameters belong to these frames.
  s_cams.json        cfg/cameras.yaml `4cam-silver / inputsz 640` structure (:212-228; rig S)
  s_cam{0..3}.png    /root/reference/2222/4cam/1/{0..3}.png (640x360, pixels unchanged)

These are data (inputs), not reference source.
"""
import json
import os
import re

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def box4(path):
    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint32)
    h, w, _ = a.shape
    assert (w, h) == (1920, 1080)
    a = a.reshape(h // 4, 4, w // 4, 4, 3).sum(axis=(1, 3))
    return ((a + 8) // 16).astype(np.uint8)


def last_record_old_format(path, source="2222/cameraparaout_1.txt (last record)"):
    lines = [l.strip() for l in open(path) if l.strip()]
    idx = max(i for i, l in enumerate(lines) if ":" in l)
    rec = lines[idx:]
    stamp = rec[0].rstrip(":")
    K = [float(v) for v in rec[1].rstrip(",").split(",")]
    assert len(K) == 9
    Rs = []
    for l in rec[2:6]:
        r = [float(v) for v in l.rstrip(",").split(",")]
        assert len(r) == 9
        Rs.append(r)
    scale = float(rec[6])
    return {"timestamp": stamp, "K": K, "R": Rs, "scale": scale,
            "width": 480, "height": 270, "source": source}


def structure(path, sttype="4cam-black", inputsz=960, height=540):
    txt = open(path).read()
    # locate the structure block: sttype <sttype> ... inputsz: <inputsz>
    blocks = txt.split("\n -\n")
    for b in blocks:
        if f"sttype: {sttype}" in b and f"inputsz: {inputsz}" in b:
            cams = re.findall(r"cams:\s*\[([^\]]*)\]", b, flags=re.S)
            cuts = re.findall(r"^\s*cut:\s*\[([^\]]*)\]", b, flags=re.M)
            out = []
            for c, cut in zip(cams, cuts):
                vals = [float(v) for v in c.replace("\n", " ").split(",") if v.strip()]
                assert len(vals) == 18 * 2 + 1
                out.append({"cams": vals, "cut": [int(v) for v in cut.split(",")]})
            return {"width": inputsz, "height": height, "num_images": 2, "stitchers": out,
                    "source": f"cfg/cameras.yaml structure lijing/imx390/{sttype}/undistor/120/{inputsz}"}
    raise SystemExit("structure not found")


def main():
    for i in range(4):
        im = box4(f"{REF}/2222/{i + 1}.png")
        Image.fromarray(im, "RGB").save(f"{OUT}/c1_cam{i}.png", optimize=True)
    json.dump(last_record_old_format(f"{REF}/2222/cameraparaout_1.txt"),
              open(f"{OUT}/c1_cams.json", "w"), indent=1)
    json.dump(structure(f"{REF}/cfg/cameras.yaml"), open(f"{OUT}/r_cams.json", "w"), indent=1)
    for i in range(8):
        a = np.asarray(Image.open(f"{REF}/2222/258st/{i + 1}.png").convert("RGB"), dtype=np.uint32)
        assert a.shape == (360, 640, 3)
        a = a.reshape(180, 2, 320, 2, 3).sum(axis=(1, 3))
        Image.fromarray(((a + 2) // 4).astype(np.uint8), "RGB").save(f"{OUT}/st258_cam{i}.png", optimize=True)
    # rig S: the 4cam-silver / inputsz 640 structure (cfg/cameras.yaml:212-228) and ITS frames 2222/4cam/1/0..3.png (640x360)
    json.dump(structure(f"{REF}/cfg/cameras.yaml", "4cam-silver", 640, 360), open(f"{OUT}/s_cams.json", "w"), indent=1)
    for i in range(4):
        im = Image.open(f"{REF}/2222/4cam/1/{i}.png").convert("RGB")
        assert im.size == (640, 360)
        im.save(f"{OUT}/s_cam{i}.png", optimize=True)
    for i in range(4):
        im = box4(f"{REF}/2222/{i + 5}.png")
        Image.fromarray(im, "RGB").save(f"{OUT}/c1b_cam{i}.png", optimize=True)
    json.dump(last_record_old_format(f"{REF}/2222/cameraparaout_2.txt", "2222/cameraparaout_2.txt (last record)"),
              open(f"{OUT}/c1b_cams.json", "w"), indent=1)
    for i in range(4):
        im = Image.open(f"{REF}/2222/4cam/{i}.png").convert("RGB")
        assert im.size == (960, 540)
        im.save(f"{OUT}/r_cam{i}.png", optimize=True)


if __name__ == "__main__":
    main()
