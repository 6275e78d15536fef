"""Does "within 1 LSB per channel of OpenCV's CPU blender" (the north star's parity bar) survive every LEGITIMATE OpenCV build?
(VERDICT r04 "next round" #2; CPU only.)  OpenCV's own output is platform-defined in two places on this path:

  * cv::pyrDown CV_32F (the blend weights) associates its five-tap sum differently in scalar code, in the SSE2 / NEON vertical
    bodies and in the universal-intrinsics horizontal + vertical bodies of late 3.4.x (oracle/pano_oracle.c, po_set_pyrdown32f_variant);
  * the projectors call the platform's sinf / cosf / atan2f / acosf, which are not correctly rounded and differ between libms in the
    last ulp (po_set_trig_perturbation: +1 ulp, -1 ulp, or -1 / 0 / +1 by a hash of the argument - a libm that disagrees two times in
    three, far more often than glibc x86-64 and glibc aarch64 do).

The oracle is run over associations x libm models x both projectors on the bundled rigs (c1, c1b: 2222/1..8.png; R, S: 2222/4cam) and
on one group of config 2 at full size, each panorama against the one the shipped arithmetic gives (scalar order, this box's libm -
what the HIP path reproduces bit for bit).  Findings, asserted below and written to tests/_build/oracle_variants_report.json
(committed as profiles/r05_oracle_variants_report.json):

  * the f32 association ALONE never moves a panorama value by more than 1 LSB (a few hundred values of a million by exactly 1);
  * a libm that disagrees in the last ulp moves a 1/32-pixel bucket of cv::remap at a few dozen pixels per camera; where that
    happens on a strong edge the value moves by 2 - 3 LSB, and where it flips a NEAREST-warped mask pixel on a seam by more: a
    handful of values per million (bound asserted: 2e-4 of the values).  That spread is OpenCV against OpenCV - no restatement can be
    closer to "OpenCV" than OpenCV's builds are to each other - and it is what the pin loader's end-to-end tolerance is set to."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_png_bgr
from helpers import c2_group, synth_frame
import pin_stages as ps

ASSOCIATIONS = [("scalar", (0, 8, 0, 4)), ("sse2 body 8", (1, 8, 0, 4)), ("neon body 8", (2, 8, 0, 4)), ("universal 4 + horizontal 4 fused", (1, 4, 2, 4)),
                ("universal 16 + horizontal 16", (1, 16, 1, 16))]
LIBMS = [("this box", (0, 0)), ("+1 ulp", (1, 0)), ("-1 ulp", (2, 0)), ("hashed -1/0/+1, seed 1", (3, 1)), ("hashed -1/0/+1, seed 2", (3, 2))]
OUTLIERS = 2e-4


def _pano(po, g, kind, bands):
    masks = po.prepare_masks_voronoi(kind, g["w"], g["h"], g["K"], g["R"], g["scale"])
    return po.compose(g["frames"], g["K"], g["R"], g["scale"], masks, bands, kind=kind)[0]


def _sweep(po, name, g, kinds, bands, report):
    for kind in kinds:
        po.set_pyrdown32f_variant()
        po.set_trig_perturbation()
        base = _pano(po, g, kind, bands)
        for aname, a in ASSOCIATIONS:
            for lname, t in LIBMS:
                po.set_pyrdown32f_variant(*a)
                po.set_trig_perturbation(*t)
                p = _pano(po, g, kind, bands)
                row = {"rig": name, "projector": ("spherical", "cylindrical")[kind], "bands": bands, "association": aname, "libm": lname,
                       "values": int(base.size)}
                if p.shape != base.shape:   # a ROI integer moved: panoramas are not comparable pixel by pixel
                    row["roi_changed"] = [list(base.shape[:2]), list(p.shape[:2])]
                else:
                    d = np.abs(p.astype(np.int16) - base.astype(np.int16))
                    row.update(max_diff=int(d.max()), differ=int((d > 0).sum()), beyond_1_lsb=int((d > 1).sum()))
                report.append(row)
    po.set_pyrdown32f_variant()
    po.set_trig_perturbation()


def test_one_lsb_claim_over_legitimate_opencv_builds(po):
    po.set_threads(min(8, os.cpu_count() or 1))
    report = []
    try:
        groups = ps.load_groups(GOLDEN, load_png_bgr)
        for name in ("c1", "c1b", "r0", "r1", "s0", "s1"):
            g = groups[name]
            _sweep(po, name, g, (0, 1), 4 if name.startswith("c1") else 3, report)
        c2 = c2_group()
        c2["frames"] = [synth_frame(c2["w"], c2["h"], 42 + i) for i in range(4)]
        _sweep(po, "config 2, one group of 4 x 1080p", c2, (0,), 5, report)
    finally:
        po.set_pyrdown32f_variant()
        po.set_trig_perturbation()
        po.set_threads(1)
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    comparable = [r for r in report if "roi_changed" not in r]
    summary = {
        "runs": len(report), "roi_changed_runs": len(report) - len(comparable),
        "association_alone": {"max_diff": max(r["max_diff"] for r in comparable if r["libm"] == "this box"),
                              "most_values_moved": max(r["differ"] / r["values"] for r in comparable if r["libm"] == "this box")},
        "with_libm_models": {"max_diff": max(r["max_diff"] for r in comparable),
                             "worst_fraction_beyond_1_lsb": max(r["beyond_1_lsb"] / r["values"] for r in comparable),
                             "worst_count_beyond_1_lsb": max(r["beyond_1_lsb"] for r in comparable),
                             "worst_fraction_moved_at_all": max(r["differ"] / r["values"] for r in comparable)},
    }
    json.dump({"summary": summary, "runs": report}, open(os.path.join(out, "oracle_variants_report.json"), "w"), indent=1)
    print(json.dumps(summary))
    # (1) the f32 association alone: never beyond 1 LSB
    for r in comparable:
        if r["libm"] == "this box":
            assert r["max_diff"] <= 1, r
    # (2) every model: no ROI integer moves on these rigs, and the values beyond 1 LSB are a counted handful
    assert len(comparable) == len(report), [r for r in report if "roi_changed" in r]
    for r in comparable:
        assert r["beyond_1_lsb"] <= OUTLIERS * r["values"], r
    assert OUTLIERS == ps.E2E_OUTLIER_FRACTION   # the pin loader's end-to-end tolerance is this measured spread, not a guess
