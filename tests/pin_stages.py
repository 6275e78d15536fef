"""The stage schema of the OpenCV pin, and its loader (TEST INFRASTRUCTURE; VERDICT r04 "next round" #1).

tools/opencv_pin/pin.cpp, run ONCE by a holder of OpenCV 3.4.x, writes for every fixture group the arrays named below - every
stage of the reference's call sequence (include/ocvstitcher.hpp:975-1136 initSeam, :1141-1216 process, :1218-1261 updateMask,
src/stitching_detailed.cpp:841 exposure apply, src/master.cpp:321-326 stacking) as RAW data (.npy), not as hashes - into
tests/golden/opencv/<group>/.  `compare_group` then decides parity stage by stage:

  * every oracle stage is run on OPENCV'S OWN INPUT of that stage (its maps, its warps, its masks ...), so a difference in one stage
    cannot cascade into the verdict on the next: the first stage reported as DIVERGES is where the restatement is wrong;
  * integer stages (remap, resize, dilate, seam finders, int16 pyramids, accumulation, the panorama from pinned levels) must be EXACT;
  * the projector's float maps depend on the platform's libm (sinf / cosf / atan2f / acosf are not correctly rounded): compared in
    pixels, TOLERATED up to `MAP_PIXELS`, with the number of pixels whose 1/32-pixel bucket (cvRound(32 x)) flips reported;
  * cv::pyrDown CV_32F associates its sum differently in scalar, SSE2, NEON and universal-intrinsics builds (oracle/pano_oracle.c):
    the unit stage `unit/pyrdown32f_*` finds WHICH association the pinned build ran, the oracle is switched to it, and everything
    downstream of the weights is then demanded exact; when no association reproduces the pinned weights the stage DIVERGES;
  * end to end (the oracle from the frames, its own libm) the panorama must be within the north star's 1 LSB per channel but for a
    counted handful of values where a one-ulp map difference flips a 1/32-pixel bucket (`E2E_OUTLIER_FRACTION`).

`write_with_oracle` writes the SAME files from the oracle under a chosen build model (association + libm perturbation): the
synthetic stand-in the loader's own tests run on.  It pins nothing - only a run of pin.cpp does."""
import json
import os
from collections import OrderedDict

import numpy as np

MAP_PIXELS = 2e-3               # |difference| of a map coordinate, in pixels, that a libm disagreeing in the last ulp of sin / cos / atan2 can
                                # cause (1e-7 relative at coordinates of a few thousand pixels); a 1/32-pixel bucket of cv::remap is 0.031
E2E_OUTLIER_FRACTION = 2e-4     # values of the end-to-end panorama allowed beyond 1 LSB (1-ulp map differences at strong edges)
BLEND_LEVEL_RUNS = ("b4", "rig")  # the blend runs whose pyramid levels are dumped (the others: panorama only)

# the associations of cv::pyrDown CV_32F the oracle can follow: (name, (vertical, vbody, horizontal, hbody))
PYRDOWN32F_VARIANTS = (
    ("scalar", (0, 8, 0, 4)),
    ("sse2 / universal intrinsics, vertical body of 8", (1, 8, 0, 4)),
    ("universal intrinsics, vertical body of 4", (1, 4, 0, 4)),
    ("universal intrinsics, vertical body of 16", (1, 16, 0, 4)),
    ("neon, vertical body of 8", (2, 8, 0, 4)),
    ("universal intrinsics, vertical 4 + horizontal 4", (1, 4, 1, 4)),
    ("universal intrinsics, vertical 4 + horizontal 4 fused", (1, 4, 2, 4)),
    ("universal intrinsics, vertical 8 + horizontal 8", (1, 8, 1, 8)),
    ("universal intrinsics, vertical 8 + horizontal 8 fused", (1, 8, 2, 8)),
    ("universal intrinsics, vertical 16 + horizontal 16 fused", (1, 16, 2, 16)),
)


# ---- fixture groups --------------------------------------------------------------------------------------------------------------
def load_groups(golden_dir, load_png_bgr):
    """the fixture groups pin.cpp runs: c1, c1b (4 x 480x270, shared K) and the two stitchers of rigs R and S"""
    groups = OrderedDict()
    for prefix in ("c1", "c1b"):
        d = json.load(open(os.path.join(golden_dir, f"{prefix}_cams.json")))
        groups[prefix] = {
            "name": prefix, "n": 4, "w": 480, "h": 270, "kind": 0, "K": [d["K"]] * 4, "R": d["R"], "scale": d["scale"],
            "frames": [load_png_bgr(os.path.join(golden_dir, f"{prefix}_cam{i}.png")) for i in range(4)],
            # (tag, mask set, bands, strength, gains, cut)
            "runs": ([("bNO", "voronoi", -1, 0.0, False, None), ("b0", "voronoi", 0, 0.0, False, None), ("b2", "voronoi", 2, 0.0, False, None)]
                     if prefix == "c1" else []) +
                    [("b4", "voronoi", 4, 0.0, False, None)] +
                    ([("b2cut", "voronoi", 2, 0.0, False, (100, 20, 1000, 200))] if prefix == "c1" else []) +
                    [("gc4", "graphcut", 4, 0.0, False, None), ("gain4", "voronoi", 4, 0.0, True, None)],
        }
    for prefix in ("r", "s"):
        r = json.load(open(os.path.join(golden_dir, f"{prefix}_cams.json")))
        for s, st in enumerate(r["stitchers"]):
            v = st["cams"]
            groups[f"{prefix}{s}"] = {
                "name": f"{prefix}{s}", "n": 2, "w": r["width"], "h": r["height"], "kind": 0, "K": [v[0:9], v[18:27]],
                "R": [v[9:18], v[27:36]], "scale": v[-1],
                "frames": [load_png_bgr(os.path.join(golden_dir, f"{prefix}_cam{2 * s + i}.png")) for i in range(2)],
                "runs": [("rig", "graphcut", -2, 1.0, False, tuple(st["cut"]))],   # stitcherBlenderStrength: 1, the yaml cut
            }
    return groups


# ---- single stages, each from explicit inputs (the loader feeds them the PINNED inputs) ---------------------------------------------
def _seam_aspect(g):
    return min(1.0, (1e5 / (g["h"] * g["w"])) ** 0.5)          # ocvstitcher.hpp:298


def _projector(po, g, i, seam=False):
    K = np.asarray(g["K"][i], np.float32).copy()
    scale = np.float32(g["scale"])
    if seam:
        swa = _seam_aspect(g)
        f = np.float32(swa)
        K[0] *= f; K[2] *= f; K[4] *= f; K[5] *= f              # ocvstitcher.hpp:1008-1012
        scale = np.float32(np.float64(scale) * swa)              # static_cast<float>(warped_image_scale * seam_work_aspect), :1001
    return po.projector(g["kind"], scale, K, np.asarray(g["R"][i], np.float32))


def _gain_apply(po, img, gain_full):
    import ctypes as C
    out = np.ascontiguousarray(img, np.uint8).copy()
    gf = np.ascontiguousarray(gain_full, np.float32)
    po.lib().po_gain_apply_8uc3(out.ctypes.data_as(C.c_void_p), out.shape[1], out.shape[0], gf.ctypes.data_as(C.c_void_p))
    return out


def _bands(po, g, run, rois):
    tag, _, bands, strength, _, _ = run
    if bands != -2:
        return bands
    full = po.result_roi([r[:2] for r in rois], [r[2:] for r in rois])
    return po.bands_from_strength(full[2], full[3], strength)   # ocvstitcher.hpp:1188-1195


def _blend_run(po, g, run, rois, warps, masks, levels):
    """MultiBandBlender prepare / feed x n / blend on explicit warps and masks: {name: array}"""
    out = OrderedDict()
    nb = _bands(po, g, run, rois)
    b = po.Blender(nb)
    b.prepare([r[:2] for r in rois], [r[2:] for r in rois])
    for i in range(g["n"]):
        b.feed(warps[i].astype(np.int16), masks[i], rois[i][:2])
    if levels and nb >= 0:
        for l in range(b.num_bands() + 1):
            out[f"laplace_l{l}"] = b.level_laplace(l)
            out[f"weights_l{l}"] = b.level_weights(l)
    res, rmask = b.blend()
    out["result"] = res
    out["result_mask"] = rmask
    pano = np.clip(res, 0, 255).astype(np.uint8)               # convertTo(CV_8U), ocvstitcher.hpp:1208
    cut = run[5]
    if cut is not None:
        pano = np.ascontiguousarray(pano[cut[1]:cut[1] + cut[3], cut[0]:cut[0] + cut[2]])   # :1210
    out["pano"] = pano
    return out


def compute_group(po, g):
    """every stage of group g from the oracle alone, in pipeline order: OrderedDict name -> array"""
    n, w, h = g["n"], g["w"], g["h"]
    o = OrderedDict()
    swa = _seam_aspect(g)
    rois, warps, full_masks = [], [], []
    for i in range(n):
        p = _projector(po, g, i)
        rois.append(po.warp_roi(p, w, h))
    o["roi"] = np.asarray(rois, np.int32)
    for i in range(n):
        p = _projector(po, g, i)
        xm, ym = po.build_maps(p, w, h)
        o[f"cam{i}/xmap"], o[f"cam{i}/ymap"] = xm, ym
        warps.append(po.remap(g["frames"][i], xm, ym, po.INTER_LINEAR, po.BORDER_REFLECT))
        o[f"cam{i}/warp"] = warps[-1]
        full_masks.append(po.remap(np.full((h, w), 255, np.uint8), xm, ym, po.INTER_NEAREST, po.BORDER_CONSTANT))
        o[f"cam{i}/full_mask"] = full_masks[-1]
    seam_warps, seam_masks, corners = [], [], []
    for i in range(n):
        sf = po.resize_linear_exact_fxy(g["frames"][i], swa, swa)
        o[f"cam{i}/seam_frame"] = sf
        ps = _projector(po, g, i, seam=True)
        sxm, sym = po.build_maps(ps, sf.shape[1], sf.shape[0])
        o[f"cam{i}/seam_xmap"], o[f"cam{i}/seam_ymap"] = sxm, sym
        corners.append(po.warp_roi(ps, sf.shape[1], sf.shape[0])[:2])
        seam_warps.append(po.remap(sf, sxm, sym, po.INTER_LINEAR, po.BORDER_REFLECT))
        seam_masks.append(po.remap(np.full(sf.shape[:2], 255, np.uint8), sxm, sym, po.INTER_NEAREST, po.BORDER_CONSTANT))
        o[f"cam{i}/seam_warp"], o[f"cam{i}/seam_mask_warp"] = seam_warps[-1], seam_masks[-1]
    o["seam_corners"] = np.asarray(corners, np.int32)
    sizes = [(m.shape[1], m.shape[0]) for m in seam_masks]
    found = {"voronoi": po.voronoi_find(corners, sizes, seam_masks), "graphcut": po.graphcut_find(corners, seam_warps, seam_masks)}
    blend_masks = {}
    for kind, ms in found.items():
        blend_masks[kind] = []
        for i in range(n):
            o[f"cam{i}/{kind}_seam_mask"] = ms[i]
            bm = po.resize_linear_exact(po.dilate3x3(ms[i]), rois[i][2], rois[i][3]) & full_masks[i]   # :1097-1101
            blend_masks[kind].append(bm)
            o[f"cam{i}/{kind}_blend_mask"] = bm
    need_gains = any(r[4] for r in g["runs"])
    gain_warps = None
    if need_gains:
        maps, ok = po.gain_blocks_feed(corners, seam_warps, seam_masks)
        assert ok
        gain_warps = []
        for i in range(n):
            o[f"cam{i}/gain_map"] = maps[i]
            gain_warps.append(_gain_apply(po, warps[i], po.resize_linear_32f(maps[i], rois[i][2], rois[i][3])))
            o[f"cam{i}/warp_gain"] = gain_warps[-1]
    # unit stages: the pyramid primitives on real data, by themselves
    u_in = warps[0].astype(np.int16)
    o["unit/pyrdown16s_in"] = u_in
    o["unit/pyrdown16s_out"] = po.pyr_down_16s(u_in)
    o["unit/pyrup16s_out"] = po.pyr_up_16s(o["unit/pyrdown16s_out"])
    kind0 = g["runs"][0][1]
    wm = blend_masks[kind0][0].astype(np.float32) * np.float32(1.0 / 255.0)   # mask.convertTo(CV_32F, 1./255.) (blenders.cpp feed)
    o["unit/pyrdown32f_in"] = wm
    for l in range(1, 4):
        wm = po.pyr_down_32f(wm)
        o[f"unit/pyrdown32f_l{l}"] = wm
    for run in g["runs"]:
        tag, mkind, _, _, gains, _ = run
        br = _blend_run(po, g, run, rois, gain_warps if gains else warps, blend_masks[mkind], tag in BLEND_LEVEL_RUNS)
        for k, v in br.items():
            o[f"blend_{tag}/{k}"] = v
    return o


def write_group(out_dir, name, arrays, meta=None):
    d = os.path.join(out_dir, name)
    os.makedirs(d, exist_ok=True)
    man = OrderedDict()
    for k, v in arrays.items():
        fn = k.replace("/", "__") + ".npy"
        np.save(os.path.join(d, fn), np.ascontiguousarray(v))
        man[k] = {"file": fn, "dtype": str(v.dtype), "shape": list(v.shape)}
    json.dump({"group": name, "meta": meta or {}, "arrays": man}, open(os.path.join(d, "manifest.json"), "w"), indent=1)


def read_group(out_dir, name):
    d = os.path.join(out_dir, name)
    man = json.load(open(os.path.join(d, "manifest.json")))
    arrays = OrderedDict()
    for k, e in man["arrays"].items():
        a = np.load(os.path.join(d, e["file"]))
        assert list(a.shape) == e["shape"] and str(a.dtype) == e["dtype"], (k, a.shape, a.dtype, e)
        arrays[k] = a
    return arrays, man.get("meta", {})


def write_with_oracle(po, out_dir, groups, variant=(0, 8, 0, 4), trig=(0, 0), meta=None):
    """the stand-in for a run of pin.cpp: the same files, from the oracle under a build model"""
    po.set_pyrdown32f_variant(*variant)
    po.set_trig_perturbation(*trig)
    try:
        for name, g in groups.items():
            write_group(out_dir, name, compute_group(po, g),
                        dict(meta or {}, generator="oracle (SYNTHETIC stand-in, pins nothing)", pyrdown32f_variant=list(variant), trig=list(trig)))
    finally:
        po.set_pyrdown32f_variant()
        po.set_trig_perturbation()


# ---- comparison ----------------------------------------------------------------------------------------------------------------------
def _ulps(a, b):
    """distance in units in the last place between two f32 arrays (0 where both are equal, NaN-safe)"""
    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-(2 ** 31)) - ia, ia)
    ib = np.where(ib < 0, np.int64(-(2 ** 31)) - ib, ib)
    return np.abs(ia - ib)


def _hist(d, edges=(0, 1, 2, 3, 4, 8, 16, 64, 256)):
    d = np.asarray(d).reshape(-1)
    out = OrderedDict()
    for lo, hi in zip(edges, edges[1:] + (None,)):
        cnt = int(((d >= lo) & (d < hi)).sum()) if hi is not None else int((d >= lo).sum())
        if cnt:
            out[str(lo) if hi == lo + 1 else (f"{lo}..{hi - 1}" if hi is not None else f">={lo}")] = cnt
    return out


class Stage:
    def __init__(self, name, what, status, detail):
        self.name, self.what, self.status, self.detail = name, what, status, detail

    def line(self):
        return "%-9s %-34s %s  %s" % (self.status, self.name, self.what, self.detail)


def _exact(name, what, want, got):
    if want.shape != got.shape:
        return Stage(name, what, "DIVERGES", "shape OpenCV %s, oracle %s" % (want.shape, got.shape))
    if np.array_equal(want, got):
        return Stage(name, what, "EXACT", "%d values" % want.size)
    d = np.abs(want.astype(np.int64) - got.astype(np.int64))
    y = np.argwhere(d > 0)[0]
    return Stage(name, what, "DIVERGES", "%d of %d values differ, max |diff| %d, first at %s; histogram of |diff| %s" %
                 (int((d > 0).sum()), d.size, int(d.max()), tuple(int(v) for v in y), dict(_hist(d))))


def _within(name, what, want, got, tol, outlier_fraction=0.0):
    if want.shape != got.shape:
        return Stage(name, what, "DIVERGES", "shape OpenCV %s, oracle %s" % (want.shape, got.shape))
    d = np.abs(want.astype(np.int64) - got.astype(np.int64))
    over = int((d > tol).sum())
    detail = "max |diff| %d, %d of %d values differ, %d beyond %d; histogram of |diff| %s" % (int(d.max()), int((d > 0).sum()), d.size, over, tol, dict(_hist(d)))
    if not d.any():
        return Stage(name, what, "EXACT", "%d values" % d.size)
    if over <= outlier_fraction * d.size:
        return Stage(name, what, "TOLERATED", detail)
    return Stage(name, what, "DIVERGES", detail)


def _ulp_stage(name, what, want, got, max_ulps):
    if want.shape != got.shape:
        return Stage(name, what, "DIVERGES", "shape OpenCV %s, oracle %s" % (want.shape, got.shape))
    u = _ulps(want, got)
    if not u.any():
        return Stage(name, what, "EXACT", "%d values" % u.size)
    detail = "max %d ulp, %d of %d values differ; histogram of ulps %s" % (int(u.max()), int((u > 0).sum()), u.size, dict(_hist(u)))
    return Stage(name, what, "TOLERATED" if int(u.max()) <= max_ulps else "DIVERGES", detail)


def _map_stage(name, what, want, got):
    """a float coordinate map: the difference that matters is in pixels (ulps explode where a coordinate passes through zero)"""
    if want.shape != got.shape:
        return Stage(name, what, "DIVERGES", "shape OpenCV %s, oracle %s" % (want.shape, got.shape))
    if np.array_equal(want, got):
        return Stage(name, what, "EXACT", "%d values" % want.size)
    fin = np.isfinite(want) & np.isfinite(got)
    d = np.abs(want.astype(np.float64) - got.astype(np.float64))[fin]
    flips = int((np.rint(want[fin].astype(np.float64) * 32) != np.rint(got[fin].astype(np.float64) * 32)).sum())
    bad_inf = int((np.isfinite(want) != np.isfinite(got)).sum())
    detail = "max |diff| %.3g px, %d of %d values differ, %d flip their 1/32-pixel bucket" % (float(d.max()) if d.size else 0.0, int((d > 0).sum()), want.size, flips)
    return Stage(name, what, "TOLERATED" if (not d.size or float(d.max()) <= MAP_PIXELS) and not bad_inf else "DIVERGES", detail)


def compare_group(po, g, pin):
    """the oracle against the pinned arrays of one group: list of Stage in pipeline order.  Leaves the oracle on its defaults."""
    n, w, h = g["n"], g["w"], g["h"]
    S = []
    po.set_pyrdown32f_variant()
    po.set_trig_perturbation()
    swa = _seam_aspect(g)
    try:
        # -- geometry: the platform's libm is in these
        rois_or = np.asarray([po.warp_roi(_projector(po, g, i), w, h) for i in range(n)], np.int32)
        S.append(_within("roi", "RotationWarper::warpRoi (libm)", pin["roi"], rois_or, 0) if np.array_equal(pin["roi"], rois_or)
                 else _within("roi", "RotationWarper::warpRoi (libm)", pin["roi"], rois_or, 1, 1.0))
        rois = [tuple(int(v) for v in r) for r in pin["roi"]]
        same_roi = np.array_equal(pin["roi"], rois_or)
        for i in range(n):
            if same_roi:
                xm, ym = po.build_maps(_projector(po, g, i), w, h)
                S.append(_map_stage(f"cam{i}/xmap", "buildMaps x (libm)", pin[f"cam{i}/xmap"], xm))
                S.append(_map_stage(f"cam{i}/ymap", "buildMaps y (libm)", pin[f"cam{i}/ymap"], ym))
            # -- remap on OpenCV's own maps: fixed point, must be exact
            S.append(_exact(f"cam{i}/warp", "remap LINEAR / REFLECT on the pinned maps", pin[f"cam{i}/warp"],
                            po.remap(g["frames"][i], pin[f"cam{i}/xmap"], pin[f"cam{i}/ymap"], po.INTER_LINEAR, po.BORDER_REFLECT)))
            S.append(_exact(f"cam{i}/full_mask", "remap NEAREST / CONSTANT on the pinned maps", pin[f"cam{i}/full_mask"],
                            po.remap(np.full((h, w), 255, np.uint8), pin[f"cam{i}/xmap"], pin[f"cam{i}/ymap"], po.INTER_NEAREST, po.BORDER_CONSTANT)))
        # -- seam scale
        for i in range(n):
            sf = po.resize_linear_exact_fxy(g["frames"][i], swa, swa)
            S.append(_exact(f"cam{i}/seam_frame", "resize INTER_LINEAR_EXACT (Size(), fx, fy)", pin[f"cam{i}/seam_frame"], sf))
            psf = pin[f"cam{i}/seam_frame"]
            if pin[f"cam{i}/seam_xmap"].shape == tuple(po.warp_roi(_projector(po, g, i, seam=True), psf.shape[1], psf.shape[0])[:1:-1]):
                sxm, sym = po.build_maps(_projector(po, g, i, seam=True), psf.shape[1], psf.shape[0])
                S.append(_map_stage(f"cam{i}/seam_xmap", "buildMaps x at the seam scale (libm)", pin[f"cam{i}/seam_xmap"], sxm))
                S.append(_map_stage(f"cam{i}/seam_ymap", "buildMaps y at the seam scale (libm)", pin[f"cam{i}/seam_ymap"], sym))
            S.append(_exact(f"cam{i}/seam_warp", "remap LINEAR / REFLECT at the seam scale", pin[f"cam{i}/seam_warp"],
                            po.remap(psf, pin[f"cam{i}/seam_xmap"], pin[f"cam{i}/seam_ymap"], po.INTER_LINEAR, po.BORDER_REFLECT)))
            S.append(_exact(f"cam{i}/seam_mask_warp", "remap NEAREST / CONSTANT at the seam scale", pin[f"cam{i}/seam_mask_warp"],
                            po.remap(np.full(psf.shape[:2], 255, np.uint8), pin[f"cam{i}/seam_xmap"], pin[f"cam{i}/seam_ymap"], po.INTER_NEAREST,
                                     po.BORDER_CONSTANT)))
        corners = [tuple(int(v) for v in c) for c in pin["seam_corners"]]
        seam_warps = [pin[f"cam{i}/seam_warp"] for i in range(n)]
        seam_masks = [pin[f"cam{i}/seam_mask_warp"] for i in range(n)]
        sizes = [(m.shape[1], m.shape[0]) for m in seam_masks]
        found = {"voronoi": po.voronoi_find(corners, sizes, seam_masks), "graphcut": po.graphcut_find(corners, seam_warps, seam_masks)}
        what = {"voronoi": "VoronoiSeamFinder::find", "graphcut": "GraphCutSeamFinder(COST_COLOR)::find"}
        for kind in ("voronoi", "graphcut"):
            for i in range(n):
                if f"cam{i}/{kind}_seam_mask" not in pin:
                    continue
                S.append(_exact(f"cam{i}/{kind}_seam_mask", what[kind] + " on the pinned seam-scale warps", pin[f"cam{i}/{kind}_seam_mask"], found[kind][i]))
                bm = po.resize_linear_exact(po.dilate3x3(pin[f"cam{i}/{kind}_seam_mask"]), rois[i][2], rois[i][3]) & pin[f"cam{i}/full_mask"]
                S.append(_exact(f"cam{i}/{kind}_blend_mask", "dilate 3x3, resize INTER_LINEAR_EXACT, AND", pin[f"cam{i}/{kind}_blend_mask"], bm))
        # -- exposure
        if "cam0/gain_map" in pin:
            maps, ok = po.gain_blocks_feed(corners, seam_warps, seam_masks)
            for i in range(n):
                S.append(_ulp_stage(f"cam{i}/gain_map", "BlocksGainCompensator::feed (f64 normal equations, f32 maps)", pin[f"cam{i}/gain_map"], maps[i], 64))
                ga = _gain_apply(po, pin[f"cam{i}/warp"], po.resize_linear_32f(pin[f"cam{i}/gain_map"], rois[i][2], rois[i][3]))
                S.append(_within(f"cam{i}/warp_gain", "BlocksGainCompensator::apply on the pinned gain map", pin[f"cam{i}/warp_gain"], ga, 1, 1e-3))
        # -- the pyramid primitives by themselves
        S.append(_exact("unit/pyrdown16s_out", "cv::pyrDown CV_16S", pin["unit/pyrdown16s_out"], po.pyr_down_16s(pin["unit/pyrdown16s_in"])))
        S.append(_exact("unit/pyrup16s_out", "cv::pyrUp CV_16S", pin["unit/pyrup16s_out"], po.pyr_up_16s(pin["unit/pyrdown16s_out"])))
        kind0 = g["runs"][0][1]
        S.append(_ulp_stage("unit/pyrdown32f_in", "mask.convertTo(CV_32F, 1./255.)", pin["unit/pyrdown32f_in"],
                            pin[f"cam0/{kind0}_blend_mask"].astype(np.float32) * np.float32(1.0 / 255.0), 0))
        matched = None
        tried = []
        for vname, v in PYRDOWN32F_VARIANTS:
            po.set_pyrdown32f_variant(*v)
            src, worst = pin["unit/pyrdown32f_in"], 0
            for l in range(1, 4):
                got = po.pyr_down_32f(src)
                worst = max(worst, int(_ulps(pin[f"unit/pyrdown32f_l{l}"], got).max()))
                src = pin[f"unit/pyrdown32f_l{l}"]
            tried.append((vname, worst))
            if worst == 0:
                matched = (vname, v)
                break
        if matched:
            S.append(Stage("unit/pyrdown32f", "cv::pyrDown CV_32F, three levels", "EXACT", "association of the pinned build: " + matched[0]))
            po.set_pyrdown32f_variant(*matched[1])
        else:
            po.set_pyrdown32f_variant()
            S.append(Stage("unit/pyrdown32f", "cv::pyrDown CV_32F, three levels", "DIVERGES",
                           "no association the oracle knows reproduces the pinned weights: max ulps per association %s" % tried))
        # -- the blender on OpenCV's own warps and masks (under the association just found: exact, else within one count)
        tol = 0 if matched else 1
        for run in g["runs"]:
            tag, mkind, _, _, gains, _ = run
            warps = [pin[f"cam{i}/warp_gain" if gains else f"cam{i}/warp"] for i in range(n)]
            masks = [pin[f"cam{i}/{mkind}_blend_mask"] for i in range(n)]
            br = _blend_run(po, g, run, rois, warps, masks, tag in BLEND_LEVEL_RUNS)
            for k, v in br.items():
                key = f"blend_{tag}/{k}"
                if key not in pin:
                    S.append(Stage(key, "MultiBandBlender", "DIVERGES", "missing from the pin (band count differs?)"))
                elif k.startswith("weights_"):
                    S.append(_ulp_stage(key, "MultiBandBlender dst_band_weights_", pin[key], v, 0 if matched else 4))
                elif tol == 0:
                    S.append(_exact(key, "MultiBandBlender " + k, pin[key], v))
                else:
                    S.append(_within(key, "MultiBandBlender " + k, pin[key], v, 1, 1.0 if k != "pano" else 0.0))
        # -- end to end: the oracle by itself (its libm, the matched association), from the frames
        mine = compute_group(po, g) if same_roi else None
        for run in g["runs"]:
            key = f"blend_{run[0]}/pano"
            if mine is None:
                S.append(Stage("e2e/" + key, "frames -> panorama, all oracle", "DIVERGES", "ROIs differ: panoramas are not comparable pixel by pixel"))
            else:
                S.append(_within("e2e/" + key, "frames -> panorama, all oracle (north star: within 1 LSB)", pin[key], mine[key], 1, E2E_OUTLIER_FRACTION))
    finally:
        po.set_pyrdown32f_variant()
        po.set_trig_perturbation()
    return S


def compare_stack(po, pin_r0, pin_r1, pinned_stack):
    """src/master.cpp:321-326 on the two pinned half panoramas"""
    return _exact("stacked", "resize INTER_LINEAR + vconcat + divider (master.cpp:321-326)", pinned_stack,
                  po.stack_master(pin_r0["blend_rig/pano"], pin_r1["blend_rig/pano"]))


def first_divergence(stages):
    for s in stages:
        if s.status == "DIVERGES":
            return s
    return None


def report(stages):
    return "\n".join(s.line() for s in stages)
