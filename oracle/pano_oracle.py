"""ctypes binding of oracle/libpano_oracle.so (CPU restatement, TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  PARITY UNPINNED - see pano_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpano_oracle.so")

SPHERICAL, CYLINDRICAL = 0, 1
INTER_NEAREST, INTER_LINEAR = 0, 1
BORDER_CONSTANT, BORDER_REFLECT, BORDER_REFLECT_101 = 0, 2, 4


def build(force=False):
    src = os.path.join(_HERE, "pano_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpano_oracle.so"])
    return _SO


class Projector(C.Structure):
    _fields_ = [("kind", C.c_int), ("scale", C.c_float), ("k", C.c_float * 9), ("rinv", C.c_float * 9),
                ("r_kinv", C.c_float * 9), ("k_rinv", C.c_float * 9)]


class FrontEnd(C.Structure):
    _fields_ = [("raw_w", C.c_int), ("raw_h", C.c_int), ("undist_w", C.c_int), ("undist_h", C.c_int),
                ("K", C.c_double * 9), ("dist", C.c_double * 4), ("rect", C.c_int * 4), ("out_w", C.c_int), ("out_h", C.c_int)]


def front_end(raw_wh, undist_wh, K, dist, rect, out_wh):
    fe = FrontEnd()
    fe.raw_w, fe.raw_h = raw_wh; fe.undist_w, fe.undist_h = undist_wh; fe.out_w, fe.out_h = out_wh
    for i in range(9):
        fe.K[i] = float(K[i])
    for i in range(4):
        fe.dist[i] = float(dist[i]); fe.rect[i] = int(rect[i])
    return fe


def optimal_new_camera_matrix(K, dist, w, h):
    Kc = (C.c_double * 9)(*[float(v) for v in K]); dc = (C.c_double * 4)(*[float(v) for v in dist]); out = (C.c_double * 9)()
    lib().po_optimal_new_camera_matrix(Kc, dc, int(w), int(h), out)
    return list(out)


class ComposeArgs(C.Structure):
    _fields_ = [("n", C.c_int), ("kind", C.c_int), ("src_w", C.c_int), ("src_h", C.c_int),
                ("frames", C.POINTER(C.c_void_p)), ("K9s", C.POINTER(C.c_float)), ("R9s", C.POINTER(C.c_float)),
                ("scale", C.c_float), ("masks", C.POINTER(C.c_void_p)), ("num_bands", C.c_int),
                ("gain_maps", C.POINTER(C.c_void_p)), ("cut", C.c_int * 4), ("front", C.POINTER(FrontEnd))]


_lib = None
_libs = {}


def _load(path):
    L = C.CDLL(path)
    L.po_blender_create.restype = C.c_void_p
    L.po_blender_level_laplace.restype = C.c_void_p
    L.po_blender_level_weights.restype = C.c_void_p
    L.po_bands_from_strength.argtypes = [C.c_int, C.c_int, C.c_float]
    return L


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = _libs["check"] = _load(_SO)
    return _lib


def build_timed(out_dir=None):
    """the SAME source built the way BASELINE.md states the timed CPU leg: -O3 -march=native (still -ffp-contract=off, no fast
    math: the arithmetic is unchanged), OpenMP.  A second library, so that the checker build (-O2, portable) is untouched; built
    on the host it is timed on (-march=native), into a scratch directory."""
    import tempfile
    # a directory of its own, mode 0700, made for this build: nobody else on a shared box can have put a file or a link at the path
    out = os.path.join(out_dir or tempfile.mkdtemp(prefix="pano_oracle_timed_"), "libpano_oracle_timed.so")
    src = os.path.join(_HERE, "pano_oracle.c")
    subprocess.check_call(["gcc", "-O3", "-march=native", "-std=gnu99", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fopenmp",
                           "-shared", "-o", out, src, "-lm"])
    return out


def select_build(which, path=None):
    """which = "check" (the checker build every test uses) or "timed" (build_timed(): bench.py's cpu_baseline leg only, after it has
    shown that build to reproduce the checker's bytes)"""
    global _lib
    lib()
    if which == "timed" and "timed" not in _libs:
        _libs["timed"] = _load(path or build_timed())
    _lib = _libs[which]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f9(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(9))


def set_threads(n):
    lib().po_set_threads(int(n))


def set_pyrdown32f_variant(vertical=0, vbody=8, horizontal=0, hbody=4):
    """which association of cv::pyrDown CV_32F the oracle follows (pano_oracle.c): (0, *, 0, *) = scalar = the default"""
    lib().po_set_pyrdown32f_variant(int(vertical), int(vbody), int(horizontal), int(hbody))


def set_trig_perturbation(mode=0, seed=0):
    """libm disagreement model for the projectors (0 none, 1 +1 ulp, 2 -1 ulp, 3 hashed -1/0/+1)"""
    lib().po_set_trig_perturbation(int(mode), C.c_uint(int(seed)))


def projector(kind, scale, K, R):
    p = Projector()
    K = _f9(K); R = _f9(R)
    lib().po_projector_set(C.byref(p), int(kind), C.c_float(scale), _p(K), _p(R))
    return p


def map_forward(p, x, y):
    u = C.c_float(); v = C.c_float()
    lib().po_map_forward(C.byref(p), C.c_float(x), C.c_float(y), C.byref(u), C.byref(v))
    return u.value, v.value


def map_backward(p, u, v):
    x = C.c_float(); y = C.c_float()
    lib().po_map_backward(C.byref(p), C.c_float(u), C.c_float(v), C.byref(x), C.byref(y))
    return x.value, y.value


def warp_roi(p, w, h):
    r = (C.c_int * 4)()
    lib().po_warp_roi(C.byref(p), int(w), int(h), r)
    return tuple(r)


def result_roi(corners, sizes):
    c = np.ascontiguousarray(np.asarray(corners, dtype=np.int32).reshape(-1))
    s = np.ascontiguousarray(np.asarray(sizes, dtype=np.int32).reshape(-1))
    r = (C.c_int * 4)()
    lib().po_result_roi(len(c) // 2, _p(c), _p(s), r)
    return tuple(r)


def build_maps(p, w, h):
    r = warp_roi(p, w, h)
    xm = np.empty((r[3], r[2]), np.float32); ym = np.empty((r[3], r[2]), np.float32)
    lib().po_build_maps(C.byref(p), int(w), int(h), _p(xm), _p(ym))
    return xm, ym


def remap(src, xmap, ymap, interp, border):
    src = np.ascontiguousarray(src)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dh, dw = xmap.shape
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.uint8)
    lib().po_remap_8u(_p(src), w, h, C.c_size_t(w * cn), cn, _p(np.ascontiguousarray(xmap)),
                      _p(np.ascontiguousarray(ymap)), dw, dh, int(interp), int(border), _p(dst), C.c_size_t(dw * cn))
    return dst


def warp(p, src, interp=INTER_LINEAR, border=BORDER_REFLECT):
    """RotationWarper::warp -> (corner, warped)"""
    src = np.ascontiguousarray(src)
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    r = warp_roi(p, w, h)
    dst = np.empty((r[3], r[2]) if src.ndim == 2 else (r[3], r[2], cn), np.uint8)
    corner = (C.c_int * 2)()
    lib().po_warp_8u(C.byref(p), _p(src), w, h, C.c_size_t(w * cn), cn, int(interp), int(border), _p(dst), corner)
    return (corner[0], corner[1]), dst


def pyr_down_16s(a):
    a = np.ascontiguousarray(a, dtype=np.int16)
    h, w = a.shape[:2]; cn = 1 if a.ndim == 2 else a.shape[2]
    d = np.empty(((h + 1) // 2, (w + 1) // 2) + (() if a.ndim == 2 else (cn,)), np.int16)
    lib().po_pyr_down_16s(_p(a), w, h, cn, _p(d))
    return d


def pyr_down_32f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    h, w = a.shape
    d = np.empty(((h + 1) // 2, (w + 1) // 2), np.float32)
    lib().po_pyr_down_32f(_p(a), w, h, _p(d))
    return d


def pyr_up_16s(a):
    a = np.ascontiguousarray(a, dtype=np.int16)
    h, w = a.shape[:2]; cn = 1 if a.ndim == 2 else a.shape[2]
    d = np.empty((2 * h, 2 * w) + (() if a.ndim == 2 else (cn,)), np.int16)
    lib().po_pyr_up_16s(_p(a), w, h, cn, _p(d))
    return d


def dilate3x3(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    d = np.empty_like(a)
    lib().po_dilate3x3_8u(_p(a), a.shape[1], a.shape[0], _p(d))
    return d


def resize_linear_exact(a, dw, dh):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    h, w = a.shape[:2]; cn = 1 if a.ndim == 2 else a.shape[2]
    d = np.empty((dh, dw) + (() if a.ndim == 2 else (cn,)), np.uint8)
    lib().po_resize_linear_exact_8u(_p(a), w, h, cn, _p(d), int(dw), int(dh))
    return d


def resize_linear_exact_fxy(a, fx, fy):
    """resize(a, Size(), fx, fy, INTER_LINEAR_EXACT): dsize = cvRound(ssize*f), sampling grid 1/f"""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    h, w = a.shape[:2]; cn = 1 if a.ndim == 2 else a.shape[2]
    L = lib()
    L.po_resize_dsize.restype = C.c_int
    dw = L.po_resize_dsize(int(w), C.c_double(fx)); dh = L.po_resize_dsize(int(h), C.c_double(fy))
    d = np.empty((dh, dw) + (() if a.ndim == 2 else (cn,)), np.uint8)
    L.po_resize_linear_exact_8u_fxy(_p(a), w, h, cn, _p(d), int(dw), int(dh), C.c_double(fx), C.c_double(fy))
    return d


def resize_linear_32f(a, dw, dh):
    a = np.ascontiguousarray(a, dtype=np.float32)
    d = np.empty((dh, dw), np.float32)
    lib().po_resize_linear_32f(_p(a), a.shape[1], a.shape[0], _p(d), int(dw), int(dh))
    return d


def distance_l1(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    d = np.empty(a.shape, np.float32)
    lib().po_distance_l1(_p(a), a.shape[1], a.shape[0], _p(d))
    return d


def resize_linear_8u(a, dw, dh):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    h, w = a.shape[:2]; cn = 1 if a.ndim == 2 else a.shape[2]
    d = np.empty((dh, dw) + (() if a.ndim == 2 else (cn,)), np.uint8)
    lib().po_resize_linear_8u(_p(a), w, h, cn, _p(d), int(dw), int(dh))
    return d


def stack_master(up, down):
    up = np.ascontiguousarray(up, np.uint8); down = np.ascontiguousarray(down, np.uint8)
    out = np.empty((2 * down.shape[0], down.shape[1], 3), np.uint8)
    lib().po_stack_master(_p(up), up.shape[1], up.shape[0], _p(down), down.shape[1], down.shape[0], _p(out))
    return out


def stack_finalcut(up, down, finalcut):
    up = np.ascontiguousarray(up, np.uint8); down = np.ascontiguousarray(down, np.uint8)
    w = min(up.shape[1], down.shape[1]); h = min(up.shape[0], down.shape[0]) - 2 * finalcut
    out = np.empty((2 * h, w, 3), np.uint8)
    lib().po_stack_finalcut(_p(up), up.shape[1], up.shape[0], _p(down), down.shape[1], down.shape[0], int(finalcut), _p(out))
    return out


def bands_from_strength(w, h, strength):
    return lib().po_bands_from_strength(int(w), int(h), C.c_float(strength))


def voronoi_find(corners, sizes, masks):
    n = len(masks)
    c = np.ascontiguousarray(np.asarray(corners, dtype=np.int32).reshape(-1))
    s = np.ascontiguousarray(np.asarray(sizes, dtype=np.int32).reshape(-1))
    ms = [np.ascontiguousarray(m, dtype=np.uint8).copy() for m in masks]
    arr = (C.c_void_p * n)(*[m.ctypes.data for m in ms])
    lib().po_voronoi_find(n, _p(c), _p(s), arr)
    return ms


def prepare_masks_voronoi(kind, w, h, Ks, Rs, scale):
    Ks = np.ascontiguousarray(np.asarray(Ks, dtype=np.float32).reshape(-1, 9))
    Rs = np.ascontiguousarray(np.asarray(Rs, dtype=np.float32).reshape(-1, 9))
    n = Ks.shape[0]
    masks = []
    for i in range(n):
        r = warp_roi(projector(kind, scale, Ks[i], Rs[i]), w, h)
        masks.append(np.zeros((r[3], r[2]), np.uint8))
    arr = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
    lib().po_prepare_masks_voronoi(n, int(kind), int(w), int(h), _p(Ks), _p(Rs), C.c_float(scale), arr)
    return masks


def graphcut_find(corners, images, masks):
    """detail::GraphCutSeamFinder(COST_COLOR)::find on 8UC3 images: returns the updated masks"""
    n = len(masks)
    c = np.ascontiguousarray(np.asarray(corners, dtype=np.int32).reshape(-1))
    ims = [np.ascontiguousarray(m, dtype=np.uint8) for m in images]
    ms = [np.ascontiguousarray(m, dtype=np.uint8).copy() for m in masks]
    s = np.ascontiguousarray(np.asarray([[m.shape[1], m.shape[0]] for m in ms], dtype=np.int32).reshape(-1))
    ia = (C.c_void_p * n)(*[m.ctypes.data for m in ims])
    ma = (C.c_void_p * n)(*[m.ctypes.data for m in ms])
    lib().po_graphcut_find(n, _p(c), _p(s), ia, ma)
    return ms


def gc_grid_max_flow(term, wh, wv):
    """test hook: GCGraph<float>::maxFlow on a 4-connected grid; (flow, inSourceSegment labels)"""
    term = np.ascontiguousarray(term, np.float32); wh = np.ascontiguousarray(wh, np.float32); wv = np.ascontiguousarray(wv, np.float32)
    h, w = term.shape
    lab = np.zeros((h, w), np.uint8)
    f = lib().po_gc_grid_max_flow
    f.restype = C.c_float
    return float(f(w, h, _p(term), _p(wh), _p(wv), _p(lab))), lab


def prepare_masks_graphcut(frames, Ks, Rs, scale, kind=SPHERICAL):
    """ocvStitcher::updateMask with GraphCutSeamFinder(COST_COLOR): frames -> blend masks"""
    Ks = np.ascontiguousarray(np.asarray(Ks, dtype=np.float32).reshape(-1, 9))
    Rs = np.ascontiguousarray(np.asarray(Rs, dtype=np.float32).reshape(-1, 9))
    n = Ks.shape[0]
    fr = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
    h, w = fr[0].shape[:2]
    masks = []
    for i in range(n):
        r = warp_roi(projector(kind, scale, Ks[i], Rs[i]), w, h)
        masks.append(np.zeros((r[3], r[2]), np.uint8))
    fa = (C.c_void_p * n)(*[f.ctypes.data for f in fr])
    arr = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
    lib().po_prepare_masks_graphcut(n, int(kind), int(w), int(h), fa, _p(Ks), _p(Rs), C.c_float(scale), arr)
    return masks


def gain_feed(corners, images, masks):
    """detail::GainCompensator::feed: one gain (float64) per image"""
    n = len(images)
    c = np.ascontiguousarray(np.asarray(corners, dtype=np.int32).reshape(-1))
    ims = [np.ascontiguousarray(m, dtype=np.uint8) for m in images]
    ms = [np.ascontiguousarray(m, dtype=np.uint8) for m in masks]
    s = np.ascontiguousarray(np.asarray([[m.shape[1], m.shape[0]] for m in ims], dtype=np.int32).reshape(-1))
    ia = (C.c_void_p * n)(*[m.ctypes.data for m in ims])
    ma = (C.c_void_p * n)(*[m.ctypes.data for m in ms])
    ist = (C.c_size_t * n)(*[m.shape[1] * 3 for m in ims])
    mst = (C.c_size_t * n)(*[m.shape[1] for m in ms])
    g = np.zeros(n, np.float64)
    ok = lib().po_gain_feed(n, _p(c), _p(s), ia, ist, ma, mst, _p(g))
    return g, bool(ok)


def gain_blocks_feed(corners, images, masks, bl_w=32, bl_h=32):
    """detail::BlocksGainCompensator(bl_w, bl_h)::feed: one float32 gain map per image"""
    n = len(images)
    c = np.ascontiguousarray(np.asarray(corners, dtype=np.int32).reshape(-1))
    ims = [np.ascontiguousarray(m, dtype=np.uint8) for m in images]
    ms = [np.ascontiguousarray(m, dtype=np.uint8) for m in masks]
    s = np.ascontiguousarray(np.asarray([[m.shape[1], m.shape[0]] for m in ims], dtype=np.int32).reshape(-1))
    maps = [np.zeros(((m.shape[0] + bl_h - 1) // bl_h, (m.shape[1] + bl_w - 1) // bl_w), np.float32) for m in ims]
    ia = (C.c_void_p * n)(*[m.ctypes.data for m in ims])
    ma = (C.c_void_p * n)(*[m.ctypes.data for m in ms])
    ga = (C.c_void_p * n)(*[m.ctypes.data for m in maps])
    ok = lib().po_gain_blocks_feed(n, _p(c), _p(s), ia, ma, int(bl_w), int(bl_h), ga)
    return maps, bool(ok)


def estimate_gains(frames, Ks, Rs, scale, kind=SPHERICAL, bl_w=32, bl_h=32):
    """the compensator feed of ocvStitcher::initSeam from stitcher-size frames: (gain maps, seam-scale tile sizes)"""
    Ks = np.ascontiguousarray(np.asarray(Ks, dtype=np.float32).reshape(-1, 9))
    Rs = np.ascontiguousarray(np.asarray(Rs, dtype=np.float32).reshape(-1, 9))
    n = Ks.shape[0]
    fr = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
    h, w = fr[0].shape[:2]
    swa = min(1.0, (1e5 / (h * w)) ** 0.5)
    ssw, ssh = int(np.rint(w * swa)), int(np.rint(h * swa))
    maps = []
    for i in range(n):
        K = Ks[i].copy()
        f = np.float32(swa)
        K[0] *= f; K[2] *= f; K[4] *= f; K[5] *= f
        r = warp_roi(projector(kind, np.float32(scale * swa), K, Rs[i]), ssw, ssh)
        maps.append(np.zeros(((r[3] + bl_h - 1) // bl_h, (r[2] + bl_w - 1) // bl_w), np.float32))
    sizes = np.zeros(2 * n, np.int32)
    fa = (C.c_void_p * n)(*[f.ctypes.data for f in fr])
    ga = (C.c_void_p * n)(*[m.ctypes.data for m in maps])
    ok = lib().po_estimate_gains(n, int(kind), int(w), int(h), fa, _p(Ks), _p(Rs), C.c_float(scale), int(bl_w), int(bl_h),
                                 _p(sizes), ga)
    assert ok, "singular gain system"
    return maps, sizes.reshape(-1, 2)


class Blender:
    """cv::detail::MultiBandBlender (num_bands >= 0) / Blender::NO (num_bands == -1)."""

    def __init__(self, num_bands):
        self.h = C.c_void_p(lib().po_blender_create(int(num_bands)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().po_blender_destroy(self.h)
            self.h = None

    def prepare(self, corners, sizes):
        c = np.ascontiguousarray(np.asarray(corners, dtype=np.int32).reshape(-1))
        s = np.ascontiguousarray(np.asarray(sizes, dtype=np.int32).reshape(-1))
        lib().po_blender_prepare(self.h, len(c) // 2, _p(c), _p(s))

    def feed(self, img16s, mask, tl):
        img16s = np.ascontiguousarray(img16s, dtype=np.int16)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        h, w = mask.shape
        assert img16s.shape == (h, w, 3)
        lib().po_blender_feed(self.h, _p(img16s), _p(mask), w, h, int(tl[0]), int(tl[1]))

    def blend(self):
        r = self.dst_roi_final()
        dst = np.empty((r[3], r[2], 3), np.int16)
        m = np.empty((r[3], r[2]), np.uint8)
        lib().po_blender_blend(self.h, _p(dst), _p(m))
        return dst, m

    def num_bands(self):
        return lib().po_blender_num_bands(self.h)

    def dst_roi(self):
        r = (C.c_int * 4)(); lib().po_blender_dst_roi(self.h, r); return tuple(r)

    def dst_roi_final(self):
        r = (C.c_int * 4)(); lib().po_blender_dst_roi_final(self.h, r); return tuple(r)

    def level_size(self, lvl):
        r = (C.c_int * 2)(); lib().po_blender_level_size(self.h, lvl, r); return tuple(r)

    def level_laplace(self, lvl):
        w, h = self.level_size(lvl)
        ptr = lib().po_blender_level_laplace(self.h, lvl)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int16)), shape=(h, w, 3)).copy()

    def level_weights(self, lvl):
        w, h = self.level_size(lvl)
        ptr = lib().po_blender_level_weights(self.h, lvl)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(h, w)).copy()

    def last_tile(self):
        r = (C.c_int * 4)(); t = (C.c_int * 4)()
        lib().po_blender_last_tile(self.h, r, t)
        return tuple(r), tuple(t)


def compose(frames, Ks, Rs, scale, masks, num_bands, kind=SPHERICAL, gain_maps=None, cut=None, front=None):
    """ocvStitcher::process.  Returns (pano u8 HxWx3, (warp_ms, feed_ms, blend_ms)).
    front: list of FrontEnd (fused undistort): frames are then raw captured frames."""
    n = len(frames)
    frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
    masks = [np.ascontiguousarray(m, dtype=np.uint8) for m in masks]
    h, w = frames[0].shape[:2]
    if front is not None:
        w, h = front[0].out_w, front[0].out_h
    Ks = np.ascontiguousarray(np.asarray(Ks, dtype=np.float32).reshape(n, 9))
    Rs = np.ascontiguousarray(np.asarray(Rs, dtype=np.float32).reshape(n, 9))
    a = ComposeArgs()
    a.n = n; a.kind = kind; a.src_w = w; a.src_h = h
    fa = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    ma = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
    a.frames = C.cast(fa, C.POINTER(C.c_void_p)); a.masks = C.cast(ma, C.POINTER(C.c_void_p))
    a.K9s = Ks.ctypes.data_as(C.POINTER(C.c_float)); a.R9s = Rs.ctypes.data_as(C.POINTER(C.c_float))
    a.scale = scale; a.num_bands = int(num_bands)
    if gain_maps is not None:
        gain_maps = [np.ascontiguousarray(g, dtype=np.float32) for g in gain_maps]
        ga = (C.c_void_p * n)(*[g.ctypes.data for g in gain_maps])
        a.gain_maps = C.cast(ga, C.POINTER(C.c_void_p))
    if front is not None:
        fa = (FrontEnd * n)(*front)
        a.front = C.cast(fa, C.POINTER(FrontEnd))
    rois = [warp_roi(projector(kind, scale, Ks[i], Rs[i]), w, h) for i in range(n)]
    full = result_roi([r[:2] for r in rois], [r[2:] for r in rois])
    if cut is None:
        cut = (0, 0, full[2], full[3])
    for i in range(4):
        a.cut[i] = int(cut[i])
    out = np.empty((cut[3], cut[2], 3), np.uint8)
    wh = (C.c_int * 2)()
    rc = lib().po_compose(C.byref(a), _p(out), wh)
    if rc != 0:
        raise ValueError("cut rectangle outside the panorama")
    ms = (C.c_double * 3)()
    lib().po_last_timings(ms)
    return out, tuple(ms)
