"""img-stitching_amd - MI355X-native panorama composition (the `ocvStitcher::process` hot path of
LeRoii/Img-Stitching) behind the C-ABI of include/pano.h.

This module is the Python host-side mirror used by the tests and bench.py: a ctypes binding
(`Context`) plus `Stitcher`, which keeps the reference class's surface
(`init` / `calibration` / `process`, reference include/ocvstitcher.hpp:262, :592, :1141).
The C++ mirror for drop-in use from master.cpp-style code is csrc/stitcher.hpp.

There is no CPU fallback: if libpano_hip.so is missing or no GPU is present the compute calls raise.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PANO_LIB selects another build of the same C-ABI (an experimental build next to the product one); default = product
LIB_PATH = os.environ.get("PANO_LIB") or os.path.join(_HERE, "libpano_hip.so")

SPHERICAL, CYLINDRICAL = 0, 1
BANDS_NO_BLEND, BANDS_FROM_STRENGTH = -1, -2
NUM_STAGES = 4   # PANO_NUM_STAGES: warp, pyramid, blend, level-0 blend launch alone
RET_OK, RET_ERR = 0, -1
MAX_CAMS = 8

_STATUS = {0: "PANO_OK", -1: "PANO_ERR", -2: "PANO_EINVAL", -3: "PANO_ESTATE", -4: "PANO_EHIP",
           -5: "PANO_ENODEVICE", -6: "PANO_EWRAP", -7: "PANO_ENOMEM"}


class PanoError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__(f"{_STATUS.get(status, status)}: {msg}")


class Undistort(C.Structure):
    _fields_ = [("raw_w", C.c_int), ("raw_h", C.c_int), ("undist_w", C.c_int), ("undist_h", C.c_int),
                ("K", C.c_double * 9), ("dist", C.c_double * 4), ("rect", C.c_int * 4)]


class Config(C.Structure):
    _fields_ = [("num_images", C.c_int), ("width", C.c_int), ("height", C.c_int), ("projector", C.c_int),
                ("warped_image_scale", C.c_float), ("blend_strength", C.c_float), ("num_bands", C.c_int),
                ("cut", C.c_int * 4), ("device", C.c_int)]


def build(force=False):
    """Compile csrc/ for gfx950 into libpano_hip.so (in-tree)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".hip", ".cpp", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "pano.h"))
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", src_dir, "-s"] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def load_library():
    """dlopen libpano_hip.so.  torch (when installed) bundles its own libamdhip64 with the same SONAME;
    importing it first makes both share one HIP runtime so torch device pointers are valid here."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PanoError(-4, f"{LIB_PATH} is missing - run __graft_entry__.build(); there is no CPU fallback")
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    lib.pano_last_error.restype = C.c_char_p
    lib.pano_version.restype = C.c_char_p
    lib.pano_kernel_source_id.restype = C.c_char_p
    lib.pano_last_error.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def kernel_source_id():
    """16 hex digits naming the device code the loaded library was built from (pano_kernel_source_id)"""
    return load_library().pano_kernel_source_id().decode()


# every symbol include/pano.h declares (checked by tests/test_abi.py against the header text)
MAX_FRAME_SLOTS = 4  # PANO_MAX_FRAME_SLOTS

EXPORTS = [
    "pano_create", "pano_destroy", "pano_last_error", "pano_version", "pano_set_camera", "pano_verify_cameras", "pano_debug_graph_stats", "pano_debug_graphcut_dump", "pano_frame_streams",
    "pano_set_cameras_from_list", "pano_load_camera_file", "pano_get_camera", "pano_save_camera_file", "pano_prepare", "pano_get_roi", "pano_get_pano_rect",
    "pano_get_num_bands", "pano_get_feed_tile", "pano_set_cut", "pano_get_output_size", "pano_set_mask",
    "pano_build_masks_voronoi", "pano_build_masks_graphcut", "pano_refresh_masks_begin", "pano_refresh_masks_poll", "pano_refresh_masks_wait", "pano_get_mask", "pano_set_gain_map", "pano_estimate_gains", "pano_get_gain_map", "pano_set_undistort", "pano_get_new_camera_matrix", "pano_warp", "pano_warp_mask", "pano_compose",
    "pano_compose_host", "pano_host_alloc", "pano_host_free", "pano_compose_pair", "pano_set_frame_slots", "pano_select_frame_slot", "pano_feed_cameras", "pano_get_pyramid_slots", "pano_blend", "pano_feed_cameras_host", "pano_blend_host", "pano_rccl_unique_id", "pano_rccl_comm_create", "pano_rccl_comm_destroy", "pano_gather_slots", "pano_rccl_comm_count", "pano_rccl_library", "pano_stack_master", "pano_stack_finalcut", "pano_stack_master_host", "pano_stack_finalcut_host", "pano_stream_input", "pano_stream_output",
    "pano_stream_submit", "pano_stream_wait", "pano_set_profiling",
    "pano_get_stage_ms", "pano_get_stage_stats", "pano_get_warp_bytes", "pano_get_live_rect", "pano_get_source_rect", "pano_get_live_gap", "pano_get_warp_table_stats", "pano_debug_get_level", "pano_debug_get_weights",
    "pano_debug_get_canvas_weights", "pano_debug_get_canvas", "pano_probe_copy", "pano_kernel_source_id", "pano_get_exchange_stats",
]


def _vp(x):
    return C.c_void_p(int(x))


class HostBuffer:
    """page-locked host memory (pano_host_alloc) viewed as a numpy array; pano_compose_host DMAs such buffers directly"""

    def __init__(self, shape, dtype=np.uint8):
        self.lib = load_library()
        self.lib.pano_host_alloc.restype = C.c_void_p
        self.lib.pano_host_alloc.argtypes = [C.c_size_t]
        self.lib.pano_host_free.argtypes = [C.c_void_p]
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = self.lib.pano_host_alloc(n)
        if not self.ptr:
            raise PanoError(-3, "pano_host_alloc failed")
        self.array = np.ctypeslib.as_array((C.c_uint8 * n).from_address(self.ptr)).view(dtype).reshape(shape)

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.lib.pano_host_free(self.ptr)
            self.ptr = None

    __del__ = close


class Context:
    """Thin RAII wrapper of pano_ctx."""

    def __init__(self, num_images, width, height, scale=0.0, projector=SPHERICAL, num_bands=BANDS_FROM_STRENGTH,
                 blend_strength=5.0, cut=(0, 0, 0, 0), device=0):
        self.lib = load_library()
        cfg = Config()
        cfg.num_images = num_images; cfg.width = width; cfg.height = height; cfg.projector = projector
        cfg.warped_image_scale = scale; cfg.blend_strength = blend_strength; cfg.num_bands = num_bands
        for i in range(4):
            cfg.cut[i] = int(cut[i])
        cfg.device = device
        self.h = C.c_void_p()
        self.n = num_images
        self.width, self.height = width, height
        self.frame_w, self.frame_h = width, height   # raw size once a front end is set
        st = self.lib.pano_create(C.byref(cfg), C.byref(self.h))
        if st != 0:
            self.h = None
            raise PanoError(st, "pano_create failed (no GPU / bad config); there is no CPU fallback")

    def close(self):
        if getattr(self, "h", None):
            self.lib.pano_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, st):
        if st != 0:
            raise PanoError(st, (self.lib.pano_last_error(self.h) or b"").decode())

    # -- parameters
    def set_camera(self, i, K, R):
        K = np.ascontiguousarray(np.asarray(K, np.float32).reshape(9))
        R = np.ascontiguousarray(np.asarray(R, np.float32).reshape(9))
        self._ck(self.lib.pano_set_camera(self.h, i, _vp(K.ctypes.data), _vp(R.ctypes.data)))

    def verify_cameras(self, K_est, R_est, ex_thres, in_thres):
        """verifyCamParams (ocvstitcher.hpp:365-421): (ok, first failing camera or -1)"""
        K = np.ascontiguousarray(np.asarray(K_est, np.float32).reshape(-1))
        R = np.ascontiguousarray(np.asarray(R_est, np.float32).reshape(-1))
        assert K.size == 9 * self.n and R.size == 9 * self.n
        worst = C.c_int(-1)
        self.lib.pano_verify_cameras.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        st = self.lib.pano_verify_cameras(self.h, _vp(K.ctypes.data), _vp(R.ctypes.data), ex_thres, in_thres, C.addressof(worst))
        if st not in (0, -1):
            self._ck(st)
        return st == 0, worst.value

    def set_cameras_from_list(self, text):
        self._ck(self.lib.pano_set_cameras_from_list(self.h, text.encode()))

    def load_camera_file(self, path):
        self._ck(self.lib.pano_load_camera_file(self.h, os.fsencode(path)))

    def get_camera(self, i):
        """(K[9], R[9], warped_image_scale) as float32 values"""
        K = (C.c_float * 9)(); R = (C.c_float * 9)(); sc = C.c_float()
        self._ck(self.lib.pano_get_camera(self.h, i, K, R, C.byref(sc)))
        return list(K), list(R), sc.value

    def save_camera_file(self, path):
        self._ck(self.lib.pano_save_camera_file(self.h, os.fsencode(path)))

    def set_undistort(self, cam, raw_wh, undist_wh, K, dist, rect):
        u = Undistort()
        u.raw_w, u.raw_h = raw_wh; u.undist_w, u.undist_h = undist_wh
        for i in range(9):
            u.K[i] = float(K[i])
        for i in range(4):
            u.dist[i] = float(dist[i]); u.rect[i] = int(rect[i])
        self._ck(self.lib.pano_set_undistort(self.h, cam, C.byref(u)))
        self.frame_w, self.frame_h = raw_wh

    def new_camera_matrix(self, cam):
        out = (C.c_double * 9)()
        self._ck(self.lib.pano_get_new_camera_matrix(self.h, cam, out)); return list(out)

    def prepare(self):
        self._ck(self.lib.pano_prepare(self.h))

    def roi(self, i):
        r = (C.c_int * 4)(); self._ck(self.lib.pano_get_roi(self.h, i, r)); return tuple(r)

    def pano_rect(self):
        r = (C.c_int * 4)(); self._ck(self.lib.pano_get_pano_rect(self.h, r)); return tuple(r)

    def num_bands(self):
        v = C.c_int(); self._ck(self.lib.pano_get_num_bands(self.h, C.byref(v))); return v.value

    def feed_tile(self, i):
        r = (C.c_int * 4)(); t = (C.c_int * 4)()
        self._ck(self.lib.pano_get_feed_tile(self.h, i, r, t)); return tuple(r), tuple(t)

    def set_cut(self, cut):
        r = (C.c_int * 4)(*[int(v) for v in cut]); self._ck(self.lib.pano_set_cut(self.h, r))

    def output_size(self):
        w = C.c_int(); h = C.c_int()
        self._ck(self.lib.pano_get_output_size(self.h, C.byref(w), C.byref(h))); return w.value, h.value

    # -- masks / gains
    def set_mask(self, i, mask):
        mask = np.ascontiguousarray(mask, np.uint8)
        self._ck(self.lib.pano_set_mask(self.h, i, _vp(mask.ctypes.data), mask.shape[1], mask.shape[0],
                                        C.c_size_t(mask.strides[0])))

    def build_masks_voronoi(self):
        self._ck(self.lib.pano_build_masks_voronoi(self.h))

    def build_masks_graphcut(self, frames):
        """ocvStitcher::updateMask with GraphCutSeamFinder(COST_COLOR) (ocvstitcher.hpp:1218-1261) on stitcher-size frames"""
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        assert len(frames) == self.n
        ptrs = (C.c_void_p * self.n)(*[f.ctypes.data for f in frames])
        strides = (C.c_size_t * self.n)(*[f.strides[0] for f in frames])
        self._ck(self.lib.pano_build_masks_graphcut(self.h, ptrs, strides))

    def refresh_masks_begin(self, frames):
        """updateMask beside the frame loop (pano_refresh_masks_begin): the frames are consumed before this returns"""
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        assert len(frames) == self.n
        ptrs = (C.c_void_p * self.n)(*[f.ctypes.data for f in frames])
        strides = (C.c_size_t * self.n)(*[f.strides[0] for f in frames])
        self._ck(self.lib.pano_refresh_masks_begin(self.h, ptrs, strides))

    def refresh_masks_poll(self):
        """True the one time the refreshed masks get installed"""
        done = C.c_int(0)
        self._ck(self.lib.pano_refresh_masks_poll(self.h, C.byref(done)))
        return bool(done.value)

    def refresh_masks_wait(self):
        self._ck(self.lib.pano_refresh_masks_wait(self.h))

    def get_mask(self, i):
        r = self.roi(i)
        m = np.empty((r[3], r[2]), np.uint8)
        self._ck(self.lib.pano_get_mask(self.h, i, _vp(m.ctypes.data), C.c_size_t(r[2]))); return m

    def set_gain_map(self, i, gain):
        if gain is None:
            self._ck(self.lib.pano_set_gain_map(self.h, i, None, 0, 0)); return
        g = np.ascontiguousarray(gain, np.float32)
        self._ck(self.lib.pano_set_gain_map(self.h, i, _vp(g.ctypes.data), g.shape[1], g.shape[0]))

    def estimate_gains(self, frames, block=(32, 32)):
        """ExposureCompensator(GAIN_BLOCKS)::feed as initSeam runs it (ocvstitcher.hpp:981-1032) on stitcher-size
        frames; the maps are installed for the next compose and returned"""
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        assert len(frames) == self.n
        ptrs = (C.c_void_p * self.n)(*[f.ctypes.data for f in frames])
        strides = (C.c_size_t * self.n)(*[f.strides[0] for f in frames])
        self._ck(self.lib.pano_estimate_gains(self.h, ptrs, strides, int(block[0]), int(block[1])))
        return [self.gain_map(i) for i in range(self.n)]

    def gain_map(self, i):
        gw, gh = C.c_int(0), C.c_int(0)
        self._ck(self.lib.pano_get_gain_map(self.h, i, None, C.byref(gw), C.byref(gh)))
        if gw.value == 0:
            return None
        g = np.empty((gh.value, gw.value), np.float32)
        self._ck(self.lib.pano_get_gain_map(self.h, i, _vp(g.ctypes.data), C.byref(gw), C.byref(gh)))
        return g

    # -- per-frame, host buffers (cv::Mat in / cv::Mat out)
    def compose_host(self, frames, out=None):
        """process(imgs, ret): host frames (rows may be strided views) in, host panorama out.  `out`: a (h, w, 3) uint8 array
        to write into (rows may be strided) instead of a fresh one"""
        frames = [f if (f.dtype == np.uint8 and f.ndim == 3 and f.strides[1:] == (3, 1)) else np.ascontiguousarray(f, np.uint8)
                  for f in frames]
        assert len(frames) == self.n
        for f in frames:
            if f.shape != (self.frame_h, self.frame_w, 3):
                raise PanoError(-2, "frame shape")
        w, h = self.output_size()
        if out is None:
            out = np.empty((h, w, 3), np.uint8)
        elif out.shape != (h, w, 3) or out.dtype != np.uint8 or out.strides[1:] != (3, 1):
            raise PanoError(-2, "output shape")
        ptrs = (C.c_void_p * self.n)(*[f.ctypes.data for f in frames])
        strides = (C.c_size_t * self.n)(*[f.strides[0] for f in frames])
        self._ck(self.lib.pano_compose_host(self.h, ptrs, strides, _vp(out.ctypes.data), C.c_size_t(out.strides[0])))
        return out

    # -- per-frame, device pointers (ints); stream = hipStream_t as int (0 = null stream)
    def compose(self, d_frames, strides, d_out, out_stride, stream=0):
        ptrs = (C.c_void_p * self.n)(*[int(p) for p in d_frames])
        st = (C.c_size_t * self.n)(*[int(s) for s in strides])
        self._ck(self.lib.pano_compose(self.h, ptrs, st, _vp(d_out), C.c_size_t(out_stride), _vp(stream)))

    def compose_pair(self, other, d_frames_a, strides_a, d_out_a, out_stride_a, d_frames_b, strides_b, d_out_b, out_stride_b,
                     stream=0):
        """both stitchers (upper / lower group) in one launch sequence"""
        pa = (C.c_void_p * self.n)(*[int(p) for p in d_frames_a]); sa = (C.c_size_t * self.n)(*[int(s) for s in strides_a])
        pb = (C.c_void_p * other.n)(*[int(p) for p in d_frames_b]); sb = (C.c_size_t * other.n)(*[int(s) for s in strides_b])
        self._ck(self.lib.pano_compose_pair(self.h, other.h, pa, sa, _vp(d_out_a), C.c_size_t(out_stride_a), pb, sb,
                                            _vp(d_out_b), C.c_size_t(out_stride_b), _vp(stream)))

    def feed_cameras(self, cam_bits, d_frames, strides, stream=0):
        ptrs = (C.c_void_p * self.n)(*[int(p) for p in d_frames])
        st = (C.c_size_t * self.n)(*[int(s) for s in strides])
        self._ck(self.lib.pano_feed_cameras(self.h, C.c_uint(cam_bits), ptrs, st, _vp(stream)))

    def blend(self, d_out, out_stride, stream=0):
        self._ck(self.lib.pano_blend(self.h, _vp(d_out), C.c_size_t(out_stride), _vp(stream)))

    def pyramid_slots(self):
        base = C.c_void_p(); sz = C.c_size_t()
        self._ck(self.lib.pano_get_pyramid_slots(self.h, C.byref(base), C.byref(sz))); return base.value, sz.value

    def warp(self, i, d_src, src_stride, d_dst, dst_stride, stream=0):
        self._ck(self.lib.pano_warp(self.h, i, _vp(d_src), C.c_size_t(src_stride), _vp(d_dst), C.c_size_t(dst_stride),
                                    _vp(stream)))

    def warp_mask(self, i, d_dst, dst_stride, stream=0):
        self._ck(self.lib.pano_warp_mask(self.h, i, _vp(d_dst), C.c_size_t(dst_stride), _vp(stream)))

    # -- streaming slots: numpy views of the library's pinned buffers
    def frame_streams(self, n):
        """(n hipStream_t handles as ints - probed to run side by side -, how many sit on a hardware queue of their own)"""
        arr = (C.c_void_p * n)(); d = C.c_int()
        self.lib.pano_frame_streams.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self._ck(self.lib.pano_frame_streams(self.h, n, arr, C.addressof(d)))
        return [int(arr[i]) for i in range(n)], d.value

    def graphcut_dump(self, path):
        self.lib.pano_debug_graphcut_dump.argtypes = [C.c_void_p, C.c_char_p]
        self._ck(self.lib.pano_debug_graphcut_dump(self.h, os.fsencode(path) if path else None))

    def graph_stats(self):
        """(graphs held or -1 when replay is off, hipGraphLaunch calls so far)"""
        n = C.c_int(); r = C.c_uint64()
        self.lib.pano_debug_graph_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._ck(self.lib.pano_debug_graph_stats(self.h, C.addressof(n), C.addressof(r)))
        return n.value, r.value

    def stream_input(self, slot, cam):
        p = C.c_void_p(); st = C.c_size_t()
        self._ck(self.lib.pano_stream_input(self.h, slot, cam, C.byref(p), C.byref(st)))
        buf = (C.c_uint8 * (st.value * self.frame_h)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(self.frame_h, st.value)[:, :self.frame_w * 3].reshape(
            self.frame_h, self.frame_w, 3)

    def stream_output(self, slot):
        p = C.c_void_p(); st = C.c_size_t()
        self._ck(self.lib.pano_stream_output(self.h, slot, C.byref(p), C.byref(st)))
        w, h = self.output_size()
        buf = (C.c_uint8 * (st.value * h)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(h, st.value)[:, :w * 3].reshape(h, w, 3)

    def stream_submit(self, slot):
        self._ck(self.lib.pano_stream_submit(self.h, slot))

    def stream_wait(self, slot):
        self._ck(self.lib.pano_stream_wait(self.h, slot))

    # -- caller-side assembly (device pointers)
    def stack_master(self, d_up, up_w, up_h, up_stride, d_down, dw, dh, d_stride, d_out, out_stride, stream=0):
        self._ck(self.lib.pano_stack_master(self.h, _vp(d_up), up_w, up_h, C.c_size_t(up_stride), _vp(d_down), dw, dh,
                                            C.c_size_t(d_stride), _vp(d_out), C.c_size_t(out_stride), _vp(stream)))

    def stack_finalcut(self, d_up, up_w, up_h, up_stride, d_down, dw, dh, d_stride, finalcut, d_out, out_stride, stream=0):
        self._ck(self.lib.pano_stack_finalcut(self.h, _vp(d_up), up_w, up_h, C.c_size_t(up_stride), _vp(d_down), dw, dh,
                                              C.c_size_t(d_stride), int(finalcut), _vp(d_out), C.c_size_t(out_stride),
                                              _vp(stream)))

    def feed_cameras_host(self, cam_bits, frames):
        """pano_feed_cameras with host frames (list indexed by camera; entries of unselected cameras may be None)"""
        keep = [np.ascontiguousarray(f, np.uint8) if f is not None else None for f in frames]
        ptrs = (C.c_void_p * self.n)(*[(f.ctypes.data if f is not None else 0) for f in keep])
        st = (C.c_size_t * self.n)(*[(f.strides[0] if f is not None else 0) for f in keep])
        self._ck(self.lib.pano_feed_cameras_host(self.h, int(cam_bits), ptrs, st))

    def blend_host(self):
        w, h = self.output_size()
        out = np.empty((h, w, 3), np.uint8)
        self._ck(self.lib.pano_blend_host(self.h, _vp(out.ctypes.data), C.c_size_t(out.strides[0])))
        return out

    # -- the sharded exchange over RCCL (one process per GPU)
    @staticmethod
    def rccl_unique_id():
        """128 bytes from ncclGetUniqueId, to be handed to every rank (e.g. torch.distributed.broadcast_object_list)"""
        buf = (C.c_char * 128)()
        st = load_library().pano_rccl_unique_id(buf)
        if st != 0:
            raise PanoError(st, "pano_rccl_unique_id (is librccl.so there?)")
        return bytes(buf)

    def rccl_comm_create(self, unique_id, world, rank):
        comm = C.c_void_p()
        self._ck(self.lib.pano_rccl_comm_create(self.h, C.c_char_p(unique_id), int(world), int(rank), C.byref(comm)))
        return comm

    def rccl_comm_destroy(self, comm):
        """returns the status (a test double reports sends that never met their receive here)"""
        return int(self.lib.pano_rccl_comm_destroy(comm))

    def rccl_comm_count(self, comm):
        """ncclCommCount: the ranks RCCL itself says the communicator spans"""
        n = C.c_int(0)
        self._ck(self.lib.pano_rccl_comm_count(comm, C.byref(n)))
        return n.value

    @staticmethod
    def rccl_library():
        """the name the RCCL library was opened by (PANO_RCCL_LIB, or the system's librccl.so); '' when none loads"""
        lib = load_library()
        lib.pano_rccl_library.restype = C.c_char_p
        return (lib.pano_rccl_library() or b"").decode()

    def gather_slots(self, comm, rank, root, owner_rank, stream=0):
        arr = (C.c_int * self.n)(*[int(r) for r in owner_rank])
        self._ck(self.lib.pano_gather_slots(self.h, comm, int(rank), int(root), arr, _vp(stream)))

    def stack_master_host(self, up, down):
        """master.cpp:321-326 on host arrays"""
        up = np.ascontiguousarray(up, np.uint8); down = np.ascontiguousarray(down, np.uint8)
        out = np.empty((2 * down.shape[0], down.shape[1], 3), np.uint8)
        self._ck(self.lib.pano_stack_master_host(self.h, _vp(up.ctypes.data), up.shape[1], up.shape[0], C.c_size_t(up.strides[0]),
                                                 _vp(down.ctypes.data), down.shape[1], down.shape[0], C.c_size_t(down.strides[0]),
                                                 _vp(out.ctypes.data), C.c_size_t(out.strides[0])))
        return out

    def stack_finalcut_host(self, up, down, finalcut):
        """panocamimpl.cpp:354-360 on host arrays"""
        up = np.ascontiguousarray(up, np.uint8); down = np.ascontiguousarray(down, np.uint8)
        w, h = min(up.shape[1], down.shape[1]), min(up.shape[0], down.shape[0]) - 2 * finalcut
        out = np.empty((2 * h, w, 3), np.uint8)
        self._ck(self.lib.pano_stack_finalcut_host(self.h, _vp(up.ctypes.data), up.shape[1], up.shape[0], C.c_size_t(up.strides[0]),
                                                   _vp(down.ctypes.data), down.shape[1], down.shape[0], C.c_size_t(down.strides[0]),
                                                   int(finalcut), _vp(out.ctypes.data), C.c_size_t(out.strides[0])))
        return out

    # -- measurement
    def set_profiling(self, on):
        self._ck(self.lib.pano_set_profiling(self.h, int(bool(on))))

    def stage_ms(self):
        ms = (C.c_float * NUM_STAGES)(); self._ck(self.lib.pano_get_stage_ms(self.h, ms)); return tuple(ms)

    def stage_stats(self, reset=True):
        ms = (C.c_double * NUM_STAGES)(); n = (C.c_uint64 * NUM_STAGES)()
        self._ck(self.lib.pano_get_stage_stats(self.h, ms, n, int(bool(reset)))); return tuple(ms), tuple(n)

    def warp_bytes(self):
        a = C.c_uint64(); b = C.c_uint64()
        self._ck(self.lib.pano_get_warp_bytes(self.h, C.byref(a), C.byref(b))); return a.value, b.value

    def source_rect(self, i):
        """(byte x0, row y0, byte width, rows) of camera i's frame that the warp reads with the present masks: what the host
        entries upload"""
        r = (C.c_int * 4)()
        self._ck(self.lib.pano_get_source_rect(self.h, int(i), r)); return tuple(r)

    def live_rect(self, i, level):
        r = (C.c_int * 4)()
        self._ck(self.lib.pano_get_live_rect(self.h, int(i), int(level), r)); return tuple(r)

    def live_gap(self, i, level):
        """(first dead column, dead column count) inside live_rect: the middle of a +-pi straddler's tile, (0, 0) = none"""
        g = (C.c_int * 2)()
        self._ck(self.lib.pano_get_live_gap(self.h, i, level, g)); return tuple(g)

    def set_frame_slots(self, n):
        self._ck(self.lib.pano_set_frame_slots(self.h, int(n)))

    def select_frame_slot(self, k):
        self._ck(self.lib.pano_select_frame_slot(self.h, int(k)))

    def warp_table_stats(self):
        a = C.c_uint64(); nb = C.c_uint64(); nf = C.c_uint64()
        self._ck(self.lib.pano_get_warp_table_stats(self.h, C.byref(a), C.byref(nb), C.byref(nf)))
        return {"table_bytes": a.value, "blocks": nb.value, "blocks_checked": nf.value}

    def exchange_stats(self):
        """what pano_gather_slots moves: {"packed_bytes_per_camera": [...], "slot_bytes", "bytes_moved"}"""
        per = (C.c_uint64 * self.n)(); sb = C.c_uint64(); mv = C.c_uint64()
        self._ck(self.lib.pano_get_exchange_stats(self.h, per, C.byref(sb), C.byref(mv)))
        return {"packed_bytes_per_camera": [int(v) for v in per], "slot_bytes": int(sb.value), "bytes_moved": int(mv.value)}

    PROBE_COPY_F4, PROBE_COPY_K1_SHAPE, PROBE_COPY_F4_FLAT = 0, 1, 2

    def probe_copy(self, kind, units, sets=1, reps=50):
        """device-copy ceiling measured by the library (pano_probe_copy): {"GBps", "us_per_launch", "bytes_per_launch"}"""
        g = C.c_double(); us = C.c_double(); b = C.c_uint64()
        self._ck(self.lib.pano_probe_copy(self.h, int(kind), C.c_uint64(int(units)), int(sets), int(reps), C.byref(g), C.byref(us), C.byref(b)))
        return {"GBps": g.value, "us_per_launch": us.value, "bytes_per_launch": b.value}

    # -- stage inspection
    def debug_level(self, i, level):
        w = C.c_int(); h = C.c_int()
        self._ck(self.lib.pano_debug_get_level(self.h, i, level, None, C.byref(w), C.byref(h)))
        a = np.empty((h.value, w.value, 3), np.int16)
        self._ck(self.lib.pano_debug_get_level(self.h, i, level, _vp(a.ctypes.data), C.byref(w), C.byref(h)))
        return a

    def debug_weights(self, i, level):
        w = C.c_int(); h = C.c_int()
        self._ck(self.lib.pano_debug_get_weights(self.h, i, level, None, C.byref(w), C.byref(h)))
        a = np.empty((h.value, w.value), np.float32)
        self._ck(self.lib.pano_debug_get_weights(self.h, i, level, _vp(a.ctypes.data), C.byref(w), C.byref(h)))
        return a

    def debug_canvas_weights(self, level):
        w = C.c_int(); h = C.c_int()
        self._ck(self.lib.pano_debug_get_canvas_weights(self.h, level, None, C.byref(w), C.byref(h)))
        a = np.empty((h.value, w.value), np.float32)
        self._ck(self.lib.pano_debug_get_canvas_weights(self.h, level, _vp(a.ctypes.data), C.byref(w), C.byref(h)))
        return a

    def debug_canvas(self, level):
        w = C.c_int(); h = C.c_int()
        self._ck(self.lib.pano_debug_get_canvas(self.h, level, None, C.byref(w), C.byref(h)))
        a = np.empty((h.value, w.value, 3), np.int16)
        self._ck(self.lib.pano_debug_get_canvas(self.h, level, _vp(a.ctypes.data), C.byref(w), C.byref(h)))
        return a


class Stitcher:
    """Python mirror of `ocvStitcher` (reference include/ocvstitcher.hpp:254-1306) for the compose path.

    init(cfg)          <- init(yamlPath) (:262): size, num_images, blend strength, default cams + cut
    calibration(imgs)  <- calibration(imgs) (:592): K/R are fixed, so this is initSeam's mask half (:975-1139)
                          with the Voronoi seam finder; returns RET_OK / RET_ERR
    process(imgs)      <- process(imgs, ret) (:1141): returns the cut BGR panorama
    """

    def __init__(self):
        self.ctx = None

    def init(self, cfg):
        """cfg: dict with outPutWidth, outPutHeight, num_images, stitcherBlenderStrength, cams (18N+1 list or
        comma string), optional cut, projector, num_bands, device."""
        try:
            cams = cfg["cams"]
            text = cams if isinstance(cams, str) else ",".join(repr(float(v)) for v in cams)
            self.ctx = Context(int(cfg["num_images"]), int(cfg["outPutWidth"]), int(cfg["outPutHeight"]),
                               projector=int(cfg.get("projector", SPHERICAL)),
                               num_bands=int(cfg.get("num_bands", BANDS_FROM_STRENGTH)),
                               blend_strength=float(cfg.get("stitcherBlenderStrength", 5.0)),
                               cut=tuple(cfg.get("cut", (0, 0, 0, 0))), device=int(cfg.get("device", 0)))
            self.ctx.set_cameras_from_list(text)
            self.ctx.prepare()
        except (PanoError, KeyError, ValueError):
            self.ctx = None
            return RET_ERR
        return RET_OK

    def calibration(self, imgs=None):
        if self.ctx is None:
            return RET_ERR
        try:
            self.ctx.build_masks_voronoi()
        except PanoError:
            return RET_ERR
        return RET_OK

    def process(self, imgs):
        return self.ctx.compose_host(imgs)
