"""Second, independent restatement (NumPy, array-at-a-time) of the OpenCV-3.4 semantics in
pano_oracle.c.  TEST INFRASTRUCTURE ONLY.  Its job is to cross-check the C oracle: two restatements
written differently that agree bit for bit is the only pin available (PARITY UNPINNED, see
pano_oracle.h).  Trigonometry goes through the host libm's sinf/cosf/atan2f/acosf via ctypes so that
the f32 bits match a C build on the same machine.

Follows: stitching/detail/warpers_inl.hpp (projectors, buildMaps), imgproc/imgwarp.cpp (remap fixed
point), imgproc/pyramids.cpp, stitching/src/blenders.cpp; reference call sites
include/ocvstitcher.hpp:1171 (warp), :1180 (16S), :1198-1207 (prepare/feed/blend).
"""
import ctypes
import ctypes.util
import math

import numpy as np

f32 = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library("m"))
for _n in ("sinf", "cosf", "acosf", "sqrtf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
_libm.atan2f.restype = ctypes.c_float
_libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
PI_F = f32(math.pi)


def _vec(fn, a):
    a = np.asarray(a, f32)
    return np.array([fn(float(v)) for v in a.ravel()], f32).reshape(a.shape)


class Projector:
    def __init__(self, kind, scale, K, R):
        K = np.asarray(K, f32).reshape(3, 3)
        R = np.asarray(R, f32).reshape(3, 3)
        self.kind, self.scale, self.k, self.rinv = kind, f32(scale), K, R.T.copy()
        Kd = K.astype(np.float64)
        # cv::invert 3x3: double cofactors -> float
        det = (Kd[0, 0] * (Kd[1, 1] * Kd[2, 2] - Kd[1, 2] * Kd[2, 1]) - Kd[0, 1] * (Kd[1, 0] * Kd[2, 2] - Kd[1, 2] * Kd[2, 0])
               + Kd[0, 2] * (Kd[1, 0] * Kd[2, 1] - Kd[1, 1] * Kd[2, 0]))
        d = 1.0 / det
        cof = np.empty((3, 3))
        for i in range(3):
            for j in range(3):
                r = [a for a in range(3) if a != j]
                c = [a for a in range(3) if a != i]
                # adjugate entry (i, j) = (-1)^(i+j) * minor(j, i)
                m = Kd[r[0], c[0]] * Kd[r[1], c[1]] - Kd[r[0], c[1]] * Kd[r[1], c[0]]
                cof[i, j] = m if (i + j) % 2 == 0 else -m
        kinv = (cof * d).astype(f32)
        self.r_kinv = self._mm(R, kinv)
        self.k_rinv = self._mm(K, self.rinv)

    @staticmethod
    def _mm(a, b):
        out = np.empty((3, 3), f32)
        for i in range(3):
            for j in range(3):
                out[i, j] = f32(f32(a[i, 0] * b[0, j]) + f32(a[i, 1] * b[1, j])) + f32(a[i, 2] * b[2, j])
        return out

    def forward(self, x, y):
        x = np.asarray(x, f32); y = np.asarray(y, f32)
        m = self.r_kinv
        x_ = m[0, 0] * x + m[0, 1] * y + m[0, 2]
        y_ = m[1, 0] * x + m[1, 1] * y + m[1, 2]
        z_ = m[2, 0] * x + m[2, 1] * y + m[2, 2]
        u = self.scale * np.array([_libm.atan2f(float(a), float(b)) for a, b in zip(x_.ravel(), z_.ravel())], f32).reshape(x_.shape)
        if self.kind == 0:
            w = y_ / np.sqrt(x_ * x_ + y_ * y_ + z_ * z_, dtype=f32)
            w = np.where(w == w, w, f32(0))
            v = self.scale * (PI_F - _vec(_libm.acosf, w))
        else:
            v = self.scale * y_ / np.sqrt(x_ * x_ + z_ * z_, dtype=f32)
        return u.astype(f32), v.astype(f32)

    def roi(self, w, h):
        xs = np.arange(w, dtype=f32); ys = np.arange(h, dtype=f32)
        pts = [(xs, np.zeros(w, f32)), (xs, np.full(w, h - 1, f32)), (np.zeros(h, f32), ys), (np.full(h, w - 1, f32), ys)]
        us, vs = zip(*[self.forward(a, b) for a, b in pts])
        u = np.concatenate(us); v = np.concatenate(vs)
        tl = [int(u.min()), int(v.min())]; br = [int(u.max()), int(v.max())]   # C truncation toward zero
        if self.kind == 0:
            tlf = [f32(tl[0]), f32(tl[1])]; brf = [f32(br[0]), f32(br[1])]
            for sign, pole in ((1, f32(math.pi * float(self.scale))), (-1, f32(0))):
                x, y, z = self.rinv[0, 1], f32(sign) * self.rinv[1, 1], self.rinv[2, 1]
                if y > 0:
                    x_ = (self.k[0, 0] * x + self.k[0, 1] * y) / z + self.k[0, 2]
                    y_ = self.k[1, 1] * y / z + self.k[1, 2]
                    if 0 < x_ < w and 0 < y_ < h:
                        tlf = [min(tlf[0], f32(0)), min(tlf[1], pole)]
                        brf = [max(brf[0], f32(0)), max(brf[1], pole)]
            tl = [int(tlf[0]), int(tlf[1])]; br = [int(brf[0]), int(brf[1])]
        return tl[0], tl[1], br[0] - tl[0] + 1, br[1] - tl[1] + 1

    def maps(self, w, h):
        """buildMaps through the separable factors of mapBackward."""
        x0, y0, rw, rh = self.roi(w, h)
        u = (np.arange(rw, dtype=f32) + f32(x0)) / self.scale
        v = (np.arange(rh, dtype=f32) + f32(y0)) / self.scale
        su, cu = _vec(_libm.sinf, u), _vec(_libm.cosf, u)
        if self.kind == 0:
            sv, cv = _vec(_libm.sinf, PI_F - v), _vec(_libm.cosf, PI_F - v)
            x_ = sv[:, None] * su[None, :]
            y_ = np.broadcast_to(cv[:, None], (rh, rw))
            z_ = sv[:, None] * cu[None, :]
        else:
            x_ = np.broadcast_to(su[None, :], (rh, rw))
            y_ = np.broadcast_to(v[:, None], (rh, rw))
            z_ = np.broadcast_to(cu[None, :], (rh, rw))
        m = self.k_rinv
        X = (m[0, 0] * x_ + m[0, 1] * y_) + m[0, 2] * z_
        Y = (m[1, 0] * x_ + m[1, 1] * y_) + m[1, 2] * z_
        Z = (m[2, 0] * x_ + m[2, 1] * y_) + m[2, 2] * z_
        pos = Z > 0
        Zs = np.where(pos, Z, f32(1))
        X = np.where(pos, X / Zs, f32(-1)).astype(f32)
        Y = np.where(pos, Y / Zs, f32(-1)).astype(f32)
        return X, Y


def _cvround(a):
    r = np.rint(a.astype(np.float64))          # half-to-even
    bad = ~((a >= -2147483648.0) & (a < 2147483648.0))
    return np.where(bad, -2147483648, r).astype(np.int64)


def _reflect(p, n):
    if n == 1:
        return np.zeros_like(p)
    q = np.mod(p, 2 * n)
    return np.where(q < n, q, 2 * n - 1 - q)


def _reflect101(p, n):
    if n == 1:
        return np.zeros_like(p)
    q = np.mod(p, 2 * n - 2)
    return np.where(q < n, q, 2 * n - 2 - q)


def remap_linear_reflect(src, X, Y):
    h, w = src.shape[:2]
    sx = _cvround(X * f32(32)); sy = _cvround(Y * f32(32))
    a = sx & 31; b = sy & 31
    ix = np.clip(sx >> 5, -32768, 32767); iy = np.clip(sy >> 5, -32768, 32767)
    x0 = _reflect(ix, w); x1 = _reflect(ix + 1, w); y0 = _reflect(iy, h); y1 = _reflect(iy + 1, h)
    s = src.astype(np.int64)
    w00 = ((32 - a) * (32 - b) * 32)[..., None]; w01 = (a * (32 - b) * 32)[..., None]
    w10 = ((32 - a) * b * 32)[..., None]; w11 = (a * b * 32)[..., None]
    acc = s[y0, x0] * w00 + s[y0, x1] * w01 + s[y1, x0] * w10 + s[y1, x1] * w11
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def pyr_down_16s(a):
    a = a.astype(np.int64)
    h, w = a.shape[:2]
    dh, dw = (h + 1) // 2, (w + 1) // 2
    k = np.array([1, 4, 6, 4, 1], np.int64)
    xi = np.stack([_reflect101(2 * np.arange(dw) + t - 2, w) for t in range(5)])
    yi = np.stack([_reflect101(2 * np.arange(dh) + t - 2, h) for t in range(5)])
    hor = sum(k[t] * a[:, xi[t]] for t in range(5))
    ver = sum(k[t] * hor[yi[t]] for t in range(5))
    return np.clip((ver + 128) >> 8, -32768, 32767).astype(np.int16)


def pyr_up_16s(a):
    a = a.astype(np.int64)

    def up1(s, axis):
        s = np.moveaxis(s, axis, 0)
        n = s.shape[0]
        prev = s[[1 if n > 1 else 0] + list(range(n - 1))]
        nxt = s[list(range(1, n)) + [n - 1]]
        out = np.empty((2 * n,) + s.shape[1:], np.int64)
        out[0::2] = prev + 6 * s + nxt
        out[1::2] = 4 * (s + nxt)
        return np.moveaxis(out, 0, axis)

    r = up1(up1(a, 1), 0)
    return np.clip((r + 32) >> 6, -32768, 32767).astype(np.int16)


def pyr_down_32f(a):
    a = a.astype(f32)
    h, w = a.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    xi = [_reflect101(2 * np.arange(dw) + t - 2, w) for t in range(5)]
    yi = [_reflect101(2 * np.arange(dh) + t - 2, h) for t in range(5)]
    hor = a[:, xi[2]] * f32(6) + (a[:, xi[1]] + a[:, xi[3]]) * f32(4) + a[:, xi[0]] + a[:, xi[4]]
    ver = hor[yi[2]] * f32(6) + (hor[yi[1]] + hor[yi[3]]) * f32(4) + hor[yi[0]] + hor[yi[4]]
    return (ver * f32(1.0 / 256)).astype(f32)


def _trunc16(x):
    """static_cast<short>(float) then short storage: truncate toward zero"""
    return np.trunc(x).astype(np.int64)


def _wrap16(x):
    return ((x + 32768) % 65536) - 32768


class MultiBand:
    def __init__(self, bands):
        self.req = bands

    def prepare(self, corners, sizes):
        tl = np.min(np.array(corners), 0); br = np.max(np.array(corners) + np.array(sizes), 0)
        self.final = (int(tl[0]), int(tl[1]), int(br[0] - tl[0]), int(br[1] - tl[1]))
        nb = min(self.req, int(math.ceil(math.log(max(self.final[2:])) / math.log(2.0))))
        self.nb = nb
        m = 1 << nb
        W = self.final[2] + (m - self.final[2] % m) % m
        H = self.final[3] + (m - self.final[3] % m) % m
        self.roi = (self.final[0], self.final[1], W, H)
        self.lap = [np.zeros((H >> i, W >> i, 3), np.int64) for i in range(nb + 1)]
        self.wgt = [np.zeros((H >> i, W >> i), f32) for i in range(nb + 1)]

    def feed(self, img, mask, tl):
        nb, R = self.nb, self.roi
        h, w = mask.shape
        m = 1 << nb
        gap = 3 * m
        tlx = max(R[0], tl[0] - gap); tly = max(R[1], tl[1] - gap)
        brx = min(R[0] + R[2], tl[0] + w + gap); bry = min(R[1] + R[3], tl[1] + h + gap)
        tlx = R[0] + (((tlx - R[0]) >> nb) << nb); tly = R[1] + (((tly - R[1]) >> nb) << nb)
        W = brx - tlx; H = bry - tly
        W += (m - W % m) % m; H += (m - H % m) % m
        brx, bry = tlx + W, tly + H
        dx = max(brx - (R[0] + R[2]), 0); dy = max(bry - (R[1] + R[3]), 0)
        tlx -= dx; brx -= dx; tly -= dy; bry -= dy
        top, left = tl[1] - tly, tl[0] - tlx
        ys = _reflect(np.arange(H) - top, h); xs = _reflect(np.arange(W) - left, w)
        g = [img.astype(np.int16)[ys][:, xs]]
        for i in range(nb):
            g.append(pyr_down_16s(g[i]))
        lap = [np.clip(g[i].astype(np.int64) - pyr_up_16s(g[i + 1]).astype(np.int64), -32768, 32767) for i in range(nb)]
        lap.append(g[nb].astype(np.int64))
        wm = np.zeros((H, W), f32)
        wm[top:top + h, left:left + w] = mask.astype(f32) * f32(1.0 / 255.0)
        ws = [wm]
        for i in range(nb):
            ws.append(pyr_down_32f(ws[i]))
        x0, y0, x1, y1 = tlx - R[0], tly - R[1], brx - R[0], bry - R[1]
        for i in range(nb + 1):
            prod = _trunc16(lap[i].astype(f32) * ws[i][..., None])
            self.lap[i][y0:y1, x0:x1] = _wrap16(self.lap[i][y0:y1, x0:x1] + prod)
            self.wgt[i][y0:y1, x0:x1] += ws[i]
            x0 //= 2; y0 //= 2; x1 //= 2; y1 //= 2
        self.last_tile = ((tlx - R[0], tly - R[1], W, H), (top, bry - tl[1] - h, left, brx - tl[0] - w))

    def blend(self):
        eps = f32(1e-5)
        nb = self.nb
        lv = [_trunc16(self.lap[i].astype(f32) / (self.wgt[i] + eps)[..., None]) for i in range(nb + 1)]
        for i in range(nb, 0, -1):
            lv[i - 1] = np.clip(lv[i - 1] + pyr_up_16s(lv[i].astype(np.int16)).astype(np.int64), -32768, 32767)
        fw, fh = self.final[2], self.final[3]
        mask = self.wgt[0][:fh, :fw] > eps
        out = np.where(mask[..., None], lv[0][:fh, :fw], 0).astype(np.int16)
        return out, (mask * 255).astype(np.uint8)
