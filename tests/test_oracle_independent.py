"""Evidence for the warp that shares no code and no source with the restatement ("every parity claim is HIP ==
the builder's restatement").  cv::detail::RotationWarper::warp is, mathematically, a bilinear resampling of the frame along the
inverse spherical / cylindrical projection.  Here that is evaluated from the geometry alone - float64, no quantisation, no fixed
point, straight from K, R and the scale: the direction of panorama pixel (u, v), its image point K R^-1 d, the real-valued bilinear
sample with BORDER_REFLECT - and compared with the oracle's warp (OpenCV's fixed-point remap as restated: coordinates rounded to 1/32
pixel, Q15 weights, + 16384 >> 15).  The two may differ by the rounding of the result (0.5) plus what moving the sample point by at most
1/64 pixel in x and in y can change: (1/64 + 1/64) x the range of the four taps.  EVERY pixel must lie inside that bound - a wrong
projector, a transposed rotation, an off-by-one in the ROI, a wrong border rule or a wrong weight order each break it at once."""
import numpy as np
import pytest


def _reflect(p, n):
    p = np.where(p < 0, -p - 1, p)
    return np.where(p >= n, 2 * n - 1 - p, p)


def _ideal_warp(img, K, R, scale, roi, cylindrical):
    x0, y0, w, h = roi
    U, V = np.meshgrid((x0 + np.arange(w)) / scale, (y0 + np.arange(h)) / scale)
    if cylindrical:
        dx, dy, dz = np.sin(U), V, np.cos(U)
    else:
        sv = np.sin(np.pi - V)
        dx, dy, dz = sv * np.sin(U), np.cos(np.pi - V), sv * np.cos(U)
    M = K @ np.linalg.inv(R)
    X = M[0, 0] * dx + M[0, 1] * dy + M[0, 2] * dz
    Y = M[1, 0] * dx + M[1, 1] * dy + M[1, 2] * dz
    Z = M[2, 0] * dx + M[2, 1] * dy + M[2, 2] * dz
    ok = Z > 0
    px = np.where(ok, X / np.where(ok, Z, 1.0), -1.0)
    py = np.where(ok, Y / np.where(ok, Z, 1.0), -1.0)
    H, W = img.shape[:2]
    ix, iy = np.floor(px).astype(int), np.floor(py).astype(int)
    fx, fy = (px - ix)[..., None], (py - iy)[..., None]
    # taps far outside the frame (behind the camera: the projector returns (-1, -1)) fold like everything else
    xa, xb = _reflect(np.clip(ix, -W, 2 * W - 1), W), _reflect(np.clip(ix + 1, -W, 2 * W - 1), W)
    ya, yb = _reflect(np.clip(iy, -H, 2 * H - 1), H), _reflect(np.clip(iy + 1, -H, 2 * H - 1), H)
    im = img.astype(np.float64)
    p00, p01, p10, p11 = im[ya, xa], im[ya, xb], im[yb, xa], im[yb, xb]
    val = (p00 * (1 - fx) + p01 * fx) * (1 - fy) + (p10 * (1 - fx) + p11 * fx) * fy
    rng = np.maximum.reduce([p00, p01, p10, p11]) - np.minimum.reduce([p00, p01, p10, p11])
    near = (ix >= -W) & (ix < 2 * W - 1) & (iy >= -H) & (iy < 2 * H - 1)   # one fold of BORDER_REFLECT covers these
    return val, rng, near


@pytest.mark.parametrize("kind", [0, 1])
def test_warp_is_the_bilinear_resampling_along_the_inverse_projection(po, c1, rig_r_real, kind):
    rigs = [(c1, range(4))] + ([(rig_r_real[0], range(2))] if kind == 0 else [])
    worst, total, beyond_one = 0.0, 0, 0
    for g, cams in rigs:
        scale = float(np.float32(g["scale"]))
        for i in cams:
            K = np.asarray(g["K"][i], np.float64).reshape(3, 3)
            R = np.asarray(g["R"][i], np.float64).reshape(3, 3)
            p = po.projector(kind, g["scale"], g["K"][i], g["R"][i])
            roi = po.warp_roi(p, g["w"], g["h"])
            _, got = po.warp(p, g["frames"][i])
            val, rng, near = _ideal_warp(g["frames"][i], K, R, scale, roi, kind == 1)
            diff = np.abs(got.astype(np.float64) - val)
            bound = 0.5 + rng / 32.0 + 1e-6
            bad = (diff > bound) & near[..., None]
            assert not bad.any(), (kind, i, int(bad.sum()), float(diff[bad].max()))
            worst = max(worst, float(diff[near].max()))
            total += int(near.sum()) * 3
            beyond_one += int((diff[near] > 1).sum())
    print("projector %d: %d values, every one within 0.5 + range / 32 of the real-valued resampling; max |diff| %.2f, %.2f %% beyond 1"
          % (kind, total, worst, 100.0 * beyond_one / total))
    assert beyond_one < 0.02 * total


def test_blend_is_burt_adelson_with_opencv_s_two_truncations(po):
    """MultiBandBlender against the algorithm it implements (Burt & Adelson's multiresolution spline), written here in float64 over
    scipy's separable convolution - no integer arithmetic, no line of the restatement: Gaussian pyramids with [1 4 6 4 1] / 16 and
    mirror borders, Laplacian = level - expand(next level), weights = Gaussian pyramid of mask / 255, per level sum(L w) / (sum(w) +
    1e-5), collapse by expand + add.  The only OpenCV-specific ingredients modelled are the two conversions the reference's code spells
    out - static_cast<short>(lap * w) in feed and static_cast<short>(acc / (W + eps)) in blend, both truncating toward zero; everything
    else (the (v + 128) >> 8 and (v + 32) >> 6 roundings of the 16-bit pyramids) stays exact real arithmetic.  The oracle's result must
    agree with that to the accumulated rounding of its integer pyramids - mean < 0.6, max < 3.5 counts - on the whole canvas, borders
    included.  (Without the two truncations the same float blend sits 1.5 - 2.3 counts away on average, up to 9: OpenCV's blender
    loses up to one count of magnitude per band to them - a property of the reference's arithmetic, reproduced, not corrected.)"""
    from scipy.ndimage import convolve1d
    from helpers import synth_frame
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16

    def down(a):
        return convolve1d(convolve1d(a, k, axis=0, mode="mirror"), k, axis=1, mode="mirror")[::2, ::2]

    def up(a, shape):
        z = np.zeros(shape + a.shape[2:], np.float64)
        z[::2, ::2] = a
        return convolve1d(convolve1d(z, 2 * k, axis=0, mode="mirror"), 2 * k, axis=1, mode="mirror")

    def blend(imgs, masks, nb, truncate):
        num, den = [None] * (nb + 1), [None] * (nb + 1)
        for im, m in zip(imgs, masks):
            G, W = [im.astype(np.float64)], [m.astype(np.float64) / 255.0]
            for _ in range(nb):
                G.append(down(G[-1]))
                W.append(down(W[-1]))
            L = [G[l] - up(G[l + 1], G[l].shape[:2]) for l in range(nb)] + [G[nb]]
            for l in range(nb + 1):
                t = L[l] * W[l][..., None]
                if truncate:
                    t = np.trunc(np.round(L[l]) * W[l][..., None])
                num[l] = t if num[l] is None else num[l] + t
                den[l] = W[l] if den[l] is None else den[l] + W[l]
        nrm = lambda l: np.trunc(num[l] / (den[l][..., None] + 1e-5)) if truncate else num[l] / (den[l][..., None] + 1e-5)
        out = nrm(nb)
        for l in range(nb - 1, -1, -1):
            out = nrm(l) + up(out, num[l].shape[:2])
        return out

    H, W = 256, 384
    a, b = synth_frame(W, H, 5), synth_frame(W, H, 9)
    ma = np.zeros((H, W), np.uint8)
    ma[:, :W // 2] = 255
    # a soft, slanted seam as well: weights strictly between 0 and 1 over a band
    yy, xx = np.mgrid[0:H, 0:W]
    ms = np.clip((xx - W // 2 + (yy - H // 2) // 3) * 8 + 128, 0, 255).astype(np.uint8)
    for name, m1 in (("hard vertical seam", ma), ("soft slanted seam", ms)):
        m2 = (255 - m1).astype(np.uint8)
        for nb in (2, 4):
            bl = po.Blender(nb)
            bl.prepare([(0, 0), (0, 0)], [(W, H), (W, H)])
            bl.feed(a.astype(np.int16), m1, (0, 0))
            bl.feed(b.astype(np.int16), m2, (0, 0))
            res, _ = bl.blend()
            d = np.abs(res.astype(np.float64) - blend([a, b], [m1, m2], nb, True))
            pure = np.abs(res.astype(np.float64) - blend([a, b], [m1, m2], nb, False))
            print("%s, %d bands: against the float blend with the two truncations mean %.2f max %.2f; without them mean %.2f max %.2f"
                  % (name, nb, d.mean(), d.max(), pure.mean(), pure.max()))
            assert d.mean() < 0.6 and d.max() < 3.5, (name, nb, d.mean(), d.max())
            assert pure.mean() > d.mean()
