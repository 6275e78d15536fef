import importlib, sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import pano_oracle as po
from helpers import synth_frame
pano = importlib.import_module("img-stitching_amd")
from test_gpu_parity import _rig, make_ctx
d, bands, kind = _rig(3, 333, 187, 250.0, 20.0, -20.0, 2.0), 8, 0
frames = [synth_frame(d["w"], d["h"], 3 + i) for i in range(d["n"])]
ctx = make_ctx(pano, d, kind, num_bands=bands)
ctx.build_masks_voronoi()
masks = [ctx.get_mask(i) for i in range(d["n"])]
got = ctx.compose_host(frames)
want, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, bands, kind=kind)
print('nb', ctx.num_bands(), 'pano', ctx.pano_rect(), 'mismatch px', (got != want).any(axis=2).sum(), 'maxdiff', np.abs(got.astype(int)-want.astype(int)).max())
ys, xs = np.nonzero((got != want).any(axis=2)); print('bbox', xs.min(), xs.max(), ys.min(), ys.max())
for i in range(d["n"]):
    p = po.projector(kind, d["scale"], d["K"][i], d["R"][i])
    warped = po.warp(p, frames[i])[1].astype(np.int16)
    (tx, ty, tw, th), (top, bottom, left, right) = ctx.feed_tile(i)
    def refl(idx, n):
        q = np.mod(idx, 2 * n); return np.where(q < n, q, 2 * n - 1 - q)
    g = warped[refl(np.arange(-top, warped.shape[0] + bottom), warped.shape[0])][:, refl(np.arange(-left, warped.shape[1] + right), warped.shape[1])]
    wgt = np.zeros((th, tw), np.float32)
    wgt[top:top + warped.shape[0], left:left + warped.shape[1]] = masks[i].astype(np.float32) * np.float32(1.0 / 255.0)
    print('cam', i, 'tile', (tx,ty,tw,th), (top,bottom,left,right))
    for l in range(ctx.num_bands() + 1):
        gl = ctx.debug_level(i, l); wl = ctx.debug_weights(i, l)
        print('  level', l, g.shape, 'G ok', np.array_equal(gl, g), 'W ok', np.array_equal(wl, wgt), '' if np.array_equal(gl,g) else np.argwhere((gl!=g).any(axis=2))[:5].tolist())
        if l < ctx.num_bands():
            g = po.pyr_down_16s(g); wgt = po.pyr_down_32f(wgt)
