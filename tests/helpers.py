"""shared synthetic inputs for the tests and bench.py (no reference files needed at run time)"""
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def synth_frame(w, h, seed):
    """deterministic BGR8 frame: PCG64 noise low-passed 3x3, mixed with smooth ramps and hard edges so that
    interpolation, pyramid and seam arithmetic all see structure"""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8).astype(np.float32)
    p = np.pad(img, ((1, 1), (1, 1), (0, 0)), mode="edge")
    lp = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)) / 9.0
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    ramp = 110.0 + 70.0 * np.sin(xx / w * 9.0 + seed)[..., None] + 60.0 * np.cos((yy / h * 5.0)[..., None] + np.arange(3, dtype=np.float32))
    checker = (((xx // 37).astype(np.int32) + (yy // 29).astype(np.int32)) % 2 * 60.0)[..., None]
    return np.clip(0.4 * lp + 0.6 * ramp + checker - 30.0, 0, 255).astype(np.uint8)


def ry(deg):
    t = math.radians(deg)
    c, s = math.cos(t), math.sin(t)
    return [c, 0.0, s, 0.0, 1.0, 0.0, -s, 0.0, c]


def c2_group(w=1920, h=1080, f=1002.416):
    """config 2, one group: 4 cameras at yaw +67.5, +22.5, -22.5, -67.5 degrees"""
    K = [f, 0.0, w / 2.0, 0.0, f, h / 2.0, 0.0, 0.0, 1.0]
    return {"K": [K] * 4, "R": [ry(a) for a in (67.5, 22.5, -22.5, -67.5)], "scale": f, "w": w, "h": h, "n": 4}


def c4_rig(w=3840, h=2160, f=3534.0):
    """config 4: 4 x 4K, cylindrical, same yaws"""
    d = c2_group(w, h, f)
    return d


def c4_gain_map(gw, gh, seed):
    """a smooth field of block gains in [0.8, 1.25] - a 5 x 4 grid of seeded values, bilinearly
    interpolated to the gw x gh blocks of BlocksGainCompensator (32 x 32-pixel blocks of the warped tile)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    coarse = rng.random((4, 5))
    ys = np.linspace(0, 3, gh)[:, None]; xs = np.linspace(0, 4, gw)[None, :]
    y0 = np.minimum(ys.astype(int), 2); x0 = np.minimum(xs.astype(int), 3)
    fy = ys - y0; fx = xs - x0
    f = (coarse[y0, x0] * (1 - fy) * (1 - fx) + coarse[y0, x0 + 1] * (1 - fy) * fx
         + coarse[y0 + 1, x0] * fy * (1 - fx) + coarse[y0 + 1, x0 + 1] * fy * fx)
    return (0.8 + 0.45 * f).astype(np.float32)


def build_fake_rccl():
    """tests/src/fake_rccl.cpp -> tests/_build/libfake_rccl.so (hipcc, host code only; rebuilt when the source is newer)"""
    import subprocess
    src = os.path.join(ROOT, "tests", "src", "fake_rccl.cpp")
    out_dir = os.path.join(ROOT, "tests", "_build")
    out = os.path.join(out_dir, "libfake_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "-O1", "-std=c++17", "-shared", "-fPIC", src, "-o", out + ".tmp", "-lrt", "-lpthread"])
        os.replace(out + ".tmp", out)
    return out


def read_graphcut_dump(path):
    """pano_debug_graphcut_dump's records: (i, j, term, wh, wv, labels) with (H, W) arrays"""
    raw = open(path, "rb").read()
    out, o = [], 0
    while o < len(raw):
        i, j, W, H = np.frombuffer(raw, np.int32, 4, o); o += 16
        n = int(W) * int(H)
        f = np.frombuffer(raw, np.float32, 3 * n, o).reshape(3, H, W); o += 12 * n
        lab = np.frombuffer(raw, np.uint8, n, o).reshape(H, W); o += n
        out.append((int(i), int(j), f[0], f[1], f[2], lab))
    return out


def min_cut_capacity(term, wh, wv, lab):
    """capacity of the cut (source side = lab == 1) of the grid graph of GraphCutSeamFinder::Impl::findInPair: terminal edges of
    weight |term| (to the source where term > 0, to the sink where term < 0), wh between (y, x) and (y, x + 1), wv between (y, x)
    and (y + 1, x).  All weights are integers carried in f32"""
    t, s_ = term.astype(np.int64), lab.astype(bool)
    cut = int((-t[s_ & (t < 0)]).sum() + t[~s_ & (t > 0)].sum())
    cut += int(wh.astype(np.int64)[:, :-1][s_[:, :-1] != s_[:, 1:]].sum())
    cut += int(wv.astype(np.int64)[:-1, :][s_[:-1, :] != s_[1:, :]].sum())
    return cut


def scipy_max_flow(term, wh, wv, sides=False):
    """max-flow value of the grid graph; sides=True: also the (H, W) masks of the vertices the source reaches / that reach the sink
    in the residual graph"""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import breadth_first_order, maximum_flow
    H, W = term.shape
    n = H * W
    idx = np.arange(n).reshape(H, W)
    t = term.astype(np.int64).reshape(-1)
    rows = [np.full(int((t > 0).sum()), n), idx.reshape(-1)[t < 0]]
    cols = [idx.reshape(-1)[t > 0], np.full(int((t < 0).sum()), n + 1)]
    caps = [t[t > 0], -t[t < 0]]
    a, b, c = idx[:, :-1].reshape(-1), idx[:, 1:].reshape(-1), wh.astype(np.int64)[:, :-1].reshape(-1)
    rows += [a, b]; cols += [b, a]; caps += [c, c]
    a, b, c = idx[:-1, :].reshape(-1), idx[1:, :].reshape(-1), wv.astype(np.int64)[:-1, :].reshape(-1)
    rows += [a, b]; cols += [b, a]; caps += [c, c]
    g = csr_matrix((np.concatenate(caps).astype(np.int32), (np.concatenate(rows), np.concatenate(cols))), shape=(n + 2, n + 2))
    r = maximum_flow(g, n, n + 1)
    if not sides:
        return int(r.flow_value)
    resid = (g - r.flow).tocsr()
    resid.data = np.maximum(resid.data, 0)
    resid.eliminate_zeros()
    src = np.zeros(n + 2, bool); src[breadth_first_order(resid, n, return_predecessors=False)] = True
    snk = np.zeros(n + 2, bool); snk[breadth_first_order(resid.T.tocsr(), n + 1, return_predecessors=False)] = True
    return int(r.flow_value), src[:n].reshape(H, W), snk[:n].reshape(H, W)
