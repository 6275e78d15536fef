"""GPU parity tests (run with -m gpu on an MI355X): the HIP path through the C-ABI against the CPU
oracle on the same inputs.  Everything on this path is integer / fixed-point or exactly-ordered f32,
so the bar is BIT-EXACT (tolerance 0), stage by stage and end to end."""
import os

import numpy as np
import pytest

from helpers import c2_group, c4_rig, ry, synth_frame, min_cut_capacity, read_graphcut_dump, scipy_max_flow

pytestmark = pytest.mark.gpu


def make_ctx(pano, d, kind=0, **kw):
    ctx = pano.Context(d["n"], d["w"], d["h"], scale=d["scale"], projector=kind, device=0, **kw)
    for i in range(d["n"]):
        ctx.set_camera(i, d["K"][i], d["R"][i])
    ctx.prepare()
    return ctx


def oracle_masks(po, d, kind=0):
    return po.prepare_masks_voronoi(kind, d["w"], d["h"], d["K"], d["R"], d["scale"])


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("kind", [0, 1])
def test_warp_stage_bit_exact(pano, po, torch, c1, kind):
    """RotationWarper::warp (INTER_LINEAR, BORDER_REFLECT) and the NEAREST/CONSTANT mask warp"""
    ctx = make_ctx(pano, c1, kind, num_bands=2)
    for i in range(4):
        p = po.projector(kind, c1["scale"], c1["K"][i], c1["R"][i])
        corner, want = po.warp(p, c1["frames"][i])
        r = ctx.roi(i)
        assert r[:2] == corner and want.shape == (r[3], r[2], 3)
        src = torch.from_numpy(c1["frames"][i]).cuda()
        dst = torch.zeros((r[3], r[2], 3), dtype=torch.uint8, device="cuda")
        ctx.warp(i, src.data_ptr(), 480 * 3, dst.data_ptr(), r[2] * 3, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(dst.cpu().numpy(), want)
        _, wantm = po.warp(p, np.full((270, 480), 255, np.uint8), po.INTER_NEAREST, po.BORDER_CONSTANT)
        dm = torch.zeros((r[3], r[2]), dtype=torch.uint8, device="cuda")
        ctx.warp_mask(i, dm.data_ptr(), r[2], torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(dm.cpu().numpy(), wantm)


def test_voronoi_masks_bit_exact(pano, po, c1, rig_r):
    ctx = make_ctx(pano, c1, 0, num_bands=2)
    ctx.build_masks_voronoi()
    want = oracle_masks(po, c1)
    for i in range(4):
        assert np.array_equal(ctx.get_mask(i), want[i])
    st = rig_r["stitchers"][1]
    v = st["cams"]
    d = {"n": 2, "w": 960, "h": 540, "scale": v[-1], "K": [v[0:9], v[18:27]], "R": [v[9:18], v[27:36]]}
    ctx = make_ctx(pano, d, 0, num_bands=3)
    ctx.build_masks_voronoi()
    want = oracle_masks(po, d)
    for i in range(2):
        assert np.array_equal(ctx.get_mask(i), want[i])


@pytest.mark.parametrize("bands", [0, 2, 4])
def test_stages_and_compose_c1_bit_exact(pano, po, c1, bands):
    """config 1 end to end + every intermediate the oracle exposes"""
    masks = oracle_masks(po, c1)
    ctx = make_ctx(pano, c1, 0, num_bands=bands)
    for i in range(4):
        ctx.set_mask(i, masks[i])
    got = ctx.compose_host(c1["frames"])
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, bands)
    assert got.shape == want.shape == (257, 1333, 3)
    assert np.array_equal(got, want)
    # per-camera Gaussian levels == pyrDown chain of the REFLECT-bordered warped image; weights likewise
    rois = [ctx.roi(i) for i in range(4)]
    b = po.Blender(bands)
    b.prepare([r[:2] for r in rois], [r[2:] for r in rois])
    for i in range(4):
        p = po.projector(0, c1["scale"], c1["K"][i], c1["R"][i])
        warped = po.warp(p, c1["frames"][i])[1].astype(np.int16)
        (tx, ty, tw, th), (top, bottom, left, right) = ctx.feed_tile(i)
        ys = np.abs(np.arange(-top, warped.shape[0] + bottom))
        # BORDER_REFLECT index map
        def refl(idx, n):
            q = np.mod(idx, 2 * n)
            return np.where(q < n, q, 2 * n - 1 - q)
        g = warped[refl(np.arange(-top, warped.shape[0] + bottom), warped.shape[0])][:, refl(np.arange(-left, warped.shape[1] + right), warped.shape[1])]
        wgt = np.zeros((th, tw), np.float32)
        wgt[top:top + warped.shape[0], left:left + warped.shape[1]] = masks[i].astype(np.float32) * np.float32(1.0 / 255.0)
        for l in range(bands + 1):
            # the pyramid is only produced where something downstream reads it (pano_get_live_rect)
            lx, ly, lw, lh = ctx.live_rect(i, l)
            assert lw > 0 and lh > 0
            assert np.array_equal(ctx.debug_level(i, l)[ly:ly + lh, lx:lx + lw], g[ly:ly + lh, lx:lx + lw]), (i, l)
            assert np.array_equal(ctx.debug_weights(i, l), wgt), (i, l)
            if l < bands:
                g = po.pyr_down_16s(g)
                wgt = po.pyr_down_32f(wgt)
        b.feed(warped, masks[i], rois[i][:2])
    for l in range(bands + 1):
        assert np.array_equal(ctx.debug_canvas_weights(l), b.level_weights(l))
    # idempotence: a second frame through the same context gives the same panorama
    assert np.array_equal(ctx.compose_host(c1["frames"]), want)


def test_graph_replay(pano, po, torch, c1, monkeypatch):
    """PANO_GRAPH=1: the frame's launch sequence captured once per buffer set and replayed"""
    monkeypatch.setenv("PANO_GRAPH", "1")
    masks = oracle_masks(po, c1)
    ctx = make_ctx(pano, c1, 0, num_bands=4)
    for i in range(4):
        ctx.set_mask(i, masks[i])
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 4)
    st = torch.cuda.Stream()
    d = [torch.from_numpy(f).cuda() for f in c1["frames"]]
    out = torch.zeros((257, 1333, 3), dtype=torch.uint8, device="cuda")
    for rep in range(3):   # capture, then two replays; new frame content between replays
        out.zero_()
        torch.cuda.synchronize()
        ctx.compose([t.data_ptr() for t in d], [480 * 3] * 4, out.data_ptr(), 1333 * 3, st.cuda_stream)
        st.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
    d[0].copy_(torch.from_numpy(c1["frames"][1]).cuda())
    torch.cuda.synchronize()
    frames2 = [c1["frames"][1]] + c1["frames"][1:]
    want2, _ = po.compose(frames2, c1["K"], c1["R"], c1["scale"], masks, 4)
    ctx.compose([t.data_ptr() for t in d], [480 * 3] * 4, out.data_ptr(), 1333 * 3, st.cuda_stream)
    st.synchronize()
    assert np.array_equal(out.cpu().numpy(), want2)
    # graphs are keyed by frame slot: the same caller buffers composed in another slot capture their own graph
    ctx.set_frame_slots(2)
    for slot in (0, 1, 0, 1):
        out.zero_()
        torch.cuda.synchronize()
        ctx.select_frame_slot(slot)
        ctx.compose([t.data_ptr() for t in d], [480 * 3] * 4, out.data_ptr(), 1333 * 3, st.cuda_stream)
        st.synchronize()
        assert np.array_equal(out.cpu().numpy(), want2), slot


def test_on_the_fly_warp_kernel(pano, po, c1, monkeypatch):
    """frames up to 2048 x 2048 use the static remap table; PANO_WARP_ON_THE_FLY=1 selects the projecting
    kernel that larger frames use - both must give the oracle's bits"""
    masks = oracle_masks(po, c1)
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 3)
    for flag in ("1", "0"):
        monkeypatch.setenv("PANO_WARP_ON_THE_FLY", flag)
        ctx = make_ctx(pano, c1, 0, num_bands=3)
        for i in range(4):
            ctx.set_mask(i, masks[i])
        assert np.array_equal(ctx.compose_host(c1["frames"]), want)


def test_no_blend_and_cut_and_cylindrical(pano, po, c1, rig_r):
    masks = oracle_masks(po, c1)
    ctx = make_ctx(pano, c1, 0, num_bands=pano.BANDS_NO_BLEND)
    for i in range(4):
        ctx.set_mask(i, masks[i])
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, -1)
    assert np.array_equal(ctx.compose_host(c1["frames"]), want)
    # cut (m_cutParams) on rig R stitcher 0 with the band rule at strength 1 -> 3 bands
    st = rig_r["stitchers"][0]
    v = st["cams"]
    d = {"n": 2, "w": 960, "h": 540, "scale": v[-1], "K": [v[0:9], v[18:27]], "R": [v[9:18], v[27:36]]}
    frames = [synth_frame(960, 540, 11 + i) for i in range(2)]
    m = oracle_masks(po, d)
    ctx = make_ctx(pano, d, 0, num_bands=pano.BANDS_FROM_STRENGTH, blend_strength=1.0, cut=st["cut"])
    assert ctx.num_bands() == 3
    for i in range(2):
        ctx.set_mask(i, m[i])
    want, _ = po.compose(frames, d["K"], d["R"], d["scale"], m, 3, cut=st["cut"])
    got = ctx.compose_host(frames)
    assert got.shape == (250, 1430, 3) and np.array_equal(got, want)
    # cylindrical projector
    mc = oracle_masks(po, c1, 1)
    ctx = make_ctx(pano, c1, 1, num_bands=3)
    for i in range(4):
        ctx.set_mask(i, mc[i])
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], mc, 3, kind=1)
    assert np.array_equal(ctx.compose_host(c1["frames"]), want)


def test_ragged_masks_and_extreme_frames(pano, po, c1):
    """edge cases: empty mask for one camera, soft (non 0/255) masks, all-black and all-white frames"""
    rng = np.random.default_rng(5)
    ctx = make_ctx(pano, c1, 0, num_bands=3)
    rois = [ctx.roi(i) for i in range(4)]
    masks = [rng.integers(0, 256, size=(r[3], r[2]), dtype=np.uint8) for r in rois]
    masks[1][:] = 0
    masks[2][:, : rois[2][2] // 2] = 255
    for i in range(4):
        ctx.set_mask(i, masks[i])
    frames = [np.zeros((270, 480, 3), np.uint8), np.full((270, 480, 3), 255, np.uint8), c1["frames"][2],
              rng.integers(0, 256, size=(270, 480, 3), dtype=np.uint8)]
    want, _ = po.compose(frames, c1["K"], c1["R"], c1["scale"], masks, 3)
    assert np.array_equal(ctx.compose_host(frames), want)


def test_exposure_gain_apply(pano, po, c1):
    """BlocksGainCompensator::apply between warp and feed (stitching_detailed.cpp:841)"""
    rng = np.random.default_rng(7)
    masks = oracle_masks(po, c1)
    ctx = make_ctx(pano, c1, 0, num_bands=2)
    gains, full = [], []
    for i in range(4):
        r = ctx.roi(i)
        g = (0.8 + 0.45 * rng.random(((r[3] + 31) // 32, (r[2] + 31) // 32))).astype(np.float32)
        gains.append(g)
        full.append(po.resize_linear_32f(g, r[2], r[3]))
        ctx.set_mask(i, masks[i])
        ctx.set_gain_map(i, g)
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 2, gain_maps=full)
    assert np.array_equal(ctx.compose_host(c1["frames"]), want)
    ctx.set_gain_map(0, None)


def test_exposure_gain_apply_fine_maps(pano, po, c1):
    """K1's gain path has two branches (pano_warp.hip): the patch's gain rows staged in LDS (a 16-row patch reads at most
    kGainRows = 4 rows of the horizontally resized map), and a per-lane fallback from global memory for maps finer than that
    (grow_base == -1).  Block maps of 32-pixel blocks only ever take the first.  Here: maps of th / 3 rows (every patch takes the
    fallback), th / 6 rows (patches of both kinds in one camera, by where the rows fall) and the usual th / 32, one camera each,
    plus a camera without a map - against the oracle's apply on the same maps"""
    rng = np.random.default_rng(11)
    masks = oracle_masks(po, c1)
    ctx = make_ctx(pano, c1, 0, num_bands=2)
    full = []
    for i, div in enumerate((3, 6, 32, None)):
        r = ctx.roi(i)
        ctx.set_mask(i, masks[i])
        if div is None:
            full.append(np.ones((r[3], r[2]), np.float32))   # the oracle wants a map per camera: x 1.0f is the identity
            continue
        g = (0.8 + 0.45 * rng.random(((r[3] + div - 1) // div, (r[2] + div - 1) // div))).astype(np.float32)
        full.append(po.resize_linear_32f(g, r[2], r[3]))
        ctx.set_gain_map(i, g)
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 2, gain_maps=full)
    assert np.array_equal(ctx.compose_host(c1["frames"]), want)
    # the same through the device entry of both stitchers at once (pano_compose_pair: the gain instantiation with 8 cameras)
    ctx2 = make_ctx(pano, c1, 0, num_bands=2)
    for i in range(4):
        ctx2.set_mask(i, masks[i])
    import torch
    d = [torch.from_numpy(f).cuda() for f in c1["frames"]]
    ow, oh = ctx.output_size()
    o = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    st = torch.cuda.current_stream().cuda_stream
    ptr = [t.data_ptr() for t in d]
    ctx.compose_pair(ctx2, ptr, [480 * 3] * 4, o[0].data_ptr(), ow * 3, ptr, [480 * 3] * 4, o[1].data_ptr(), ow * 3, st)
    torch.cuda.synchronize()
    assert np.array_equal(o[0].cpu().numpy(), want)
    want2, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 2)
    assert np.array_equal(o[1].cpu().numpy(), want2)


def test_device_entry_and_sharded_feed(pano, po, torch, c1):
    """pano_compose on device pointers == pano_feed_cameras in two shards + pano_blend (the multi-GPU split)"""
    masks = oracle_masks(po, c1)
    ctx = make_ctx(pano, c1, 0, num_bands=4)
    for i in range(4):
        ctx.set_mask(i, masks[i])
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 4)
    s = torch.cuda.current_stream().cuda_stream
    d = [torch.from_numpy(f).cuda() for f in c1["frames"]]
    out = torch.zeros((257, 1333, 3), dtype=torch.uint8, device="cuda")
    ctx.compose([t.data_ptr() for t in d], [480 * 3] * 4, out.data_ptr(), 1333 * 3, s)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    base, slot = ctx.pyramid_slots()
    assert base and slot % 4096 == 0
    out.zero_()
    ctx.feed_cameras(0b0101, [t.data_ptr() for t in d], [480 * 3] * 4, s)
    ctx.feed_cameras(0b1010, [t.data_ptr() for t in d], [480 * 3] * 4, s)
    ctx.blend(out.data_ptr(), 1333 * 3, s)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


def test_c2_full_size_bit_exact(pano, po):
    """config 2 (one group of the 8x1080p rig: 4 x 1920x1080, 5 bands) against the oracle at full size"""
    d = c2_group()
    frames = [synth_frame(1920, 1080, 42 + i) for i in range(4)]
    ctx = make_ctx(pano, d, 0, num_bands=5)
    ctx.build_masks_voronoi()
    masks = [ctx.get_mask(i) for i in range(4)]
    want_masks = oracle_masks(po, d)
    for i in range(4):
        assert np.array_equal(masks[i], want_masks[i])
    po.set_threads(8)
    want, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, 5)
    po.set_threads(1)
    got = ctx.compose_host(frames)
    assert got.shape == (991, 3893, 3)
    assert np.array_equal(got, want)


def test_c4_cylindrical_7_bands_properties(pano, po):
    """config 4 shape (4 x 4K, cylindrical, 7 bands, gains): size-independent properties at full size -
    determinism, cut == crop of the full panorama - plus an oracle comparison on a band of rows"""
    d = c4_rig()
    frames = [synth_frame(3840, 2160, 7 + i) for i in range(4)]
    ctx = make_ctx(pano, d, 1, num_bands=7)
    ctx.build_masks_voronoi()
    full = ctx.compose_host(frames)
    assert full.shape[1] == ctx.pano_rect()[2] and full.shape[0] == ctx.pano_rect()[3]
    assert np.array_equal(full, ctx.compose_host(frames))
    ctx.set_cut((1000, 300, 6000, 900))
    assert np.array_equal(ctx.compose_host(frames), full[300:1200, 1000:7000])
    assert full[:, :, :].max() > 0


def test_c4_full_size_bit_exact(pano, po):
    """config 4 at full size (4 x 3840x2160, cylindrical, 7 bands, block gains: the table form of K1 with box-relative codes,
    gain-applying instantiation) against the oracle"""
    d = c4_rig()
    rng = np.random.default_rng(11)
    frames = [synth_frame(3840, 2160, 7 + i) for i in range(4)]
    ctx = make_ctx(pano, d, 1, num_bands=7)
    ctx.build_masks_voronoi()
    masks = [ctx.get_mask(i) for i in range(4)]
    full = []
    for i in range(4):
        r = ctx.roi(i)
        g = (0.8 + 0.45 * rng.random(((r[3] + 31) // 32, (r[2] + 31) // 32))).astype(np.float32)
        ctx.set_gain_map(i, g)
        full.append(po.resize_linear_32f(g, r[2], r[3]))
    got = ctx.compose_host(frames)
    po.set_threads(16)
    want, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, 7, kind=1, gain_maps=full)
    po.set_threads(1)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_c1_against_committed_golden(pano, c1):
    """the HIP path against the committed golden vectors (tests/golden/c1_golden.json, c1_pano_b4.png) - no oracle
    code runs in this test"""
    import hashlib
    import json
    import os
    from conftest import GOLDEN, load_png_bgr
    g = json.load(open(os.path.join(GOLDEN, "c1_golden.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    for nb in (-1, 0, 2, 4):
        ctx = make_ctx(pano, c1, 0, num_bands=nb)
        assert [list(ctx.roi(i)) for i in range(4)] == g["rois"]
        ctx.build_masks_voronoi()
        assert [sha(ctx.get_mask(i)) for i in range(4)] == g["mask_sha256"]
        out = ctx.compose_host(c1["frames"])
        assert sha(out) == g["pano_sha256"][str(nb)]
        if nb == 4:
            assert np.array_equal(out, load_png_bgr(os.path.join(GOLDEN, "c1_pano_b4.png")))
        if nb == 2:
            ctx.set_cut(g["cut"])
            assert sha(ctx.compose_host(c1["frames"])) == g["pano_cut_sha256"]
    # the reference's seam finder and exposure compensator against the committed vectors
    ctx = make_ctx(pano, c1, 0, num_bands=4)
    ctx.build_masks_graphcut(c1["frames"])
    assert [sha(ctx.get_mask(i)) for i in range(4)] == g["graphcut_mask_sha256"]
    assert sha(ctx.compose_host(c1["frames"])) == g["graphcut_pano_b4_sha256"]
    ctx.build_masks_voronoi()
    gains = ctx.estimate_gains(c1["frames"])
    assert [list(x.shape) for x in gains] == g["gain_map_shape"] and [sha(x) for x in gains] == g["gain_map_sha256"]
    assert sha(ctx.compose_host(c1["frames"])) == g["gain_pano_b4_sha256"]


def _rig(n, w, h, f, yaw0, step, pitch_deg=0.0):
    import math
    from helpers import ry
    K = [f, 0.0, w / 2.0, 0.0, f, h / 2.0, 0.0, 0.0, 1.0]
    Rs = []
    for i in range(n):
        r = np.array(ry(yaw0 + i * step), np.float64).reshape(3, 3)
        t = math.radians(pitch_deg * ((-1) ** i))
        rx = np.array([[1, 0, 0], [0, math.cos(t), -math.sin(t)], [0, math.sin(t), math.cos(t)]])
        Rs.append((r @ rx).astype(np.float32).reshape(9).tolist())
    return {"K": [K] * n, "R": Rs, "scale": f, "w": w, "h": h, "n": n}


@pytest.mark.parametrize("case", ["8cams", "odd_size_max_bands", "tiny", "two_bands_one_cam"])
def test_extreme_shapes(pano, po, case):
    """PANO_MAX_CAMS cameras in one context, odd frame sizes, the maximum band count on a small panorama (levels
    shrink to a couple of pixels), a 1-camera context"""
    if case == "8cams":
        d, bands, kind = _rig(8, 320, 180, 300.0, 35.0, -10.0, 3.0), 3, 0
    elif case == "odd_size_max_bands":
        d, bands, kind = _rig(3, 333, 187, 250.0, 20.0, -20.0, 2.0), 8, 0
    elif case == "tiny":
        d, bands, kind = _rig(2, 37, 23, 30.0, 10.0, -20.0), 2, 1
    else:
        d, bands, kind = _rig(1, 200, 120, 180.0, 0.0, 0.0), 2, 0
    frames = [synth_frame(d["w"], d["h"], 3 + i) for i in range(d["n"])]
    ctx = make_ctx(pano, d, kind, num_bands=bands)
    ctx.build_masks_voronoi()
    masks = [ctx.get_mask(i) for i in range(d["n"])]
    want_masks = oracle_masks(po, d, kind)
    for i in range(d["n"]):
        assert np.array_equal(masks[i], want_masks[i])
    rois = [ctx.roi(i) for i in range(d["n"])]
    b = po.Blender(bands)
    b.prepare([r[:2] for r in rois], [r[2:] for r in rois])
    assert ctx.num_bands() == b.num_bands()
    want, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, bands, kind=kind)
    got = ctx.compose_host(frames)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_randomised_rigs(pano, po):
    """seeded random rigs: 2-5 cameras, random focal / yaw step / pitch / roll, odd frame sizes, both projectors,
    0-6 bands, random cut, soft random masks on half of the cases"""
    import math
    # PANO_FUZZ_SEED / PANO_FUZZ_CASES: the same test over other seeds and more cases (a builder's soak: tools/fuzz_rigs.sh)
    seed, cases = int(os.environ.get("PANO_FUZZ_SEED", "2024")), int(os.environ.get("PANO_FUZZ_CASES", "40"))
    rng = np.random.default_rng(seed)
    done = 0
    for case in range(cases):
        n = int(rng.integers(2, 6))
        w, h = int(rng.integers(40, 400)), int(rng.integers(30, 260))
        f = float(rng.uniform(0.6, 1.6)) * w
        step = float(rng.uniform(8.0, 30.0)) * (1 if rng.random() < 0.5 else -1)
        yaw0 = -step * (n - 1) / 2 + float(rng.uniform(-5, 5))
        kind = int(rng.integers(0, 2))
        bands = int(rng.integers(0, 7))
        K = [f, 0.0, w / 2.0 + float(rng.uniform(-5, 5)), 0.0, f * float(rng.uniform(0.95, 1.05)), h / 2.0, 0.0, 0.0, 1.0]
        Rs = []
        for i in range(n):
            a, b, c = math.radians(yaw0 + i * step), math.radians(float(rng.uniform(-6, 6))), math.radians(float(rng.uniform(-4, 4)))
            ry_ = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
            rx_ = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
            rz_ = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
            Rs.append((ry_ @ rx_ @ rz_).astype(np.float32).reshape(9).tolist())
        d = {"K": [K] * n, "R": Rs, "scale": f * float(rng.uniform(0.9, 1.1)), "w": w, "h": h, "n": n}
        try:
            ctx = make_ctx(pano, d, kind, num_bands=bands)
        except pano.PanoError as e:
            assert e.status == -6   # a camera wraps the seam: rejected, never computed wrongly
            continue
        frames = [synth_frame(w, h, 100 + case * 8 + i) for i in range(n)]
        if case % 2:
            masks = [rng.integers(0, 256, size=(ctx.roi(i)[3], ctx.roi(i)[2]), dtype=np.uint8) for i in range(n)]
            for i in range(n):
                ctx.set_mask(i, masks[i])
        else:
            ctx.build_masks_voronoi()
            masks = [ctx.get_mask(i) for i in range(n)]
            want_masks = oracle_masks(po, d, kind)
            for i in range(n):
                assert np.array_equal(masks[i], want_masks[i]), case
        pr = ctx.pano_rect()
        cut = None
        if case % 3 == 0 and pr[2] > 20 and pr[3] > 20:
            cx, cy = int(rng.integers(0, pr[2] // 2)), int(rng.integers(0, pr[3] // 2))
            cut = (cx, cy, int(rng.integers(1, pr[2] - cx + 1)), int(rng.integers(1, pr[3] - cy + 1)))
            ctx.set_cut(cut)
        want, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, bands, kind=kind, cut=cut)
        got = ctx.compose_host(frames)
        assert got.shape == want.shape, case
        assert np.array_equal(got, want), (case, n, w, h, kind, bands, cut)
        done += 1
    assert done >= cases * 5 // 8


def test_caller_side_assembly(pano, po, torch, c1):
    """pano_stack_master / pano_stack_finalcut against the oracle (cv::resize INTER_LINEAR + vconcat + divider)"""
    rng = np.random.default_rng(9)
    ctx = make_ctx(pano, c1, 0, num_bands=0)
    s = torch.cuda.current_stream().cuda_stream
    for (uw, uh, dw, dh) in ((1430, 250, 1470, 250), (333, 77, 200, 91), (640, 120, 640, 120)):
        up = rng.integers(0, 256, size=(uh, uw, 3), dtype=np.uint8); down = rng.integers(0, 256, size=(dh, dw, 3), dtype=np.uint8)
        tu, td = torch.from_numpy(up).cuda(), torch.from_numpy(down).cuda()
        out = torch.zeros((2 * dh, dw, 3), dtype=torch.uint8, device="cuda")
        ctx.stack_master(tu.data_ptr(), uw, uh, uw * 3, td.data_ptr(), dw, dh, dw * 3, out.data_ptr(), dw * 3, s)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), po.stack_master(up, down))
        w, h = min(uw, dw), min(uh, dh) - 8
        out2 = torch.zeros((2 * h, w, 3), dtype=torch.uint8, device="cuda")
        ctx.stack_finalcut(tu.data_ptr(), uw, uh, uw * 3, td.data_ptr(), dw, dh, dw * 3, 4, out2.data_ptr(), w * 3, s)
        torch.cuda.synchronize()
        assert np.array_equal(out2.cpu().numpy(), po.stack_finalcut(up, down, 4))
        # the same on host buffers (what master.cpp holds after process())
        assert np.array_equal(ctx.stack_master_host(up, down), po.stack_master(up, down))
        assert np.array_equal(ctx.stack_finalcut_host(up, down, 4), po.stack_finalcut(up, down, 4))


def test_streaming_slots(pano, po, c1):
    """pano_stream_*: pinned double-buffered H2D -> compose -> D2H; two frames in flight, results per slot"""
    masks = oracle_masks(po, c1)
    ctx = make_ctx(pano, c1, 0, num_bands=3)
    for i in range(4):
        ctx.set_mask(i, masks[i])
    sets = [c1["frames"], c1["frames"][::-1]]
    wants = [po.compose(f, c1["K"], c1["R"], c1["scale"], masks, 3)[0] for f in sets]
    for rep in range(3):
        for s in range(2):
            for i in range(4):
                ctx.stream_input(s, i)[:] = sets[s][i]
            ctx.stream_submit(s)
        with pytest.raises(pano.PanoError):
            ctx.stream_submit(0)          # still in flight
        for s in range(2):
            ctx.stream_wait(s)
            assert np.array_equal(ctx.stream_output(s), wants[s])


@pytest.mark.parametrize("rw,rh", [(1920, 1080), (2880, 1620)])
def test_fused_undistort_front_end(pano, po, rig_r, rw, rh):
    """raw 1920x1080 (and 2880x1620: source boxes past column 2048) frames -> (undistort 960x540, crop, resize,
    resize) -> spherical warp, as ONE composed map sampled once; lens = cameras.yaml sensing/imx390/fov120/960, rig R stitcher 0.  Parity is against the oracle
    of the fused map (this is a different resampling from the reference's five-pass chain)."""
    K = [4.890925118101495e+02, 0, 4.940763211103715e+02, 0, 4.912630345468579e+02, 2.865820139005963e+02, 0, 0, 1]
    dist = [-0.2838, 0.0628, 0, 0]
    rect = (70, 66, 885, 410)
    st = rig_r["stitchers"][0]
    v = st["cams"]
    d = {"n": 2, "w": 960, "h": 540, "scale": v[-1], "K": [v[0:9], v[18:27]], "R": [v[9:18], v[27:36]]}
    raw = [synth_frame(rw, rh, 21 + i) for i in range(2)]
    for flag in ("0", "1"):          # remap-table and projecting K1 variants
        os.environ["PANO_WARP_ON_THE_FLY"] = flag
        try:
            ctx = pano.Context(2, 960, 540, scale=d["scale"], num_bands=3, cut=st["cut"], device=0)
            for i in range(2):
                ctx.set_camera(i, d["K"][i], d["R"][i])
                ctx.set_undistort(i, (rw, rh), (960, 540), K, dist, rect)
            ctx.prepare()
        finally:
            os.environ.pop("PANO_WARP_ON_THE_FLY", None)
        ctx.build_masks_voronoi()
        masks = [ctx.get_mask(i) for i in range(2)]
        assert all(np.array_equal(a, b) for a, b in zip(masks, oracle_masks(po, d)))   # masks: stitcher frame geometry
        fe = [po.front_end((rw, rh), (960, 540), K, dist, rect, (960, 540)) for _ in range(2)]
        want, _ = po.compose(raw, d["K"], d["R"], d["scale"], masks, 3, cut=st["cut"], front=fe)
        got = ctx.compose_host(raw)
        assert got.shape == (250, 1430, 3) and np.array_equal(got, want)


def test_compose_pair(pano, po, torch, c1, rig_r):
    """pano_compose_pair (both stitchers per launch) == two pano_compose calls == oracle; also a pair whose level
    structure differs (falls back to sequential composition)"""
    s = torch.cuda.current_stream().cuda_stream
    # rig R: stitcher 0 and 1 have different geometry (different ROIs, canvases 1456x528 vs 1488x512) but both 3 bands
    ctxs, wants, frames_d, outs = [], [], [], []
    for k in range(2):
        st = rig_r["stitchers"][k]
        v = st["cams"]
        d = {"n": 2, "w": 960, "h": 540, "scale": v[-1], "K": [v[0:9], v[18:27]], "R": [v[9:18], v[27:36]]}
        frames = [synth_frame(960, 540, 31 + 2 * k + i) for i in range(2)]
        ctx = make_ctx(pano, d, 0, num_bands=pano.BANDS_FROM_STRENGTH, blend_strength=1.0, cut=st["cut"])
        ctx.build_masks_voronoi()
        masks = [ctx.get_mask(i) for i in range(2)]
        wants.append(po.compose(frames, d["K"], d["R"], d["scale"], masks, 3, cut=st["cut"])[0])
        ctxs.append(ctx)
        frames_d.append([torch.from_numpy(f).cuda() for f in frames])
        w, h = ctx.output_size()
        outs.append(torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda"))
    for rep in range(2):
        for o in outs:
            o.zero_()
        ctxs[0].compose_pair(ctxs[1], [t.data_ptr() for t in frames_d[0]], [960 * 3] * 2, outs[0].data_ptr(), outs[0].shape[1] * 3,
                             [t.data_ptr() for t in frames_d[1]], [960 * 3] * 2, outs[1].data_ptr(), outs[1].shape[1] * 3, s)
        torch.cuda.synchronize()
        for k in range(2):
            assert np.array_equal(outs[k].cpu().numpy(), wants[k]), k
    # mismatched structure: 2 bands vs 4 bands on config 1
    masks = oracle_masks(po, c1)
    ca, cb = make_ctx(pano, c1, 0, num_bands=2), make_ctx(pano, c1, 0, num_bands=4)
    for i in range(4):
        ca.set_mask(i, masks[i]); cb.set_mask(i, masks[i])
    fd = [torch.from_numpy(f).cuda() for f in c1["frames"]]
    oa = torch.zeros((257, 1333, 3), dtype=torch.uint8, device="cuda"); ob = torch.zeros_like(oa)
    ca.compose_pair(cb, [t.data_ptr() for t in fd], [480 * 3] * 4, oa.data_ptr(), 1333 * 3,
                    [t.data_ptr() for t in fd], [480 * 3] * 4, ob.data_ptr(), 1333 * 3, s)
    torch.cuda.synchronize()
    assert np.array_equal(oa.cpu().numpy(), po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 2)[0])
    assert np.array_equal(ob.cpu().numpy(), po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 4)[0])


def test_cut_alignment_on_vector_path(pano, po, rig_r):
    """cuts whose origin / size are not multiples of the 4 x 2 blocks of the vector blend kernel (canvas >= 600 kpx)"""
    st = rig_r["stitchers"][0]
    v = st["cams"]
    d = {"n": 2, "w": 960, "h": 540, "scale": v[-1], "K": [v[0:9], v[18:27]], "R": [v[9:18], v[27:36]]}
    frames = [synth_frame(960, 540, 51 + i) for i in range(2)]
    ctx = make_ctx(pano, d, 0, num_bands=3)
    ctx.build_masks_voronoi()
    masks = [ctx.get_mask(i) for i in range(2)]
    full, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, 3)
    assert np.array_equal(ctx.compose_host(frames), full)
    for cut in ((1, 1, 5, 3), (3, 7, 1449, 516), (10, 136, 1430, 250), (1451, 522, 1, 1), (2, 0, 6, 523), (0, 521, 1452, 2),
                (13, 3, 17, 9)):
        ctx.set_cut(cut)
        got = ctx.compose_host(frames)
        assert np.array_equal(got, full[cut[1]:cut[1] + cut[3], cut[0]:cut[0] + cut[2]]), cut


def test_frame_slots_overlapping_frames(pano, po, torch):
    """pano_set_frame_slots / pano_select_frame_slot: frames composed into different slots on different streams
    (the launch chains of consecutive frames overlap on the GPU) give the same bytes as one frame at a time"""
    d = c2_group(w=640, h=360, f=334.0)
    ctx = make_ctx(pano, d, 0, num_bands=4)
    ctx.build_masks_voronoi()
    n, nslot = d["n"], 3
    frames = [[synth_frame(d["w"], d["h"], 500 + 10 * k + i) for i in range(n)] for k in range(6)]
    frames_d = [[torch.from_numpy(f).cuda() for f in fr] for fr in frames]
    ow, oh = ctx.output_size()
    s0 = torch.cuda.current_stream().cuda_stream
    ref = []
    for fr in frames_d:  # one frame at a time, slot 0
        out = torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda")
        ctx.compose([t.data_ptr() for t in fr], [d["w"] * 3] * n, out.data_ptr(), ow * 3, s0)
        torch.cuda.synchronize()
        ref.append(out.cpu().numpy())
    masks = [ctx.get_mask(i) for i in range(n)]
    assert np.array_equal(ref[0], po.compose(frames[0], d["K"], d["R"], d["scale"], masks, 4)[0])
    ctx.set_frame_slots(nslot)
    streams = [torch.cuda.Stream() for _ in range(nslot)]
    outs = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in frames]
    torch.cuda.synchronize()
    for rep in range(2):
        for k, fr in enumerate(frames_d):
            ctx.select_frame_slot(k % nslot)
            ctx.compose([t.data_ptr() for t in fr], [d["w"] * 3] * n, outs[k].data_ptr(), ow * 3, streams[k % nslot].cuda_stream)
        torch.cuda.synchronize()
        for k in range(len(frames)):
            assert np.array_equal(outs[k].cpu().numpy(), ref[k]), (rep, k)
            outs[k].zero_()
    # back to one slot
    ctx.set_frame_slots(1)
    out = torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda")
    ctx.compose([t.data_ptr() for t in frames_d[1]], [d["w"] * 3] * n, out.data_ptr(), ow * 3, s0)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref[1])
    with pytest.raises(Exception):
        ctx.select_frame_slot(2)


def test_live_rects_follow_the_masks(pano, po, c1, monkeypatch):
    """pano_get_live_rect: the part of each pyramid level the library produces.  It follows the masks (whole tile
    until the first compose after a change), PANO_FULL_TILES=1 switches the skipping off, and the panorama is the
    oracle's either way"""
    monkeypatch.delenv("PANO_FULL_TILES", raising=False)
    masks = oracle_masks(po, c1)
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 4)
    ctx = make_ctx(pano, c1, 0, num_bands=4)
    (tx, ty, tw, th), (top, bottom, left, right) = ctx.feed_tile(0)
    assert ctx.live_rect(0, 0) == (0, 0, tw, th)          # no masks yet: everything
    for i in range(4):
        ctx.set_mask(i, masks[i])
    assert np.array_equal(ctx.compose_host(c1["frames"]), want)
    shrunk = 0
    for i in range(4):
        (tx, ty, tw, th), (top, bottom, left, right) = ctx.feed_tile(i)
        ys, xs = np.nonzero(masks[i])
        for l in range(5):
            lx, ly, lw, lh = ctx.live_rect(i, l)
            assert 0 <= lx and 0 <= ly and lx + lw <= (tw >> l) and ly + lh <= (th >> l)
            # the pixels that carry weight at level 0 are inside, and so is their footprint at the coarser levels
            assert lx <= (xs.min() + left) >> l and ((xs.max() + left) >> l) < lx + lw
            assert ly <= (ys.min() + top) >> l and ((ys.max() + top) >> l) < ly + lh
        shrunk += ctx.live_rect(i, 0)[2] < tw
    assert shrunk >= 2                                     # the outer cameras lose a good part of their tiles
    # a mask change resets to the whole tile until the next compose; the result follows the new masks
    m0 = masks[0].copy(); m0[:, : m0.shape[1] // 2] = 0
    ctx.set_mask(0, m0)
    (tx, ty, tw, th), _ = ctx.feed_tile(0)
    assert ctx.live_rect(0, 0) == (0, 0, tw, th)
    want2, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], [m0] + masks[1:], 4)
    assert np.array_equal(ctx.compose_host(c1["frames"]), want2)
    assert ctx.live_rect(0, 0)[2] < tw
    # whole tiles on request
    monkeypatch.setenv("PANO_FULL_TILES", "1")
    full = make_ctx(pano, c1, 0, num_bands=4)
    for i in range(4):
        full.set_mask(i, masks[i])
    assert np.array_equal(full.compose_host(c1["frames"]), want)
    (tx, ty, tw, th), _ = full.feed_tile(1)
    assert full.live_rect(1, 0) == (0, 0, tw, th)


@pytest.mark.parametrize("offset,pad", [(0, 0), (4, 0), (8, 16), (1, 0), (2, 5), (4, 4)])
def test_frame_pointer_alignment_and_strides(pano, po, torch, c1, offset, pad):
    """cv::Mat views: frames that start at any byte offset with any row stride.  4-byte aligned frames with strides
    that are a multiple of 16 take the LDS warp kernel (also when the pointer is not 16-byte aligned), everything else
    the general kernel - same bytes either way"""
    masks = oracle_masks(po, c1)
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 3)
    ctx = make_ctx(pano, c1, 0, num_bands=3)
    for i in range(4):
        ctx.set_mask(i, masks[i])
    stride = 480 * 3 + pad
    bufs, ptrs = [], []
    for f in c1["frames"]:
        host = np.zeros(offset + stride * 270 + 64, np.uint8)
        rows = host[offset:offset + stride * 270].reshape(270, stride)
        rows[:, :480 * 3] = f.reshape(270, 480 * 3)
        rows[:, 480 * 3:] = 0xA5           # stride padding must never be sampled
        b = torch.from_numpy(host).cuda()
        bufs.append(b)
        ptrs.append(b.data_ptr() + offset)
    out = torch.zeros((257, 1333, 3), dtype=torch.uint8, device="cuda")
    ctx.compose(ptrs, [stride] * 4, out.data_ptr(), 1333 * 3, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("offset,pad", [(1, 0), (2, 7), (3, 1), (4, 4), (0, 13)])
def test_output_pointer_alignment_and_stride(pano, po, torch, c1, offset, pad):
    """the panorama may land anywhere: any byte offset, any row stride >= 3 * width; bytes between rows stay untouched"""
    masks = oracle_masks(po, c1)
    want, _ = po.compose(c1["frames"], c1["K"], c1["R"], c1["scale"], masks, 3)
    ctx = make_ctx(pano, c1, 0, num_bands=3)
    for i in range(4):
        ctx.set_mask(i, masks[i])
    d = [torch.from_numpy(f).cuda() for f in c1["frames"]]
    stride = 1333 * 3 + pad
    buf = torch.full((offset + stride * 257 + 16,), 0x5A, dtype=torch.uint8, device="cuda")
    ctx.compose([t.data_ptr() for t in d], [480 * 3] * 4, buf.data_ptr() + offset, stride, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    rows = host[offset:offset + stride * 257].reshape(257, stride)
    assert np.array_equal(rows[:, :1333 * 3].reshape(257, 1333, 3), want)
    assert (rows[:, 1333 * 3:] == 0x5A).all() and (host[:offset] == 0x5A).all() and (host[offset + stride * 257:] == 0x5A).all()


def test_randomised_rigs_through_the_lds_warp_kernel(pano, po, torch):
    """the same kind of seeded random rigs, but frame widths that are multiples of 16 pixels and device-resident frames:
    these take the table kernel with LDS-staged source boxes (packed table, escapes, box fall-backs to global taps, the
    byte-wise taps at the end of the frame) and the live rects; roll / pitch up to 25 degrees make large, skewed boxes"""
    import math
    seed, cases = int(os.environ.get("PANO_FUZZ_SEED", "77")), int(os.environ.get("PANO_FUZZ_CASES", "30"))
    rng = np.random.default_rng(seed)
    st = torch.cuda.current_stream().cuda_stream
    done = 0
    for case in range(cases):
        n = int(rng.integers(2, 5))
        w, h = 16 * int(rng.integers(3, 40)), int(rng.integers(24, 300))
        f = float(rng.uniform(0.5, 1.8)) * w
        step = float(rng.uniform(8.0, 35.0)) * (1 if rng.random() < 0.5 else -1)
        yaw0 = -step * (n - 1) / 2 + float(rng.uniform(-5, 5))
        kind = int(rng.integers(0, 2))
        bands = int(rng.integers(0, 7))
        big = case % 4 == 0
        K = [f, 0.0, w / 2.0 + float(rng.uniform(-5, 5)), 0.0, f * float(rng.uniform(0.95, 1.05)), h / 2.0, 0.0, 0.0, 1.0]
        Rs = []
        for i in range(n):
            a = math.radians(yaw0 + i * step)
            b = math.radians(float(rng.uniform(-25, 25) if big else rng.uniform(-6, 6)))
            c = math.radians(float(rng.uniform(-25, 25) if big else rng.uniform(-4, 4)))
            ry_ = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
            rx_ = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
            rz_ = np.array([[math.cos(c), -math.sin(c), 0], [math.sin(c), math.cos(c), 0], [0, 0, 1]])
            Rs.append((ry_ @ rx_ @ rz_).astype(np.float32).reshape(9).tolist())
        d = {"K": [K] * n, "R": Rs, "scale": f * float(rng.uniform(0.9, 1.1)), "w": w, "h": h, "n": n}
        try:
            ctx = make_ctx(pano, d, kind, num_bands=bands)
        except pano.PanoError as e:
            assert e.status == -6
            continue
        frames = [synth_frame(w, h, 300 + case * 8 + i) for i in range(n)]
        ctx.build_masks_voronoi()
        masks = [ctx.get_mask(i) for i in range(n)]
        want, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, bands, kind=kind)
        fd = [torch.from_numpy(fr).cuda() for fr in frames]
        ow, oh = ctx.output_size()
        out = torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda")
        for rep in range(2):   # the second frame runs with the live rects of the masks
            out.zero_()
            ctx.compose([t.data_ptr() for t in fd], [w * 3] * n, out.data_ptr(), ow * 3, st)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), want), (case, rep, n, w, h, kind, bands)
        done += 1
    assert done >= cases * 3 // 5


@pytest.mark.parametrize("w,h,f", [(2048, 1152, 1069.0), (2560, 1440, 1336.0)])
def test_largest_frames_the_remap_table_takes(pano, po, torch, w, h, f):
    """the table codes are relative to each workgroup's source box, so frames beyond the 11-bit tap indices of one
    code (2048) stay on the table path: 2048-wide frames exercise xs = 2047 in a box at the origin, 2560-wide ones
    boxes whose origin is past 2048, and both the last bytes of the frame"""
    d = c2_group(w=w, h=h, f=f)
    d = {**d, "n": 2, "K": d["K"][:2], "R": d["R"][1:3]}
    ctx = make_ctx(pano, d, 0, num_bands=4)
    ctx.build_masks_voronoi()
    masks = [ctx.get_mask(i) for i in range(2)]
    frames = [synth_frame(d["w"], d["h"], 77 + i) for i in range(2)]
    want, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, 4)
    fd = [torch.from_numpy(f).cuda() for f in frames]
    ow, oh = ctx.output_size()
    out = torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda")
    for rep in range(2):
        out.zero_()
        ctx.compose([t.data_ptr() for t in fd], [d["w"] * 3] * 2, out.data_ptr(), ow * 3, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want), rep
    stats = ctx.warp_table_stats()
    assert stats["table_bytes"] > 0 and stats["blocks"] > 0      # the table path, not the projecting kernel


@pytest.mark.parametrize("case", ["c1", "c1_dark_cylindrical", "c2_small_blocks", "full_size_seam"])
def test_gain_estimation_bit_exact(pano, po, c1, case):
    """ExposureCompensator(GAIN_BLOCKS)::feed as initSeam runs it (ocvstitcher.hpp:981-1032) - resize
    INTER_LINEAR_EXACT, seam-scale LINEAR/REFLECT and NEAREST warps, the pairwise overlap sums (f64, sequential per
    pair) on the GPU, OpenCV's LU and the two smoothing passes on the host: gain maps bit-equal to the oracle's, and the
    panorama composed with them equal to the oracle's composed with its own"""
    kind, block, bands = 0, (32, 32), 2
    if case == "c1":
        d, frames = c1, c1["frames"]
    elif case == "c1_dark_cylindrical":
        d, kind = c1, 1
        frames = [f.copy() for f in c1["frames"]]
        frames[1] = (frames[1].astype(np.uint16) * 5 // 8).astype(np.uint8)
        frames[3] = np.clip(frames[3].astype(np.float32) * 1.3, 0, 255).astype(np.uint8)
    elif case == "c2_small_blocks":
        d = c2_group(w=1280, h=720, f=668.0)
        frames = [np.clip(synth_frame(1280, 720, 60 + i).astype(np.float32) * (0.75 + 0.15 * i), 0, 255).astype(np.uint8) for i in range(4)]
        block = (16, 24)
    else:   # frames below 0.1 Mpx: seam_work_aspect = 1, no resize, seam scale == compose scale
        d = c2_group(w=320, h=200, f=167.0)
        frames = [synth_frame(320, 200, 90 + i) for i in range(4)]
        frames[2] = (frames[2] // 2).astype(np.uint8)
    ctx = make_ctx(pano, d, kind, num_bands=bands)
    assert ctx.gain_map(0) is None
    got = ctx.estimate_gains(frames, block)
    want, _ = po.estimate_gains(frames, d["K"], d["R"], d["scale"], kind, *block)
    for i in range(d["n"]):
        assert got[i].shape == want[i].shape and np.array_equal(got[i], want[i]), (case, i)
    assert max(float(np.abs(g - 1).max()) for g in got) > 0.02          # the estimate is not the trivial one
    masks = oracle_masks(po, d, kind)
    full = []
    for i in range(d["n"]):
        r = ctx.roi(i)
        ctx.set_mask(i, masks[i])
        full.append(po.resize_linear_32f(want[i], r[2], r[3]))
    ref, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, bands, kind=kind, gain_maps=full)
    assert np.array_equal(ctx.compose_host(frames), ref)
    # a second estimate replaces the first; removing the maps restores the uncompensated panorama
    again = ctx.estimate_gains(frames, block)
    assert all(np.array_equal(a, b) for a, b in zip(again, got))
    for i in range(d["n"]):
        ctx.set_gain_map(i, None)
    plain, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, bands, kind=kind)
    assert np.array_equal(ctx.compose_host(frames), plain)


def ring_of_eight(w=480, h=270, f=250.6, first=157.5):
    """eight cameras 45 degrees apart in ONE stitcher (the reference splits them 2 x 4, README.md:27-29).  first = 157.5:
    the cameras at +-157.5 degrees both straddle +-pi and RotationWarper::warpRoi gives them the full panorama width
    (Voronoi then hands each of them one end); first = 180: one camera looks straight at the seam and keeps both ends"""
    K = [f, 0.0, w / 2.0, 0.0, f, h / 2.0, 0.0, 0.0, 1.0]
    return {"n": 8, "w": w, "h": h, "scale": f, "K": [K] * 8, "R": [ry(first - 45.0 * i) for i in range(8)]}


@pytest.mark.parametrize("w,h,f,bands,first", [(480, 270, 250.6, 4, 157.5), (480, 270, 250.6, 4, 180.0), (1920, 1080, 1002.416, 5, 180.0)])
def test_single_ring_of_eight_cameras(pano, po, monkeypatch, w, h, f, bands, first):
    """a 360-degree ring in ONE context.  The +-pi straddlers get RotationWarper::warpRoi's full-width
    ROIs, Voronoi runs between tiles that overlap everywhere; masks and panorama bit-equal to the oracle, with the dead
    middle of the straddlers' tiles stepped over (pano_get_live_gap) and with whole tiles (PANO_FULL_TILES=1)"""
    d = ring_of_eight(w, h, f, first)
    full_w = int(2 * np.pi * f)
    masks = oracle_masks(po, d)
    frames = [synth_frame(w, h, 30 + i) for i in range(8)]
    want, _ = po.compose(frames, d["K"], d["R"], d["scale"], masks, bands)
    for full_tiles in ("0", "1"):
        monkeypatch.setenv("PANO_FULL_TILES", full_tiles)
        ctx = make_ctx(pano, d, 0, num_bands=bands)
        assert ctx.roi(0)[2] >= full_w - 1 and ctx.roi(3)[2] < full_w // 3
        ctx.build_masks_voronoi()
        for i in range(8):
            assert np.array_equal(ctx.get_mask(i), masks[i]), i
        for rep in range(2):
            assert np.array_equal(ctx.compose_host(frames), want), (full_tiles, rep)
        gaps = [ctx.live_gap(i, 0) for i in range(8)]
        two_ended = [i for i in range(8) if ctx.roi(i)[2] >= full_w - 1 and (masks[i][:, :w // 4] != 0).any() and (masks[i][:, -w // 4:] != 0).any()]
        if full_tiles == "1":
            assert all(g == (0, 0) for g in gaps)
        else:
            assert all(gaps[i] == (0, 0) for i in range(8) if i not in two_ended)
            for i in two_ended:     # a straddler whose mask lives at both ends: most of its width is dead
                assert gaps[i][1] > full_w // 2, (i, gaps[i])
                x, y, lw, lh = ctx.live_rect(i, 0)
                assert x <= gaps[i][0] and gaps[i][0] + gaps[i][1] <= x + lw
    assert two_ended or first != 180.0


@pytest.mark.parametrize("case", ["c1", "c1_cylindrical", "rig_r", "c2_1080p", "ties", "half_scale"])
def test_graphcut_masks_bit_exact(pano, po, c1, rig_r, case, tmp_path):
    """the reference's own seam finder: ocvStitcher::updateMask with GraphCutSeamFinder(COST_COLOR)
    (ocvstitcher.hpp:1218-1261, :1033-1035) - seam-scale warps, graph weights and mask update on the GPU, OpenCV's
    Boykov-Kolmogorov max-flow on the host: blend masks bit-equal to the oracle's, and the panorama composed under them"""
    kind, bands = 0, 3
    if case == "c1":
        d, frames = c1, c1["frames"]
    elif case == "c1_cylindrical":
        d, frames, kind = c1, c1["frames"], 1
    elif case == "rig_r":
        v = rig_r["stitchers"][0]["cams"]
        d = {"n": 2, "w": 960, "h": 540, "scale": v[-1], "K": [v[0:9], v[18:27]], "R": [v[9:18], v[27:36]]}
        frames = [synth_frame(960, 540, 70 + i) for i in range(2)]
    elif case == "c2_1080p":
        d = c2_group()
        frames = [synth_frame(1920, 1080, 80 + i) for i in range(4)]
    elif case == "half_scale":
        # 800 x 500 = 4e5 pixels: seam_work_aspect = sqrt(1e5 / 4e5) is EXACTLY 0.5, where cv::resize(INTER_LINEAR_EXACT) hands the
        # frame to INTER_AREA's 2 x 2 box (which OpenCV documents as equal to the bit-exact bilinear there: tests/test_oracle.py)
        d = c2_group(w=800, h=500, f=417.7)
        frames = [synth_frame(800, 500, 170 + i) for i in range(4)]
    else:   # flat frames: every edge costs the same and the labels of the vertices no tree holds decide
        d = c2_group(w=640, h=360, f=334.0)
        frames = [np.full((360, 640, 3), 90 + 20 * i, np.uint8) for i in range(4)]
    ctx = make_ctx(pano, d, kind, num_bands=bands)
    dump = str(tmp_path / "graphs.bin")
    ctx.graphcut_dump(dump)
    ctx.build_masks_graphcut(frames)
    ctx.graphcut_dump(None)
    want = po.prepare_masks_graphcut(frames, d["K"], d["R"], d["scale"], kind)
    for i in range(d["n"]):
        got = ctx.get_mask(i)
        assert got.shape == want[i].shape and np.array_equal(got, want[i]), (case, i, int((got != want[i]).sum()))
    # Evidence that is not sibling against sibling: on every pair's graph AS THE GPU BUILT IT, checked with an
    # independent max-flow (SciPy's Dinic): every vertex the source still reaches in the residual graph is labelled source, every
    # vertex that still reaches the sink is labelled sink (what ALL minimum cuts share), and - wherever image content makes the
    # costs distinct - the labelling IS a minimum cut: its capacity equals the max-flow value.  (On the flat "ties" frames
    # GCGraph's labelling is not one: vertices that end in neither search tree keep the label of the tree they were last in,
    # cv::detail::GCGraph::inSourceSegment reads `t == 0`, and a patchwork of such vertices cuts through unsaturated edges - found
    # here; the product reproduces it because the reference's masks are the target, and the oracle comparison above covers
    # which labels those are.)
    pairs = read_graphcut_dump(dump)
    assert len(pairs) >= d["n"] - 1
    for (i, j, term, wh, wv, lab) in pairs:
        flow, src_side, snk_side = scipy_max_flow(term, wh, wv, sides=True)
        assert (lab[src_side] == 1).all() and (lab[snk_side] == 0).all(), (case, i, j)
        cap = min_cut_capacity(term, wh, wv, lab)
        assert cap >= flow and (cap == flow or case == "ties"), (case, i, j, cap, flow)
    ref, _ = po.compose(frames, d["K"], d["R"], d["scale"], want, bands, kind=kind)
    assert np.array_equal(ctx.compose_host(frames), ref)
    # the seams depend on the frames: other content, other masks; and the Voronoi masks come back on request
    if case == "c1":
        other = [np.ascontiguousarray(f[::-1]) for f in frames]
        ctx.build_masks_graphcut(other)
        w2 = po.prepare_masks_graphcut(other, d["K"], d["R"], d["scale"], kind)
        assert all(np.array_equal(ctx.get_mask(i), w2[i]) for i in range(4))
        assert any(not np.array_equal(w2[i], want[i]) for i in range(4))
        ctx.build_masks_voronoi()
        vor = oracle_masks(po, d, kind)
        assert all(np.array_equal(ctx.get_mask(i), vor[i]) for i in range(4))


def test_mask_refresh_with_frames_in_flight(pano, po, torch):
    """the updateMask cadence (ocvstitcher.hpp:1152-1159) under frames in flight: the weights are shared by all frame
    slots, so a mask change while earlier frames are still running has to let them finish under the OLD masks - every
    frame equals the oracle under the masks that were current when it was submitted"""
    d = c2_group(w=960, h=540, f=501.0)
    n, nslot = d["n"], 4
    ctx = make_ctx(pano, d, 0, num_bands=4)
    ctx.build_masks_voronoi()
    ctx.set_frame_slots(nslot)
    vor = [ctx.get_mask(i) for i in range(n)]
    frames = [[synth_frame(d["w"], d["h"], 700 + 10 * k + i) for i in range(n)] for k in range(10)]
    frames_d = [[torch.from_numpy(f).cuda() for f in fr] for fr in frames]
    ow, oh = ctx.output_size()
    streams = [torch.cuda.Stream() for _ in range(nslot)]
    outs = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in frames]
    gc_masks = po.prepare_masks_graphcut(frames[5], d["K"], d["R"], d["scale"])
    assert any(not np.array_equal(a, b) for a, b in zip(vor, gc_masks))
    torch.cuda.synchronize()
    for k, fr in enumerate(frames_d):
        if k == 5:      # no synchronisation by the caller: frames 2..4 are still in flight
            ctx.build_masks_graphcut(frames[5])
        ctx.select_frame_slot(k % nslot)
        ctx.compose([t.data_ptr() for t in fr], [d["w"] * 3] * n, outs[k].data_ptr(), ow * 3, streams[k % nslot].cuda_stream)
    torch.cuda.synchronize()
    for k in range(len(frames)):
        want, _ = po.compose(frames[k], d["K"], d["R"], d["scale"], vor if k < 5 else gc_masks, 4)
        assert np.array_equal(outs[k].cpu().numpy(), want), k


def test_mask_refresh_beside_the_frame_loop(pano, po, torch):
    """pano_refresh_masks_*: the graph cuts run on a thread of the library while frames keep being composed; frames composed
    before the masks are installed equal the oracle under the old masks, frames after it under the graph-cut masks - the same
    masks pano_build_masks_graphcut gives; one refresh at a time; a refresh under way ends before pano_destroy"""
    d = c2_group(w=960, h=540, f=501.0)
    n = d["n"]
    ctx = make_ctx(pano, d, 0, num_bands=4)
    ctx.build_masks_voronoi()
    vor = [ctx.get_mask(i) for i in range(n)]
    frames = [[synth_frame(d["w"], d["h"], 900 + 10 * k + i) for i in range(n)] for k in range(3)]
    gc_masks = po.prepare_masks_graphcut(frames[0], d["K"], d["R"], d["scale"])
    ctx.refresh_masks_begin(frames[0])
    with pytest.raises(Exception):
        ctx.refresh_masks_begin(frames[0])          # one at a time
    frames[0][0][:] = 0                              # the caller's buffers are its own again after begin()
    before = []
    installed = False
    for k in range(400):                             # the frame loop goes on; poll() never blocks
        if ctx.refresh_masks_poll():
            installed = True
            break
        before.append(ctx.compose_host(frames[1]))
    if not installed:
        ctx.refresh_masks_wait()
    assert all(np.array_equal(ctx.get_mask(i), gc_masks[i]) for i in range(n))
    want_old, _ = po.compose(frames[1], d["K"], d["R"], d["scale"], vor, 4)
    assert all(np.array_equal(b, want_old) for b in before[:3])
    want_new, _ = po.compose(frames[2], d["K"], d["R"], d["scale"], gc_masks, 4)
    assert np.array_equal(ctx.compose_host(frames[2]), want_new)
    assert not ctx.refresh_masks_poll()              # nothing under way: not an error
    # the synchronous entry waits for a refresh under way and then does its own
    ctx.refresh_masks_begin(frames[2])
    ctx.build_masks_graphcut(frames[1])
    gc1 = po.prepare_masks_graphcut(frames[1], d["K"], d["R"], d["scale"])
    assert all(np.array_equal(ctx.get_mask(i), gc1[i]) for i in range(n))
    ctx.refresh_masks_begin(frames[2])               # left running: destroying the context joins it
    ctx.close()


def test_bench_shape_bit_exact(pano, po, torch):
    """the exact launch shape bench.py times (config 2): pano_compose_pair over 2 x 4 x 1080p, 5 bands, Voronoi seams,
    pano_set_frame_slots(4) with step k in slot k % 4 on stream k % 4 - three distinct frame sets dealt over 12 steps in
    flight, EVERY panorama of every step against the oracle"""
    g = c2_group()
    W, H, NG, NC, F = g["w"], g["h"], 2, 4, 4
    ctxs = []
    for grp in range(NG):
        ctx = make_ctx(pano, g, 0, num_bands=5)
        ctx.build_masks_voronoi()
        ctx.set_frame_slots(F)
        ctxs.append(ctx)
    ow, oh = ctxs[0].output_size()
    assert (ow, oh) == (3893, 991)
    masks = [ctxs[0].get_mask(i) for i in range(NC)]
    sets = [[[synth_frame(W, H, 900 + 100 * s + grp * NC + i) for i in range(NC)] for grp in range(NG)] for s in range(3)]
    sets_d = [[[torch.from_numpy(f).cuda() for f in fr] for fr in st] for st in sets]
    po.set_threads(16)
    wants = [[po.compose(st[grp], g["K"], g["R"], g["scale"], masks, 5)[0] for grp in range(NG)] for st in sets]
    po.set_threads(1)
    steps = 12
    streams = [torch.cuda.Stream() for _ in range(F)]
    outs = [[torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(NG)] for _ in range(steps)]
    strides = [W * 3] * NC
    torch.cuda.synchronize()
    for rep in range(2):
        for k in range(steps):
            f = k % F
            fr = sets_d[k % 3]
            ctxs[0].select_frame_slot(f)
            ctxs[1].select_frame_slot(f)
            ctxs[0].compose_pair(ctxs[1], [t.data_ptr() for t in fr[0]], strides, outs[k][0].data_ptr(), ow * 3,
                                 [t.data_ptr() for t in fr[1]], strides, outs[k][1].data_ptr(), ow * 3, streams[f].cuda_stream)
        torch.cuda.synchronize()
        for k in range(steps):
            for grp in range(NG):
                got = outs[k][grp].cpu().numpy()
                assert np.array_equal(got, wants[k % 3][grp]), (rep, k, grp, int((got != wants[k % 3][grp]).sum()))
                outs[k][grp].zero_()


def test_bundled_set_second_half_c1b(pano, po, c1b):
    """the other half of the bundled set (2222/5..8.png under the last record of 2222/cameraparaout_2.txt): every stage
    against the oracle and against the committed golden vectors (tests/golden/c1b_golden.json)"""
    import hashlib
    import json
    from conftest import GOLDEN
    g = json.load(open(os.path.join(GOLDEN, "c1b_golden.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    d = c1b
    ctx = make_ctx(pano, d, 0, num_bands=4)
    assert [list(ctx.roi(i)) for i in range(4)] == g["rois"]
    ctx.build_masks_voronoi()
    vor = oracle_masks(po, d)
    for i in range(4):
        assert np.array_equal(ctx.get_mask(i), vor[i]) and sha(vor[i]) == g["mask_sha256"][i]
    got = ctx.compose_host(d["frames"])
    assert [got.shape[1], got.shape[0]] == g["pano_size"]
    assert np.array_equal(got, po.compose(d["frames"], d["K"], d["R"], d["scale"], vor, 4)[0])
    assert sha(got) == g["pano_sha256"]["4"]
    # the reference's own seam finder and compensator on these frames
    ctx.build_masks_graphcut(d["frames"])
    gc = po.prepare_masks_graphcut(d["frames"], d["K"], d["R"], d["scale"])
    for i in range(4):
        assert np.array_equal(ctx.get_mask(i), gc[i])
    assert [sha(ctx.get_mask(i)) for i in range(4)] == g["graphcut_mask_sha256"]
    assert sha(ctx.compose_host(d["frames"])) == g["graphcut_pano_b4_sha256"]
    ctx.build_masks_voronoi()
    gains = ctx.estimate_gains(d["frames"])
    want, _ = po.estimate_gains(d["frames"], d["K"], d["R"], d["scale"])
    assert all(np.array_equal(a, b) for a, b in zip(gains, want))
    assert [sha(x) for x in gains] == g["gain_map_sha256"]
    assert sha(ctx.compose_host(d["frames"])) == g["gain_pano_b4_sha256"]


@pytest.mark.parametrize("which", ["r", "s"])
def test_rig_r_on_its_real_frames(pano, po, torch, rig_r_real, rig_s_real, which):
    """rig R (cfg/cameras.yaml 4cam-black/960) on the REAL frames 2222/4cam/0..3.png, and rig S (4cam-silver/640, :212-228) on
    2222/4cam/1/0..3.png, driven like replay.cpp:206-290: two 2-camera stitchers from the 18N+1 lists, graph-cut masks from the
    frames (calibration), bands from strength 1, yaml cut, process, then master.cpp:321-326's resize + vconcat + divider -
    against the oracle and the committed vectors"""
    import hashlib
    import json
    from conftest import GOLDEN, load_png_bgr
    rig = rig_r_real if which == "r" else rig_s_real
    g = json.load(open(os.path.join(GOLDEN, f"{which}_golden.json")))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    halves, halves_d = [], []
    for st, gs in zip(rig, g["stitchers"]):
        ctx = pano.Context(2, st["w"], st["h"], scale=1.0, num_bands=pano.BANDS_FROM_STRENGTH, blend_strength=1.0, cut=st["cut"], device=0)
        ctx.set_cameras_from_list(",".join(repr(float(x)) for x in st["cams"]))   # the yaml's `cams: [...]` text
        ctx.prepare()
        assert [list(ctx.roi(i)) for i in range(2)] == gs["rois"] and ctx.num_bands() == gs["bands"] == (3 if which == "r" else 2)
        ctx.build_masks_graphcut(st["frames"])
        gc = po.prepare_masks_graphcut(st["frames"], st["K"], st["R"], st["scale"])
        for i in range(2):
            assert np.array_equal(ctx.get_mask(i), gc[i])
        assert [sha(ctx.get_mask(i)) for i in range(2)] == gs["graphcut_mask_sha256"]
        got = ctx.compose_host(st["frames"])
        want, _ = po.compose(st["frames"], st["K"], st["R"], st["scale"], gc, gs["bands"], cut=tuple(st["cut"]))
        assert [got.shape[1], got.shape[0]] == gs["pano_cut_size"]
        assert np.array_equal(got, want) and sha(got) == gs["pano_cut_sha256"]
        halves.append(got)
        halves_d.append(torch.from_numpy(got).cuda())
    (uh, uw), (dh, dw) = halves[0].shape[:2], halves[1].shape[:2]
    out = torch.zeros((2 * dh, dw, 3), dtype=torch.uint8, device="cuda")
    ctx.stack_master(halves_d[0].data_ptr(), uw, uh, uw * 3, halves_d[1].data_ptr(), dw, dh, dw * 3, out.data_ptr(), dw * 3,
                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    stacked = out.cpu().numpy()
    assert np.array_equal(stacked, po.stack_master(halves[0], halves[1]))
    assert sha(stacked) == g["stack_master_sha256"] and np.array_equal(stacked, load_png_bgr(os.path.join(GOLDEN, f"{which}_stacked.png")))


def test_host_entry_pageable_pinned_and_strided(pano, po, torch, c1):
    """pano_compose_host = process(vector<Mat>&, Mat&): pageable caller memory (staged through page-locked buffers by the copy
    threads), page-locked caller memory (pano_host_alloc: DMA'd directly), padded row strides on both sides, two stitchers
    called from two threads at once like master.cpp:314-318, and a selected frame slot > 0 left alone - all the oracle's bytes"""
    import threading
    masks = oracle_masks(po, c1)
    sets = [c1["frames"], [np.ascontiguousarray(f[::-1]) for f in c1["frames"]]]
    wants = [po.compose(f, c1["K"], c1["R"], c1["scale"], masks, 3)[0] for f in sets]
    ctxs = []
    for k in range(2):
        ctx = make_ctx(pano, c1, 0, num_bands=3)
        for i in range(4):
            ctx.set_mask(i, masks[i])
        ctxs.append(ctx)
    ctx = ctxs[0]
    ow, oh = ctx.output_size()
    # pageable, tight
    assert np.array_equal(ctx.compose_host(sets[0]), wants[0])
    # pageable, padded rows in and out (views into wider arrays)
    wide = [np.zeros((270, 480 + 7, 3), np.uint8) for _ in range(4)]
    for w_, f in zip(wide, sets[1]):
        w_[:, :480] = f
    out_w = np.full((oh, ow + 5, 3), 77, np.uint8)
    got = ctx.compose_host([w_[:, :480] for w_ in wide], out=out_w[:, :ow])
    assert np.array_equal(got, wants[1]) and (out_w[:, ow:] == 77).all()
    # page-locked frames and output: no staging
    pin = [pano.HostBuffer((270, 480, 3)) for _ in range(4)]
    pout = pano.HostBuffer((oh, ow, 3))
    for rep in range(2):
        for b, f in zip(pin, sets[rep]):
            b.array[:] = f
        pout.array[:] = 0
        assert np.array_equal(ctx.compose_host([b.array for b in pin], out=pout.array), wants[rep])
    # mixed: two cameras page-locked, two pageable, pageable output
    assert np.array_equal(ctx.compose_host([pin[0].array, sets[1][1], pin[2].array, sets[1][3]]), wants[1])
    # a frame slot > 0 selected by the caller stays selected and untouched (the host entry works in slot 0)
    ctx.set_frame_slots(2)
    ctx.select_frame_slot(1)
    fd = [torch.from_numpy(f).cuda() for f in sets[0]]
    o1 = torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda")
    st1 = torch.cuda.Stream()
    ctx.compose([t.data_ptr() for t in fd], [480 * 3] * 4, o1.data_ptr(), ow * 3, st1.cuda_stream)
    assert np.array_equal(ctx.compose_host(sets[1]), wants[1])
    ctx.compose([t.data_ptr() for t in fd], [480 * 3] * 4, o1.data_ptr(), ow * 3, st1.cuda_stream)   # still slot 1
    torch.cuda.synchronize()
    assert np.array_equal(o1.cpu().numpy(), wants[0])
    # two stitchers from two threads, many frames
    res = [[None] * 6 for _ in range(2)]

    def run(k):
        for j in range(6):
            res[k][j] = ctxs[k].compose_host(sets[(j + k) % 2])

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        for j in range(6):
            assert np.array_equal(res[k][j], wants[(j + k) % 2]), (k, j)
    for b in pin + [pout]:
        b.close()


def test_host_entries_upload_only_what_the_warp_reads(pano, po):
    """pano_get_source_rect: with masks set, the host entries (pano_compose_host, pano_stream_submit) upload only the frame bytes
    the warp's live patches tap.  Proof that nothing outside that rectangle is ever read: frames whose bytes OUTSIDE it are
    garbage give the oracle's panorama of the clean frames, through both entries; and the rectangle really is smaller than the
    frame (config 2 geometry: each camera keeps about 70 % of its columns)"""
    d = c2_group(w=960, h=540, f=501.2)
    clean = [synth_frame(960, 540, 300 + i) for i in range(4)]
    ctx = make_ctx(pano, d, 0, num_bands=4)
    full = [ctx.source_rect(i) for i in range(4)]
    assert all(r == (0, 0, 2880, 540) for r in full)            # no masks yet: every patch is live, whole frames
    ctx.build_masks_voronoi()
    masks = [ctx.get_mask(i) for i in range(4)]
    want, _ = po.compose(clean, d["K"], d["R"], d["scale"], masks, 4)
    assert np.array_equal(ctx.compose_host(clean), want)        # the first frame after new masks: weights rebuilt, rects shrink
    rects = [ctx.source_rect(i) for i in range(4)]
    share = sum(r[2] * r[3] for r in rects) / (4 * 2880 * 540)
    assert all(r[0] % 64 == 0 and r[2] % 64 == 0 and r[2] > 0 for r in rects) and 0.4 < share < 0.9, (rects, share)
    rng = np.random.default_rng(5)
    dirty = []
    for f, (x0, y0, w, rows) in zip(clean, rects):
        g = rng.integers(0, 256, f.shape, dtype=np.uint8)
        flat, gflat = f.reshape(540, 2880), g.reshape(540, 2880)
        gflat[y0:y0 + rows, x0:x0 + w] = flat[y0:y0 + rows, x0:x0 + w]
        dirty.append(g)
    assert np.array_equal(ctx.compose_host(dirty), want)
    pin = [pano.HostBuffer((540, 960, 3)) for _ in range(4)]   # page-locked caller frames: the direct DMA path
    for b, f in zip(pin, dirty):
        b.array[:] = f
    assert np.array_equal(ctx.compose_host([b.array for b in pin]), want)
    for s in range(2):
        for i in range(4):
            ctx.stream_input(s, i)[:] = dirty[i]
        ctx.stream_submit(s)
        ctx.stream_wait(s)
        assert np.array_equal(ctx.stream_output(s), want)
    # new masks (one camera dropped): the rectangles follow, the first frame behind them is uploaded whole
    ctx.set_mask(1, np.zeros_like(masks[1]))
    m2 = [masks[0], np.zeros_like(masks[1]), masks[2], masks[3]]
    want2, _ = po.compose(clean, d["K"], d["R"], d["scale"], m2, 4)
    assert np.array_equal(ctx.compose_host(clean), want2)
    assert ctx.source_rect(1)[3] == 0                            # nothing of camera 1 is read any more
    for b in pin:
        b.close()


def test_frame_streams_run_side_by_side_and_compose_right(pano, po, torch, c1):
    """pano_frame_streams: four streams of the library's own, probed against each other with spin kernels so that each sits on a
    hardware queue of its own (the runtime's default gives a process four) - and frames composed on them, four in flight, are the
    oracle's.  A second call hands out the same streams"""
    masks = oracle_masks(po, c1)
    ctx = make_ctx(pano, c1, 0, num_bands=3)
    for i in range(4):
        ctx.set_mask(i, masks[i])
    streams, distinct = ctx.frame_streams(4)
    # the handles are hard; how many hardware queues the probe found is a wall-clock measurement (GPU_MAX_HW_QUEUES, other
    # users of the device, host jitter) and, like every clock in this suite, asserted only under PANO_STRICT_TIMING=1
    assert len(set(streams)) == 4 and all(streams) and 1 <= distinct <= 4, (streams, distinct)
    print("pano_frame_streams: %d of 4 flight streams on hardware queues of their own" % distinct)
    if os.environ.get("PANO_STRICT_TIMING") == "1":
        assert 3 <= distinct <= 4, distinct
    assert ctx.frame_streams(4)[0] == streams and ctx.frame_streams(2)[0] == streams[:2]
    ctx.set_frame_slots(4)
    sets = [c1["frames"], [np.ascontiguousarray(f[::-1]) for f in c1["frames"]], [np.ascontiguousarray(f[:, ::-1]) for f in c1["frames"]]]
    wants = [po.compose(fr, c1["K"], c1["R"], c1["scale"], masks, 3)[0] for fr in sets]
    dev = [[torch.from_numpy(f).cuda() for f in fr] for fr in sets]
    ow, oh = ctx.output_size()
    outs = [torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    for rnd in range(3):
        for k in range(4):
            ctx.select_frame_slot(k)
            ctx.compose([t.data_ptr() for t in dev[(k + rnd) % 3]], [480 * 3] * 4, outs[k].data_ptr(), ow * 3, streams[k])
        torch.cuda.synchronize()
        for k in range(4):
            assert np.array_equal(outs[k].cpu().numpy(), wants[(k + rnd) % 3]), (rnd, k)


def test_258st_frames_bit_exact(pano, po, st258):
    """the last of the bundled inputs, 2222/258st/1..8.png (320x180 after a 2x2 box), as two 4-camera groups: Voronoi masks and the
    4-band panorama equal the oracle's and the committed hashes; graph-cut masks from these frames equal the oracle's too"""
    import hashlib
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    for d in st258:
        ctx = make_ctx(pano, d, 0, num_bands=4)
        ctx.build_masks_voronoi()
        masks = [ctx.get_mask(i) for i in range(4)]
        assert [sha(m) for m in masks] == d["golden"]["mask_sha256"]
        got = ctx.compose_host(d["frames"])
        assert sha(got) == d["golden"]["pano_b4_sha256"]
        ctx.build_masks_graphcut(d["frames"])
        gc = po.prepare_masks_graphcut(d["frames"], d["K"], d["R"], d["scale"])
        assert all(np.array_equal(ctx.get_mask(i), gc[i]) for i in range(4))
        want, _ = po.compose(d["frames"], d["K"], d["R"], d["scale"], gc, 4)
        assert np.array_equal(ctx.compose_host(d["frames"]), want)
