"""CPU tests of the product's boundary: the C-ABI library loads, exports every symbol include/pano.h
declares, and its host-side geometry (plan-only contexts: no GPU, no compute) matches the oracle.
Compute entry points must refuse to run without a device - there is no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import c2_group, c4_rig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(pano):
    pano.build()
    return pano.load_library()


def test_exports_match_header(pano, lib):
    hdr = open(os.path.join(ROOT, "include", "pano.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(pano_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(pano.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.pano_version()


def make_plan(pano, d, kind, **kw):
    ctx = pano.Context(d["n"], d["w"], d["h"], scale=d["scale"], projector=kind, device=-1, **kw)
    for i in range(d["n"]):
        ctx.set_camera(i, d["K"][i], d["R"][i])
    ctx.prepare()
    return ctx


@pytest.mark.parametrize("name,kind,bands", [("c1", 0, 2), ("c1", 0, 4), ("c1", 0, 0), ("c1", 1, 3), ("c2", 0, 5), ("c4", 1, 7)])
def test_plan_matches_oracle(pano, po, c1, name, kind, bands):
    d = {"c1": c1, "c2": c2_group(), "c4": c4_rig()}[name]
    ctx = make_plan(pano, d, kind, num_bands=bands)
    rois = [po.warp_roi(po.projector(kind, d["scale"], d["K"][i], d["R"][i]), d["w"], d["h"]) for i in range(d["n"])]
    assert [ctx.roi(i) for i in range(d["n"])] == rois
    b = po.Blender(bands)
    b.prepare([r[:2] for r in rois], [r[2:] for r in rois])
    assert ctx.pano_rect() == b.dst_roi_final() and ctx.num_bands() == b.num_bands()
    for i in range(d["n"]):
        # feed() geometry only (a 1-row dummy image is enough to exercise the tile box arithmetic)
        img = np.zeros((rois[i][3], rois[i][2], 3), np.int16)
        b.feed(img, np.zeros((rois[i][3], rois[i][2]), np.uint8), rois[i][:2])
        assert ctx.feed_tile(i) == b.last_tile()
    assert ctx.output_size() == b.dst_roi_final()[2:]
    src, dst = ctx.warp_bytes()
    assert src == d["n"] * d["w"] * d["h"] * 3
    assert dst == sum(ctx.feed_tile(i)[0][2] * ctx.feed_tile(i)[0][3] * 3 for i in range(d["n"]))


def test_band_rule_and_cut(pano, po, c1, rig_r):
    for s in (1.0, 3.0, 5.0):
        ctx = make_plan(pano, c1, 0, num_bands=pano.BANDS_FROM_STRENGTH, blend_strength=s)
        assert ctx.num_bands() == po.bands_from_strength(1333, 257, s)
    ctx = make_plan(pano, c1, 0, num_bands=pano.BANDS_FROM_STRENGTH, blend_strength=0.01)
    assert ctx.num_bands() == -1      # Blender::NO
    st = rig_r["stitchers"][0]
    ctx = pano.Context(2, 960, 540, num_bands=pano.BANDS_FROM_STRENGTH, blend_strength=1.0, cut=st["cut"], device=-1)
    ctx.set_cameras_from_list(",".join(repr(v) for v in st["cams"]))
    ctx.prepare()
    assert ctx.pano_rect() == (-721, 497, 1452, 523) and ctx.num_bands() == 3
    assert ctx.output_size() == (1430, 250)
    with pytest.raises(pano.PanoError):
        ctx.set_cut((0, 0, 2000, 10))


def test_camera_file_loader(pano, c1, tmp_path):
    # old 7-line shared-K record (2222/cameraparaout_1.txt) preceded by an older record that must be skipped
    K = ",".join(repr(v) for v in c1["K"][0]) + ","
    rec = ["2021-01-01-00-00-00:", "1,0,0,0,1,0,0,0,1,"] + ["1,0,0,0,1,0,0,0,1,"] * 4 + ["100"]
    rec += ["2021-11-17-10-25-21:", K] + [",".join(repr(v) for v in r) + "," for r in c1["R"]] + [repr(c1["scale"])]
    p = tmp_path / "cameraparaout_1.txt"
    p.write_text("\n".join(rec) + "\n")
    ctx = pano.Context(4, 480, 270, num_bands=2, device=-1)
    ctx.load_camera_file(str(p))
    ctx.prepare()
    assert ctx.pano_rect() == (-1121, 475, 1333, 257)
    # new format written by saveCameraParams: N lines of 18 values + scale
    rec = ["2022-10-10-16-25-01:"] + [K + ",".join(repr(v) for v in r) + "," for r in c1["R"]] + [repr(c1["scale"])]
    p.write_text("\n".join(rec) + "\n")
    ctx = pano.Context(4, 480, 270, num_bands=2, device=-1)
    ctx.load_camera_file(str(p))
    ctx.prepare()
    assert ctx.pano_rect() == (-1121, 475, 1333, 257)
    with pytest.raises(pano.PanoError):
        pano.Context(4, 480, 270, device=-1).load_camera_file(str(tmp_path / "missing.txt"))


def test_camera_file_writer_roundtrip(pano, c1, tmp_path):
    """saveCameraParams format (ocvstitcher.hpp:522-562): append a record, read it back with the loader"""
    p = tmp_path / "cameraparaout_0.txt"
    ctx = pano.Context(4, 480, 270, scale=c1["scale"], num_bands=2, device=-1)
    for i in range(4):
        ctx.set_camera(i, c1["K"][i], c1["R"][i])
    ctx.save_camera_file(str(p))
    ctx.save_camera_file(str(p))                      # logs are append-only: two records now
    lines = p.read_text().strip().splitlines()
    assert len(lines) == 12 and lines[0].endswith(":") and lines[6].endswith(":")
    assert lines[1].count(",") == 18 and lines[1].startswith("391.047,0,240,0,391.047,135,0,0,1,")
    assert lines[5] == "381.719"
    ctx.prepare()
    ctx2 = pano.Context(4, 480, 270, num_bands=2, device=-1)
    ctx2.load_camera_file(str(p))
    ctx2.prepare()
    assert ctx2.pano_rect() == ctx.pano_rect() and [ctx2.roi(i) for i in range(4)] == [ctx.roi(i) for i in range(4)]


def test_errors_and_no_cpu_fallback(pano, c1):
    with pytest.raises(pano.PanoError):
        pano.Context(0, 480, 270, device=-1)
    with pytest.raises(pano.PanoError):
        pano.Context(9, 480, 270, device=-1)
    ctx = pano.Context(4, 480, 270, scale=c1["scale"], num_bands=2, device=-1)
    with pytest.raises(pano.PanoError):   # cameras missing
        ctx.prepare()
    with pytest.raises(pano.PanoError):   # bad index
        ctx.set_camera(7, c1["K"][0], c1["R"][0])
    with pytest.raises(pano.PanoError):   # list length
        ctx.set_cameras_from_list("1,2,3")
    ctx = make_plan(pano, c1, 0, num_bands=2)
    # every compute entry refuses on a plan-only context: PANO_ENODEVICE, never a host computation
    for fn in (lambda: ctx.compose_host(c1["frames"]), lambda: ctx.build_masks_voronoi(),
               lambda: ctx.set_mask(0, np.zeros((254, 422), np.uint8)), lambda: ctx.blend(0, 0),
               lambda: ctx.warp(0, 0, 0, 0, 0), lambda: ctx.pyramid_slots(), lambda: ctx.debug_level(0, 0),
               lambda: ctx.set_frame_slots(2), lambda: ctx.select_frame_slot(0)):
        with pytest.raises(pano.PanoError) as e:
            fn()
        assert e.value.status == -5
    # the plan-side queries work without a device: before any mask exists every pixel of every level is live
    (tx, ty, tw, th), _ = ctx.feed_tile(0)
    assert ctx.live_rect(0, 0) == (0, 0, tw, th) and ctx.live_rect(0, 2) == (0, 0, tw >> 2, th >> 2)
    for bad in ((9, 0), (0, 9), (-1, 0)):
        with pytest.raises(pano.PanoError) as e:
            ctx.live_rect(*bad)
        assert e.value.status == -2


def test_camera_across_the_projection_seam(pano, po, monkeypatch):
    """a camera looking backwards straddles the +-pi seam: it gets RotationWarper::warpRoi's full-width ROI (SURVEY
    8(f)-4); PANO_WRAP_IS_ERROR=1 keeps the refusal that enforces the reference's 2 x 4 grouping (README.md:27-29)"""
    from helpers import ry
    K = [1002.416, 0, 960, 0, 1002.416, 540, 0, 0, 1]
    ctx = pano.Context(1, 1920, 1080, scale=1002.416, num_bands=2, device=-1)
    ctx.set_camera(0, K, ry(180.0))
    ctx.prepare()
    assert ctx.roi(0) == po.warp_roi(po.projector(po.SPHERICAL, 1002.416, K, ry(180.0)), 1920, 1080)
    assert ctx.roi(0)[2] >= int(2 * 3.14159265 * 1002.416) - 1
    monkeypatch.setenv("PANO_WRAP_IS_ERROR", "1")
    ctx = pano.Context(1, 1920, 1080, scale=1002.416, num_bands=2, device=-1)
    ctx.set_camera(0, K, ry(180.0))
    with pytest.raises(pano.PanoError) as e:
        ctx.prepare()
    assert e.value.status == -6


def test_new_camera_matrix_matches_oracle(pano, po, rig_r):
    """getOptimalNewCameraMatrix(alpha=1) of the undistort front end (nvcam.hpp:830): host planner == oracle"""
    K = [4.890925118101495e+02, 0, 4.940763211103715e+02, 0, 4.912630345468579e+02, 2.865820139005963e+02, 0, 0, 1]
    dist = [-0.2838, 0.0628, 0, 0]
    ctx = pano.Context(2, 960, 540, device=-1)
    ctx.set_undistort(0, (1920, 1080), (960, 540), K, dist, (70, 66, 885, 410))
    assert ctx.new_camera_matrix(0) == po.optimal_new_camera_matrix(K, dist, 960, 540)
    with pytest.raises(pano.PanoError):
        ctx.set_undistort(1, (8200, 2160), (960, 540), K, dist, (70, 66, 885, 410))   # raw frame too large for the table
    ctx.set_cameras_from_list(",".join(repr(v) for v in rig_r["stitchers"][0]["cams"]))
    with pytest.raises(pano.PanoError):
        ctx.prepare()                                                                  # front end on one camera only


def _ry(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)


def test_cameras_are_validated_at_the_boundary(pano, c1, tmp_path):
    """VERDICT r03 #8 (ocvstitcher.hpp:365-421 / SURVEY 5): non-finite values, a scaled R, a sheared R and a reflected R are
    refused with PANO_EINVAL and a reason; the context keeps what it had; the reference's own records pass"""
    ctx = pano.Context(4, 480, 270, scale=c1["scale"], device=-1)
    K, R = np.asarray(c1["K"][0], np.float32).reshape(3, 3), np.asarray(c1["R"][1], np.float32).reshape(3, 3)
    ctx.set_camera(0, K, R)
    bad = []
    k = K.copy(); k[0, 2] = np.nan; bad.append((k, R, "finite"))
    r = R.copy(); r[1, 1] = np.inf; bad.append((K, r, "finite"))
    k = K.copy(); k[0, 0] = 0; bad.append((k, R, "focal"))
    k = K.copy(); k[1, 1] = -391; bad.append((k, R, "focal"))
    bad.append((K, R * 1.01, "orthonormal"))
    r = R.copy(); r[0, 1] += 0.01; bad.append((K, r, "orthonormal"))
    bad.append((K, R @ np.diag([1, -1, 1]).astype(np.float32), "reflection"))
    bad.append((K, np.zeros((3, 3), np.float32), "orthonormal"))
    for k, r, why in bad:
        with pytest.raises(pano.PanoError) as e:
            ctx.set_camera(0, k, r)
        assert e.value.status == -2 and why in str(e.value), (why, str(e.value))
    gk, gr, _ = ctx.get_camera(0)
    assert np.array_equal(np.asarray(gk, np.float32).reshape(3, 3), K) and np.array_equal(np.asarray(gr, np.float32).reshape(3, 3), R)
    # a rotation perturbed inside the tolerance (six significant digits, like the reference's text files) is accepted
    ctx.set_camera(1, K, np.round(_ry(-42.9).astype(np.float64), 5).astype(np.float32))
    # the list and the file loaders are all or nothing: camera 2 of 4 is bad -> nothing changes
    good = [np.concatenate([K.reshape(-1), _ry(-20.0 * i).reshape(-1)]) for i in range(4)]
    vals = np.concatenate(good + [np.float32([c1["scale"]])])
    ctx.set_cameras_from_list(",".join("%.9g" % v for v in vals))
    before = [ctx.get_camera(i) for i in range(4)]
    broken = [g.copy() for g in good]
    broken[2][9:] *= 1.5
    with pytest.raises(pano.PanoError):
        ctx.set_cameras_from_list(",".join("%.9g" % v for v in np.concatenate(broken + [np.float32([c1["scale"]])])))
    p = tmp_path / "cameraparaout_9.txt"
    p.write_text("2026-01-01-00-00-00:\n" + "".join(",".join("%.9g" % v for v in b) + ",\n" for b in broken) + "%g\n" % c1["scale"])
    with pytest.raises(pano.PanoError):
        ctx.load_camera_file(str(p))
    after = [ctx.get_camera(i) for i in range(4)]
    assert all(np.array_equal(np.asarray(a[0]), np.asarray(b[0])) and np.array_equal(np.asarray(a[1]), np.asarray(b[1])) for a, b in zip(before, after))
    with pytest.raises(pano.PanoError):
        ctx.set_cameras_from_list(",".join("%.9g" % v for v in np.concatenate(good + [np.float32([-1.0])])))


def test_verify_cameras_like_the_reference(pano, c1):
    """verifyCamParams (ocvstitcher.hpp:394-417): Euler-angle distance in degrees against stitcherCameraExThres, (fx, fy)
    distance against stitcherCameraInThres; RET_ERR keeps the defaults"""
    n = 4
    ctx = pano.Context(n, 480, 270, scale=c1["scale"], device=-1)
    K = np.asarray(c1["K"][0], np.float32).reshape(3, 3)
    Rs = [_ry(-45.0 * i) for i in range(n)]
    for i in range(n):
        ctx.set_camera(i, K, Rs[i])
    Ks = np.stack([K] * n)
    assert ctx.verify_cameras(Ks, np.stack(Rs), 5.0, 50.0) == (True, -1)
    # camera 2 turned by 3 degrees: inside a 5-degree threshold, outside a 2-degree one
    Re = [r.copy() for r in Rs]
    Re[2] = _ry(-90.0 + 3.0)
    assert ctx.verify_cameras(Ks, np.stack(Re), 5.0, 50.0) == (True, -1)
    assert ctx.verify_cameras(Ks, np.stack(Re), 2.0, 50.0) == (False, 2)
    # the reference's own formula on the same pair
    def euler(R):
        R = R.astype(np.float64)
        sy = np.float32(np.sqrt(R[0, 0] ** 2 + R[1, 0] ** 2))
        return np.float32(np.rad2deg([np.arctan2(R[2, 1], R[2, 2]), np.arctan2(-R[2, 0], sy), np.arctan2(R[1, 0], R[0, 0])]))
    d = float(np.linalg.norm(euler(Rs[2]).astype(np.float64) - euler(Re[2]).astype(np.float64)))
    assert ctx.verify_cameras(Ks, np.stack(Re), d * 1.001, 50.0)[0] and not ctx.verify_cameras(Ks, np.stack(Re), d * 0.999, 50.0)[0]
    # focal lengths: (fx + 30, fy + 40) is 50 away
    Ke = Ks.copy(); Ke[1, 0, 0] += 30; Ke[1, 1, 1] += 40
    assert ctx.verify_cameras(Ke, np.stack(Rs), 5.0, 50.5) == (True, -1)
    assert ctx.verify_cameras(Ke, np.stack(Rs), 5.0, 49.5) == (False, 1)
    Ke[3, 0, 0] = np.nan
    assert ctx.verify_cameras(Ke, np.stack(Rs), 5.0, 1e9) == (False, 3)
