"""the C++ mirror of ocvStitcher (csrc/stitcher.hpp) and the replay example: compile with plain g++ against
the C-ABI, read the reference's two YAML shapes (a stitcher cfg + a cameras.yaml `structures:` entry written
from the committed r_cams.json fixture), check the geometry in plan mode (CPU) and the frame loop on the GPU"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_cfgs(tmp_path, rig_r):
    cams = tmp_path / "cameras.yaml"
    lines = ["cameras:", " -", "  vendor: lijing   # unrelated list", "structures:", " -", "  vendor: lijing",
             "  sensor: imx390", "  sttype: 4cam-black", "  undistor: true", "  fov: 120", "  inputsz: 640", "  params:",
             "   -", "    cams: [0]", "    cut: [0, 0, 10, 10]", " -", "  vendor: lijing", "  sensor: imx390",
             "  sttype: 4cam-black", "  undistor: true", "  fov: 120", "  inputsz: 960", "  params:"]
    for st in rig_r["stitchers"]:
        v = [repr(x) for x in st["cams"]]
        lines += ["   -", "    cams: [" + ",".join(v[:20]) + ",", "            " + ",".join(v[20:]) + "]",
                  "    cut: [%d, %d, %d, %d]" % tuple(st["cut"])]
    cams.write_text("\n".join(lines) + "\n")
    cfg = tmp_path / "stitcher.yaml"
    cfg.write_text("\n".join([
        "# camera vendor", "vendor: lijing", "sensor: imx390", "sttype: 4cam-black", "undistor: true",
        "outPutWidth: 960 # 800", "outPutHeight: 540", "fov: 120", 'cameraparams: "%s"' % cams, "num_images: 2",
        'camcfgpath: "%s/"' % tmp_path, "stitcherMatchConf: 0.3", "stitcherAdjusterConf: 0.7",
        "stitcherBlenderStrength: 1", "stitcherCameraExThres: 30e2", "stitcherCameraInThres: 500e2", "initMode: 2"]) + "\n")
    return cfg


@pytest.fixture(scope="module")
def replay_bin(tmp_path_factory, pano):
    pano.build()
    out = tmp_path_factory.mktemp("bin") / "replay"
    lib_dir = os.path.join(ROOT, "img-stitching_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(ROOT, "examples", "replay.cpp"), "-o", str(out),
                           "-L" + lib_dir, "-lpano_hip", "-Wl,-rpath," + lib_dir, "-lpthread"])
    return str(out)


def test_replay_plan_mode(replay_bin, rig_r, tmp_path):
    cfg = write_cfgs(tmp_path, rig_r)
    r = subprocess.run([replay_bin, str(cfg), "--plan"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    # rig R stitcher 0 pano 1452x523, cut 1430x250, strength 1 -> 3 bands; stitcher 1 1484x509
    assert "stitcher 0: pano 1452x523 at (-721,497), output 1430x250, bands 3" in r.stdout
    assert "stitcher 1: pano 1484x509 at (-733,523), output 1470x250, bands 3" in r.stdout


@pytest.mark.gpu
def test_replay_frames_on_gpu(replay_bin, rig_r, tmp_path):
    cfg = write_cfgs(tmp_path, rig_r)
    r = subprocess.run([replay_bin, str(cfg), "--frames", "2"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "wrote final.ppm 1470x500" in r.stdout
    assert os.path.getsize(tmp_path / "final.ppm") > 1470 * 500 * 3
    # the REAL rig-R frames (2222/4cam/0..3.png as committed fixtures) through the whole C++ flow - init(yaml) x2, calibration
    # (graph-cut masks), two process() threads, pano_stack_master_host - must give the oracle's stacked image byte for byte
    import numpy as np
    from conftest import GOLDEN, load_png_bgr
    ppms = []
    for i in range(4):
        a = load_png_bgr(os.path.join(GOLDEN, f"r_cam{i}.png"))[:, :, ::-1]
        p = tmp_path / f"r{i}.ppm"
        with open(p, "wb") as f:
            f.write(b"P6\n960 540\n255\n" + np.ascontiguousarray(a).tobytes())
        ppms.append(str(p))
    r = subprocess.run([replay_bin, str(cfg), "--frames", "2"] + ppms, capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr + r.stdout
    raw = open(tmp_path / "final.ppm", "rb").read()
    head = b"P6\n1470 500\n255\n"
    assert raw.startswith(head)
    got = np.frombuffer(raw[len(head):], np.uint8).reshape(500, 1470, 3)[:, :, ::-1]
    assert np.array_equal(got, load_png_bgr(os.path.join(GOLDEN, "r_stacked.png")))
    # exposure compensation on: calibration() estimates the block gains (seam-scale tiles of rig R are about 340 x 218
    # -> 11 x 7 blocks of 32), process() applies them
    r = subprocess.run([replay_bin, str(cfg), "--exposure", "--frames", "1"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "stitcher 0: exposure gain maps 11x7 blocks" in r.stdout and "wrote final.ppm 1470x500" in r.stdout
    # the geometry-only Voronoi seam finder instead of the reference's graph cut
    r = subprocess.run([replay_bin, str(cfg), "--voronoi", "--frames", "1"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0 and "wrote final.ppm 1470x500" in r.stdout, r.stderr + r.stdout


@pytest.mark.gpu
def test_replay_loop_with_mask_refresh_beside_it_unpaced(replay_bin, rig_r, tmp_path):
    """the capture loop of src/master.cpp with a graph-cut mask refresh every 10 frames running BESIDE it
    (pano::Stitcher::asyncMaskRefresh -> pano_refresh_masks_*), not paced - no wall clock in this test: every frame is composed,
    and the stacked image behind three refreshes from the same real frames is the oracle's, byte for byte (the paced twin of this
    run lives in test_zz_gpu_paced_loops.py, last in the collection order)"""
    import numpy as np
    from conftest import GOLDEN, load_png_bgr
    cfg = write_cfgs(tmp_path, rig_r)
    ppms = []
    for i in range(4):
        a = load_png_bgr(os.path.join(GOLDEN, f"r_cam{i}.png"))[:, :, ::-1]
        p = tmp_path / f"r{i}.ppm"
        with open(p, "wb") as f:
            f.write(b"P6\n960 540\n255\n" + np.ascontiguousarray(a).tobytes())
        ppms.append(str(p))
    r = subprocess.run([replay_bin, str(cfg), "--frames", "40", "--refresh-every", "10", "--async-refresh"] + ppms,
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr + r.stdout
    assert len([l for l in r.stdout.splitlines() if l.startswith("frame ")]) == 40      # dropped 0: nothing here can drop one
    raw = open(tmp_path / "final.ppm", "rb").read()
    head = b"P6\n1470 500\n255\n"
    assert raw.startswith(head)
    got = np.frombuffer(raw[len(head):], np.uint8).reshape(500, 1470, 3)[:, :, ::-1]
    assert np.array_equal(got, load_png_bgr(os.path.join(GOLDEN, "r_stacked.png")))


@pytest.fixture(scope="module")
def cut_bin(tmp_path_factory, pano):
    pano.build()
    out = tmp_path_factory.mktemp("bin") / "mirror_cut_harness"
    lib_dir = os.path.join(ROOT, "img-stitching_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "src", "mirror_cut_harness.cpp"), "-o",
                           str(out), "-L" + lib_dir, "-lpano_hip", "-Wl,-rpath," + lib_dir, "-lpthread"])
    return str(out)


def test_mirror_cut_rule_per_init_mode(cut_bin, rig_r, tmp_path):
    """m_cutParams as the reference handles it: init(yaml) loads the structure's cut in EVERY mode
    (ocvstitcher.hpp:333-337); mode 3's success path initCamParams -> initSeam (:627-628) keeps it; only a successful initAll
    rewrites it to [0, (rows - cut_h) / 2, cols, cut_h] (:959-964) - the mirror's calibration(imgs, K_est, R_est, scale)."""
    cfg = write_cfgs(tmp_path, rig_r)
    run = lambda *a: subprocess.run([cut_bin, str(cfg), "0"] + list(a), capture_output=True, text=True, cwd=tmp_path)
    r = run("plain")                                   # mode 2 as written
    assert r.returncode == 0 and "pano 1452x523 output 1430x250" in r.stdout, r.stdout + r.stderr
    txt = cfg.read_text().replace("initMode: 2", "initMode: 3")
    cfg.write_text(txt)
    r = run("plain")                                   # mode 3, no cameraparaout_0.txt: the defaults, the yaml cut
    assert r.returncode == 0 and "pano 1452x523 output 1430x250" in r.stdout, r.stdout + r.stderr
    # mode 3 with a record on file (new format, ocvstitcher.hpp:522-562): ITS cameras, still the yaml cut
    cams = rig_r["stitchers"][0]["cams"]
    rec = ["2022-10-10-16-25-01:"]
    for i in range(2):
        v = list(cams[18 * i:18 * i + 18])
        v[0] = v[4] = 498.0                            # another focal than the defaults' 501.208: proof of where K came from
        rec.append(",".join(repr(float(x)) for x in v) + ",")
    rec.append("499.5")
    (tmp_path / "cameraparaout_0.txt").write_text("\n".join(rec) + "\n")
    r = run("plain")
    assert r.returncode == 0 and "output 1430x250" in r.stdout and "fx0 498 scale 499.5" in r.stdout, r.stdout + r.stderr
    # a caller-side bundle adjustment that verifyCamParams accepts (1 degree off the defaults): initAll's cut - the full width,
    # the yaml height centred
    r = run("estimated", "1.0")
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"pano (\d+)x(\d+) output (\d+)x(\d+)", r.stdout)
    pw, ph, ow, oh = map(int, m.groups())
    assert (ow, oh) == (pw, 250) and ph >= 250, r.stdout
    # ... and one it refuses (the yaml's thresholds are 30e2 degrees / 500e2 px: nothing fails them; tighten them)
    cfg.write_text(txt.replace("stitcherCameraExThres: 30e2", "stitcherCameraExThres: 0.5"))
    r = run("estimated", "1.0")
    assert r.returncode == 1 and "calibration RET_ERR" in r.stdout, r.stdout + r.stderr


@pytest.fixture(scope="module")
def sharded_bin(tmp_path_factory, pano):
    """examples/sharded_replay.cpp: the camera-sharded flow (feed -> pano_gather_slots over RCCL -> blend) for a C++ caller,
    plain g++ against the C-ABI - the RCCL call path of the library is compiled and linked here, on the CPU box too"""
    pano.build()
    out = tmp_path_factory.mktemp("bin") / "sharded_replay"
    lib_dir = os.path.join(ROOT, "img-stitching_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(ROOT, "examples", "sharded_replay.cpp"), "-o", str(out),
                           "-I" + os.path.join(ROOT, "include"), "-L" + lib_dir, "-lpano_hip", "-Wl,-rpath," + lib_dir, "-lpthread"])
    return str(out)


def test_sharded_replay_builds_and_rccl_loads(sharded_bin, pano):
    r = subprocess.run([sharded_bin], capture_output=True, text=True)
    assert r.returncode == 2 and "usage: sharded_replay <rank> <world> <id-file>" in r.stderr
    # librccl.so is found and ncclGetUniqueId answers without a GPU: the exchange is there to be called
    uid = pano.Context.rccl_unique_id()
    assert len(uid) == 128 and any(uid)


@pytest.mark.gpu
def test_sharded_replay_single_rank_on_gpu(sharded_bin, pano, tmp_path):
    """world = 1 walks feed -> (no exchange) -> blend on the box's one GPU; the checksum of the panorama must be the one the
    ordinary host entry gives for the same frames"""
    import zlib
    import numpy as np
    from helpers import c2_group
    r = subprocess.run([sharded_bin, "--single", "2"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr + r.stdout
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame ")]
    assert len(lines) == 2 and "3893x991 panorama" in lines[0]
    g = c2_group()
    ctx = pano.Context(4, 1920, 1080, scale=np.float32(1002.416), num_bands=5, device=0)
    for i in range(4):
        ctx.set_camera(i, g["K"][i], g["R"][i])
    ctx.prepare(); ctx.build_masks_voronoi()
    k = np.arange(1920 * 1080 * 3, dtype=np.uint64)
    frames = [(((k * 7 + np.uint64(c * 31)) >> np.uint64(3)) & np.uint64(0xff)).astype(np.uint8).reshape(1080, 1920, 3) for c in range(4)]
    got = ctx.compose_host(frames)
    ctx.feed_cameras_host(0b1111, frames)
    assert np.array_equal(got, ctx.blend_host())          # feed + blend with host buffers == the ordinary host entry
    b = got.reshape(-1).astype(np.uint64)
    w = (np.arange(b.size, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(1)) & np.uint64(0xffffffff)
    want = int(((b * w) & np.uint64(0xffffffff)).sum() & np.uint64(0xffffffff))
    assert "checksum %08x" % want in lines[0], (lines[0], "%08x" % want)
