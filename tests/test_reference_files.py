"""The loaders and the C++ mirror's YAML reader on the reference's OWN data files (build container only: skipped where
/root/reference does not exist, e.g. on the GPU box).  Nothing is copied: the files are read where they lie.

  * pano_load_camera_file on 2222/cameraparaout_{1,2}.txt, cfg/390camcfg/*.txt, cfg/424camcfg/*.txt - append-only logs
    that mix record shapes; the LAST record must come back, equal to an independent Python parse
    and to what tests/golden/make_inputs.py committed
  * the latent bug of initCamParams (ocvstitcher.hpp:486: skip 6*(records-1) lines, right only while every record
    has 6 lines) demonstrated on cfg/390camcfg/cameraparaout_1.txt
  * pano::Stitcher::init on cfg/stitcher-imx390cfg.yaml + cfg/cameras.yaml (examples/replay --plan)
"""
import json
import os
import re
import subprocess

import numpy as np
import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")


def f32(vals):
    return np.asarray([float(v) for v in vals], np.float64).astype(np.float32)


def last_record(path):
    """independent parse: the lines after the last 'timestamp:' line.  Returns (K list per camera, R list, scale)"""
    lines = [l.strip() for l in open(path) if l.strip()]
    idx = max(i for i, l in enumerate(lines) if l.endswith(":"))
    body = lines[idx + 1:]
    rows = [[v for v in l.split(",") if v.strip()] for l in body]
    scale = float(rows[-1][0][:len(rows[-1][0])])
    rows = rows[:-1]
    if all(len(r) == 18 for r in rows):             # saveCameraParams format: K and R per line
        return [f32(r[:9]) for r in rows], [f32(r[9:]) for r in rows], scale
    assert len(rows[0]) == 9 and all(len(r) == 9 for r in rows)   # old format: one shared K, then one R per line
    return [f32(rows[0])] * (len(rows) - 1), [f32(r) for r in rows[1:]], scale


CASES = [("2222/cameraparaout_1.txt", 4, 480, 270), ("2222/cameraparaout_2.txt", 4, 480, 270),
         ("cfg/390camcfg/cameraparaout_0.txt", 2, 960, 540), ("cfg/390camcfg/cameraparaout_1.txt", 2, 960, 540),
         ("cfg/390camcfg/cameraparaout_2.txt", 2, 960, 540), ("cfg/424camcfg/cameraparaout_1.txt", 4, 640, 360),
         ("cfg/424camcfg/cameraparaout_2.txt", 4, 640, 360)]


@pytest.mark.parametrize("rel,n,w,h", CASES)
def test_loader_returns_the_last_record(pano, rel, n, w, h):
    path = os.path.join(REF, rel)
    Ks, Rs, scale = last_record(path)
    assert len(Ks) == n
    ctx = pano.Context(n, w, h, num_bands=2, device=-1)
    ctx.load_camera_file(path)
    for i in range(n):
        K, R, sc = ctx.get_camera(i)
        assert np.array_equal(np.float32(K), Ks[i]) and np.array_equal(np.float32(R), Rs[i]), (rel, i)
        assert np.float32(sc) == np.float32(scale)
    # a context with another camera count refuses the record instead of reading across records
    with pytest.raises(pano.PanoError):
        pano.Context(n + 1, w, h, device=-1).load_camera_file(path)


def test_loader_agrees_with_committed_fixture_inputs(pano):
    from conftest import GOLDEN
    for rel, fix in (("2222/cameraparaout_1.txt", "c1_cams.json"), ("2222/cameraparaout_2.txt", "c1b_cams.json")):
        d = json.load(open(os.path.join(GOLDEN, fix)))
        ctx = pano.Context(4, 480, 270, num_bands=2, device=-1)
        ctx.load_camera_file(os.path.join(REF, rel))
        for i in range(4):
            K, R, sc = ctx.get_camera(i)
            assert np.array_equal(np.float32(K), f32(d["K"])) and np.array_equal(np.float32(R), f32(d["R"][i]))
            assert np.float32(sc) == np.float32(d["scale"])
    # and the geometry that follows from the first one: SURVEY appendix C
    ctx.load_camera_file(os.path.join(REF, "2222/cameraparaout_1.txt"))
    ctx.prepare()
    assert ctx.pano_rect() == (-1121, 475, 1333, 257)


def test_reference_skip_rule_misreads_the_mixed_log():
    """initCamParams (ocvstitcher.hpp:452-520) counts the records (lines with ':'), skips 6*(records-1) lines and reads a
    timestamp + num_images lines of 18 values + scale.  cfg/390camcfg/cameraparaout_1.txt starts with 6-line records
    (4 cameras) and ends with 4-line records (2 cameras): the skip does not land on the last record - which is why
    pano_load_camera_file searches for the last timestamp instead."""
    path = os.path.join(REF, "cfg/390camcfg/cameraparaout_1.txt")
    lines = open(path).read().split("\n")
    records = sum(1 for l in lines if ":" in l)
    at = 6 * (records - 1)                           # where the reference would expect the last timestamp
    last = max(i for i, l in enumerate(lines) if ":" in l)
    assert at != last
    lens = {}
    stamps = [i for i, l in enumerate(lines) if ":" in l] + [len([l for l in lines if l.strip()])]
    for a, b in zip(stamps, stamps[1:]):
        lens[b - a] = lens.get(b - a, 0) + 1
    assert len(lens) >= 2 and 6 in lens and 4 in lens   # mixed record shapes: the premise of the skip rule does not hold
    # here the skip even runs past the end of the file (2412 > 2182 lines): `fin >> str` then fails, the split of the stale
    # string has not 18 values and initCamParams returns RET_ERR ("preset parameter incorrect, init all!")
    assert at >= len(lines)


@pytest.fixture(scope="module")
def replay_bin(tmp_path_factory, pano):
    pano.build()
    out = tmp_path_factory.mktemp("bin") / "replay"
    lib_dir = os.path.join(ROOT, "img-stitching_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(ROOT, "examples", "replay.cpp"), "-o", str(out),
                           "-L" + lib_dir, "-lpano_hip", "-Wl,-rpath," + lib_dir, "-lpthread"])
    return str(out)


def reference_cfg(tmp_path, **override):
    """cfg/stitcher-imx390cfg.yaml as committed, with only the Jetson-absolute `cameraparams:` path (and what the test
    overrides) rewritten - every other line is the reference's"""
    txt = open(os.path.join(REF, "cfg/stitcher-imx390cfg.yaml")).read()
    txt = re.sub(r'(?m)^cameraparams:.*$', 'cameraparams: "%s"' % os.path.join(REF, "cfg/cameras.yaml"), txt)
    for k, v in override.items():
        txt, cnt = re.subn(r'(?m)^%s:[^#\n]*' % k, '%s: %s ' % (k, v), txt)
        assert cnt == 1, k
    p = tmp_path / "stitcher-imx390cfg.yaml"
    p.write_text(txt)
    return p


def test_stitcher_init_on_the_reference_yaml_960(replay_bin, tmp_path):
    """the self-consistent rig-R entry (cameras.yaml:230-246, inputsz 960): only outPutWidth/Height differ from the yaml as
    committed.  SURVEY appendix C integers."""
    cfg = reference_cfg(tmp_path, outPutWidth=960, outPutHeight=540)
    r = subprocess.run([replay_bin, str(cfg), "--plan"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert "stitcher 0: pano 1452x523 at (-721,497), output 1430x250, bands 3" in r.stdout
    assert "stitcher 1: pano 1484x509 at (-733,523), output 1470x250, bands 3" in r.stdout


def test_stitcher_init_on_the_reference_yaml_as_committed(replay_bin, tmp_path):
    """the yaml as committed selects `inputsz 720` (outPutWidth 720), whose `cams` were pasted from the 960 entry and whose
    stitcher-1 cut (width 1421) exceeds the panorama those cameras give at 720x405: init() finds the entry,
    calibration() refuses the cut - the reference would run into cv::Mat::operator()(Rect)'s assertion in process()
    (ocvstitcher.hpp:1210)"""
    cfg = reference_cfg(tmp_path)
    r = subprocess.run([replay_bin, str(cfg), "--plan"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1
    assert "stitcher 1 calibration failed" in r.stderr and "cut" in r.stderr
    # an entry with `cams: [0]` placeholders (4cam-silver / 720) is found by init() and refused for its parameter count
    cfg = reference_cfg(tmp_path, sttype="4cam-silver")
    r = subprocess.run([replay_bin, str(cfg), "--plan"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "calibration failed" in r.stderr


def test_stitcher_mode_3_on_the_reference_yaml_keeps_the_yaml_cut(replay_bin, tmp_path):
    """cfg/stitcher-imx390cfg.yaml with `initMode: 3`: the reference loads the structure's cut in every mode
    (ocvstitcher.hpp:333-337) and mode 3's initCamParams -> initSeam path (:627-628) never rewrites it (:959-964 is initAll's).
    The reference's own cfg/390camcfg/ logs show what that means: cameraparaout_0.txt / _1.txt end in records of ANOTHER
    calibration (panoramas 1026 and 1038 wide at 960 x 540), which the structure's 1430-wide cut does not fit - the reference would
    die in cv::Mat::operator()(Rect) (:1210), the mirror refuses at calibration(); cameraparaout_2.txt ends in a record the cut
    does fit (1510 x 527)."""
    import shutil
    cfg = reference_cfg(tmp_path, outPutWidth=960, outPutHeight=540, initMode=3,
                        camcfgpath='"%s/"' % os.path.join(REF, "cfg/390camcfg"))
    r = subprocess.run([replay_bin, str(cfg), "--plan"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "stitcher 0 calibration failed" in r.stderr and "cut" in r.stderr, r.stderr
    # the record the cut fits, offered to both stitchers: ITS cameras (not the yaml defaults' 1452x523 / 1484x509), the yaml cuts
    d = tmp_path / "camcfg"
    d.mkdir()
    for i in (0, 1):
        shutil.copy(os.path.join(REF, "cfg/390camcfg/cameraparaout_2.txt"), d / ("cameraparaout_%d.txt" % i))
    cfg = reference_cfg(tmp_path, outPutWidth=960, outPutHeight=540, initMode=3, camcfgpath='"%s/"' % d)
    r = subprocess.run([replay_bin, str(cfg), "--plan"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert "stitcher 0: pano 1510x527 at (-762,829), output 1430x250, bands 3" in r.stdout, r.stdout
    assert "stitcher 1: pano 1510x527 at (-762,829), output 1470x250, bands 3" in r.stdout, r.stdout
    # no record on file: the defaults (where the reference ends after its fallbacks, :639-643), the same cuts
    cfg = reference_cfg(tmp_path, outPutWidth=960, outPutHeight=540, initMode=3, camcfgpath='"%s/"' % tmp_path)
    r = subprocess.run([replay_bin, str(cfg), "--plan"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    assert "stitcher 0: pano 1452x523 at (-721,497), output 1430x250, bands 3" in r.stdout
    assert "stitcher 1: pano 1484x509 at (-733,523), output 1470x250, bands 3" in r.stdout
