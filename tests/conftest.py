import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pano():
    """the product package (directory name has a hyphen, so import by string)"""
    return importlib.import_module("img-stitching_amd")


@pytest.fixture(scope="session")
def po():
    """the CPU oracle (test infrastructure)"""
    import pano_oracle
    pano_oracle.build()
    return pano_oracle


def load_png_bgr(path):
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


@pytest.fixture(scope="session")
def c1():
    """config 1: 4 x 480x270 frames of 2222/1..4.png + last record of 2222/cameraparaout_1.txt"""
    d = json.load(open(os.path.join(GOLDEN, "c1_cams.json")))
    frames = [load_png_bgr(os.path.join(GOLDEN, f"c1_cam{i}.png")) for i in range(4)]
    return {"frames": frames, "K": [d["K"]] * 4, "R": d["R"], "scale": d["scale"], "w": 480, "h": 270, "n": 4}


@pytest.fixture(scope="session")
def c1b():
    """the other half of the bundled set: 4 x 480x270 frames of 2222/5..8.png + last record of 2222/cameraparaout_2.txt"""
    d = json.load(open(os.path.join(GOLDEN, "c1b_cams.json")))
    frames = [load_png_bgr(os.path.join(GOLDEN, f"c1b_cam{i}.png")) for i in range(4)]
    return {"frames": frames, "K": [d["K"]] * 4, "R": d["R"], "scale": d["scale"], "w": 480, "h": 270, "n": 4}


def _rig_real(rig, prefix):
    out = []
    for s, st in enumerate(rig["stitchers"]):
        v = st["cams"]
        out.append({"n": 2, "w": rig["width"], "h": rig["height"], "scale": v[-1], "K": [v[0:9], v[18:27]], "R": [v[9:18], v[27:36]],
                    "cut": st["cut"], "cams": v, "prefix": prefix,
                    "frames": [load_png_bgr(os.path.join(GOLDEN, f"{prefix}_cam{2 * s + i}.png")) for i in range(2)]})
    return out


@pytest.fixture(scope="session")
def rig_r_real(rig_r):
    """rig R with its REAL frames 2222/4cam/0..3.png (960x540): replay.cpp:211-215 gives 0,1 to the "up" stitcher and 2,3
    to the "down" one.  Returns the two stitchers as dicts like c1 (+ "cut")"""
    return _rig_real(rig_r, "r")


@pytest.fixture(scope="session")
def rig_s_real():
    """rig S: cfg/cameras.yaml 4cam-silver/640 (:212-228) with ITS frames 2222/4cam/1/0..3.png (640x360), like rig_r_real"""
    return _rig_real(json.load(open(os.path.join(GOLDEN, "s_cams.json"))), "s")


@pytest.fixture(scope="session")
def rig_r():
    """reference rig R: cfg/cameras.yaml 4cam-black/960, stitcher 0 and 1 (2 cams each) + cut"""
    return json.load(open(os.path.join(GOLDEN, "r_cams.json")))


from helpers import build_fake_rccl  # noqa: E402  (pytest-free: __graft_entry__.build() uses it too)


@pytest.fixture(scope="session")
def fake_rccl_lib():
    """the RCCL test double: libpano_hip.so opens it instead of librccl.so when PANO_RCCL_LIB names it"""
    return build_fake_rccl()


@pytest.fixture(scope="session")
def st258():
    """the remaining bundled frames, 2222/258st/1..8.png (2x2 box to 320x180), as two groups of four under the config-1 / config-1b
    parameters scaled by 2/3 (tests/golden/make_inputs.py: a pairing of this repo's own - the tree holds no parameters for them)"""
    g = json.load(open(os.path.join(GOLDEN, "st258_golden.json")))
    out = []
    for k, (grp, prefix) in enumerate(zip(g["groups"], ("c1", "c1b"))):
        d = json.load(open(os.path.join(GOLDEN, f"{prefix}_cams.json")))
        out.append({"n": 4, "w": 320, "h": 180, "K": [grp["K"]] * 4, "R": d["R"], "scale": grp["scale"], "golden": grp,
                    "frames": [load_png_bgr(os.path.join(GOLDEN, f"st258_cam{4 * k + i}.png")) for i in range(4)]})
    return out
