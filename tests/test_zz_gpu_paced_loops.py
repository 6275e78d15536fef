"""Wall-clock loops (BASELINE config 5: frames offered at 60 fps), collected LAST (file name) so that scheduler noise on a shared
box can never mark a parity test untested under `pytest -x`.

What is HARD here is what does not depend on the clock: every offered tick is either composed or counted as dropped, the sampled
panoramas equal the oracle's for the masks in force at their tick, the refreshed masks were installed.  What depends on the clock
(dropped ticks, achieved fps, latencies) is REPORTED - printed and written to gpurun_out/paced_loops_report.json - and asserted
only under PANO_STRICT_TIMING=1 (the builder's own runs; figures in DESIGN.md section 6)."""
import importlib.util
import json
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
STRICT = os.environ.get("PANO_STRICT_TIMING", "0") == "1"


def _report(name, r):
    print(name, json.dumps(r))
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        p = os.path.join(d, "paced_loops_report.json")
        all_ = json.load(open(p)) if os.path.exists(p) else {}
        all_[name] = r
        json.dump(all_, open(p, "w"), indent=1)
    except OSError:
        pass


def _harness():
    spec = importlib.util.spec_from_file_location("stream_60fps", os.path.join(ROOT, "tools", "stream_60fps.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_paced_60fps_stream(pano, po):
    """BASELINE config 5 in small: 2 x 4 x 960x540 frames offered at 60 fps for one second through pano_stream_* (page-locked
    slots, two panoramas in flight; tools/stream_60fps.py is the harness): the sampled panoramas are the oracle's"""
    r = _harness().run(fps=60.0, frames=60, width=960, height=540, bands=4, check=True)
    _report("stream_960x540_60_ticks", r)
    assert r["frames_composed"] + r["dropped"] == 60, r
    assert r["sampled_frames_equal_oracle"] is True, r
    if STRICT:
        assert r["dropped"] == 0 and r["achieved_fps"] > 55.0, r


def test_paced_60fps_stream_at_full_size_with_mask_refresh(pano, po):
    """BASELINE config 5 at its stated size: 8 x 1920x1080 frames offered at 60 fps for ten seconds (600 ticks) through
    pano_stream_*, with the reference's mask refresh every 200 frames (include/ocvstitcher.hpp:1152-1159) running BESIDE the loop
    (pano_refresh_masks_begin / _poll: graph cuts on a thread of the library).  The capture loop it stands for is
    src/master.cpp:302-411.  Panoramas sampled before, between and after the refreshes are the oracle's for the masks in force at
    their tick; both stitchers' refreshed masks were installed"""
    r = _harness().run(fps=60.0, frames=600, width=1920, height=1080, bands=5, check=True, refresh_every=200, refresh_async=True)
    _report("stream_8x1080p_600_ticks_refresh_beside_the_loop", r)
    assert r["frames_composed"] + r["dropped"] == 600, r
    assert r["mask_refresh"]["masks_installed"] >= 2, r            # both stitchers' refreshes came through (a refresh due on a dropped tick begins with the next composed one)
    # a sample tick that is dropped hands its sample to the next composed tick: at least the first two samples exist whatever the clock does
    assert r["sampled_frames_equal_oracle"] is True and len(r["sampled_frames"]) >= 2, r
    if STRICT:
        assert r["dropped"] <= 2 and r["achieved_fps"] > 58.0 and r["sampled_frames"] == [0, 300, 599], r


def test_replay_paced_loop_with_mask_refresh_beside_it(pano, rig_r, tmp_path_factory, tmp_path):
    """examples/replay.cpp: the capture loop of src/master.cpp paced at 60 fps with a graph-cut mask refresh every 30 frames beside
    the loop (pano::Stitcher::asyncMaskRefresh -> pano_refresh_masks_*): every tick is accounted for"""
    from test_cpp_mirror import write_cfgs
    pano.build()
    out = tmp_path_factory.mktemp("bin") / "replay"
    lib_dir = os.path.join(ROOT, "img-stitching_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(ROOT, "examples", "replay.cpp"), "-o", str(out),
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(lib_dir, "csrc"), "-L" + lib_dir, "-lpano_hip",
                           "-Wl,-rpath," + lib_dir, "-lpthread"])
    cfg = write_cfgs(tmp_path, rig_r)
    r = subprocess.run([str(out), str(cfg), "--frames", "120", "--fps", "60", "--refresh-every", "30", "--async-refresh"],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr + r.stdout
    m = re.search(r"120 frames offered, (\d+) composed, dropped (\d+), achieved ([0-9.]+) fps", r.stdout)
    assert m and int(m.group(1)) + int(m.group(2)) == 120 and "beside the loop" in r.stdout, r.stdout
    _report("replay_cpp_120_ticks_refresh_every_30", {"composed": int(m.group(1)), "dropped": int(m.group(2)), "achieved_fps": float(m.group(3))})
    if STRICT:
        assert int(m.group(2)) <= 2, r.stdout


def test_config5_as_baseline_states_it(pano, po):
    """BASELINE.json configs[4] in its stated combination, in ONE run (VERDICT r03 #7a): 8 x RAW 1920x1080 frames offered at 60 fps
    through pano_stream_* (page-locked double-buffered slots: H2D || compose || D2H), the undistort -> crop -> resize front end of
    include/nvcam.hpp:898-921 fused into the warp (pano_set_undistort), the frame's launch sequence replayed as a hipGraph
    (PANO_GRAPH=1), 120 ticks.  Hard: the sampled panoramas equal the oracle's of the fused map, every tick is accounted for, and the
    graph path really ran (graphs held, one replay per composed frame).  A process of its own: PANO_GRAPH is read at pano_prepare."""
    import sys
    env = dict(os.environ, PANO_GRAPH="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stream_60fps.py"), "--frames", "120", "--raw", "--check"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    _report("config5_raw_undistort_hipgraph_120_ticks", r)
    assert "fused undistort front end" in r["config"] and "hipGraph replay" in r["config"], r["config"]
    assert r["frames_composed"] + r["dropped"] == 120, r
    assert r["sampled_frames_equal_oracle"] is True and len(r["sampled_frames"]) >= 2, r
    for g in r["hipgraph"]:     # two page-locked slots -> two buffer sets -> two graphs per stitcher; every frame after the captures is a replay
        assert g["graphs_held"] == 2 and g["replays"] >= r["frames_composed"], r["hipgraph"]
    if STRICT:
        assert r["dropped"] == 0 and r["achieved_fps"] > 58.0 and r["sampled_frames"] == [0, 60, 119], r
