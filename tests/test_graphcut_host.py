"""the product's host-side max-flow (csrc/pano_graphcut.hpp, what pano_build_masks_graphcut runs between its two GPU
kernels) as a plain C++ unit: built with g++ under ASan/UBSan, fed random grids - with and without ties - and compared
label for label with the oracle's restatement of GCGraph<float>::maxFlow"""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_max_flow_matches_oracle_label_for_label(po, tmp_path):
    exe = tmp_path / "gc_harness"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           os.path.join(ROOT, "tests", "src", "graphcut_host_harness.cpp"), "-o", str(exe)])
    rng = np.random.default_rng(11)
    grids = []
    for trial in range(60):
        W, H = int(rng.integers(2, 48)), int(rng.integers(2, 48))
        kind = trial % 3
        if kind == 0:      # random terminals and capacities
            term = np.where(rng.random((H, W)) < 0.3, rng.integers(-50, 50, (H, W)), 0)
            wh, wv = rng.integers(1, 30, (H, W)), rng.integers(1, 30, (H, W))
        elif kind == 1:    # the seam finder's shape: source strip left, sink strip right, large capacities
            term = np.zeros((H, W)); term[:, :2] = 10000; term[:, -2:] = -10000
            wh, wv = rng.integers(1, 390000, (H, W)), rng.integers(1, 390000, (H, W))
        else:              # flat costs: every column is a minimum cut, the free vertices keep their last tree
            term = np.zeros((H, W)); term[:, :1] = 10000; term[:, -1:] = -10000
            wh, wv = np.ones((H, W)), np.ones((H, W))
            wh[rng.random((H, W)) < 0.1] = 1001
        grids.append((term.astype(np.float32), wh.astype(np.float32), wv.astype(np.float32)))
    path = tmp_path / "grids.bin"
    with open(path, "wb") as f:
        for term, wh, wv in grids:
            np.asarray(term.shape[::-1], np.int32).tofile(f)
            term.tofile(f); wh.tofile(f); wv.tofile(f)
    out = subprocess.run([str(exe), str(path)], capture_output=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    got = np.frombuffer(out.stdout, np.uint8)
    o = 0
    for k, (term, wh, wv) in enumerate(grids):
        _, lab = po.gc_grid_max_flow(term, wh, wv)
        n = lab.size
        assert np.array_equal(got[o:o + n], lab.reshape(-1)), k
        o += n
    assert o == got.size


def test_copy_pool_under_thread_sanitizer(tmp_path):
    """the copy-thread pool behind pano_compose_host (csrc/pano_hostcopy.hpp): three caller threads at once, built with
    -fsanitize=thread - no data race, no lost wake-up, every byte where it belongs"""
    exe = tmp_path / "copypool_harness"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-fsanitize=thread", "-fno-omit-frame-pointer",
                           os.path.join(ROOT, "tests", "src", "copypool_harness.cpp"), "-o", str(exe), "-lpthread"])
    for nthreads in ("8", "1", "3"):
        out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, PANO_HOST_THREADS=nthreads, TSAN_OPTIONS="halt_on_error=1"))
        assert out.returncode == 0 and "bad 0" in out.stdout, out.stdout[-500:] + out.stderr[-3000:]
