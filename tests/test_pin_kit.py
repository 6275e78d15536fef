"""tools/opencv_pin/pin.cpp cannot be compiled here (no OpenCV), but the parts of it that do not touch OpenCV can: its SHA-256, its
reader of the *_cams.json fixtures and its .npy writer are cut out of the source, compiled with g++ and checked against hashlib /
json / numpy - so that the day someone runs the kit its hashes, its camera parameters and its raw stage files are what this
repository's loader expects.  The whole file is also run through `g++ -fsyntax-only` against declarations-only mock headers
(tests/mock_opencv: they pin nothing)."""
import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_pin_kit_helpers_against_hashlib_and_json(tmp_path):
    s = open(os.path.join(ROOT, "tools", "opencv_pin", "pin.cpp")).read()
    sha = s[s.index("// ---- SHA-256"):s.index("// the bytes numpy")]
    js = s[s.index("// ---- the little of JSON"):s.index("Mat_<float> mat3")]
    prog = """#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
""" + sha + "}\nnamespace {\n" + js + """}
int main(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {   // hash every file named
        Sha256 h; const std::string t = slurp(argv[i]); h.update((const uint8_t*)t.data(), t.size()); std::printf("%s\\n", h.hex().c_str());
    }
    const std::string j = slurp(argv[1]);
    for (const char* k : {"K", "R", "scale"}) { for (double v : numbers_after(j, k)) std::printf("%.17g ", v); std::printf("\\n"); }
    const std::string r = slurp(argv[2]);
    size_t pos = r.find("\\"stitchers\\"");
    for (int s = 0; s < 2; s++) {
        size_t e1 = 0, e2 = 0;
        for (double v : numbers_after(r, "cams", pos, &e1)) std::printf("%.17g ", v);
        std::printf("| ");
        for (double v : numbers_after(r, "cut", pos, &e2)) std::printf("%.17g ", v);
        std::printf("\\n");
        pos = std::max(e1, e2);
    }
    std::printf("%.17g %.17g\\n", numbers_after(r, "width")[0], numbers_after(r, "height")[0]);
    return 0;
}
"""
    src, exe = tmp_path / "pin_helpers.cpp", tmp_path / "pin_helpers"
    src.write_text(prog)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", str(src), "-o", str(exe)])
    files = [os.path.join(GOLDEN, "c1_cams.json"), os.path.join(GOLDEN, "r_cams.json"), os.path.join(GOLDEN, "c1_cam0.png")]
    out = subprocess.check_output([str(exe)] + files).decode().splitlines()
    for f, got in zip(files, out[:3]):
        assert got == hashlib.sha256(open(f, "rb").read()).hexdigest(), f
    c1, r = json.load(open(files[0])), json.load(open(files[1]))
    nums = lambda line: [float(x) for x in line.split()]
    assert nums(out[3]) == [float(v) for v in c1["K"]]
    assert nums(out[4]) == [float(v) for row in c1["R"] for v in row]
    assert nums(out[5]) == [float(c1["scale"])]
    for s in range(2):
        cams, cut = out[6 + s].split("|")
        assert nums(cams) == [float(v) for v in r["stitchers"][s]["cams"]] and nums(cut) == [float(v) for v in r["stitchers"][s]["cut"]]
    assert nums(out[8]) == [float(r["width"]), float(r["height"])]


def test_pin_kit_syntax_against_mock_headers():
    """g++ -fsyntax-only over the whole kit against tests/mock_opencv (declarations of the OpenCV 3.4 API slice the kit uses, no
    definitions; PINS NOTHING - tests/mock_opencv/README.md): ill-formed C++, typos and missing includes are caught in the CPU gate
    instead of on the one day a holder of OpenCV runs the kit"""
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "tests", "mock_opencv"),
                        os.path.join(ROOT, "tools", "opencv_pin", "pin.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_pin_kit_npy_writer_is_readable_by_numpy(tmp_path):
    """the kit's raw stage writer (StageDir: NumPy format 1.0 + manifest.json) cut out of pin.cpp and compiled against a ten-line
    stand-in for cv::Mat: what it writes must load with numpy.load and pass tests/pin_stages.read_group's shape / dtype checks"""
    import sys
    import numpy as np
    s = open(os.path.join(ROOT, "tools", "opencv_pin", "pin.cpp")).read()
    sd = s[s.index("struct StageDir {"):s.index("struct Run {")]
    prog = """#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#define CV_VERSION "none (stand-in)"
enum { CV_8U = 0, CV_16S = 3, CV_32S = 4, CV_32F = 5, ACCESS_READ = 1 };
struct Mat {   // the few members StageDir::put touches
    int rows, cols, ch, dep; std::vector<unsigned char> store; unsigned char* data;
    Mat(int r, int c, int channels, int depth_) : rows(r), cols(c), ch(channels), dep(depth_), store((size_t)r * c * channels * esz1()), data(store.data()) {}
    size_t esz1() const { return dep == CV_8U ? 1 : dep == CV_16S ? 2 : 4; }
    bool isContinuous() const { return true; }
    Mat clone() const { return *this; }
    int depth() const { return dep; } int channels() const { return ch; }
    size_t elemSize() const { return esz1() * ch; } size_t total() const { return (size_t)rows * cols; }
};
struct UMat { Mat m; Mat getMat(int) const { return m; } };
""" + sd + """
int main(int argc, char** argv) {
    StageDir d(argv[1], "g0");
    Mat a(5, 7, 3, CV_8U); for (size_t i = 0; i < a.store.size(); i++) a.data[i] = (unsigned char)(i * 3);
    Mat b(4, 6, 1, CV_32F); for (int i = 0; i < 24; i++) ((float*)b.data)[i] = 0.25f * i - 1.f;
    Mat c(3, 2, 3, CV_16S); for (int i = 0; i < 18; i++) ((short*)c.data)[i] = (short)(i * 1000 - 9000);
    Mat e(2, 4, 1, CV_32S); for (int i = 0; i < 8; i++) ((int*)e.data)[i] = i - 100000;
    d.put("cam0/warp", a); d.put("unit/pyrdown32f_in", b); d.put("blend_b4/laplace_l0", c); d.put("roi", e);
    d.finish();
    return 0;
}
"""
    src, exe = tmp_path / "npy.cpp", tmp_path / "npy"
    src.write_text(prog)
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", str(src), "-o", str(exe)])
    subprocess.check_call([str(exe), str(tmp_path)])
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pin_stages
    arrays, meta = pin_stages.read_group(str(tmp_path), "g0")
    assert meta["generator"] == "tools/opencv_pin/pin.cpp"
    assert arrays["cam0/warp"].dtype == np.uint8 and arrays["cam0/warp"].shape == (5, 7, 3)
    assert np.array_equal(arrays["cam0/warp"].reshape(-1), (np.arange(105) * 3).astype(np.uint8))
    assert arrays["unit/pyrdown32f_in"].dtype == np.float32 and np.array_equal(arrays["unit/pyrdown32f_in"].reshape(-1), 0.25 * np.arange(24, dtype=np.float32) - 1)
    assert arrays["blend_b4/laplace_l0"].dtype == np.int16 and arrays["blend_b4/laplace_l0"].shape == (3, 2, 3)
    assert np.array_equal(arrays["blend_b4/laplace_l0"].reshape(-1), (np.arange(18) * 1000 - 9000).astype(np.int16))
    assert arrays["roi"].dtype == np.int32 and np.array_equal(arrays["roi"].reshape(-1), np.arange(8, dtype=np.int32) - 100000)
