"""tools/opencv_pin/pin.cpp cannot be compiled here (no OpenCV), but the parts of it that do not touch OpenCV can: its SHA-256 and its
reader of the *_cams.json fixtures are cut out of the source, compiled with g++ and checked against hashlib / json - so that the day
someone runs the kit its hashes and its camera parameters are the ones this repository's golden vectors were made with."""
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
