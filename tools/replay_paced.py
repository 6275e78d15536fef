import json, os, subprocess, sys, pathlib, tempfile
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_cpp_mirror import write_cfgs
rig = json.load(open(os.path.join(ROOT, "tests", "golden", "r_cams.json")))
tmp = pathlib.Path(tempfile.mkdtemp())
cfg = write_cfgs(tmp, rig)
lib = os.path.join(ROOT, "img-stitching_amd")
exe = str(tmp / "replay")
subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "examples", "replay.cpp"), "-o", exe, "-L" + lib, "-lpano_hip", "-Wl,-rpath," + lib, "-lpthread"])
for extra in ([], ["--async-refresh"]):
    r = subprocess.run([exe, str(cfg), "--frames", "1200", "--fps", "120", "--refresh-every", "200"] + extra, capture_output=True, text=True, cwd=tmp)
    print([l for l in r.stdout.splitlines() if l.startswith("paced")], r.stderr[-300:])
