# HBM bytes of EVERY kernel of the compose path per 8-camera frame: FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md, HBM section; gfx950: FETCH_SIZE x2) over tools/frames_one_at_a_time.py.  Usage: bash tools/pmc_traffic_all.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/taF $R/gpurun_out/taW
timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/taF -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py > $R/gpurun_out/taF.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/taW -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py > $R/gpurun_out/taW.log 2>&1 || exit 1
python3 - <<'PY'
import csv, collections, json, os
R = os.environ["GRAFT_REPO_ROOT"]
out = collections.defaultdict(dict)
frames = None
for tag, name in (("taF", "FETCH_SIZE"), ("taW", "WRITE_SIZE")):
    agg, calls = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f"{R}/gpurun_out/{tag}/run_counter_collection.csv")):
        if r["Counter_Name"] != name: continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k] += float(r["Counter_Value"]); calls[k] += 1
    frames = calls[[k for k in calls if "warp_tiles_lut" in k][0]]
    for k in agg:
        if calls[k] >= frames:
            out[k][name + "_KB_per_frame_raw"] = round(agg[k] / frames, 1)
            out[k]["launches_per_frame"] = round(calls[k] / frames, 2)
tot = 0.0
for k, v in out.items():
    v["hbm_MB_per_frame"] = round((2 * v.get("FETCH_SIZE_KB_per_frame_raw", 0) + v.get("WRITE_SIZE_KB_per_frame_raw", 0)) * 1024 / 1e6, 2)
    tot += v["hbm_MB_per_frame"]
out["_total_hbm_MB_per_frame"] = round(tot, 1)
# the figures belong to the kernel sources they were counted on: bench.py prints them only beside a library of the same id
import importlib, sys
sys.path.insert(0, R)
kid = importlib.import_module("img-stitching_amd").kernel_source_id()
out["kernel_source_id"] = kid
out["workload"] = "tools/frames_one_at_a_time.py: config 2 (8 x 1920x1080, 5 bands), both stitchers per launch sequence, one frame at a time"
json.dump(out, open(R + "/gpurun_out/traffic_all.json", "w"), indent=1)
k1 = [k for k in out if "warp_tiles_lut" in k][0]
json.dump({"kernel_source_id": kid, "kernel": k1, "FETCH_SIZE_KB_raw": out[k1].get("FETCH_SIZE_KB_per_frame_raw"), "WRITE_SIZE_KB": out[k1].get("WRITE_SIZE_KB_per_frame_raw"),
           "hbm_bytes_per_launch_8cam": int(out[k1]["hbm_MB_per_frame"] * 1e6),
           "source": "tools/pmc_traffic_all.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/frames_one_at_a_time.py; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM)"},
          open(R + "/gpurun_out/warp_traffic_current.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
