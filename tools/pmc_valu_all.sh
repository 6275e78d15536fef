# one SQ pass over whole frames: vector instructions, waves and busy cycles of EVERY kernel of the compose path
# (tools/frames_one_at_a_time.py = 110 8-camera frames, one at a time).  Usage on the GPU box: bash tools/pmc_valu_all.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/sqall
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES -d $R/gpurun_out/sqall -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py > $R/gpurun_out/sqall.log 2>&1 || exit 1
python3 - <<'PY'
import csv, collections, json, os
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(R + "/gpurun_out/sqall/run_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": calls[k] += 1
frames = calls[[k for k in calls if "warp_tiles_lut" in k][0]]
out = {k: dict({c: round(v / frames) for c, v in cs.items()}, launches_per_frame=round(calls[k] / frames, 2)) for k, cs in agg.items() if calls[k] >= frames}
tot = sum(v["SQ_INSTS_VALU"] for v in out.values())
for k in out: out[k]["valu_share"] = round(out[k]["SQ_INSTS_VALU"] / tot, 3)
out["_per_frame_total_valu"] = tot
json.dump(out, open(R + "/gpurun_out/sqall_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
