# rocprofv3 kernel stats of bench.py composing one frame at a time; prints the per-kernel table.
# Usage on the GPU box: bash tools/prof1.sh [tag]   (extra env vars pass through)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-p1}
rm -rf $R/gpurun_out/$T
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$T -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-host-paths --frames-in-flight 1 > $R/gpurun_out/$T.json 2> $R/gpurun_out/$T.log || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/$T/**/run_kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pano::" in r["Name"] and int(r["Calls"]) >= 300]
for r in rows:
    print("%-70s calls %5s avg %8.2f us" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
python3 -c "
import json
d = json.loads(open('$R/gpurun_out/$T.json').read().strip().splitlines()[-1])
print('value', d['value'], 'stage_us', d['stage_us_per_launch'])
"
