# A/B of the pyrDown tail on the GPU box: tests first, then bench.py one frame at a time and with frames in flight per variant
set -e
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests -m gpu -x -q > gpurun_out/r2b_tests.log 2>&1 || { tail -40 gpurun_out/r2b_tests.log; exit 1; }
tail -3 gpurun_out/r2b_tests.log
fi
for rep in 1 2; do
CFGS=${CFGS:-0:64:0:32 0:64:2:32 1:64:2:32 2:64:2:32 2:32:2:32 1:32:2:32}
for cfg in $CFGS; do
  IFS=: read hd hts b ts <<< "$cfg"
  export PANO_PYR_HEAD=$hd PANO_PYR_HEAD_TS=$hts
  PANO_PYR_TAIL=$b PANO_PYR_TAIL_TS=$ts python bench.py --frames-in-flight 1 --no-cpu-baseline --no-host-paths --steps 300 > gpurun_out/r2b_one.json 2>gpurun_out/r2b_err.log
  PANO_PYR_TAIL=$b PANO_PYR_TAIL_TS=$ts python bench.py --no-cpu-baseline --no-host-paths --no-isolated-pass --steps 300 > gpurun_out/r2b_fl.json 2>>gpurun_out/r2b_err.log
  python - <<PY
import json
a=json.loads(open("gpurun_out/r2b_one.json").read().strip().splitlines()[-1])
b=json.loads(open("gpurun_out/r2b_fl.json").read().strip().splitlines()[-1])
print("head=$hd/$hts tail base=$b ts=$ts  one-at-a-time %8.1f (%.1f us)   in flight %8.1f (%.1f us)" % (a["value"], 1e3*a["ms_per_step"], b["value"], 1e3*b["ms_per_step"]))
PY
done
done
