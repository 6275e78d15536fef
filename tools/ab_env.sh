# A/B of environment knobs on the GPU box: VARIANTS="A=1 B=2|A=0" (variants separated by |), each run one frame at a time and in flight
IFS='|' read -ra V <<< "${VARIANTS:-PANO_NOP=0}"
for rep in 1 2; do
for v in "${V[@]}"; do
  env $v python bench.py --frames-in-flight 1 --no-cpu-baseline --no-host-paths --steps 300 > gpurun_out/abe_one.json 2>gpurun_out/abe_err.log || exit 1
  env $v python bench.py --no-cpu-baseline --no-host-paths --no-isolated-pass --steps 300 > gpurun_out/abe_fl.json 2>>gpurun_out/abe_err.log || exit 1
  python - <<PY
import json
a=json.loads(open("gpurun_out/abe_one.json").read().strip().splitlines()[-1])
b=json.loads(open("gpurun_out/abe_fl.json").read().strip().splitlines()[-1])
s=a.get("stage_us_per_launch",{})
print("%-40s one-at-a-time %8.1f (%.1f us) L0 %.1f K1 %.1f | in flight %8.1f (%.1f us)" % ("$v", a["value"], 1e3*a["ms_per_step"], s.get("blend_level0",0), s.get("warp",0), b["value"], 1e3*b["ms_per_step"]))
PY
done
done
