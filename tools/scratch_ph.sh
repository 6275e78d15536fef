fmt='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], d["value"], d["ms_per_step"], "K1", r["avg_launch_us"], r["frac"], "cold", r["cold"]["frac"], "L0", r["blend_level0"]["avg_launch_us"], "in-flight K1", r["in_timed_region"]["avg_launch_us"])'
for i in 1 2; do
for v in "none:--preheat 0" "steps0.1:--preheat 0.1" "steps0.5:--preheat 0.5" "steps2:--preheat 2" "matmul0.5:--preheat 0.5 --preheat-kind matmul"; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-host-paths --no-c4 ${v#*:} 2>/dev/null | python -c "$fmt" ${v%%:*} || exit 1
done; done
