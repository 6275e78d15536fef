# the driver's form of the bench (--steps 20 --warmup 5: 1.2 ms timed) five times in a row on one box: how much of a figure is the box's noise
for i in 1 2 3 4 5; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-paths --no-c4 > gpurun_out/r05_driver_form_$i.json 2>/dev/null; python - <<PY
import json
b=json.load(open("gpurun_out/r05_driver_form_$i.json")); r=b["roofline"]
print($i, b["value"], b["ms_per_step"], "from_idle", b["from_idle"]["value"], "cold", r["avg_launch_us"], r["frac"], "traffic", r["traffic"])
PY
done
