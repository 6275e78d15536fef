#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of the frames-in-flight loop: the steady-state window (the last 300 K1 launches of
the timed pass), GPU busy fraction (union of kernel intervals), mean number of kernels resident at once, per-kernel mean
duration in that window and the idle gaps."""
import csv, collections, json, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
k1 = [i for i, r in enumerate(rows) if "warp_tiles_lut" in r[2]]
# bench.py: warm-up, timed pass (300), event pass (300); take the middle 200 frames of the timed pass
n = len(k1)
first_timed = n - 600 if n >= 620 else 0
lo, hi = rows[k1[first_timed + 50]][0], rows[k1[first_timed + 250]][0]
win = [r for r in rows if r[0] >= lo and r[0] < hi]
ev = []
for s, e, _ in win:
    ev.append((s, 1)); ev.append((min(e, hi), -1))
ev.sort()
busy = 0; conc_area = 0; depth = 0; last = lo; gaps = []
for t, d in ev:
    if depth > 0:
        busy += t - last; conc_area += depth * (t - last)
    elif t > last:
        gaps.append(t - last)
    depth += d; last = t
per = collections.defaultdict(list)
for s, e, k in win: per[k].append(e - s)
out = {"window_us": round((hi - lo) / 1e3, 1), "frames": 200, "us_per_frame": round((hi - lo) / 200e3, 2),
       "gpu_busy_fraction": round(busy / (hi - lo), 4), "mean_kernels_resident_when_busy": round(conc_area / max(busy, 1), 2),
       "idle_gaps": len(gaps), "idle_us_per_frame": round(sum(gaps) / 200e3, 2),
       "kernels": {k: {"per_frame": round(len(v) / 200, 2), "mean_us": round(sum(v) / len(v) / 1e3, 2), "sum_us_per_frame": round(sum(v) / 200e3, 2)} for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))}}
out["sum_of_kernel_us_per_frame"] = round(sum(v["sum_us_per_frame"] for v in out["kernels"].values()), 1)
print(json.dumps(out, indent=1))
