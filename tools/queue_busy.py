#!/usr/bin/env python3
"""per hardware queue: busy fraction, kernels per frame and mean kernel duration, from a rocprofv3 kernel trace (steady-state window)"""
import csv, collections, json, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void pano::", "").replace("pano::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r["Queue_Id"]))
rows.sort()
k1 = [i for i, r in enumerate(rows) if "warp_tiles_lut" in r[2]]
lo, hi = rows[k1[len(k1) // 2]][0], rows[k1[len(k1) // 2 + 200]][0]
win = [r for r in rows if lo <= r[0] < hi]
out = {"us_per_frame": round((hi - lo) / 200 / 1e3, 2), "queues": {}}
for q in sorted(set(r[3] for r in win)):
    rs = [r for r in win if r[3] == q]
    busy = sum(min(e, hi) - s for s, e, _, _ in rs)
    per = collections.defaultdict(list)
    for s, e, n, _ in rs: per[n].append(e - s)
    out["queues"][q] = {"busy": round(busy / (hi - lo), 3), "kernels": {n: [len(v), round(sum(v) / len(v) / 1e3, 1)] for n, v in per.items()}}
print(json.dumps(out, indent=1))
