#!/usr/bin/env python3
"""What the GPU is doing while F frames are in flight, from a rocprofv3 kernel trace of tools/inflight_time.py:
  * per queue: the gap between the end of a kernel and the start of the next one of the same queue (a dependent launch's cost);
  * time-weighted: how many kernels are resident at once, and how much of the time NO bandwidth-bound kernel (K1, pyrDown 0->1, blend
    levels 1 and 0) is among them - time in which the latency-bound small launches have the GPU to themselves;
  * per kernel: mean duration in flight.
python3 tools/timeline2.py <kernel_trace.csv> [skip_first_frames]"""
import csv, collections, json, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    short = name.split("(")[0].replace("void pano::", "").replace("pano::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "0"), r.get("Grid_Size", r.get("Grid_Size_X", "0"))))
rows.sort()
k1 = [i for i, r in enumerate(rows) if "warp_tiles_lut" in r[2]]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 600
lo, hi = rows[k1[skip]][0], rows[k1[-50]][0]
nfr = len(k1) - 50 - skip
win = [r for r in rows if lo <= r[0] < hi]
def big(name, grid):
    return "warp_tiles" in name or "<true" in name or ("pyr_down_kernel" in name and int(grid) > 1500000) or ("ordered_kernel<false" in name and int(grid) > 700000)
# label levels by grid size where one kernel name serves two levels
labels = collections.Counter((r[2], r[4]) for r in win)
ev = []
for s, e, n, q, g in win:
    b = 1 if big(n, g) else 0
    ev.append((s, 1, b)); ev.append((min(e, hi), -1, -b))
ev.sort()
depth = nbig = 0; last = lo
hist = collections.Counter(); nobig = 0; idle = 0
for t, d, b in ev:
    dt = t - last
    hist[depth] += dt
    if depth > 0 and nbig == 0: nobig += dt
    if depth == 0: idle += dt
    depth += d; nbig += b; last = t
tot = hi - lo
perq = collections.defaultdict(list)
for r in win: perq[r[3]].append(r)
gaps = []
for q, rs in perq.items():
    rs.sort()
    for a, b in zip(rs, rs[1:]):
        gaps.append(b[0] - a[1])
gaps.sort()
per = collections.defaultdict(list)
for s, e, n, q, g in win: per[(n, g)].append(e - s)
out = {"frames": nfr, "us_per_frame": round(tot / nfr / 1e3, 2), "queues": len(perq),
       "idle_fraction": round(idle / tot, 4), "no_bandwidth_bound_kernel_resident_fraction": round(nobig / tot, 4),
       "resident_kernels_time_share": {str(k): round(v / tot, 4) for k, v in sorted(hist.items())},
       "same_queue_gap_us": {"p10": gaps[len(gaps) // 10] / 1e3, "p50": gaps[len(gaps) // 2] / 1e3, "p90": gaps[len(gaps) * 9 // 10] / 1e3, "mean": round(sum(gaps) / len(gaps) / 1e3, 2),
                             "sum_per_frame": round(sum(gaps) / nfr / 1e3, 2)},
       "kernel_us_in_flight": {"%s grid %s" % k: {"n": len(v), "mean": round(sum(v) / len(v) / 1e3, 2), "per_frame": round(sum(v) / nfr / 1e3, 2)} for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))}}
print(json.dumps(out, indent=1))
