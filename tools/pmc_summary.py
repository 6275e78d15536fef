#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counter CSVs under gpurun_out/sq*/ per kernel."""
import csv, collections, glob, json, sys
out = {}
for f in sorted(glob.glob("gpurun_out/sq*/run_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if any(t in k for t in sys.argv[1:] or ["warp_tiles", "blend_level_vec", "pyr_down_kernel"]):
            out.setdefault(k, {}).update({c: round(sum(v) / len(v)) for c, v in cs.items()})
print(json.dumps(out, indent=1))
