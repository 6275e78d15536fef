#!/usr/bin/env python3
"""Takes the counter traffic that tools/pmc_traffic_all.sh just collected (gpurun_out/traffic_all.json, gpurun_out/warp_traffic_current.json)
into profiles/ - profiles/warp_traffic.json["current"], profiles/hbm_bytes_per_kernel_per_frame.json and its r05 copy - so that bench.py,
which prints counter traffic only beside the kernel source id it was counted on, finds the figures of THIS library.  Run on the GPU box
between the collection and the bench (tools/final_r05.sh), and again at home on the merged gpurun_out/."""
import json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cur = json.load(open(os.path.join(R, "gpurun_out", "warp_traffic_current.json")))
allk = os.path.join(R, "gpurun_out", "traffic_all.json")
if json.load(open(allk))["kernel_source_id"] != cur["kernel_source_id"]:
    sys.exit("adopt_traffic: the two files of gpurun_out/ were counted on different kernel sources")
p = os.path.join(R, "profiles", "warp_traffic.json")
d = json.load(open(p))
d["current"] = cur
json.dump(d, open(p, "w"), indent=1)
for name in ("hbm_bytes_per_kernel_per_frame.json", "r05_hbm_bytes_per_kernel_per_frame.json"):
    shutil.copyfile(allk, os.path.join(R, "profiles", name))
print("adopted counter traffic of kernel sources", cur["kernel_source_id"])
