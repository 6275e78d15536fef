#!/bin/bash
# every flight stream on its own share of the CUs (hipExtStreamCreateWithCUMask) against the shared device, 4 frames in flight
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/cu_mask.jsonl; : > $O
for rep in 1 2; do
  timeout -k 10 120 python3 tools/inflight_time.py 4 2000 0 >> $O
  for m in block stride halves; do
    PANO_CU_MASK=$m timeout -k 10 120 python3 tools/inflight_time.py 4 2000 0 >> $O
  done
done
PANO_CU_MASK=block timeout -k 10 120 python3 tools/inflight_time.py 2 2000 0 >> $O
PANO_CU_MASK=stride timeout -k 10 120 python3 tools/inflight_time.py 2 2000 0 >> $O
cat $O
