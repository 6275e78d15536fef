#!/bin/bash
# A/B of cache-policy hints (non-temporal stores of the panorama, non-temporal loads of level 0's inputs / of K1's frames):
# experiments/_build/libpano_nt*.so against the product, 4 frames in flight, warm and over six rotating frame sets
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/nt_ab.jsonl; : > $O
for rep in 1 2; do
  for rot in 0 1; do
    timeout -k 10 120 python3 tools/inflight_time.py 4 2000 $rot >> $O
    for v in ${NT_VARIANTS:-ntout ntio ntio_k1 ntout_k1}; do
      PANO_LIB=$PWD/experiments/_build/libpano_$v.so timeout -k 10 120 python3 tools/inflight_time.py 4 2000 $rot >> $O
    done
  done
done
cat $O
