#!/bin/bash
# fused K1 v2 (ring pixels one per lane of the four patch waves; experiments/_build/libpano_fused_v2.so) against v1 (fifth wave) and
# the product: parity suite first, then one frame at a time (stage us) and four frames in flight
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=${FUSE_LIB:-$PWD/experiments/_build/libpano_fused_v2.so}
PANO_LIB=$V timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/fuse_v2_tests.log 2>&1; tail -5 gpurun_out/fuse_v2_tests.log
grep -q " passed" gpurun_out/fuse_v2_tests.log || exit 1
grep -q "failed" gpurun_out/fuse_v2_tests.log && exit 1
O=gpurun_out/fuse_v2.jsonl; : > $O
for rep in 1 2 3; do
  echo -n "product one: " >> $O; timeout -k 10 120 python3 tools/frames_one_at_a_time.py 2>/dev/null >> $O
  echo -n "v2 one:      " >> $O; PANO_LIB=$V timeout -k 10 120 python3 tools/frames_one_at_a_time.py 2>/dev/null >> $O
  echo -n "v1 one:      " >> $O; PANO_LIB=$PWD/experiments/_build/libpano_fused_v1.so timeout -k 10 120 python3 tools/frames_one_at_a_time.py 2>/dev/null >> $O
done
for rep in 1 2 3; do
  for rot in 0 1; do
    timeout -k 10 120 python3 tools/inflight_time.py 4 2000 $rot 2>/dev/null >> $O
    PANO_LIB=$V timeout -k 10 120 python3 tools/inflight_time.py 4 2000 $rot 2>/dev/null >> $O
    PANO_LIB=$V PANO_K1_FUSE=0 timeout -k 10 120 python3 tools/inflight_time.py 4 2000 $rot 2>/dev/null >> $O
  done
done
cat $O
