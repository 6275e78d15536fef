#!/bin/bash
# parity suite on a library build (VARIANT=<name> -> experiments/_build/libpano_<name>.so), then product against it: one frame at
# a time (stage us) and four frames in flight, warm and over rotating frame sets, interleaved
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
V=$PWD/experiments/_build/libpano_${VARIANT}.so
PANO_LIB=$V timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/variant_tests.log 2>&1; tail -3 gpurun_out/variant_tests.log
grep -q " passed" gpurun_out/variant_tests.log || exit 1
grep -q "failed" gpurun_out/variant_tests.log && exit 1
O=gpurun_out/variant_ab.jsonl; : > $O
for rep in 1 2 3; do
  echo -n "product one: " >> $O; timeout -k 10 120 python3 tools/frames_one_at_a_time.py 2>/dev/null >> $O
  echo -n "$VARIANT one: " >> $O; PANO_LIB=$V timeout -k 10 120 python3 tools/frames_one_at_a_time.py 2>/dev/null >> $O
done
for rep in 1 2 3; do
  for rot in 0 1; do
    timeout -k 10 120 python3 tools/inflight_time.py 4 2000 $rot 2>/dev/null >> $O
    PANO_LIB=$V timeout -k 10 120 python3 tools/inflight_time.py 4 2000 $rot 2>/dev/null >> $O
  done
done
cat $O
