cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out/l0_ab.log; : > $O
B=$PWD/experiments/_build/libpano_base.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -1 >> $O
for rep in 1 2 3; do
  echo -n "base one: " >> $O; PANO_LIB=$B timeout -k 10 120 python3 tools/frames_one_at_a_time.py 2>/dev/null >> $O
  echo -n "new  one: " >> $O; timeout -k 10 120 python3 tools/frames_one_at_a_time.py 2>/dev/null >> $O
done
for rep in 1 2 3; do for rot in 0 1; do
  echo -n "base: " >> $O; PANO_LIB=$B timeout -k 10 120 python3 tools/inflight_time.py 3 2000 $rot 2>/dev/null >> $O
  echo -n "new : " >> $O; timeout -k 10 120 python3 tools/inflight_time.py 3 2000 $rot 2>/dev/null >> $O
done; done
cat $O
