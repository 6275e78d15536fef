timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/fuse_tests.log 2>&1; tail -12 gpurun_out/fuse_tests.log
for rep in 1 2 3; do
  echo -n "fused: "; timeout -k 10 120 python3 tools/frames_one_at_a_time.py 2>/dev/null
  echo -n "plain: "; PANO_K1_FUSE=0 timeout -k 10 120 python3 tools/frames_one_at_a_time.py 2>/dev/null
done
for rep in 1 2; do
  echo -n "fused: "; timeout -k 10 120 python3 tools/inflight_time.py 4 2000 2>/dev/null
  echo -n "plain: "; PANO_K1_FUSE=0 timeout -k 10 120 python3 tools/inflight_time.py 4 2000 2>/dev/null
done
