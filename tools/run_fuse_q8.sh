R=${GRAFT_REPO_ROOT:-$(pwd)}
export PANO_LIB=$R/experiments/_build/libpano_k1fuse.so PANO_TORCH_STREAMS=1 GPU_MAX_HW_QUEUES=8
for rep in 1 2; do
  echo -n "fused: "; timeout -k 10 120 python3 $R/tools/inflight_time.py 4 2000 2>/dev/null
  echo -n "plain: "; PANO_K1_FUSE=0 timeout -k 10 120 python3 $R/tools/inflight_time.py 4 2000 2>/dev/null
  echo -n "fused F=3: "; timeout -k 10 120 python3 $R/tools/inflight_time.py 3 2000 2>/dev/null
  echo -n "plain F=3: "; PANO_K1_FUSE=0 timeout -k 10 120 python3 $R/tools/inflight_time.py 3 2000 2>/dev/null
done
unset PANO_LIB PANO_TORCH_STREAMS GPU_MAX_HW_QUEUES
for F in 2 3 4; do echo -n "product probed F=$F: "; timeout -k 10 120 python3 $R/tools/inflight_time.py $F 2000 2>/dev/null; echo -n "product probed rotating F=$F: "; timeout -k 10 120 python3 $R/tools/inflight_time.py $F 2000 1 2>/dev/null; done
