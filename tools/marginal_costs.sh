# what each launch of a config-2 frame adds to the frame time with four frames in flight (and with one): the skip_launches variant
# with one launch left out at a time (TIMING ONLY: the pictures are wrong).  bash tools/marginal_costs.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PANO_LIB=$R/experiments/_build/libpano_skip.so
# four flight streams on four DISTINCT hardware queues (with ROCclr's default of 4 queues two of this script's streams share one: tools/queue_map.sh)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
for F in 4 1; do
  for s in 0 1 2 4 8 16 32 64 128 6 14 0; do
    PANO_SKIP=$s timeout -k 10 120 python3 $R/tools/inflight_time.py $F 2000 || exit 1
  done
done
