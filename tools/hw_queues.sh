# How many HIP streams really run side by side?  ROCclr multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4):
# frames in flight F x hardware queues Q, us per config-2 frame (tools/inflight_time.py; the library built with PANO_MAX_FRAME_SLOTS=16)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PANO_LIB=$R/experiments/_build/libpano_slots16.so
for Q in 4 8 16; do
  for F in 4 6 8 12; do
    echo -n "Q=$Q "; GPU_MAX_HW_QUEUES=$Q timeout -k 10 120 python3 $R/tools/inflight_time.py $F 2000 2>/dev/null || exit 1
  done
done
