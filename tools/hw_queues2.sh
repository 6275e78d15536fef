R=${GRAFT_REPO_ROOT:-$(pwd)}
export PANO_LIB=$R/experiments/_build/libpano_slots16.so
for Q in 5 6 8 32; do
  for F in 3 4 5; do
    echo -n "Q=$Q "; GPU_MAX_HW_QUEUES=$Q timeout -k 10 120 python3 $R/tools/inflight_time.py $F 2000 2>/dev/null || exit 1
  done
done
unset PANO_LIB
for Q in 4 8; do
  echo -n "bench Q=$Q: "; GPU_MAX_HW_QUEUES=$Q timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-host-paths --no-c4 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.0f (%.1f us) from_idle %.0f rotating %.0f  K1 cold %.2f warm %.2f  L0 %.2f  one-at-a-time %.0f' % (d['value'], 1e3*d['ms_per_step'], d['from_idle']['value'], d['rotating_inputs_panoramas_per_s'], r['avg_launch_us'], r['warm']['avg_launch_us'], r['blend_level0']['avg_launch_us'], r['one_frame_at_a_time_panoramas_per_s']))"
done
