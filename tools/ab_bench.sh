# A/B of library builds through bench.py itself (value in flight, K1 warm / cold, level 0): LIBS="base product" bash tools/ab_bench.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${ROUNDS:-2}); do
  for v in ${LIBS:-product}; do
    lib=$R/experiments/_build/libpano_$v.so; [ "$v" = product ] && lib=$R/img-stitching_amd/libpano_hip.so
    PANO_LIB=$lib timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-host-paths --no-c4 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v: value %.0f pano/s (%.1f us)  rotating %.0f  K1 warm %.2f us (%.3f)  cold %.2f us (%.3f)  L0 %.2f  one-at-a-time %.0f' % (d['value'], 1e3*d['ms_per_step'], d['rotating_inputs_panoramas_per_s'], r['avg_launch_us'], r['frac'], r['cold']['avg_launch_us'], r['cold']['frac'], r['blend_level0']['avg_launch_us'], r['one_frame_at_a_time_panoramas_per_s']))" || exit 1
  done
done
