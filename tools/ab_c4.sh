# config 4 (4 x 4K, cylindrical, 7 bands) K1 with / without gain maps for several library builds: LIBS="base product" bash tools/ab_c4.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${ROUNDS:-2}); do
  for v in ${LIBS:-product}; do
    lib=$R/experiments/_build/libpano_$v.so; [ "$v" = product ] && lib=$R/img-stitching_amd/libpano_hip.so
    for g in 1 0; do
      echo -n "$v gains=$g: "; PANO_LIB=$lib GAINS=$g timeout -k 10 200 python3 $R/tools/bench_c4.py 2>/dev/null | python3 -c "
import sys, json
a=json.loads(sys.stdin.readline()); b=json.loads(sys.stdin.readline())
print('K1 %.2f us (%.3f of 8 TB/s)  pyr %.1f  blend %.1f   %.0f pano/s in flight' % (a['k1_c4']['avg_launch_us'], a['k1_c4']['frac_of_8TBps'], a['k1_c4']['stage_us']['pyramid'], a['k1_c4']['stage_us']['blend'], b['panoramas_per_s']))" || exit 1
    done
  done
done
