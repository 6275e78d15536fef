R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for v in product canvas8; do
    lib=$R/experiments/_build/libpano_$v.so; [ "$v" = product ] && lib=$R/img-stitching_amd/libpano_hip.so
    echo -n "$v: "; PANO_LIB=$lib timeout -k 10 120 python3 $R/tools/inflight_time.py 4 2000 2>/dev/null
  done
done
for v in product canvas8; do
    lib=$R/experiments/_build/libpano_$v.so; [ "$v" = product ] && lib=$R/img-stitching_amd/libpano_hip.so
    echo -n "$v alone: "; PANO_LIB=$lib timeout -k 10 120 python3 $R/tools/frames_one_at_a_time.py 2>/dev/null
done
