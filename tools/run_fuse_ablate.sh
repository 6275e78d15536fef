R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for v in product abl1 abl2 abl3; do
    lib=$R/experiments/_build/libpano_$v.so; [ "$v" = product ] && lib=$R/img-stitching_amd/libpano_hip.so
    echo -n "$v: "; PANO_LIB=$lib timeout -k 10 120 python3 $R/tools/frames_one_at_a_time.py 2>/dev/null
  done
  echo -n "plain: "; PANO_K1_FUSE=0 timeout -k 10 120 python3 $R/tools/frames_one_at_a_time.py 2>/dev/null
done
