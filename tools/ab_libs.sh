# A/B of library builds on the GPU box, one frame at a time (tools/frames_one_at_a_time.py: K1 / pyramid / blend / level-0 us from the
# dispatch events), interleaved over ROUNDS rounds: LIBS="base k1free" bash tools/ab_libs.sh   ("product" = the in-tree library)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq 1 ${ROUNDS:-3}); do
  for v in ${LIBS:-product}; do
    lib=$R/experiments/_build/libpano_$v.so; [ "$v" = product ] && lib=$R/img-stitching_amd/libpano_hip.so
    echo -n "$v: "; PANO_LIB=$lib timeout -k 10 120 python3 $R/tools/frames_one_at_a_time.py || exit 1
  done
done
