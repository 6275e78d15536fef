# a soak of the two randomised-rig parity tests over other seeds: bash tools/fuzz_rigs.sh [cases per seed] [seeds...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=${1:-150}; shift
for S in ${@:-11 12 13}; do
  PANO_FUZZ_SEED=$S PANO_FUZZ_CASES=$C timeout -k 10 900 python -m pytest $R/tests/test_gpu_parity.py -x -q -k "randomised" 2>&1 | tail -4 || exit 1
done
