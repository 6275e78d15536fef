# rocprofv3 kernel stats of config 4 (tools/bench_c4.py).  Usage on the GPU box: bash tools/profile_c4.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pc4
F=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pc4 -o run --output-format csv -- python3 $R/tools/bench_c4.py > $R/gpurun_out/pc4.json 2> $R/gpurun_out/pc4.log || exit 1
head -12 $R/gpurun_out/pc4/run_kernel_stats.csv | cut -c1-150
