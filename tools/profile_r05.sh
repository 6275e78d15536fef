# the rocprofv3 kernel summaries of round 5 (on the GPU box: bash tools/profile_r05.sh): the default bench command (the judged
# line's own command: its kernel averages mix the in-flight and the one-at-a-time passes), the one-frame-at-a-time loop, the cold
# loop (rotating frame sets) and config 4.  Summaries land in gpurun_out/r05_*; the *_kernel_stats.csv are copied into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in default one cold c4; do rm -rf $R/gpurun_out/r05_$T; done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_default -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-host-paths > $R/gpurun_out/r05_default.json 2> $R/gpurun_out/r05_default.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_one -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-host-paths --no-c4 --frames-in-flight 1 > $R/gpurun_out/r05_one.json 2> $R/gpurun_out/r05_one.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_cold -o run --output-format csv -- python3 $R/bench.py --cold-only > $R/gpurun_out/r05_cold.json 2> $R/gpurun_out/r05_cold.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_c4 -o run --output-format csv -- python3 $R/tools/bench_c4.py > $R/gpurun_out/r05_c4.json 2> $R/gpurun_out/r05_c4.log || exit 1
echo profiles done
