# the rocprofv3 kernel summaries of round 2 (run on the GPU box: bash tools/profile_r02.sh): the default bench command, the
# one-frame-at-a-time loop and the cold (rotating frame sets) loop.  Summaries land in gpurun_out/r02_*; copy into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in default one cold; do rm -rf $R/gpurun_out/r02_$T; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_default -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-host-paths > $R/gpurun_out/r02_default.json 2> $R/gpurun_out/r02_default.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_one -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-host-paths --frames-in-flight 1 > $R/gpurun_out/r02_one.json 2> $R/gpurun_out/r02_one.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02_cold -o run --output-format csv -- python3 $R/bench.py --cold-only > $R/gpurun_out/r02_cold.json 2> $R/gpurun_out/r02_cold.log || exit 1
echo done
