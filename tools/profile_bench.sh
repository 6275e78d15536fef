# rocprofv3 kernel stats of the default bench command and of the one-frame-at-a-time command.
# Usage on the GPU box: bash tools/profile_bench.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pb2 -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-host-paths > $R/gpurun_out/pb2.json 2> $R/gpurun_out/pb2.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pb1 -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-host-paths --frames-in-flight 1 > $R/gpurun_out/pb1.json 2> $R/gpurun_out/pb1.log || exit 1
timeout -k 10 400 python3 $R/bench.py > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err || exit 1
echo done
