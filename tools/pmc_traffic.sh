# HBM traffic of the 8-camera K1 launch: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section),
# plus the kernel-trace stats of a default bench run.  Usage on the GPU box: bash tools/pmc_traffic.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/trF -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py > $R/gpurun_out/trF.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/trW -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py > $R/gpurun_out/trW.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trS -o run --output-format csv -- python3 $R/bench.py --steps 300 --no-cpu-baseline > $R/gpurun_out/trS.json 2> $R/gpurun_out/trS.log || exit 1
echo done
