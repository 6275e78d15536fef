cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for M in frames phases; do
rm -rf $R/gpurun_out/tp
MODE=$M timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/tp -o run --output-format csv -- python3 $R/tools/pipeline_time.py 600 > $R/gpurun_out/tp.log 2>&1 || { tail -5 $R/gpurun_out/tp.log; exit 1; }
echo "== $M"; python3 $R/tools/queue_busy.py $R/gpurun_out/tp/run_kernel_trace.csv
done
rm -rf $R/gpurun_out/tp
