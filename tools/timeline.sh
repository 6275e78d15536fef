# kernel timeline of the default (4 frames in flight) loop: how busy is the GPU, how many kernels run at once, how long
# does each kernel take when it shares the machine.  Usage on the GPU box: bash tools/timeline.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl
timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/tl -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-isolated-pass --steps 300 > $R/gpurun_out/tl.json 2> $R/gpurun_out/tl.log || exit 1
python3 $R/tools/timeline.py $R/gpurun_out/tl/run_kernel_trace.csv > $R/gpurun_out/timeline_summary.json || exit 1
cat $R/gpurun_out/timeline_summary.json
