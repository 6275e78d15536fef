# rocprofv3 kernel trace of the frames-in-flight loop and its summary (tools/timeline2.py): bash tools/trace_inflight.sh [F] -> gpurun_out/trace_F<F>.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F=${1:-4}
rm -rf $R/gpurun_out/trace_F$F
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_F$F -o run --output-format csv -- python3 $R/tools/inflight_time.py $F 400 > $R/gpurun_out/trace_F$F.log 2>&1 || { tail $R/gpurun_out/trace_F$F.log; exit 1; }
f=$(find $R/gpurun_out/trace_F$F -name '*kernel_trace.csv' | head -1)
python3 $R/tools/timeline2.py $f 900 > $R/gpurun_out/trace_F$F.json && cat $R/gpurun_out/trace_F$F.json
rm -rf $R/gpurun_out/trace_F$F
