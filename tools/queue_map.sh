# which hardware queue does each flight stream's work land on?  rocprofv3 kernel trace, Queue_Id of consecutive K1 dispatches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for Q in 4 8; do
  rm -rf $R/gpurun_out/qm
  GPU_MAX_HW_QUEUES=$Q timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/qm -o run --output-format csv -- python3 $R/tools/inflight_time.py 4 300 > $R/gpurun_out/qm.log 2>&1 || { tail -5 $R/gpurun_out/qm.log; exit 1; }
  python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$R/gpurun_out/qm/run_kernel_trace.csv")) if "warp_tiles_lut" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
print("inflight Q=$Q: K1 queue ids of 16 consecutive frames:", [r["Queue_Id"] for r in rows[-16:]], "distinct", len(set(r["Queue_Id"] for r in rows[-400:])))
PY
  rm -rf $R/gpurun_out/qm
  GPU_MAX_HW_QUEUES=$Q timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/qm -o run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-host-paths --no-c4 --no-isolated-pass > $R/gpurun_out/qm.log 2>&1 || { tail -5 $R/gpurun_out/qm.log; exit 1; }
  python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$R/gpurun_out/qm/run_kernel_trace.csv")) if "warp_tiles_lut" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
mid=rows[len(rows)//2-200:len(rows)//2]
print("bench Q=$Q: K1 queue ids of 16 consecutive frames (middle of the run):", [r["Queue_Id"] for r in mid[-16:]], "distinct", len(set(r["Queue_Id"] for r in mid)))
PY
done
rm -rf $R/gpurun_out/qm
