# everything the round's profiles/ come from, on the GPU box: GPU tests, copy ceilings, HBM bytes per kernel (tied to the kernel
# source id), the bench lines (default and the driver's form), rocprofv3 kernel summaries (default / one frame at a time / cold / c4),
# the K1 decomposition.  bash tools/final_r05.sh
set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r05_tests.log 2>&1 || { tail -40 gpurun_out/r05_tests.log; exit 1; }
tail -2 gpurun_out/r05_tests.log
python tools/copy_ceiling.py > gpurun_out/r05_copy_ceiling.json 2> gpurun_out/r05_copy_ceiling.err || { tail -20 gpurun_out/r05_copy_ceiling.err; exit 1; }
bash tools/pmc_traffic_all.sh > gpurun_out/r05_traffic.log 2>&1 || { tail gpurun_out/r05_traffic.log; exit 1; }
python tools/adopt_traffic.py || exit 1   # the bench prints counter traffic only beside the kernel source id it was counted on
echo traffic done
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err || { tail -20 gpurun_out/r05_bench.err; exit 1; }
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_form.json 2> gpurun_out/r05_bench_driver_form.err || { tail -20 gpurun_out/r05_bench_driver_form.err; exit 1; }
echo bench done
bash tools/profile_r05.sh
bash tools/k1_decompose.sh > gpurun_out/r05_k1_decompose.log 2>&1 || { tail gpurun_out/r05_k1_decompose.log; exit 1; }
echo all done
