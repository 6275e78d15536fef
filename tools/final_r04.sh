# everything the round's profiles/ come from, on the GPU box: GPU tests, the default bench line, rocprofv3 kernel summaries
# (default / one frame at a time / cold), HBM bytes and VALU counters per kernel.  bash tools/final_r04.sh
set -e
python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 || { tail -40 gpurun_out/final_tests.log; exit 1; }
tail -2 gpurun_out/final_tests.log
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err || { tail -20 gpurun_out/final_bench.err; exit 1; }
echo bench done
bash tools/profile_r04.sh
bash tools/pmc_traffic_all.sh > gpurun_out/final_traffic.log 2>&1 || { tail gpurun_out/final_traffic.log; exit 1; }
echo traffic done
bash tools/pmc_valu_all.sh > gpurun_out/final_valu.log 2>&1 || { tail gpurun_out/final_valu.log; exit 1; }
echo valu done
