# round 5: whole GPU suite, the copy ceilings, the bench line, HBM bytes per kernel (tied to the kernel source id)
set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r05_tests.log 2>&1 || { tail -40 gpurun_out/r05_tests.log; exit 1; }
tail -2 gpurun_out/r05_tests.log
python tools/copy_ceiling.py > gpurun_out/r05_copy_ceiling.json 2> gpurun_out/r05_copy_ceiling.err || { tail -20 gpurun_out/r05_copy_ceiling.err; exit 1; }
bash tools/pmc_traffic_all.sh > gpurun_out/r05_traffic.log 2>&1 || { tail gpurun_out/r05_traffic.log; exit 1; }
echo traffic done
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err || { tail -20 gpurun_out/r05_bench.err; exit 1; }
echo bench done
