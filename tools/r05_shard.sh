set -e
python -m pytest tests/test_gpu_sharded_ranks.py tests/test_gpu_bench_line.py tests/test_gpu_env_knobs.py -x -q > gpurun_out/r05_shard_tests.log 2>&1 || { tail -60 gpurun_out/r05_shard_tests.log; NCCL_DEBUG=INFO python -m pytest tests/test_gpu_sharded_ranks.py -x -q -k one_rank 2>&1 | grep -i "nccl\|rccl\|error\|warn" | head -40; exit 1; }
tail -3 gpurun_out/r05_shard_tests.log
bash tools/k1_decompose.sh
