# SQ counter passes over the 8-camera K1 launch (tools/frames_one_at_a_time.py).  Usage on the GPU box: bash tools/pmc_sq.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 100 python3 $R/tools/frames_one_at_a_time.py | grep warp8 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $R/gpurun_out/sq1 -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py > $R/gpurun_out/sq1.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/sq2 -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py > $R/gpurun_out/sq2.log 2>&1 || exit 1
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_CYCLES SQ_THREAD_CYCLES_VALU -d $R/gpurun_out/sq3 -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py > $R/gpurun_out/sq3.log 2>&1 || exit 1
echo done
