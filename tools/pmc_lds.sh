# LDS counters of the K1 launch for a library build: V=base bash tools/pmc_lds.sh  -> gpurun_out/lds_<V>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${V:-product}
lib=$R/experiments/_build/libpano_$V.so; [ "$V" = product ] && lib=$R/img-stitching_amd/libpano_hip.so
export PANO_LIB=$lib
timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU -d $R/gpurun_out/lds_$V -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py > $R/gpurun_out/lds_$V.log 2>&1 || exit 1
echo "pmc $V done"
