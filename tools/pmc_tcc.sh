# L2 / memory-side counters of the K1 launch for a library build: V=base bash tools/pmc_tcc.sh  -> gpurun_out/tcc_<V>_{a,b,c}/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${V:-product}
lib=$R/experiments/_build/libpano_$V.so; [ "$V" = product ] && lib=$R/img-stitching_amd/libpano_hip.so
export PANO_LIB=$lib
timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/tcc_${V}_a -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py > $R/gpurun_out/tcc_$V.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/tcc_${V}_b -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py >> $R/gpurun_out/tcc_$V.log 2>&1 || exit 1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $R/gpurun_out/tcc_${V}_c -o run --output-format csv -- python3 $R/tools/frames_one_at_a_time.py >> $R/gpurun_out/tcc_$V.log 2>&1 || exit 1
echo "tcc $V done"
