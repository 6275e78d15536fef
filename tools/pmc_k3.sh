cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PANO_LIB=$R/img-stitching_amd/libpano_hip_diag.so
for a in 0 5; do
export PANO_K3_ABL=$a
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $R/gpurun_out/sqk3_$a -o run --output-format csv -- python3 $R/tools/warp_ablate8.py > $R/gpurun_out/sqk3_$a.log 2>&1 || exit 1
done
echo done
