cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/lib_bitcompare.py zrot0 > gpurun_out/bitcmp2.txt 2>&1
tail -3 gpurun_out/bitcmp2.txt
SOL_HIP_LIB=solver-in-the-loop_amd/lib/libsol_bwwprof.so python tools/bww_row_probe.py > gpurun_out/bww_row_probe.txt 2>&1
cat gpurun_out/bww_row_probe.txt | tail -40
