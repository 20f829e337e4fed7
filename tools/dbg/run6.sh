cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SOL_HIP_LIB=solver-in-the-loop_amd/lib/libsol_bwwprof.so python tools/bww_row_probe.py > gpurun_out/bww_row_probe6.txt 2>&1
tail -26 gpurun_out/bww_row_probe6.txt | head -25
