cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/ab_lib.py --run ps0 ps3 --reps 3 > gpurun_out/ab5.txt 2>&1
tail -1 gpurun_out/ab5.txt
python tools/k3d_ab.py ps0 ps3 --reps 2 > gpurun_out/k3dab5.txt 2>&1
tail -1 gpurun_out/k3dab5.txt
SOL_HIP_LIB=solver-in-the-loop_amd/lib/libsol_bwwprof0.so python tools/bww_row_probe.py > gpurun_out/bww_row_probe5.txt 2>&1
tail -24 gpurun_out/bww_row_probe5.txt | head -23
