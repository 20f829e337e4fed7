cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/ab_lib.py --run pin0 acl1 --reps 3 > gpurun_out/ab3.txt 2>&1
tail -1 gpurun_out/ab3.txt
python tools/k3d_ab.py pin0 --reps 2 > gpurun_out/k3dab3.txt 2>&1
tail -1 gpurun_out/k3dab3.txt
SOL_HIP_LIB=solver-in-the-loop_amd/lib/libsol_bwwprof.so python tools/bww_row_probe.py > gpurun_out/bww_row_probe3.txt 2>&1
tail -28 gpurun_out/bww_row_probe3.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t3.txt
cat gpurun_out/t3.txt
