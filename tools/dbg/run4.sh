cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/lib_bitcompare.py pipe0 > gpurun_out/bitcmp4.txt 2>&1
tail -1 gpurun_out/bitcmp4.txt
python -m pytest tests -m gpu -x -q -k "train_step or determin or sol32_bench or conv3d_gradients or finite_differences or weight" 2>&1 | tail -3 > gpurun_out/t4.txt
cat gpurun_out/t4.txt
python tools/ab_lib.py --run pipe0 reqe --reps 3 > gpurun_out/ab4.txt 2>&1
tail -1 gpurun_out/ab4.txt
python tools/k3d_ab.py pipe0 --reps 2 > gpurun_out/k3dab4.txt 2>&1
tail -1 gpurun_out/k3dab4.txt
SOL_HIP_LIB=solver-in-the-loop_amd/lib/libsol_bwwprof.so python tools/bww_row_probe.py > gpurun_out/bww_row_probe4.txt 2>&1
tail -24 gpurun_out/bww_row_probe4.txt | head -23
