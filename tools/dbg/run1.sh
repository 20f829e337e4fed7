set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/lib_bitcompare.py zrot0 at1 > gpurun_out/bitcmp1.txt 2>&1
tail -5 gpurun_out/bitcmp1.txt
python -m pytest tests -m gpu -x -q -k "train_step or determin or sol32_bench or conv3d_gradients or finite_differences or weight" 2>&1 | tail -5 > gpurun_out/t1.txt
cat gpurun_out/t1.txt
python tools/ab_lib.py --run zrot0 at1 --reps 3 > gpurun_out/ab1.txt 2>&1
tail -3 gpurun_out/ab1.txt
python tools/k3d_ab.py zrot0 --reps 2 > gpurun_out/k3dab1.txt 2>&1
tail -2 gpurun_out/k3dab1.txt
