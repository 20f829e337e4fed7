cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_karman3d.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t7.txt
cat gpurun_out/t7.txt
python tools/k3d_ab.py --reps 1 > gpurun_out/k3dab7.txt 2>&1
tail -2 gpurun_out/k3dab7.txt
