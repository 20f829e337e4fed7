cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_karman3d.py tests/test_gpu_determinism.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/t8.txt
cat gpurun_out/t8.txt
python tools/k3d_thin_ab.py --reps 2 > gpurun_out/k3dthin8.txt 2>&1
tail -1 gpurun_out/k3dthin8.txt
