#!/bin/bash
# kernel trace of the karman-3d SOL-16 training step alone (ON THE GPU BOX): gpurun_out/k3dprof/k3d_train_kernel_stats.txt
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/k3dprof; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace3d -o trace -- python $R/tools/k3d_train_prof.py --steps 3 > $OUT/k3d_train_under_rocprof.json 2>$OUT/rocprof.err
DB=$(find $OUT/trace3d -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py "$DB" > $OUT/k3d_train_kernel_stats.txt
rm -rf $OUT/trace3d
