cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/k3dprof; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/k3d_prof.py <<PY
import sys, json, torch
sys.path.insert(0, "$R")
import sol_amd, bench
print(json.dumps(bench.karman3d_leg(sol_amd, torch.device("cuda", 0))))
PY
rocprofv3 --kernel-trace --stats -d $OUT/trace3d -o trace -- python /tmp/k3d_prof.py > $OUT/k3d_under_rocprof.json 2>$OUT/rocprof.err
DB=$(find $OUT/trace3d -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py "$DB" > $OUT/k3d_kernel_stats.txt
rm -rf $OUT/trace3d
