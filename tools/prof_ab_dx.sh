R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/dxprof; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 0 ]; then export SOL_CONV_NO_DX=1; else unset SOL_CONV_NO_DX; fi
  rocprofv3 --kernel-trace --stats -d $OUT/t$v -o trace -- python $R/bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $OUT/bench$v.json 2>$OUT/err$v.txt
  DB=$(find $OUT/t$v -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py "$DB" > $OUT/stats$v.txt
  python $R/tools/rocpd_gaps.py "$DB" > $OUT/gaps$v.txt
  rm -rf $OUT/t$v
done
head -14 $OUT/stats0.txt; echo; head -14 $OUT/stats1.txt
