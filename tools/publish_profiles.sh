#!/bin/bash
# Copies the artefacts of tools/collect_profiles.sh (gpurun_out/r6prof/, scratch) into profiles/ (tracked) under their round-6 names.
# usage: bash tools/publish_profiles.sh [bench-suffix]     (default suffix: b  ->  profiles/r06_bench_1gpu_b.json)
R=$(cd "$(dirname "$0")/.." && pwd); S=$R/gpurun_out/r6prof; P=$R/profiles; X=${1:-b}
cp $S/bench.json $P/r06_bench_1gpu_$X.json
cp $S/kernel_stats.txt $P/r06_bench_kernel_stats.txt
[ -f $S/k3d_kernel_stats.txt ] && cp $S/k3d_kernel_stats.txt $P/r06_karman3d_kernel_stats.txt
[ -f $S/k3d_time_b1.txt ] && cp $S/k3d_time_b1.txt $P/r06_k3d_time_b1.txt
cp $S/pmc_sq.txt $P/r06_pmc_sq.txt
[ -f $S/pmc_sq_conv3d.txt ] && cp $S/pmc_sq_conv3d.txt $P/r06_pmc_sq_conv3d.txt
cp $S/r06_pmc_traffic.json $P/r06_pmc_traffic.json
cat $S/pmc_FETCH_SIZE.txt $S/pmc_WRITE_SIZE.txt > $P/r06_pmc_summary.txt
[ -f $S/pmc3d_FETCH_SIZE.txt ] && cat $S/pmc3d_FETCH_SIZE.txt $S/pmc3d_WRITE_SIZE.txt > $P/r06_pmc_karman3d.txt
python $R/tools/design_table.py $P/r06_bench_1gpu_$X.json
