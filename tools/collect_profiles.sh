#!/bin/bash
# Round-6 measurement artefacts (same recipe as rounds 3 to 5), run ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh'):
#   1. bench.py line (unprofiled)                                   -> gpurun_out/r6prof/bench.json
#   2. rocprofv3 --kernel-trace --stats of the SAME bench command   -> kernel_stats.txt (tools/rocpd_stats.py)
#   3. PMC passes, one counter set each (FETCH_SIZE / WRITE_SIZE cannot share a pass; --pmc never combined with other
#      trace domains): HBM traffic per launch -> r06_pmc_traffic.json (with the library's source hash), and the SQ
#      counters of the conv / solver kernels -> pmc_sq.txt
# Copy what is to be judged from gpurun_out/r6prof/ into profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/r6prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5"
[ "${1:-}" = "--2d-only" ] && ARGS="--steps 10 --warmup 3 --no-cpu-baseline"
if [ "${1:-}" = "--pmc-only" ]; then      # the counter passes of the 2-D step alone (after a source change: the traffic file is tied to the build)
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --prewarm 0 --no-extras --no-cpu-baseline --no-device-state > /dev/null 2>>$OUT/rocprof.err
    python $R/tools/pmc_summary.py $OUT/pmc_$C "" --json $OUT/r06_pmc_traffic.json > $OUT/pmc_$C.txt
  done
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 \
    --output-format csv -d $OUT/pmc_sq -o pmc -- python $R/bench.py --steps 3 --warmup 1 --prewarm 0 --no-extras --no-cpu-baseline --no-device-state > /dev/null 2>>$OUT/rocprof.err
  python $R/tools/pmc_summary.py $OUT/pmc_sq "k_" > $OUT/pmc_sq.txt
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq; ls -la $OUT; exit 0
fi
python $R/bench.py $ARGS 2>$OUT/bench.err | tail -1 > $OUT/bench.json
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py $ARGS > $OUT/bench_under_rocprof.json 2>$OUT/rocprof.err
DB=$(find $OUT/trace -name "*.db" | head -1)
if [ -n "$DB" ]; then python $R/tools/rocpd_stats.py "$DB" > $OUT/kernel_stats.txt; else ls -R $OUT/trace > $OUT/kernel_stats.txt; fi
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --prewarm 0 --no-extras --no-cpu-baseline --no-device-state > /dev/null 2>>$OUT/rocprof.err
  python $R/tools/pmc_summary.py $OUT/pmc_$C "" --json $OUT/r06_pmc_traffic.json > $OUT/pmc_$C.txt
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 \
  --output-format csv -d $OUT/pmc_sq -o pmc -- python $R/bench.py --steps 3 --warmup 1 --prewarm 0 --no-extras --no-cpu-baseline --no-device-state > /dev/null 2>>$OUT/rocprof.err
python $R/tools/pmc_summary.py $OUT/pmc_sq "k_" > $OUT/pmc_sq.txt
if [ "${1:-}" = "--2d-only" ]; then rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq; ls -la $OUT; exit 0; fi     # (the karman-3d passes take ten minutes of box time)
# karman-3d: kernel trace + HBM counters of the forward roll-out and one training step
cat > /tmp/k3d_prof.py <<PY
import sys, json, torch
sys.path.insert(0, "$R")
import sol_amd, bench
print(json.dumps(bench.karman3d_leg(sol_amd, torch.device("cuda", 0))))
PY
rocprofv3 --kernel-trace --stats -d $OUT/trace3d -o trace -- python /tmp/k3d_prof.py > $OUT/k3d_under_rocprof.json 2>>$OUT/rocprof.err
DB=$(find $OUT/trace3d -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_stats.py "$DB" > $OUT/k3d_kernel_stats.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc3d_$C -o pmc -- python $R/tools/k3d_time.py 1 2 > /dev/null 2>>$OUT/rocprof.err
  python $R/tools/pmc_summary.py $OUT/pmc3d_$C "k" > $OUT/pmc3d_$C.txt
done
# SQ counters of the one-launch Conv3D kernels (three-, six-, eight-row forms; tools/k3d_time.py runs all three)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 \
  --output-format csv -d $OUT/pmc3d_sq -o pmc -- python $R/tools/k3d_time.py 1 2 > /dev/null 2>>$OUT/rocprof.err
python $R/tools/pmc_summary.py $OUT/pmc3d_sq "k_conv3d" > $OUT/pmc_sq_conv3d.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD \
  --output-format csv -d $OUT/pmc3d_sq2 -o pmc -- python $R/tools/k3d_time.py 1 2 > /dev/null 2>>$OUT/rocprof.err
python $R/tools/pmc_summary.py $OUT/pmc3d_sq2 "k_conv3d" >> $OUT/pmc_sq_conv3d.txt
python $R/tools/k3d_time.py 1 4 > $OUT/k3d_time_b1.txt 2>>$OUT/rocprof.err
rm -rf $OUT/trace $OUT/trace3d $OUT/pmc3d_sq $OUT/pmc3d_sq2 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq $OUT/pmc3d_FETCH_SIZE $OUT/pmc3d_WRITE_SIZE
ls -la $OUT
