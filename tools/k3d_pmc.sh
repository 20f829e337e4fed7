#!/bin/bash
# SQ counters of the karman-3d Conv3D kernels (three- and six-row forms), run ON THE GPU BOX: gpurun -- 'bash tools/k3d_pmc.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/k3dpmc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 \
  --output-format csv -d $OUT/a -o pmc -- python $R/tools/k3d_time.py 1 2 > /dev/null 2>$OUT/err_a.txt
python $R/tools/pmc_summary.py $OUT/a "k_conv3d" > $OUT/pmc_sq_conv3d.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD \
  --output-format csv -d $OUT/b -o pmc -- python $R/tools/k3d_time.py 1 2 > /dev/null 2>$OUT/err_b.txt
python $R/tools/pmc_summary.py $OUT/b "k_conv3d" >> $OUT/pmc_sq_conv3d.txt
tail -3 $OUT/err_b.txt
rm -rf $OUT/a $OUT/b
cat $OUT/pmc_sq_conv3d.txt
