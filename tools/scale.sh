#!/bin/bash
# The 1 -> 8 GPU weak-scaling series of BASELINE.json's metric (karman-2d 128x64 SOL-32, 6 simulations per GPU) on ONE node, plus the
# readiness checklist of DESIGN.md section 6.  Needs a node with 8 MI355X; nothing here has run on such a node (rounds 1-6 had none).
#   tools/scale.sh [steps] [warmup]          -> gpurun_out/scale_N{1,2,4,8}.json + a table on stdout
#   tools/scale.sh --dry-run                 -> the same launcher, parser and table on ONE GPU: N = 1 and N = 2 (two ranks time-sharing the
#                                               device over gloo -- bench.py falls back to gloo only when there are more ranks than devices and
#                                               REFUSES any non-nccl backend otherwise), 3 steps each.  Says nothing about xGMI; it exists so
#                                               that the first real run on an 8-GPU lease does not fail on a typo.
# Each point is the driver's own command (python -m torch.distributed.run ... bench.py --gpus N); N > 1 runs skip the single-GPU side legs.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
NS="1 2 4 8"
if [ "${1:-}" = "--dry-run" ]; then NS="1 2"; shift; set -- 3 1; export SOL_SCALE_DRY=1; fi
STEPS=${1:-20}; WARM=${2:-5}
mkdir -p gpurun_out
for N in $NS; do
  PORT=$((29500 + N))
  if [ "$N" = 1 ]; then
    python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-extras --no-cpu-baseline > gpurun_out/scale_N1.json 2> gpurun_out/scale_N1.err
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$PORT" \
      bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline > "gpurun_out/scale_N$N.json" 2> "gpurun_out/scale_N$N.err"
  fi
  echo "N=$N rc=$?" >&2
done
python - <<'PY'
import json
rows = []
import os
for n in ((1, 2) if os.environ.get("SOL_SCALE_DRY") else (1, 2, 4, 8)):
    try:
        rows.append(json.loads(open("gpurun_out/scale_N%d.json" % n).read().strip().splitlines()[-1]))
    except Exception as e:      # noqa: BLE001
        print("N=%d: no line (%s)" % (n, e))
if rows:
    base = rows[0]["value"] / rows[0]["n_gpus"]
    print("%-4s %-12s %-14s %-10s %-10s %-12s %-10s %-14s" % ("N", "ms/step", "sim-steps/s", "x of N=1", "backend", "collectives", "skew", "bit-identical"))
    for r in rows:
        dp = r.get("data_parallel") or {}
        print("%-4d %-12.3f %-14.0f %-10.2f %-10s %-12s %-10.3f %-14s" % (
            r["n_gpus"], r["ms_per_step"], r["value"], r["value"] / base, dp.get("backend", "-"), dp.get("collectives_per_step", "-"),
            r.get("rank_skew") or 0.0, dp.get("weights_bit_identical_across_ranks", "-")))
    print("checklist (DESIGN.md section 6): backend must read nccl (= RCCL), collectives 1 per step, replicas bit-identical True, "
          "all-reduce of 1.04 MB in the tens of microseconds, skew < 0.05; the north star asks for >= 6.0 at N = 8")
    if os.environ.get("SOL_SCALE_DRY"):
        print("DRY RUN: ranks shared one device over gloo -- the table proves the launcher / parser / checklist plumbing, not scaling")
PY
