#!/bin/bash
# A/B of an environment override (e.g. SOL_CONV_NO_DX=1) against the default, SEPARATE processes alternating on one box:
#   bash tools/ab_env.sh SOL_CONV_NO_DX=1 [reps] [bench args...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
VAR=$1; REPS=${2:-3}; shift; shift
for r in $(seq 1 $REPS); do
  for v in default override; do
    if [ $v = override ]; then export "$VAR"; else unset "${VAR%%=*}"; fi
    ms=$(python $R/bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "rep $r $v ($VAR): $ms ms"
  done
done
