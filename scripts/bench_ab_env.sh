#!/bin/bash
# A/B of an environment switch on the full bench (same box, alternating): scripts/bench_ab_env.sh VAR v1 v2 [bench args...]
VAR=$1; A=$2; B=$3; shift 3
for r in 1 2; do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 2 --warmup 0 --no-cpu-baseline --pipeline 0 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value'],2), 'games/s', round(d['ms_per_step']), 'ms')"
  done
done
