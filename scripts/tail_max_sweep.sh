#!/bin/bash
# round 5: one option of the tail of a batch (default: its reach, spec_max_games) against the batch time, values interleaved to average box drift
#   SWEEP="64 96 64 96" OPT=SPEC_MAX_GAMES STEPS=3 scripts/tail_max_sweep.sh
OPT=${OPT:-SPEC_MAX_GAMES}
for M in ${SWEEP:-64 96 64 96}; do
  env DIEE_$OPT=$M python3 bench.py --no-cpu-baseline --pipeline 0 --hbm-only-steps 0 --steps ${STEPS:-3} > /tmp/l.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('/tmp/l.json')); print('$OPT $M:', round(d['value'],2), round(d['ms_per_step']), d['stats']['tail'])"
done
