#!/bin/bash
# round 5: the reach of the tail of a batch (option spec_max_games) against the batch time, interleaved to average box drift
for M in ${SWEEP:-64 96 64 96}; do
  DIEE_SPEC_MAX_GAMES=$M python3 bench.py --no-cpu-baseline --pipeline 0 --hbm-only-steps 0 --steps ${STEPS:-3} > /tmp/l.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('/tmp/l.json')); print('spec_max_games $M:', round(d['value'],2), round(d['ms_per_step']), d['stats']['tail'])"
done
