#!/bin/bash
# round 5: combinations of the tail's options against the batch time (each "KEY=V KEY=V" group is one configuration; two passes, interleaved)
#   CONFIGS="SPEC_EXTRA_ROWS=2+SPEC_ROLLOUT_STEPS=24 SPEC_EXTRA_ROWS=4+SPEC_ROLLOUT_STEPS=40" scripts/tail_opts_sweep.sh
for pass in 1 2; do
for C in ${CONFIGS}; do
  E=""; for kv in ${C//+/ }; do E="$E DIEE_$kv"; done
  env $E python3 bench.py --no-cpu-baseline --pipeline 0 --hbm-only-steps 0 --steps ${STEPS:-2} > /tmp/l.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('/tmp/l.json')); print('$C:', round(d['value'],2), round(d['ms_per_step']), d['stats']['tail'])"
done; done
