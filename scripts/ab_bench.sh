#!/bin/bash
# A/B of two builds on the SAME box: bench.py with DIEE_LIB=die-e_amd/libdiee_ab.so (A) and the product library (B), alternating
for r in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export DIEE_LIB=$(pwd)/die-e_amd/libdiee_ab.so; else unset DIEE_LIB; fi
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d.get('value_pipelined',0),2), round(d['ms_per_step']))"
  done
done
