#!/bin/bash
# kernel times of the one-launch batch-norm passes for the product build and the DIEE_BN_ABLATE dev builds (libdiee_a1/a2.so)
cd /tmp && export TMPDIR=/tmp
for v in "" a1 a2; do
  if [ -n "$v" ]; then export DIEE_LIB=/root/repo/die-e_amd/libdiee_$v.so; else unset DIEE_LIB; fi
  [ -n "$v" ] && [ ! -f "$DIEE_LIB" ] && continue
  rocprofv3 --kernel-trace --stats -d /tmp/bn_$v -o o --output-format csv -- python3 /root/repo/scripts/train_step_engine_loop.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/bn_$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "bn_" in r["Name"]: print("variant '$v'", r["Name"][:30], r["Calls"], r["AverageNs"])
PY
done
