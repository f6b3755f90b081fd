#!/bin/bash
# k_expand's average launch time over the first move-steps of the headline batch, product build vs DIEE_LIB=libdiee_ab.so
cd /tmp && export TMPDIR=/tmp
for v in ab ""; do
  if [ -n "$v" ]; then export DIEE_LIB=/root/repo/die-e_amd/libdiee_$v.so; else unset DIEE_LIB; fi
  rocprofv3 --kernel-trace --stats -d /tmp/ex_$v -o o --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --pipeline 0 --max-steps 4 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/ex_$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_expand" in r["Name"] or "k_policy" in r["Name"]: print("variant '$v'", r["Name"][:24], r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
