#!/bin/bash
# kernel-by-kernel durations of the search at a small batch: scripts/tail_prof.sh TAG N [VAR=VAL ...]
TAG=$1; N=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/tp_$TAG --output-format csv -- python3 $ROOT/scripts/small_batch_loop.py $N 10 > $OUT/${TAG}_out.txt 2>/dev/null
F=$(find /tmp/tp_$TAG -name '*kernel_stats.csv' | head -1)
cp $F $OUT/${TAG}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
print("$TAG", open("$OUT/${TAG}_out.txt").read().strip())
for r in rows[:7]:
    print("   ", r['Name'][:50].ljust(50), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:8.2f} us", f"{float(r['Percentage']):6.2f}%")
PY
