#!/bin/bash
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats -d /tmp/pp --output-format csv -- python3 $ROOT/scripts/pair_time.py > /dev/null 2>&1
F=$(find /tmp/pp -name '*kernel_stats.csv' | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$F")))[:8]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:9.1f} us", f"min {float(r['MinNs'])/1e3:8.1f}", f"max {float(r['MaxNs'])/1e3:8.1f}")
PY
