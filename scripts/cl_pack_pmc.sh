#!/bin/bash
# FETCH_SIZE of the cluster tower with its clusters packed on as few XCDs as hold them (DIEE_CL_PACK=1) or one per XCD (0): scripts/cl_pack_pmc.sh N
N=${1:-4}
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
for P in 1 0; do
  export DIEE_CL_PACK=$P
  rm -rf /tmp/clp_$P
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/clp_$P --output-format csv -- python3 $ROOT/scripts/small_batch_loop.py $N 2 > /dev/null 2>&1
  F=$(find /tmp/clp_$P -name '*counter_collection.csv' | head -1)
  python3 - <<PY
import csv, collections
tot = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open("$F")):
    if r["Counter_Name"] == "FETCH_SIZE":
        k = r["Kernel_Name"][:34]; tot[k][0] += 1; tot[k][1] += float(r["Counter_Value"])
for k, (n, v) in tot.items():
    if "tower" in k: print(f"n=$N DIEE_CL_PACK=$P  {k:36s} {n:5d} dispatches  FETCH_SIZE {v / n / 1024:8.1f} MB per launch (x2 on gfx950: {2 * v / n / 1024:8.1f} MB)")
PY
done
