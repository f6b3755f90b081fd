#!/bin/bash
cd /tmp && export TMPDIR=/tmp
g=${1:-8}
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM" "SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_FLAT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set -d /tmp/eq_${g}_$tag -o o --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --pipeline 0 --games $g --max-steps 2 > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/eq_${g}_$tag/**/*counter_collection.csv",recursive=True)
if not fs: print("games $g: no counter file for $set")
else:
    acc=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "k_expand" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("games $g:", {k: round(v/n[k]/$g,1) for k,v in acc.items()}, "(per wave)")
PY
done
