#!/bin/bash
# SQ counters of k_expand (per launch averages) at 8 and 1024 games: where a wave's cycles go
cd /tmp && export TMPDIR=/tmp
for g in 8 1024; do
  for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVES"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --kernel-trace --pmc $set -d /tmp/ep_${g}_$tag -o o --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --pipeline 0 --games $g --max-steps 2 > /dev/null 2>&1
    python3 - <<PY
import csv,glob,collections
fs=glob.glob("/tmp/ep_${g}_$tag/**/*counter_collection.csv",recursive=True)
if not fs: print("games $g: no counter file for $set")
else:
    acc=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "k_expand" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print("games $g:", {k: round(v/n[k],1) for k,v in acc.items()}, "launches", max(n.values()) if n else 0)
PY
  done
done
