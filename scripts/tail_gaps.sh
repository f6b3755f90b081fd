#!/bin/bash
# Where a search iteration of the batch tail goes: kernel durations AND the idle time between consecutive kernels, from a kernel trace
# of scripts/small_batch_loop.py at N live games:   scripts/tail_gaps.sh TAG N [N ...]   -> gpurun_out/TAG_tail_gaps.txt
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_tail_gaps.txt
for N in "$@"; do
  rm -rf /tmp/tg_${TAG}_$N
  rocprofv3 --kernel-trace -d /tmp/tg_${TAG}_$N --output-format csv -- python3 $ROOT/scripts/small_batch_loop.py $N 10 > /tmp/tg_${TAG}_$N.out 2>/dev/null
  F=$(find /tmp/tg_${TAG}_$N -name '*kernel_trace.csv' | head -1)
  echo "== n=$N  $(cat /tmp/tg_${TAG}_$N.out | tail -1)" >> $OUT/${TAG}_tail_gaps.txt
  python3 $ROOT/scripts/gap_analysis.py "$F" >> $OUT/${TAG}_tail_gaps.txt
done
cat $OUT/${TAG}_tail_gaps.txt
