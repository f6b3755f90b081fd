#!/bin/bash
# Profiles of the default bench command for profiles/ (run on the GPU box from the repo root):
#   scripts/profile_bench.sh TAG    ->  gpurun_out/TAG_kernel_stats.csv, TAG_line.json, TAG_headline_kernel_stats.csv, TAG_headline_line.json, TAG_pmc_traffic*.json
# Pass 1: rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu-baseline` (whole run: headline batch + pipelined).
# Pass 2/3: --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) of the first move-step at 1024 and 32 boards.
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W=/tmp/prof_$TAG; rm -rf "$W"; mkdir -p "$W"
rocprofv3 --kernel-trace --stats -d "$W/full" --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/${TAG}_line.json" 2> "$OUT/${TAG}_err.log"
cp "$(find "$W/full" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_kernel_stats.csv"
# Pass 1b: the timed leg alone (one batch at a time, records delivered; no HBM-only repeat): k_tower16<4,8,3>'s AverageNs here is what roofline.avg_launch_us must agree with
rocprofv3 --kernel-trace --stats -d "$W/headline" --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --pipeline 0 --hbm-only-steps 0 > "$OUT/${TAG}_headline_line.json" 2>> "$OUT/${TAG}_err.log"
cp "$(find "$W/headline" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_headline_kernel_stats.csv"
for G in 1024 32; do
  for C in FETCH_SIZE WRITE_SIZE; do
    # (32 boards: the launch-per-iteration search, so that k_tower_cl<1,8> runs at exactly 32 boards in every launch -- with the
    # free-running search / the tail on, a 32-game search sends 128-row launches of k_tower_cl<4,8> whose row count varies)
    SPEC=1; [ $G = 32 ] && SPEC=0
    DIEE_SPEC_EVAL=$SPEC DIEE_FREE_EVAL=$SPEC rocprofv3 --kernel-trace --pmc $C -d "$W/pmc_${C}_$G" --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --max-steps 1 --pipeline 0 --hbm-only-steps 0 --games $G > /dev/null 2>> "$OUT/${TAG}_err.log"
  done
  F=$(find "$W/pmc_FETCH_SIZE_$G" -name '*counter_collection.csv' | head -1)
  Wr=$(find "$W/pmc_WRITE_SIZE_$G" -name '*counter_collection.csv' | head -1)
  SUF=""; [ $G = 32 ] && SUF="_32boards"
  python3 "$ROOT/scripts/pmc_traffic.py" "$F" "$Wr" $G "$OUT/${TAG}_pmc_traffic$SUF.json" "python3 bench.py --no-cpu-baseline --max-steps 1 --pipeline 0 --hbm-only-steps 0 --games $G" > "$OUT/${TAG}_pmc_summary$SUF.txt" 2>&1
done
ls -la "$OUT" | grep "$TAG"
