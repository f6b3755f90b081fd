#!/bin/bash
# MFMA-busy counters under the roofline (run on the GPU box from the repo root):
#   scripts/pmc_mfma.sh TAG  ->  gpurun_out/TAG_mfma_busy.json (1024 boards: k_tower16<4,8,3>), TAG_mfma_busy_32boards.json (k_tower_cl<1,8>)
# One rocprofv3 pass per batch size with --kernel-trace --pmc only (no other trace domain), the program directly after `--`:
#   SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE     (SQ slots: 8, GRBM: 2 -- one pass)
# and a second pass with the wave-cycle split (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16).
set -u
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
W=/tmp/pmc_$TAG; rm -rf "$W"; mkdir -p "$W"
for G in 1024 32; do
  SUF=""; [ $G = 32 ] && SUF="_32boards"
  CMD="python3 bench.py --no-cpu-baseline --pipeline 0 --hbm-only-steps 0 --max-steps 1 --games $G"
  # (32 boards: DIEE_SPEC_EVAL=0 = the launch-per-iteration search, every launch of k_tower_cl<1,8> at exactly 32 boards)
  export DIEE_SPEC_EVAL=1 DIEE_FREE_EVAL=1; [ $G = 32 ] && { export DIEE_SPEC_EVAL=0 DIEE_FREE_EVAL=0; CMD="DIEE_SPEC_EVAL=0 DIEE_FREE_EVAL=0 $CMD"; }
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$W/a_$G" --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --pipeline 0 --hbm-only-steps 0 --max-steps 1 --games $G > /dev/null 2>> "$OUT/${TAG}_pmc_err.log"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 -d "$W/b_$G" --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --pipeline 0 --hbm-only-steps 0 --max-steps 1 --games $G > /dev/null 2>> "$OUT/${TAG}_pmc_err.log"
  A=$(find "$W/a_$G" -name '*counter_collection.csv' | head -1)
  B=$(find "$W/b_$G" -name '*counter_collection.csv' | head -1)
  KA=$(find "$W/a_$G" -name '*kernel_trace.csv' | head -1)
  KB=$(find "$W/b_$G" -name '*kernel_trace.csv' | head -1)
  python3 "$ROOT/scripts/pmc_mfma.py" "$OUT/${TAG}_mfma_busy$SUF.json" $G "$CMD" "$A" "$KA" "$B" "$KB" > "$OUT/${TAG}_mfma_busy_summary$SUF.txt" 2>&1
  cat "$OUT/${TAG}_mfma_busy_summary$SUF.txt"
done
