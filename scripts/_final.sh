#!/bin/bash
set -u
TAG=r05P
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > "$OUT/${TAG}_gpu_suite.log" 2>&1; echo suite rc $?; tail -3 "$OUT/${TAG}_gpu_suite.log"
cd /tmp && export TMPDIR=/tmp
W=/tmp/prof_$TAG; rm -rf "$W"; mkdir -p "$W"
rocprofv3 --kernel-trace --stats -d "$W/headline" --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --pipeline 0 --hbm-only-steps 0 > "$OUT/${TAG}_headline_line.json" 2> "$OUT/${TAG}_err.log"
cp "$(find "$W/headline" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_headline_kernel_stats.csv"
cd "$ROOT"
python3 bench.py --steps 5 --warmup 1 > "$OUT/${TAG}_bench_line_steps5.json" 2>> "$OUT/${TAG}_err.log"
python3 -c "
import json
for f in ('headline_line','bench_line_steps5'):
    d=json.load(open('gpurun_out/r05P_'+f+'.json')); print(f, d['value'], d.get('value_pipelined'), d['roofline']['frac'], d['roofline']['end_to_end_frac'], d['stats']['tail'])"
