#!/bin/bash
# the tail of a batch: delivered games/s against how far up the path reaches (spec_max_games) and from how many live games a launch
# carries 64 / 128 rows; run on the GPU box from the repo root (development overrides through the environment, read at diee_create)
OUT=${1:-gpurun_out/tail_rows_sweep.txt}
: > "$OUT"
run() {   # label, env...
  local label=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --pipeline 0 --hbm-only-steps 0 --steps ${BENCH_STEPS:-2} > /tmp/tail_rows_line.json 2>/dev/null
  python3 - "$label" >> "$OUT" <<'PY'
import json, sys
d = json.load(open("/tmp/tail_rows_line.json"))
t = d["stats"]["tail"]
print(f"{sys.argv[1]:44s}: value {d['value']:.2f} games/s, ms_per_step {d['ms_per_step']:.0f}, tail iterations/step {t['iterations_per_step']:.0f}, "
      f"launches/step {t['launches_per_step']:.0f} ({t['launches_per_iteration']:.3f} per iteration), speculative rows/step {t['speculative_rows_per_step']:.0f}")
PY
}
run "max 16, 32 rows (round 5 first build)"   DIEE_SPEC_MAX_GAMES=16 DIEE_SPEC_ROWS64_FROM=65 DIEE_SPEC_ROWS128_FROM=65
run "max 16, 64 rows from 5, 128 from 10"     DIEE_SPEC_MAX_GAMES=16
run "max 32, 64 rows from 5, 128 from 10"     DIEE_SPEC_MAX_GAMES=32
run "max 64, 64 rows from 5, 128 from 10 (default)" DIEE_SPEC_MAX_GAMES=64
run "max 64, 64 rows from 9, 128 from 17"     DIEE_SPEC_MAX_GAMES=64 DIEE_SPEC_ROWS64_FROM=9 DIEE_SPEC_ROWS128_FROM=17
run "max 64, 64 rows from 3, 128 from 6"      DIEE_SPEC_MAX_GAMES=64 DIEE_SPEC_ROWS64_FROM=3 DIEE_SPEC_ROWS128_FROM=6
run "max 64, 64 rows from 17, 128 from 33"    DIEE_SPEC_MAX_GAMES=64 DIEE_SPEC_ROWS64_FROM=17 DIEE_SPEC_ROWS128_FROM=33
run "max 48"                                  DIEE_SPEC_MAX_GAMES=48
cat "$OUT"
