#!/bin/bash
# the tail of a batch: delivered games/s and launches per iteration against the number of virtual descents per game and launch
# (option spec_rollout_steps, here through its development override in the environment); run on the GPU box from the repo root
OUT=${1:-gpurun_out/tail_sweep.txt}
: > "$OUT"
for S in ${STEPS_LIST:-0 8 16 24 48}; do
  DIEE_SPEC_ROLLOUT_STEPS=$S python3 bench.py --no-cpu-baseline --pipeline 0 --hbm-only-steps 0 --steps ${BENCH_STEPS:-2} > /tmp/tail_sweep_line.json 2>/dev/null
  python3 - "$S" >> "$OUT" <<'PY'
import json, sys
d = json.load(open("/tmp/tail_sweep_line.json"))
t = d["stats"]["tail"]
print(f"spec_rollout_steps {sys.argv[1]:>3}: value {d['value']:.2f} games/s, ms_per_step {d['ms_per_step']:.0f}, tail iterations/step {t['iterations_per_step']:.0f}, "
      f"launches/step {t['launches_per_step']:.0f} ({t['launches_per_iteration']:.3f} per iteration), speculative rows/step {t['speculative_rows_per_step']:.0f}")
PY
done
cat "$OUT"
