#!/bin/bash
# BASELINE configs[3] (iterations=1600): L2 hit rate of the tree kernels (the PUCT walk reads SoA statistics from L2 / HBM;
# it is not staged in LDS, so this is the number that answers SURVEY 8(d) item 4) -- first move-step, 1024 games.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c4; : > "$OUT/config4_l2_hit.txt"; for IT in 100 1600; do
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/c4/it$IT --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline --max-steps 1 --pipeline 0 --hbm-only-steps 0 --iterations $IT > "$OUT/config4_line_it$IT.json" 2> "$OUT/config4_err_it$IT.log"
python3 - "$IT" "$(find /tmp/c4/it$IT -name '*counter_collection.csv' | head -1)" <<'PY' >> "$OUT/config4_l2_hit.txt"
import csv, re, sys
from collections import defaultdict
it, path = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(path)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -(kv[1].get("TCC_HIT_sum", 0) + kv[1].get("TCC_MISS_sum", 0)))[:6]:
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(f"iterations={it:>5s} {k[:40]:40s} dispatches {n[k]//2:6d}  TCC_HIT {h:.3e}  TCC_MISS {m:.3e}  L2 hit rate {h/max(h+m,1):.4f}")
PY
done
cat "$OUT/config4_l2_hit.txt"
