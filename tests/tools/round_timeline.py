#!/usr/bin/env python3
"""Development tool: the timeline of the free-running search's rounds from a rocprofv3 --kernel-trace CSV of tests/tools/free_prof.py:
per kernel of a round its mean duration and the mean idle time in front of it (the boundary between two dependent kernels of one stream).

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tests/tools/free_prof.py 600 4 1
    python tests/tools/round_timeline.py gpurun_out/tl
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(name):
    m = re.search(r"diee::([A-Za-z_0-9]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


# the last search of the run: from the last k_init_roots on
starts = [i for i, r in enumerate(rows) if "k_init_roots" in r[2]]
seg = rows[starts[-1]:] if starts else rows
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for prev, cur in zip(seg, seg[1:]):
    k = short(cur[2])
    dur[k] += (cur[1] - cur[0]) / 1e3; gap[k] += max(0, cur[0] - prev[1]) / 1e3; cnt[k] += 1
total = (seg[-1][1] - seg[0][0]) / 1e3
print(f"last search: {len(seg)} kernels, {total:.0f} us from the first start to the last end")
print(f"{'kernel':38s} {'launches':>8s} {'mean us':>9s} {'idle before, mean us':>21s} {'share of the span':>18s}")
for k in sorted(cnt, key=lambda k: -(dur[k] + gap[k])):
    print(f"{k:38s} {cnt[k]:8d} {dur[k] / cnt[k]:9.1f} {gap[k] / cnt[k]:21.1f} {100 * (dur[k] + gap[k]) / total:17.1f}%")
print(f"idle between kernels in all: {sum(gap.values()):.0f} us = {100 * sum(gap.values()) / total:.1f} % of the span")
