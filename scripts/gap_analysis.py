"""GPU idle time between the kernels of the search loop, from a rocprofv3 kernel trace CSV (argv[1]): per-kernel busy time,
gap before each kernel name, and the busy fraction of the traced window"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
busy = collections.Counter(); gap = collections.Counter(); cnt = collections.Counter()
prev_end = None; t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
for r in rows:
    n = r["Kernel_Name"]; n = n[:n.index("(")] if "(" in n else n
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy[n] += e - s; cnt[n] += 1
    if prev_end is not None and s > prev_end: gap[n] += s - prev_end
    prev_end = max(prev_end or 0, e)
tot_busy = sum(busy.values()); tot_gap = sum(gap.values())
print(f"window {(t1 - t0) / 1e6:.1f} ms, busy {tot_busy / 1e6:.1f} ms, gaps {tot_gap / 1e6:.1f} ms ({100 * tot_gap / (t1 - t0):.1f} %)")
for n, b in busy.most_common(10):
    print(f"  {n[-50:]:50s} calls {cnt[n]:6d}  avg {b / cnt[n] / 1e3:7.1f} us  gap before: avg {gap[n] / cnt[n] / 1e3:5.2f} us")
