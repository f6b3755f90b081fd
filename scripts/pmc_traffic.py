"""Aggregate the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into
profiles/rNN_pmc_traffic.json: mean KB per dispatch and kernel, FETCH_SIZE doubled (gfx950 correction for wide
coalesced read streams, MI355X_MICROARCH.md HBM section), WRITE_SIZE as reported.

usage: python scripts/pmc_traffic.py FETCH.csv WRITE.csv BOARDS OUT.json ["command line that was profiled"]"""
import csv, json, re, sys
from collections import defaultdict

fetch_csv, write_csv, boards, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
cmd = sys.argv[5] if len(sys.argv) > 5 else "python3 bench.py --no-cpu-baseline --max-steps 1"


def means(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        a = acc[name]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items()}


f, w = means(fetch_csv, "FETCH_SIZE"), means(write_csv, "WRITE_SIZE")
doc = {"_about": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) of `{cmd}`. "
                 "Units: KB as reported; FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read stream, "
                 "MI355X_MICROARCH.md HBM section), WRITE_SIZE is exact.",
       "batch_boards": boards, "kernels": {}}
for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[0])):
    e = {}
    if k in f: e["FETCH_SIZE_KB_mean"], e["dispatches"] = f[k]
    if k in w: e["WRITE_SIZE_KB_mean"] = w[k][0]
    if k in f and k in w:
        e["hbm_side_bytes_per_launch_corrected"] = (2 * f[k][0] + w[k][0]) * 1024
    doc["kernels"][k] = e
json.dump(doc, open(out, "w"), indent=1)
for k, e in list(doc["kernels"].items())[:8]:
    print(k[:60].ljust(60), {a: round(b, 1) for a, b in e.items()})
