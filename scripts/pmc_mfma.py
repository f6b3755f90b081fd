"""Aggregate rocprofv3 --kernel-trace --pmc passes into profiles/rNN_mfma_busy*.json: per kernel, mean counter values per
dispatch, mean duration, and the derived figures

    clock_mhz      = GRBM_GUI_ACTIVE / 8 / duration          (the counter sums the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back)
    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (CUs x 4 SIMDs x GRBM_GUI_ACTIVE / 8)
                     = share of the chip's SIMD-cycles in which a matrix instruction was executing
    mfma_per_wave  = SQ_VALU_MFMA_BUSY_CYCLES / 16 / waves   (v_mfma_f32_16x16x32_bf16 holds the pipe 16 cycles; 32x32x16: 32)

usage: pmc_mfma.py OUT.json BOARDS "command" COUNTERS_A.csv KTRACE_A.csv [COUNTERS_B.csv KTRACE_B.csv]"""
import csv, json, re, sys
from collections import defaultdict

out, boards, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3]
passes = [(sys.argv[i], sys.argv[i + 1]) for i in range(4, len(sys.argv) - 1, 2)]
CUS = 256


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


doc = {"_about": f"rocprofv3 --kernel-trace --pmc <counters> (one pass per counter set, no other trace domain) of `{cmd}`; "
                 "means per dispatch. SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (16 per v_mfma_f32_16x16x32_bf16, "
                 "32 per 32x32x16); GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles.",
       "batch_boards": boards, "kernels": {}}
for cpath, kpath in passes:
    dur = {}
    try:
        for r in csv.DictReader(open(kpath)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    except Exception as e:
        print("no kernel trace:", e)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dsum = defaultdict(lambda: [0.0, 0])
    seen = set()
    meta = {}
    for r in csv.DictReader(open(cpath)):
        k = short(r["Kernel_Name"])
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        meta[k] = {"grid_threads": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "vgpr": int(r.get("VGPR_Count", 0) or 0),
                   "agpr": int(r.get("Accum_VGPR_Count", 0) or 0), "lds": int(r.get("LDS_Block_Size", 0) or 0), "scratch": int(r.get("Scratch_Size", 0) or 0)}
        d = r["Dispatch_Id"]
        if (k, d) not in seen and d in dur:
            seen.add((k, d)); dsum[k][0] += dur[d]; dsum[k][1] += 1
    for k, cs in acc.items():
        e = doc["kernels"].setdefault(k, {})
        e.update(meta[k])
        for c, (s, n) in cs.items():
            e[c] = s / n; e["dispatches"] = n
        if dsum[k][1]:
            e.setdefault("duration_us", {})[",".join(sorted(cs))] = dsum[k][0] / dsum[k][1] * 1e6
for k, e in doc["kernels"].items():
    durs = e.get("duration_us", {})
    d_a = next((v for kk, v in durs.items() if "GRBM_GUI_ACTIVE" in kk), None)
    if "GRBM_GUI_ACTIVE" in e and d_a:
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        e["kernel_cycles"] = cyc
        e["clock_mhz"] = cyc / d_a
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (CUS * 4 * cyc)
            waves = e["grid_threads"] / 64
            e["mfma_busy_cycles_per_wave"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / waves
    if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in e:
                e[c + "_share_of_wave_cycles"] = e[c] / e["SQ_WAVE_CYCLES"]
json.dump(doc, open(out, "w"), indent=1)
rank = sorted(doc["kernels"].items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))
for k, e in rank[:8]:
    print(k[:48].ljust(48), {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items() if a in
          ("dispatches", "SQ_VALU_MFMA_BUSY_CYCLES", "kernel_cycles", "clock_mhz", "mfma_busy_frac", "mfma_busy_cycles_per_wave", "duration_us")})
