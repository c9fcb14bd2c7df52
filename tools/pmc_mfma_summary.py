"""Summarise one rocprofv3 pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace) over one step of bench.py into,
per kernel name: launches, average duration, the clock the chip held (GRBM_GUI_ACTIVE / 8 XCDs / duration: MI355X_MICROARCH.md, DVFS give-back)
and the share of SIMD cycles in which the MFMA pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles)).
usage: pmc_mfma_summary.py <rocprof dir> <out.json>"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
val = collections.defaultdict(lambda: collections.defaultdict(float))
name = {}
for r in csv.DictReader(open(cc)):
    val[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    name[r["Dispatch_Id"]] = r["Kernel_Name"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for did, v in val.items():
    k = name[did]
    a = agg[k]
    a["n"] += 1
    for c, x in v.items():
        a[c] += x
    if did in dur:
        a["ns"] += dur[did][1]
out = {}
for k, a in agg.items():
    n = a["n"]
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0                    # summed over the 8 XCDs
    rec = {"launches": int(n), "avg_us": round(a["ns"] / n / 1e3, 1) if a["ns"] else None,
           "clock_ghz": round(cyc / a["ns"], 3) if a["ns"] else None,
           "mfma_busy_frac_of_simd_cycles": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4) if cyc else None,
           "cu_busy_frac": round(a["SQ_BUSY_CU_CYCLES"] / (256.0 * cyc), 4) if cyc else None,
           "total_ms": round(a["ns"] / 1e6, 3)}
    out[k] = rec
top = dict(sorted(out.items(), key=lambda kv: -(kv[1]["total_ms"] or 0))[:24])
json.dump({"note": "one rocprofv3 pass: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace over bench.py --steps 1 --warmup 1 --steps-only "
                   "(2 steps traced); clock = GRBM_GUI_ACTIVE / 8 / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles); counter collection "
                   "serialises and lengthens dispatches, durations here are NOT the step's", "kernels": top}, open(sys.argv[2], "w"), indent=1)
for k, r in top.items():
    print(f"{k[:90]:90s} x{r['launches']:4d} {r['avg_us']} us  clock {r['clock_ghz']} GHz  MFMA busy {r['mfma_busy_frac_of_simd_cycles']}  CU busy {r['cu_busy_frac']}")
