"""Summarise the counter-only rocprofv3 passes of tools/pmc_attn_r05.sh: per attention kernel, every counter averaged per dispatch, plus the derived
shares the round-4 review asked for — where a wave's cycles go (issuing / stalled at issue / parked on s_waitcnt or a barrier), how much of the issue
stall is LDS, LDS bank-conflict cycles as a share of LDS-active cycles, MFMA-pipe busy share, clock.
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; SQ_VALU_MFMA_BUSY_CYCLES counts cycles;
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  usage: pmc_attn_summary.py <dir with one sub-directory per pass> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

d, out = sys.argv[1], sys.argv[2]
FILTER = sys.argv[3] if len(sys.argv) > 3 else "attn"        # substring of the kernel names to summarise (round 5: "cv_fwd" for the cost-volume forward)
NOTE = sys.argv[4] if len(sys.argv) > 4 else ("rocprofv3 --pmc passes (counters only + --kernel-trace) over tools/bench_kernels.py pmc_attn, GD_PMC_F16=1: 64 x 12 x 1370, "
                                              "fp16 operands")
val = collections.defaultdict(lambda: collections.defaultdict(float))      # kernel -> counter -> sum over dispatches
cnt = collections.defaultdict(lambda: collections.defaultdict(set))        # kernel -> counter -> dispatch ids
dur = collections.defaultdict(list)
for p in sorted(glob.glob(os.path.join(d, "*"))):
    if not os.path.isdir(p):
        continue
    for cc in glob.glob(p + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(cc)):
            k = r["Kernel_Name"]
            if FILTER not in k:
                continue
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    if os.path.basename(p) == "a":
        for kt in glob.glob(p + "/*/*kernel_trace.csv"):
            for r in csv.DictReader(open(kt)):
                if FILTER in r["Kernel_Name"]:
                    dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
res = {}
for k, cs in val.items():
    c = {name: v / max(1, len(cnt[k][name])) for name, v in cs.items()}
    rec = {"counters_per_dispatch": {n: round(v, 1) for n, v in sorted(c.items())}}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        sh = {}
        for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MFMA",
                  "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_VALU", "SQ_WAIT_IFETCH"):
            if n in c:
                sh[n + "/SQ_WAVE_CYCLES"] = round(c[n] / wc, 4)
        rec["share_of_wave_cycles"] = sh
    if c.get("SQ_LDS_IDX_ACTIVE"):
        rec["lds_bank_conflict_share_of_lds_active"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("GRBM_GUI_ACTIVE"):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        rec["kernel_cycles"] = round(cyc, 0)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            rec["mfma_busy_share_of_simd_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in c:
            rec["valu_mfma_coexec_share_of_simd_cycles"] = round(c["SQ_VALU_MFMA_COEXEC_CYCLES"] / (1024.0 * cyc), 4)
        if wc:
            rec["waves_resident_per_simd_avg"] = round(4.0 * wc / (1024.0 * cyc), 3)      # quad-cycles x 4 / SIMD cycles
        if dur.get(k):
            rec["avg_us_under_counters"] = round(sum(dur[k]) / len(dur[k]) / 1e3, 1)
            rec["clock_ghz"] = round(cyc / (sum(dur[k]) / len(dur[k])), 3)
    res[k[:100]] = rec
failed = []
fp = os.path.join(d, "failed.txt")
if os.path.exists(fp):
    failed = [l.strip() for l in open(fp) if l.strip()]
json.dump({"note": NOTE + "; values are averages per dispatch; counter collection serialises dispatches: durations here are not the step's",
           "unavailable": failed, "kernels": res}, open(out, "w"), indent=1)
for k, r in res.items():
    print(k[:70], json.dumps({x: r[x] for x in r if x != "counters_per_dispatch"}))
