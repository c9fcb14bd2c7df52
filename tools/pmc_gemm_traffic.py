"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes) over
`bench.py --steps 1 --warmup 1 --steps-only` into the per-launch HBM-side traffic of the
persistent GEMM kernel that bench.py replays as roofline.traffic.  usage: pmc_gemm_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import sys


def per_dispatch(d, counter, match):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"] and r["Counter_Name"] == counter:
            acc[r["Dispatch_Id"]] += float(r["Counter_Value"])      # summed over the counter's instances
    return acc


def per_kernel(d, counter, match):
    """{kernel name: (dispatches, MB per dispatch)} — per instantiation, to set beside that instantiation's algorithmic bytes"""
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc, ids = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"] and r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]] += float(r["Counter_Value"])
            ids[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (len(ids[k]), acc[k] / len(ids[k]) / 1e3) for k in acc}


fetch = per_dispatch(sys.argv[1], "FETCH_SIZE", "gemm_nt_persist_kernel")
write = per_dispatch(sys.argv[2], "WRITE_SIZE", "gemm_nt_persist_kernel")
nf, nw = len(fetch), len(write)
f_mb = sum(fetch.values()) / nf / 1e3        # counter unit: KB
w_mb = sum(write.values()) / nw / 1e3
out = {"kernel": "gemm_nt_persist_kernel (all instantiations of the profiled steps)", "dispatches_fetch_pass": nf,
       "dispatches_write_pass": nw, "fetch_size_raw_mb_per_launch": round(f_mb, 1), "write_size_mb_per_launch": round(w_mb, 1),
       "hbm_side_mb_per_launch": round(2 * f_mb + w_mb, 1),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1 --steps-only` "
               "(tools/prof_r03.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests of "
               "16-B-per-lane reads at 64 B), Infinity-Cache hits are counted"}
pf, pw = per_kernel(sys.argv[1], "FETCH_SIZE", "gemm_nt_persist_kernel"), per_kernel(sys.argv[2], "WRITE_SIZE", "gemm_nt_persist_kernel")
out["per_instantiation"] = {k: {"dispatches": pf[k][0], "fetch_size_raw_mb": round(pf[k][1], 1), "write_size_mb": round(pw.get(k, (0, 0.0))[1], 1)} for k in sorted(pf)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
