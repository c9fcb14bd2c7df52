"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, MI355X_MICROARCH.md "HBM") over
`tools/bench_kernels.py pmc_cv_kp | pmc_cv_full` into the HBM-side bytes per launch of the cost-volume KL forward, the file bench.py
replays as roofline_cost_volume.traffic.  usage: pmc_cv_traffic.py <fetch_dir> <write_dir> <out.json> <kp|full> <kept1> <kept2>

Width correction (MI355X_MICROARCH.md: FETCH_SIZE = 64 B x the L2's fabric read requests; a 128-byte request is tallied as 64 B, other
shapes are "uncalibrated: calibrate on a known byte count in your own access pattern").  cv_fwd_persist has three read streams:
features by LDS-DMA (8 rows x 128 B per instruction) and the direction-1 teacher rows (4 rows x 256 B per instruction) are whole
128-byte requests -> counted at 1/2; the direction-2 teacher entries are 16 rows x 64 B per instruction -> 64-byte requests, counted
1 : 1.  The two mask settings calibrate that: with (1/2, 1/2, 1) the measured counter is 0.96 x (kept 20 %) and 0.975 x (all rows) of
the bytes the kernel must read; with 1/2 for all three it would be 1.16 x and 1.36 x — one kernel, two different over-fetch ratios.
So: expected_raw = features / 2 + T1_kept / 2 + T2_kept, over_fetch = measured_raw / expected_raw, traffic = over_fetch x needed
+ the small kernels (dword reads 1 : 1) + WRITE_SIZE (exact).  Counter unit: KB."""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "cv_" not in k or "cv_tstats" in k or r["Counter_Name"] != counter:      # cv_tstats: once per cached pair, not per launch
            continue
        k = k.split("(")[0].split("<")[0].replace("void ", "").strip()
        acc[k] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return {k: acc[k] / len(n[k]) / 1e3 for k in acc}        # MB per dispatch


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
tag, kept = sys.argv[4], (int(sys.argv[5]), int(sys.argv[6]))
P, hw, C = 32, 1369, 768
if tag == "rows":
    # round 4: the kept-row forward (cv_fwd_rows_kernel<f16>: both directions as compacted row problems).  Its read streams: the other view's
    # feature rows and the gathered kept rows by LDS-DMA (8 rows x 128 B per instruction: 128-byte requests, FETCH_SIZE tallies 1/2) and the
    # kept rows' teacher rows (row-contiguous, 256 B per row and instruction: 128-byte requests, 1/2): expected_raw = needed / 2 for the tile
    # kernel; the index-list / finalize / loss kernels read dwords (1 : 1).
    feat = P * 2 * hw * C * 2 + sum(kept) * C * 2                  # every row of the other view per direction + the kept rows of the own view
    teach = sum(kept) * hw * 4
    needed = feat + teach + P * 2 * hw
    raw_tile = sum(mb for k, mb in fetch.items() if "fwd_rows" in k)
    expected_raw = (feat + teach) / 2e6
    over = raw_tile / expected_raw
    moved = over * (feat + teach) / 1e6 + sum(mb for k, mb in fetch.items() if "fwd_rows" not in k) + sum(write.values())
    out = {"what": "HBM-side traffic of the KEPT-ROW cost-volume KL forward per launch (keypoint-patch row masks; 32 pairs, hw = 1369, C = 768, fp16 feature "
                   "copies, teacher maps [P, hw, 1376] with cached row statistics, row norms from the producer): rocprofv3 --pmc FETCH_SIZE and --pmc "
                   "WRITE_SIZE in separate passes on tools/bench_kernels.py pmc_cv_rows",
           "kept_rows": kept, "raw_fetch_MB": {k: round(v, 2) for k, v in fetch.items()}, "write_MB": {k: round(v, 2) for k, v in write.items()},
           "correction": "cv_fwd_rows: features (LDS-DMA) and kept teacher rows (256 contiguous bytes per row and instruction) are 128-byte requests, "
                         "FETCH_SIZE counts 1/2 (MI355X_MICROARCH.md); small kernels 1 : 1",
           "cv_fwd_rows_expected_raw_MB": round(expected_raw, 2), "cv_fwd_rows_over_fetch": round(over, 3),
           "fwd_hbm_bytes_per_launch": int(moved * 1e6), "needed_bytes_per_launch": needed, "ratio_to_needed": round(moved * 1e6 / needed, 3)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out))
    sys.exit(0)
needed = P * (2 * hw * C * 2 + 2 * hw) + sum(kept) * hw * 4
feat_mb = P * 2 * hw * C * 2 / 1e6
t1_mb, t2_mb = kept[0] * hw * 4 / 1e6, kept[1] * hw * 4 / 1e6
raw_persist = sum(mb for k, mb in fetch.items() if "persist" in k)
expected_raw = feat_mb / 2 + t1_mb / 2 + t2_mb
over = raw_persist / expected_raw
moved = over * (feat_mb + t1_mb + t2_mb) + sum(mb for k, mb in fetch.items() if "persist" not in k) + sum(write.values())
out = {"what": f"HBM-side traffic of the cost-volume KL forward per launch ({tag} row masks; 32 pairs, hw = 1369, C = 768, bf16 features, teacher maps "
               "[P, hw, 1376] with cached row statistics, feature row norms from the producer): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
               "separate passes on tools/bench_kernels.py pmc_cv_" + tag,
       "kept_rows": kept, "raw_fetch_MB": {k: round(v, 2) for k, v in fetch.items()}, "write_MB": {k: round(v, 2) for k, v in write.items()},
       "correction": "cv_fwd_persist: features and direction-1 teacher rows are 128-byte requests (FETCH_SIZE counts 1/2), direction-2 teacher entries "
                     "64-byte requests (1 : 1) — see the docstring of tools/pmc_cv_traffic.py; small kernels 1 : 1",
       "cv_fwd_persist_expected_raw_MB": round(expected_raw, 2), "cv_fwd_persist_over_fetch": round(over, 3),
       "fwd_hbm_bytes_per_launch": int(moved * 1e6), "needed_bytes_per_launch": needed, "ratio_to_needed": round(moved * 1e6 / needed, 3)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
