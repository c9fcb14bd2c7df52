"""Idle time BETWEEN the kernels of one benched step, from a rocprofv3 --kernel-trace CSV (start / end timestamps per dispatch):
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 3 --warmup 1 --steps-only
    python3 tools/kernel_gaps.py <dir>
A step = the dispatches between two consecutive adamw_kernel launches (the optimiser closes a step).  Prints the span of the last full step, the sum of
its kernel durations, the sum of the gaps (span - busy; overlapping dispatches counted once) and where the gaps are (by the kernel that FOLLOWS the gap)."""
import collections
import csv
import glob
import os
import sys


def main(d):
    f = [p for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)]
    assert f, f"no *kernel_trace.csv under {d}"
    rows = []
    with open(f[0]) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    # the last adamw of a step: an index whose successor is not an adamw launch within the same step (two ranges -> two launches per step)
    ends = [i for k, i in enumerate(marks) if k + 1 == len(marks) or marks[k + 1] != i + 1]
    assert len(ends) >= 2, "need two optimiser steps in the trace"
    a, b = ends[-2] + 1, ends[-1] + 1
    step = rows[a:b]
    span = step[-1][1] - step[0][0]
    busy, cur_end, gaps = 0, step[0][0], collections.defaultdict(lambda: [0, 0])
    hist = collections.Counter()
    for s, e, n in step:
        if s > cur_end:
            g = s - cur_end
            key = n.split("(")[0][:60]
            gaps[key][0] += 1
            gaps[key][1] += g
            hist[min(g // 1000, 20)] += 1
            busy += e - s
            cur_end = e
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
    tot_gap = span - busy
    print(f"step: {len(step)} dispatches, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle between kernels {tot_gap / 1e6:.3f} ms ({100.0 * tot_gap / span:.1f} %), "
          f"sum of kernel durations {sum(e - s for s, e, _ in step) / 1e6:.3f} ms")
    print("gap histogram (us -> count):", dict(sorted(hist.items())))
    for k, (n, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{g / 1e3:9.1f} us in {n:4d} gaps before {k}")


if __name__ == "__main__":
    main(sys.argv[1])
