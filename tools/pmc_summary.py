import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    if "attn" not in k and "gemm" not in k and "cv_" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, d in acc.items():
    nd = len(n[k]); print(k, "dispatches", nd)
    for c, v in sorted(d.items()): print(f"   {c:32s} {v/nd:16.1f}")
