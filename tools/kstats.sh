#!/bin/bash
# usage (on the GPU box): tools/kstats.sh <tag>   -> gpurun_out/<tag>.txt : per-step kernel-time totals from a rocprofv3 kernel trace
set -e
tag=${1:-ks}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > gpurun_out/$tag.log 2>&1
python3 - gpurun_out/$tag <<'PY' | tee gpurun_out/$tag.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)/4e6
grp={'gemm_nt':0,'attn':0,'gemm_tn':0,'ln_':0,'cv_':0,'pair_rank':0,'at::native':0,'kp_gather':0}
for r in rows:
    t=float(r['TotalDurationNs'])/4e6
    for k in grp:
        if k in r['Name']: grp[k]+=t; break
print(f"total {tot:.2f} ms/step | "+" ".join(f"{k} {v:.2f}" for k,v in grp.items()))
PY
