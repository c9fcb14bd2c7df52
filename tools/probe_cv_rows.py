"""Kept-row cost-volume KL forward + backward as the MASt3R trainer runs it (P = 32, 37 x 37, D = 768, keypoint-patch masks keeping <= 300 rows per view):
per-kernel times under `rocprofv3 --kernel-trace -- python3 tools/probe_cv_rows.py [h|bf16]` + tools/rocpd_stats.py, and the wall time per pair."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gd_amd  # noqa: E402,F401
from gd_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "h"
dev = torch.device("cuda:0")
P, hw, D, NK = 32, 1369, 768, 300
torch.manual_seed(0)
Tt = torch.bfloat16 if mode == "bf16" else torch.float32
f1 = torch.randn(P, hw, D, device=dev).to(Tt).requires_grad_(True)
f2 = torch.randn(P, hw, D, device=dev).to(Tt).requires_grad_(True)
t1 = ops.pad_teacher_maps(torch.softmax(torch.randn(P, hw, hw, device=dev) * 3, -1))
t2 = ops.pad_teacher_maps(torch.softmax(torch.randn(P, hw, hw, device=dev) * 3, -1))
ts = ops.cost_volume_teacher_stats(t1, t2)
m1 = torch.zeros(P, hw, dtype=torch.bool, device=dev)
m2 = torch.zeros(P, hw, dtype=torch.bool, device=dev)
for p in range(P):
    m1[p, torch.randint(0, hw, (NK,), device=dev)] = True
    m2[p, torch.randint(0, hw, (NK,), device=dev)] = True
inv = (1.0 / f1.detach().float().norm(dim=-1).clamp_min(1e-12), 1.0 / f2.detach().float().norm(dim=-1).clamp_min(1e-12))
h16 = (f1.detach().half(), f2.detach().half()) if mode == "h" else None


def fb():
    f1.grad = f2.grad = None
    ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "mast3r", tstats=ts, inv_norms=inv, x3="h" if mode == "h" else "", h16=h16, kept_rows_max=NK).sum().backward()


for _ in range(3):
    fb()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    fb()
torch.cuda.synchronize()
print(f"{mode}: kept-row fwd+bwd {(time.perf_counter() - t0) / 10 / P * 1e6:.2f} us/pair, kept rows per view {float(m1.sum()) / P:.0f}")
