"""Dense cost-volume KL forward + backward at the benched size (P = 32, 37 x 37 grid, D = 768), every row kept: the kernels of one
forward + backward for `rocprofv3 --kernel-trace --stats -- python3 tools/probe_cv_bwd.py [bf16|h]`, and the wall time per pair."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gd_amd  # noqa: E402,F401
from gd_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
P, hw, D = 32, 1369, 768
torch.manual_seed(0)
Tt = torch.bfloat16 if mode == "bf16" else torch.float32
f1 = torch.randn(P, hw, D, device=dev).to(Tt).requires_grad_(True)
f2 = torch.randn(P, hw, D, device=dev).to(Tt).requires_grad_(True)
t1 = ops.pad_teacher_maps(torch.softmax(torch.randn(P, hw, hw, device=dev) * 3, -1))
t2 = ops.pad_teacher_maps(torch.softmax(torch.randn(P, hw, hw, device=dev) * 3, -1))
ts = ops.cost_volume_teacher_stats(t1, t2)
ones = torch.ones(P, hw, dtype=torch.bool, device=dev)
inv = (1.0 / f1.detach().float().norm(dim=-1).clamp_min(1e-12), 1.0 / f2.detach().float().norm(dim=-1).clamp_min(1e-12))
h16 = (f1.detach().half(), f2.detach().half()) if mode == "h" else None


def fb():
    f1.grad = f2.grad = None
    ops.cost_volume_kl(f1, f2, t1, t2, ones, ones, "mast3r", tstats=ts, inv_norms=inv, x3="h" if mode == "h" else "", h16=h16).sum().backward()


for _ in range(3):
    fb()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    fb()
torch.cuda.synchronize()
print(f"{mode}: dense fwd+bwd {(time.perf_counter() - t0) / reps / P * 1e6:.2f} us/pair")
