"""Plain streaming-read bandwidth of this box for reference: torch reductions / copies over the benched teacher maps' size (2 x 241 MB fp32)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gd_amd  # noqa: E402,F401
from gd_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(2, 32, 1369, 1376, device=dev)
y = torch.empty_like(x)
nb = x.numel() * 4


def t(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, f, b in (("sum (read only)", lambda: x.sum(), nb), ("max (read only)", lambda: x.amax(), nb), ("copy (read + write)", lambda: y.copy_(x), 2 * nb),
                   ("abs-sum via linalg (read only)", lambda: torch.linalg.vector_norm(x, 1), nb), ("row sums dim=-1", lambda: x.sum(-1), nb)):
    dt = t(f)
    print(f"{name:32s} {dt * 1e6:8.1f} us  {b / dt / 1e12:5.2f} TB/s")
ts = lambda: ops.cost_volume_teacher_stats(x[0], x[1])
dt = t(ts)
print(f"{'gd_cost_volume_teacher_stats':32s} {dt * 1e6:8.1f} us  {nb / dt / 1e12:5.2f} TB/s")
