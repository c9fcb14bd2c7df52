"""The dense cost-volume forward's two memory streams alone and together (anatomy build, bf16, P = 32, 37 x 37, D = 768):
python3 tools/probe_cv_streams.py — what bench.py reports as roofline_cost_volume.unmasked.attainable.memory_streams_us, plus a sweep of the grid size."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gd_amd  # noqa: E402,F401
from gd_amd import ops  # noqa: E402
from gd_amd._lib import lib  # noqa: E402

dev = torch.device("cuda:0")
P, hw, D = 32, 1369, 768
torch.manual_seed(0)
f1 = torch.randn(P, hw, D, device=dev).bfloat16()
f2 = torch.randn(P, hw, D, device=dev).bfloat16()
t1 = ops.pad_teacher_maps(torch.softmax(torch.randn(P, hw, hw, device=dev) * 3, -1))
t2 = ops.pad_teacher_maps(torch.softmax(torch.randn(P, hw, hw, device=dev) * 3, -1))
ts = ops.cost_volume_teacher_stats(t1, t2)
ones = torch.ones(P, hw, dtype=torch.bool, device=dev)
inv = (1.0 / f1.float().norm(dim=-1).clamp_min(1e-12), 1.0 / f2.float().norm(dim=-1).clamp_min(1e-12))


def t(bits, ncu=0):
    lib().gd_debug_set(b"cv_dbg", bits)
    if ncu:
        lib().gd_debug_set(b"reserve_cus", 256 - ncu)
    try:
        with torch.no_grad():
            return ops.time_on_stream(lambda: ops.cost_volume_kl(f1, f2, t1, t2, ones, ones, "mast3r", tstats=ts, inv_norms=inv), 2, 5) * 1e6
    finally:
        lib().gd_debug_set(b"cv_dbg", 0)
        lib().gd_debug_set(b"reserve_cus", 0)


for name, bits in (("whole op (anatomy build, nothing off)", 32), ("ring alone", 7), ("teacher alone", 64 | 6), ("teacher alone, no mask look-up", 32 | 64 | 6), ("ring + teacher", 6),
                   ("ring + mfma", 3), ("ring + epilogue", 5), ("ring + mfma + epilogue (no teacher)", 1), ("teacher + epilogue (no ring, no mfma)", 64 | 4),
                   ("all but the ring DMA", 64)):
    print(f"{name:44s} {t(bits):8.1f} us")
for ncu in (64, 128, 192):
    print(f"{ncu} CUs: ring alone {t(7, ncu):8.1f}  teacher alone {t(64 | 6, ncu):8.1f}  both {t(6, ncu):8.1f}")
