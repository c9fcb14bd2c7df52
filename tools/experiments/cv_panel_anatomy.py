"""Round 5: where the row-panel-stationary cost-volume forward (csrc/cv_panel.h, dense form, fp16 features as the tf32h trainer calls it) spends its time:
compile-time instantiations (make XFLAGS=-DGD_CV_PANEL_ANAT) with parts switched off, selected by GD_CV_DBG (1 no teacher loads, 2 no epilogue math, 4 no LDS reads / MFMAs, 8 no DMA), whole op,
32 pairs, every row kept, interleaved with the product kernel and the round-4 kernel in one process."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import gd_amd
from gd_amd import ops
L = gd_amd._lib.lib()
P, hw, C = 32, 1369, 768
g = torch.Generator(device="cuda").manual_seed(0)
f1 = torch.randn(P, hw, C, device="cuda", generator=g); f2 = torch.randn(P, hw, C, device="cuda", generator=g)
t1 = ops.pad_teacher_maps(torch.softmax(3 * torch.randn(P, hw, hw, device="cuda", generator=g), -1))
t2 = ops.pad_teacher_maps(torch.softmax(3 * torch.randn(P, hw, hw, device="cuda", generator=g), -1))
m1 = torch.ones(P, hw, dtype=torch.bool, device="cuda"); m2 = torch.ones_like(m1)
ts = ops.cost_volume_teacher_stats(t1, t2)
inv = (1.0 / f1.norm(dim=-1).clamp_min(1e-12), 1.0 / f2.norm(dim=-1).clamp_min(1e-12))
h16 = (f1.half(), f2.half())
run = lambda: ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "mast3r", tstats=ts, inv_norms=inv, x3="h", h16=h16)
with torch.no_grad():
    for panel, dbg in ((1, 0), (0, 0), (2, 1), (2, 2), (2, 3), (2, 4), (2, 5), (2, 6), (2, 7), (2, 8), (2, 12), (2, 15), (1, 0), (0, 0)):
        L.gd_debug_set(b"cv_panel", panel); L.gd_debug_set(b"cv_dbg", dbg)
        t = ops.time_on_stream(run, 3, 10)
        print(f"cv_panel={panel} dbg={dbg:2d} (1 no teacher | 2 no epilogue | 4 no LDS reads / MFMAs | 8 no DMA): {t * 1e6:8.1f} us for {P} pairs", flush=True)
L.gd_debug_set(b"cv_panel", 0); L.gd_debug_set(b"cv_dbg", 0)
