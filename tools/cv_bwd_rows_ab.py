"""A/B of the cost-volume backward with sparse keypoint masks at the step's shape: dense hw x hw backward (GD_CV_BWD_ROWS=0) against the kept-row form."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gd_amd  # noqa: F401,E402
from gd_amd import ops
from gd_amd.options import set_option

P, hw, C, K = 32, 1369, 768, 384
dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(0)
if os.environ.get("BIGM"):
    from gd_amd._lib import lib
    lib().gd_debug_set(b"gemm_batch_big_m", int(os.environ["BIGM"]))
ONLY = sys.argv[1] if len(sys.argv) > 1 else None      # "h" | "bf16": one engine, the backward as GD_CV_BWD_ROWS says (for rocprofv3 runs)
for fmt, dt in (("h", torch.float32), ("", torch.bfloat16)):
    if ONLY and ONLY != (fmt or "bf16"):
        continue
    f1 = torch.randn(P, hw, C, generator=gen, device=dev).to(dt).requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device=dev).to(dt).requires_grad_(True)
    t1 = ops.pad_teacher_maps(torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device=dev), -1))
    t2 = ops.pad_teacher_maps(torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device=dev), -1))
    ts = ops.cost_volume_teacher_stats(t1, t2)
    m1 = torch.zeros(P, hw, dtype=torch.bool, device=dev)
    m2 = torch.zeros(P, hw, dtype=torch.bool, device=dev)
    for p in range(P):
        m1[p, torch.randperm(hw, generator=gen, device=dev)[:330]] = True
        m2[p, torch.randperm(hw, generator=gen, device=dev)[:330]] = True
    inv = (1.0 / f1.detach().float().norm(dim=-1).clamp_min(1e-12), 1.0 / f2.detach().float().norm(dim=-1).clamp_min(1e-12))
    h16 = (f1.detach().half(), f2.detach().half()) if fmt == "h" else None

    def fb():
        f1.grad = f2.grad = None
        ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "mast3r", tstats=ts, inv_norms=inv, x3=fmt, h16=h16, kept_rows_max=K).sum().backward()

    def fwd():
        with torch.no_grad():
            ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "mast3r", tstats=ts, inv_norms=inv, x3=fmt, h16=h16, kept_rows_max=K)
    tf = ops.time_on_stream(fwd, 2, 10)
    from gd_amd.options import option
    for v in ((option("cv_bwd_rows"),) if ONLY else (0, 1)):
        set_option("cv_bwd_rows", v)
        t = ops.time_on_stream(fb, 2, 10)
        print(f"fmt={fmt or dt} cv_bwd_rows={v}: fwd {tf * 1e6:.1f} us  fwd+bwd {t * 1e6:.1f} us  (bwd {1e6 * (t - tf):.1f} us)")
