"""Micro-benchmarks of individual kernels on the GPU box (diagnostics; the contract bench is bench.py).
Usage: python tools/bench_kernels.py [gemm] [cv]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gd_amd  # noqa: E402,F401
from gd_amd import ops  # noqa: E402


def timeit(fn, warm=3, it=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


def bench_gemm():
    for dt in (torch.bfloat16, torch.float32):
        for (M, N, K) in [(87680, 2304, 768), (87680, 768, 768), (87680, 3072, 768), (87680, 768, 3072), (4096, 4096, 4096)]:
            a = torch.randn(M, K, device="cuda").to(dt)
            w = torch.randn(N, K, device="cuda").to(dt)
            out = torch.empty(M, N, device="cuda", dtype=dt)
            t = timeit(lambda: ops.gemm_nt(a, w, out=out))
            t2 = timeit(lambda: torch.matmul(a, w.t()))
            print(f"gemm_nt {str(dt)[6:]:9s} {M}x{N}x{K}: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:7.1f} TF/s   (torch/hipblaslt {2*M*N*K/t2/1e12:7.1f} TF/s)")


def bench_gemm_anat(rounds=3):
    """Anatomy of the persistent kernel's main loop (gd_debug_set("gemm_anat", v)): 0 everything, 1 no operand DMA, 2 no MFMAs,
    3 the DMA ring alone (no LDS reads, no MFMAs), 4 no C stores.  Needs a library built with `make FLAGS+=-DGD_GEMM_ANATOMY`.  Interleaved rounds in one process; reported as time and as the TFLOP/s the
    full FLOP count would be at that time."""
    L = gd_amd._lib.lib()
    for (M, N, K) in [(87680, 2304, 768), (87680, 768, 768), (87680, 3072, 768), (87680, 768, 3072), (4096, 4096, 4096), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = torch.randn(N, K, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = {v: [] for v in (0, 1, 2, 3, 4)}
        for r in range(rounds):
            for v in res:
                L.gd_debug_set(b"gemm_anat", v)
                res[v].append(timeit(lambda: ops.gemm_nt(a, w, out=out), warm=2, it=8))
        L.gd_debug_set(b"gemm_anat", 0)
        print(f"gemm_anat {M}x{N}x{K}: " + " | ".join(f"v{v} {min(x) * 1e6:8.1f} us ({2 * M * N * K / min(x) / 1e12:7.1f} TF/s-equiv)" for v, x in res.items()), flush=True)


def bench_gelu():
    M, N, K = 87680, 3072, 768
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.zeros(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t0 = timeit(lambda: ops.gemm_nt(a, w, out=out))
    t1 = timeit(lambda: ops.gemm_nt(a, w, out=out, bias=bias, act=1))
    t2 = timeit(lambda: ops.gemm_nt(a, w, out=out, bias=bias, act=1, preact=pre))
    a2 = torch.randn(M, N, device="cuda").bfloat16()
    w2 = (torch.randn(768, N, device="cuda") * 0.05).bfloat16()
    o2 = torch.empty(M, 768, device="cuda", dtype=torch.bfloat16)
    res = torch.randn(M, 768, device="cuda").bfloat16()
    t3 = timeit(lambda: ops.gemm_nt(a2, w2, out=o2))
    t4 = timeit(lambda: ops.gemm_nt(a2, w2, out=o2, bias=bias[:768].contiguous(), residual=res))
    wt = (torch.randn(N, 768, device="cuda") * 0.05).bfloat16()
    dy = torch.randn(M, 768, device="cuda").bfloat16()
    t5 = timeit(lambda: ops.gemm_nt(dy, wt, out=out))
    t6 = timeit(lambda: ops.gemm_nt(dy, wt, out=out, dact_src=pre, dact=1))
    print(f"fc1 plain {t0*1e3:.3f} ms | +bias+GELU {t1*1e3:.3f} | +preact {t2*1e3:.3f} || fc2 plain {t3*1e3:.3f} | +bias+residual {t4*1e3:.3f} || dfc2 plain {t5*1e3:.3f} | +dGELU {t6*1e3:.3f}")


def bench_cv():
    for dt in (torch.bfloat16, torch.float32):
        for P in (1, 8, 32):
            hw, C = 1369, 768
            f1 = torch.randn(P, hw, C, device="cuda").to(dt).requires_grad_(True)
            f2 = torch.randn(P, hw, C, device="cuda").to(dt).requires_grad_(True)
            t1 = torch.softmax(3 * torch.randn(P, hw, hw, device="cuda"), -1)
            t2 = torch.softmax(3 * torch.randn(P, hw, hw, device="cuda"), -1)
            m1 = torch.rand(P, hw, device="cuda") > 0.3
            m2 = torch.rand(P, hw, device="cuda") > 0.3
            es = 2 if dt == torch.bfloat16 else 4
            fwd_bytes = P * (2 * hw * C * es + 2 * hw * hw * 4 + 2 * hw)
            bwd_bytes = fwd_bytes + P * 2 * hw * C * es
            t1, t2 = ops.pad_teacher_maps(t1), ops.pad_teacher_maps(t2)
            ts = ops.cost_volume_teacher_stats(t1, t2)
            with torch.no_grad():
                tf = timeit(lambda: ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "vggt", tstats=ts))

            def fb():
                f1.grad = f2.grad = None
                ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "vggt", tstats=ts).sum().backward()
            tfb = timeit(fb)
            print(f"cost_volume {str(dt)[6:]:9s} P={P:2d}: fwd {tf*1e6/P:8.1f} us/pair {fwd_bytes/tf/1e9:8.1f} GB/s | "
                  f"fwd+bwd {tfb*1e6/P:8.1f} us/pair {(fwd_bytes+bwd_bytes)/tfb/1e9:8.1f} GB/s")


def bench_adapter():
    """fused adapter kernel against the two-GEMM formulation (M = 87680 rows of the P = 32 step)."""
    for D in (768, 1024):
        M = 87680
        x = torch.randn(M, D, device="cuda").bfloat16()
        down = (torch.randn(64, D, device="cuda") * 0.05).bfloat16()
        up = (torch.randn(D, 64, device="cuda") * 0.05).bfloat16()
        by = 2 * M * D * 2 + M * 64 * 2
        t = timeit(lambda: ops.adapter_fused(x, down, up))

        def two():
            h = ops.gemm_nt(x, down, act=2)
            return ops.gemm_nt(h, up, residual=x)
        t2 = timeit(two)
        _, hid = ops.adapter_fused(x, down, up)
        ut, dn = up.t().contiguous(), down.t().contiguous()
        tb = timeit(lambda: ops.adapter_fused(x, ut, dn, gate_src=hid))
        print(f"adapter D={D}: fused fwd {t*1e6:7.1f} us ({by/t/1e9:6.0f} GB/s)  bwd-to-input {tb*1e6:7.1f} us | two GEMMs {t2*1e6:7.1f} us")


def bench_adapter_ln():
    """Round 5: the adapter pass that also writes the next block's LayerNorm (gd_adapter_fused_h_ln) against adapter pass + LayerNorm pass, M = 87 680, D = 768."""
    M, D = 87680, 768
    x = torch.randn(M, D, device="cuda")
    down, up = (0.05 * torch.randn(64, D, device="cuda")).half(), (0.05 * torch.randn(D, 64, device="cuda")).half()
    g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    out = ops.adapter_fused_h(x, down, up)[0]
    for _ in range(3):
        ta = timeit(lambda: ops.adapter_fused_h(x, down, up))
        tl = timeit(lambda: ops.layernorm_fwd(out, g, b, 1e-6, out_dtype=torch.float16))
        tf = timeit(lambda: ops.adapter_fused_h_ln(x, down, up, g, b, 1e-6))
        print(f"adapter_ln: adapter {ta * 1e6:7.1f} us + LayerNorm {tl * 1e6:6.1f} us = {(ta + tl) * 1e6:7.1f} us | fused {tf * 1e6:7.1f} us", flush=True)


def pmc_cv():
    """one configuration, few launches: the target of `rocprofv3 --pmc ...` runs (profiles/README.md)."""
    P, hw, C = 32, 1369, 768
    f1 = torch.randn(P, hw, C, device="cuda").bfloat16().requires_grad_(True)
    f2 = torch.randn(P, hw, C, device="cuda").bfloat16().requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, device="cuda"), -1)
    m1 = torch.rand(P, hw, device="cuda") > 0.3
    m2 = torch.rand(P, hw, device="cuda") > 0.3
    t1, t2 = ops.pad_teacher_maps(t1), ops.pad_teacher_maps(t2)          # the cached-target layout (teacher_cache.py)
    ts = ops.cost_volume_teacher_stats(t1, t2)
    for _ in range(3):
        f1.grad = f2.grad = None
        ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "vggt", tstats=ts).sum().backward()
    torch.cuda.synchronize()


def pmc_cv_kp(full=False):
    """as pmc_cv, with the MASt3R trainer's row masks: patches that hold one of 300 random keypoints (about 20 % of the rows);
    forward only — what bench.py's `roofline_cost_volume` times."""
    P, hw, C, img, patch = 32, 1369, 768, 518, 14
    g = torch.Generator(device="cuda").manual_seed(0)
    f1 = torch.randn(P, hw, C, device="cuda", generator=g).bfloat16()
    f2 = torch.randn(P, hw, C, device="cuda", generator=g).bfloat16()
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, device="cuda", generator=g), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, device="cuda", generator=g), -1)
    kp1 = torch.rand(P, 300, 2, device="cuda", generator=g) * (img - 1)
    kp2 = torch.rand(P, 300, 2, device="cuda", generator=g) * (img - 1)
    m1, m2 = ops.patch_mask(kp1, img, img, patch), ops.patch_mask(kp2, img, img, patch)
    t1, t2 = ops.pad_teacher_maps(t1), ops.pad_teacher_maps(t2)
    ts = ops.cost_volume_teacher_stats(t1, t2)
    print("kept rows", int(m1.sum()), int(m2.sum()), "of", P * hw, "each")
    inv = (1.0 / f1.float().norm(dim=-1).clamp_min(1e-12), 1.0 / f2.float().norm(dim=-1).clamp_min(1e-12))     # from the producer in the step
    if full:
        m1, m2 = torch.ones_like(m1), torch.ones_like(m2)
    with torch.no_grad():
        for _ in range(3):
            ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "mast3r", tstats=ts, inv_norms=inv)
    torch.cuda.synchronize()


def pmc_cv_rows():
    """the kept-row forward as the tf32h trainer runs it (gd_cost_volume_kl_fwd_rows on the fp16 feature copies, kept_rows_max = the keypoint
    count): the target of the round-4 --pmc passes behind bench.py's masked `roofline_cost_volume.traffic`."""
    P, hw, C, img, patch = 32, 1369, 768, 518, 14
    g = torch.Generator(device="cuda").manual_seed(0)
    f1 = torch.randn(P, hw, C, device="cuda", generator=g)
    f2 = torch.randn(P, hw, C, device="cuda", generator=g)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, device="cuda", generator=g), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, device="cuda", generator=g), -1)
    kp1 = torch.rand(P, 300, 2, device="cuda", generator=g) * (img - 1)
    kp2 = torch.rand(P, 300, 2, device="cuda", generator=g) * (img - 1)
    m1, m2 = ops.patch_mask(kp1, img, img, patch), ops.patch_mask(kp2, img, img, patch)
    t1, t2 = ops.pad_teacher_maps(t1), ops.pad_teacher_maps(t2)
    ts = ops.cost_volume_teacher_stats(t1, t2)
    print("kept rows", int(m1.sum()), int(m2.sum()), "of", P * hw, "each")
    inv = (1.0 / f1.norm(dim=-1).clamp_min(1e-12), 1.0 / f2.norm(dim=-1).clamp_min(1e-12))
    h16 = (f1.half(), f2.half())
    with torch.no_grad():
        for _ in range(3):
            ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "mast3r", tstats=ts, inv_norms=inv, x3="h", h16=h16, kept_rows_max=300)
    torch.cuda.synchronize()


def bench_attn():
    for (B, N, H) in ((64, 1370, 12), (8, 6401, 12)):
        _bench_attn(B, N, H)


def _bench_attn(B, N, H):
    qkv = torch.randn(B * N, 3 * H * 64, device="cuda").bfloat16()
    dout = torch.randn(B * N, H * 64, device="cuda").bfloat16()
    o, lse = ops.attention_fwd(qkv, B, N, H)
    fl = 4.0 * B * H * N * N * 64
    t = timeit(lambda: ops.attention_fwd(qkv, B, N, H))
    print(f"attn fwd  B={B} N={N}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
    t = timeit(lambda: ops.attention_bwd(qkv, o, dout, lse, B, N, H))
    print(f"attn bwd  {t*1e6:8.1f} us  {2.5*fl/t/1e12:7.1f} TF/s (algorithmic 5 products)")


def bench_attn_x3():
    """fp32 attention: the exact-f32 MFMA kernels against the split-precision (x3) instantiation, 64 x 12 x 1370."""
    B, N, H = 64, 1370, 12
    qkv = torch.randn(B * N, 3 * H * 64, device="cuda")
    dout = torch.randn(B * N, H * 64, device="cuda")
    fl = 4.0 * B * H * N * N * 64
    for x3 in (False, True):
        o, lse = ops.attention_fwd(qkv, B, N, H, x3=x3)
        tf = timeit(lambda: ops.attention_fwd(qkv, B, N, H, x3=x3), warm=1, it=3)
        tb = timeit(lambda: ops.attention_bwd(qkv, o, dout, lse, B, N, H, x3=x3), warm=1, it=3)
        print(f"attn fp32 tensors, {'split-precision x3' if x3 else 'exact-f32 MFMA   '}: fwd {tf*1e6:8.1f} us ({fl/tf/1e12:6.1f} TF/s)  bwd {tb*1e6:8.1f} us ({2.5*fl/tb/1e12:6.1f} TF/s)", flush=True)


def bench_cva():
    """teacher cross-view attention maps (gd_cross_view_attn) vs the reference formulation in torch (per-head maps materialised)."""
    from gd_amd import teacher_glue as TG
    B, H, n, prefix = 4, 16, 1369, 5
    N = 2 * (n + prefix)
    q = torch.randn(B, H, N, 64, device="cuda").bfloat16()
    k = torch.randn(B, H, N, 64, device="cuda").bfloat16()
    out = torch.empty(2 * B, n, n, device="cuda")
    t = timeit(lambda: TG.cross_view_attention_maps(q, k, 0.125, 0.7, prefix, out=out))

    def ref():
        qs = q.float() * 0.125
        a1 = torch.softmax(qs[..., prefix:N // 2, :] @ k.float()[..., N // 2 + prefix:, :].transpose(-2, -1) / 0.7, -1)
        a2 = torch.softmax(qs[..., N // 2 + prefix:, :] @ k.float()[..., prefix:N // 2, :].transpose(-2, -1) / 0.7, -1)
        return torch.cat([a1, a2], 0).mean(1)
    t2 = timeit(ref, warm=1, it=3)
    by = 2 * B * n * n * 4 + 2 * q.numel() * 2
    print(f"cross_view_attn B={B} H={H} n={n}: {t*1e6:8.1f} us ({t*1e6/B:.1f} us/pair, {by/t/1e9:.0f} GB/s of output+inputs) | torch per-head maps {t2*1e6:8.1f} us")


def bench_tn():
    for (M, N, K, ydt, xdt) in [(87680, 8, 2304, torch.float32, torch.bfloat16), (87680, 8, 768, torch.float32, torch.bfloat16),
                                (87680, 768, 64, torch.bfloat16, torch.bfloat16), (87680, 64, 768, torch.bfloat16, torch.bfloat16),
                                (19200, 768, 6912, torch.bfloat16, torch.bfloat16)]:
        y = torch.randn(M, N, device="cuda").to(ydt)
        x = torch.randn(M, K, device="cuda").to(xdt)
        out = torch.zeros(N, K, device="cuda")
        t = timeit(lambda: ops.gemm_tn(y, x, out=out))
        by = y.numel() * y.element_size() + x.numel() * x.element_size()
        ref = y.double().t() @ x.double()
        got = ops.gemm_tn(y, x)
        err = float((got.double() - ref).norm() / ref.norm())
        print(f"gemm_tn M={M} N={N:4d} K={K:5d}: {t*1e6:8.1f} us  {by/t/1e12:5.2f} TB/s  rel err {err:.2e}")


def bench_ln():
    M, D = 87680, 768
    x = torch.randn(M, D, device="cuda").bfloat16()
    dy = torch.randn(M, D, device="cuda").bfloat16()
    dres = torch.randn(M, D, device="cuda").bfloat16()
    g, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6)
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6))
    print(f"ln fwd {t*1e6:7.1f} us  {2*M*D*2/t/1e12:5.2f} TB/s")
    t = timeit(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres))
    print(f"ln bwd (+dres) {t*1e6:7.1f} us  {4*M*D*2/t/1e12:5.2f} TB/s")


def bench_lora():
    M, D = 87680, 768
    buf = torch.randn(M, 3 * D, device="cuda").bfloat16()
    dqv = buf[:, :2 * D]
    t = torch.randn(M, 8, device="cuda")
    bt = (torch.randn(8, 2 * D, device="cuda") * 0.05).bfloat16()
    gbt = torch.zeros(8, 2 * D, device="cuda")
    tf = timeit(lambda: ops.lora_bwd_fused(dqv, t, bt, gbt))
    t1 = timeit(lambda: ops.gemm_nt(dqv, bt, out_dtype=torch.float32))
    t2 = timeit(lambda: ops.gemm_tn(t, dqv, out=gbt))
    by = M * 2 * D * 2
    y1 = torch.randn(M, D, device="cuda").bfloat16()
    ga = torch.zeros(8, D, device="cuda")
    ta = timeit(lambda: ops.skinny_tn_mfma(t, y1, ga))
    tb = timeit(lambda: ops.gemm_tn(t, y1, out=ga))
    print(f"gat (K = {D}): slab MFMA kernel {ta*1e6:6.1f} us | streaming kernel {tb*1e6:6.1f} us")
    print(f"lora bwd fused {tf*1e6:7.1f} us ({by/tf/1e12:4.2f} TB/s) | separate: dt {t1*1e6:6.1f} + gbt {t2*1e6:6.1f} = {(t1+t2)*1e6:6.1f} us")


def bench_rank():
    """depth head losses at the bench shape: 32 pairs x 2 views x 300 keypoints, D = 768."""
    P, N, D = 32, 300, 768
    g = torch.Generator(device="cuda").manual_seed(0)
    head = {"w1": torch.randn(128, D, device="cuda", generator=g) * 0.05, "b1": torch.zeros(128, device="cuda"),
            "ln_w": torch.ones(128, device="cuda"), "ln_b": torch.zeros(128, device="cuda"),
            "w2": torch.randn(1, 128, device="cuda", generator=g) * 0.1, "b2": torch.zeros(1, device="cuda")}
    head = {k: v.requires_grad_(True) for k, v in head.items()}
    feats = torch.randn(P, 2, N, D, device="cuda", generator=g).requires_grad_(True)
    d1 = torch.rand(P, N, device="cuda", generator=g) * 3
    d2 = torch.rand(P, N, device="cuda", generator=g) * 3

    def f():
        l1, intra = ops.depth_losses(feats, d1, d2, head)
        return l1, intra
    t = timeit(f)
    l1, intra = f()
    print(f"depth_losses fwd+bwd (fused): {t*1e6:8.1f} us   l1 {l1.mean().item():.6f} intra {intra.mean().item():.6f}")


def probe_gemm():
    """phase split of the persistent gemm_nt kernel (gd_gemm_phase_probe): cycles per tile in each phase."""
    import ctypes
    from gd_amd import _lib
    L = _lib.lib()
    out = (ctypes.c_ulonglong * 6)()
    cases = [("a0", 87680, 3072, 768, {}), ("a0", 87680, 768, 3072, {}), ("a0", 87680, 768, 768, {}), ("a0", 4096, 4096, 4096, {}),
             ("ba1", 87680, 3072, 768, dict(bias=True, act=1)), ("bpa1", 87680, 3072, 768, dict(bias=True, act=1, preact=True)),
             ("da0", 87680, 3072, 768, dict(dact=True)), ("bra0", 87680, 768, 3072, dict(bias=True, residual=True))]
    for tag, M, N, K, o in cases:
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = torch.randn(N, K, device="cuda").bfloat16()
        out_t = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        kw = {}
        if o.get("bias"): kw["bias"] = torch.randn(N, device="cuda")
        if o.get("act"): kw["act"] = o["act"]
        if o.get("preact"): kw["preact"] = torch.empty_like(out_t)
        if o.get("dact"): kw.update(dact_src=torch.randn(M, N, device="cuda").bfloat16(), dact=1)
        if o.get("residual"): kw["residual"] = torch.randn(M, N, device="cuda").bfloat16()
        f = lambda: ops.gemm_nt(a, w, out=out_t, **kw)
        t = timeit(f)
        L.gd_gemm_phase_probe(1, None)
        f()
        L.gd_gemm_phase_probe(0, out)
        n = max(out[3], 1)
        print(f"probe {tag:5s} {M}x{N}x{K}: {t*1e6:7.1f} us {2*M*N*K/t/1e12:7.1f} TF/s | shader-clock cycles per tile: wait {out[0]/n:8.1f}  main {out[1]/n:8.1f}  dma-issue {out[4]/n:8.1f}  epi {out[2]/n:8.1f}  tiles/blk {n/256:5.2f} | of main: stage waits (vmcnt + barrier) {out[5]/n:8.1f} = {out[5]/max(out[1],1):.3f}")


def pmc_attn():
    import os
    B, N, H = (8, 6401, 12) if os.environ.get("GD_PMC_LONG") else (64, 1370, 12)
    dt = torch.float16 if os.environ.get("GD_PMC_F16") else torch.bfloat16      # fp16 = the tf32h engine's attention operands
    qkv = torch.randn(B * N, 3 * H * 64, device="cuda").to(dt)
    dout = torch.randn(B * N, H * 64, device="cuda").to(dt)
    for _ in range(2):
        o, lse = ops.attention_fwd(qkv, B, N, H)
        ops.attention_bwd(qkv, o, dout, lse, B, N, H)
    torch.cuda.synchronize()


def pmc_gemm():
    a = torch.randn(87680, 768, device="cuda").bfloat16()
    w = torch.randn(2304, 768, device="cuda").bfloat16()
    for _ in range(3):
        ops.gemm_nt(a, w)
    torch.cuda.synchronize()


def pmc_adapter():
    M, D = 87680, 768
    x = torch.randn(M, D, device="cuda").bfloat16()
    down = (torch.randn(64, D, device="cuda") * 0.05).bfloat16()
    up = (torch.randn(D, 64, device="cuda") * 0.05).bfloat16()
    for _ in range(3):
        ops.adapter_fused(x, down, up)
    torch.cuda.synchronize()


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "cv"]
    if "pmc_adapter" in which:
        pmc_adapter()
    if "gemm" in which:
        bench_gemm()
    if "gemm_anat" in which:
        bench_gemm_anat()
    if "cv" in which:
        bench_cv()
    if "adapter_ln" in which:
        bench_adapter_ln()
    if "gelu" in which:
        bench_gelu()
    if "pmc_cv" in which:
        pmc_cv()
    if "pmc_cv_kp" in which:
        pmc_cv_kp()
    if "pmc_cv_full" in which:
        pmc_cv_kp(full=True)
    if "pmc_cv_rows" in which:
        pmc_cv_rows()
    if "attn" in which:
        bench_attn()
    if "attn_x3" in which:
        bench_attn_x3()
    if "cva" in which:
        bench_cva()
    if "tn" in which:
        bench_tn()
    if "rank" in which:
        bench_rank()
    if "ln" in which:
        bench_ln()
    if "lora" in which:
        bench_lora()
    if "probe" in which:
        probe_gemm()
    if "adapter" in which:
        bench_adapter()
    if "pmc_attn" in which:
        pmc_attn()
    if "pmc_gemm" in which:
        pmc_gemm()
