"""Round-4 diagnostics of the persistent GEMM on the GPU box (not the contract bench):
  tail   — what the last, nearly empty round of a persistent launch costs: the same (N, K) at row counts that end exactly on a round
           boundary and a few tiles past it (1029 tiles on 256 CUs is 4.02 rounds for every N = 768 GEMM of the step);
  shapes — the step's eight GEMMs stand-alone WITH the step's epilogues and dtypes (fp16 operands), long loops so that the clock settles,
           to set beside bench.py --gemm-shapes' in-step figures.
Usage: python tools/probe_r04.py [tail] [shapes] [ln]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gd_amd  # noqa: E402,F401
from gd_amd import ops  # noqa: E402


def timeit(fn, warm=5, it=40):
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


def mk(shape, dt=torch.float16, s=1.0):
    return (torch.randn(*shape, device="cuda") * s).to(dt)


def tail():
    for (N, K) in [(768, 768), (768, 3072), (768, 2304), (2304, 768), (3072, 768)]:
        tn = (N + 255) // 256
        base = None
        for tm in (256, 257, 341, 342, 343):          # row tiles: tm * tn = 768 / 771 / 1023 / 1026 / 1029 tiles at N = 768
            M = tm * 256 if tm != 343 else 87680
            a, w = mk((M, K)), mk((N, K), s=0.05)
            out = torch.empty(M, N, device="cuda", dtype=torch.float16)
            t = timeit(lambda: ops.gemm_nt(a, w, out=out))
            tiles = tm * tn
            per_round = t / (tiles / 256.0)
            if base is None:
                base = t / (tiles // 256)
            print(f"tail N={N:5d} K={K:5d} M={M:6d} tiles={tiles:5d} ({tiles / 256:6.3f} rounds): {t * 1e6:8.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TF/s  "
                  f"us per full round (first line) {base * 1e6:7.1f}", flush=True)


def shapes():
    M, D = 87680, 768
    bias3, bias1, bias4 = torch.zeros(3 * D, device="cuda"), torch.zeros(D, device="cuda"), torch.zeros(4 * D, device="cuda")
    y, o, h = mk((M, D)), mk((M, D)), mk((M, 4 * D))
    wqkv, wproj, w1, w2 = mk((3 * D, D), s=0.03), mk((D, D), s=0.03), mk((4 * D, D), s=0.03), mk((D, 4 * D), s=0.03)
    x32 = torch.randn(M, D, device="cuda")
    t8, bt = torch.randn(M, 8, device="cuda") * 0.1, torch.randn(8, 3 * D, device="cuda") * 0.1
    at = torch.randn(8, D, device="cuda") * 0.1
    pre = torch.empty(M, 4 * D, device="cuda", dtype=torch.float16)
    dqkv = mk((M, 3 * D))
    one = torch.ones(1, device="cuda")
    cases = [
        ("qkv  fp16C bias lora      ", 3 * D, D, lambda: ops.gemm_nt(y, wqkv, bias=bias3, lora_t=t8, lora_b=bt, out_dtype=torch.float16)),
        ("qkv  fp16C plain          ", 3 * D, D, lambda: ops.gemm_nt(y, wqkv, out_dtype=torch.float16)),
        ("proj f32C bias residual   ", D, D, lambda: ops.gemm_nt(o, wproj, bias=bias1, residual=x32, out_dtype=torch.float32)),
        ("proj f32C plain           ", D, D, lambda: ops.gemm_nt(o, wproj, out_dtype=torch.float32)),
        ("proj fp16C plain (dproj)  ", D, D, lambda: ops.gemm_nt(o, wproj, out_dtype=torch.float16)),
        ("fc1  fp16C bias gelu pre  ", 4 * D, D, lambda: ops.gemm_nt(y, w1, bias=bias4, act=3, preact=pre, out_dtype=torch.float16)),
        ("fc1  fp16C bias gelu      ", 4 * D, D, lambda: ops.gemm_nt(y, w1, bias=bias4, act=3, out_dtype=torch.float16)),
        ("fc1  fp16C plain          ", 4 * D, D, lambda: ops.gemm_nt(y, w1, out_dtype=torch.float16)),
        ("fc2  f32C bias residual   ", D, 4 * D, lambda: ops.gemm_nt(h, w2, bias=bias1, residual=x32, out_dtype=torch.float32)),
        ("fc2  f32C plain (dfc1)    ", D, 4 * D, lambda: ops.gemm_nt(h, w2, out_dtype=torch.float32, alpha_dev=one)),
        ("fc2  fp16C plain (dfc1 h) ", D, 4 * D, lambda: ops.gemm_nt(h, w2, out_dtype=torch.float16)),
        ("dfc2 fp16C gate           ", 4 * D, D, lambda: ops.gemm_nt(y, w1, dact_src=pre, dact=3, out_dtype=torch.float16)),
        ("dqkv f32C lora            ", D, 3 * D, lambda: ops.gemm_nt(dqkv, wqkv.t().contiguous(), lora_t=t8, lora_b=at, out_dtype=torch.float32, alpha_dev=one)),
        ("dqkv fp16C lora           ", D, 3 * D, lambda: ops.gemm_nt(dqkv, wqkv.t().contiguous(), lora_t=t8, lora_b=at, out_dtype=torch.float16)),
    ]
    wq_t = wqkv.t().contiguous()
    cases[-2] = (cases[-2][0], D, 3 * D, lambda: ops.gemm_nt(dqkv, wq_t, lora_t=t8, lora_b=at, out_dtype=torch.float32, alpha_dev=one))
    cases[-1] = (cases[-1][0], D, 3 * D, lambda: ops.gemm_nt(dqkv, wq_t, lora_t=t8, lora_b=at, out_dtype=torch.float16))
    only = os.environ.get("ONLY")
    if only:
        cases = [c for c in cases if any(o in c[0] for o in only.split(","))]
    from gd_amd._lib import lib
    for name, N, K, fn in cases:
        line = f"shape {name} {M}x{N}x{K}:"
        t = timeit(fn, warm=10, it=60)
        line += f" {t * 1e6:6.1f} us"
        print(line, flush=True)


def ln():
    M, D = 87680, 768
    x = torch.randn(M, D, device="cuda")
    gamma, beta = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    dy, dres = torch.randn(M, D, device="cuda") * 1e-5, torch.randn(M, D, device="cuda") * 1e-5
    sc = ops.amax_scale(dy, 8.0)
    dy16 = ops.cast16(dy, scale_dev=sc[0:1])
    print(f"ln_bwd f32 dy + cast      : {timeit(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres, cast_scale=sc[0:1])) * 1e6:7.1f} us")
    print(f"ln_bwd fp16 dy + cast     : {timeit(lambda: ops.layernorm_bwd(dy16, x, gamma, mean, rstd, dres=dres, cast_scale=sc[0:1], dy_scale=sc[1:2])) * 1e6:7.1f} us")
    print(f"ln_bwd f32 dy             : {timeit(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres)) * 1e6:7.1f} us")

    def with_amax():
        d = ops.layernorm_bwd(dy16, x, gamma, mean, rstd, dres=dres, dy_scale=sc[1:2], want_amax=True)
        ops.amax_take(d)
    print(f"ln_bwd fp16 dy + amax     : {timeit(with_amax) * 1e6:7.1f} us")
    print(f"amax_scale stand-alone    : {timeit(lambda: ops.amax_scale(dy, 8.0)) * 1e6:7.1f} us")


def adapter():
    """adapter_persist_h_kernel forward and backward-to-input at the step's size (fp32 x / out, fp16 operands)."""
    M, D = 87680, 768
    x = torch.randn(M, D, device="cuda")
    down, up = (torch.randn(64, D, device="cuda") * 0.05).half(), (torch.randn(D, 64, device="cuda") * 0.05).half()
    _, hid, _ = ops.adapter_fused_h(x, down, up)
    dout = torch.randn(M, D, device="cuda") * 1e-5
    sc = ops.amax_scale(dout, 8.0)
    ut, dt = up.t().contiguous(), down.t().contiguous()
    tf = timeit(lambda: ops.adapter_fused_h(x, down, up), warm=5, it=50)
    tb = timeit(lambda: ops.adapter_fused_h(dout, ut, dt, gate_src=hid, in_scale=sc[0:1], alpha_dev=sc[1:2], copy_scale=sc[0:1], want_copy=True), warm=5, it=50)
    bf, bb = M * D * 8 + M * 64 * 2, M * D * 10 + M * 64 * 4
    print(f"adapter_h fwd {tf * 1e6:7.1f} us ({bf / tf / 1e12:5.2f} TB/s)   bwd-to-input + fp16 copy {tb * 1e6:7.1f} us ({bb / tb / 1e12:5.2f} TB/s)   lib {os.environ.get('GD_HIP_LIB', 'default')}")


if __name__ == "__main__":
    what = sys.argv[1:] or ["tail", "shapes", "ln"]
    for w in what:
        {"tail": tail, "shapes": shapes, "ln": ln, "adapter": adapter}[w]()
