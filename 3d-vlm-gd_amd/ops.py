"""Thin tensor-level wrappers over the C ABI (include/gd_hip.h).  Caller-owned memory, current
stream, no hidden allocation inside the library: outputs and workspaces are torch.empty here."""
import torch

from . import _lib
from ._lib import check, dtype_code, lib, ptr, stream


def _req(cond, msg):
    if not cond:
        raise _lib.GdHipError(msg)


def gemm_nt(a, w, out=None, *, out_dtype=None, alpha=1.0, bias=None, lora_t=None, lora_b=None, preact=None,
            act=0, dact_src=None, dact=0, residual=None, accumulate=False):
    """out[M,N] = epilogue(alpha * a[M,K] @ w[N,K]^T).  a, w: same dtype (f32 | bf16), last dim contiguous.
    Batched when a is 3-D ([B,M,K] x [B,N,K] -> [B,M,N], no epilogue tensors).
    Epilogue order: +bias[N] (f32) -> +lora_t[M,r] @ lora_b[r,N] (f32) -> store preact -> act (1 GELU, 2 ReLU)
    -> *act'(dact_src) (1 dGELU(pre), 2 src>0) -> +residual -> +out (accumulate)."""
    _req(a.is_cuda and w.is_cuda and a.dtype == w.dtype, "gemm_nt: a and w must be CUDA tensors of one dtype")
    _req(a.stride(-1) == 1 and w.stride(-1) == 1, "gemm_nt: a and w must be contiguous along K")
    batched = a.dim() == 3
    if batched:
        B, M, K = a.shape
        N = w.shape[1]
        _req(w.shape[0] == B and w.shape[2] == K, "gemm_nt: batched shape mismatch")
        sA, sW, lda, ldw = a.stride(0), w.stride(0), a.stride(1), w.stride(1)
    else:
        (M, K), N, B = a.shape, w.shape[0], 1
        _req(w.shape[1] == K, f"gemm_nt: K mismatch {a.shape} x {w.shape}")
        sA = sW = 0
        lda, ldw = a.stride(0), w.stride(0)
    if out is None:
        odt = out_dtype or a.dtype
        out = torch.empty((B, M, N) if batched else (M, N), dtype=odt, device=a.device)
    _req(out.stride(-1) == 1, "gemm_nt: out must be contiguous along N")
    cdt = dtype_code(out)
    for t, name in ((preact, "preact"), (dact_src, "dact_src"), (residual, "residual")):
        if t is not None:
            _req(t.dtype == out.dtype and t.stride(-1) == 1 and tuple(t.shape) == (M, N), f"gemm_nt: bad {name}")
    for t, name in ((bias, "bias"), (lora_t, "lora_t"), (lora_b, "lora_b")):
        if t is not None:
            _req(t.dtype == torch.float32 and t.is_contiguous(), f"gemm_nt: {name} must be contiguous fp32")
    rt = 0
    if lora_t is not None:
        rt = lora_t.shape[1]
        _req(tuple(lora_t.shape) == (M, rt) and tuple(lora_b.shape) == (rt, N), "gemm_nt: bad lora shapes")
    rc = lib().gd_gemm_nt(ptr(a), ptr(w), ptr(out), M, N, K, lda, ldw, out.stride(-2), B, sA, sW,
                          out.stride(0) if batched else 0, dtype_code(a), cdt, float(alpha), ptr(bias), ptr(lora_t),
                          ptr(lora_b), rt, ptr(preact), preact.stride(0) if preact is not None else 0, int(act),
                          ptr(dact_src), dact_src.stride(0) if dact_src is not None else 0, int(dact), ptr(residual),
                          residual.stride(0) if residual is not None else 0, 1 if accumulate else 0, stream())
    check(rc, "gd_gemm_nt")
    return out


def gemm_tn(y, x, out=None, *, alpha=1.0):
    """out[N,K] (fp32) += alpha * y[M,N]^T @ x[M,K]   (weight gradients; fp32 accumulation)."""
    _req(y.is_cuda and x.is_cuda and y.stride(-1) == 1 and x.stride(-1) == 1 and y.shape[0] == x.shape[0],
         "gemm_tn: bad operands")
    M, N = y.shape
    K = x.shape[1]
    if out is None:
        out = torch.zeros((N, K), dtype=torch.float32, device=y.device)
    _req(out.dtype == torch.float32 and out.stride(-1) == 1, "gemm_tn: out must be fp32, contiguous rows")
    rc = lib().gd_gemm_tn(ptr(y), ptr(x), ptr(out), M, N, K, y.stride(0), x.stride(0), out.stride(0), dtype_code(y),
                          dtype_code(x), float(alpha), stream())
    check(rc, "gd_gemm_tn")
    return out


VARIANTS = {"vggt": 0, "mast3r": 1}


class _CostVolumeKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, t1, t2, m1, m2, variant):
        P, hw, C = f1.shape
        f1, f2 = f1.contiguous(), f2.contiguous()
        t1, t2 = t1.contiguous().float(), t2.contiguous().float()
        m1 = m1.contiguous().to(torch.uint8)
        m2 = m2.contiguous().to(torch.uint8)
        dt = dtype_code(f1)
        loss = torch.empty(P, dtype=torch.float32, device=f1.device)
        stats = torch.empty(P, 2, hw, 4, dtype=torch.float32, device=f1.device)
        ws = torch.empty(lib().gd_cost_volume_kl_workspace_bytes(P, hw, C, dt, 0), dtype=torch.uint8, device=f1.device)
        rc = lib().gd_cost_volume_kl_fwd(ptr(f1), ptr(f2), ptr(t1), ptr(t2), ptr(m1), ptr(m2), P, hw, C,
                                         VARIANTS[variant], dt, ptr(loss), ptr(stats), ptr(ws), stream())
        check(rc, "gd_cost_volume_kl_fwd")
        ctx.save_for_backward(f1, f2, t1, t2, m1, m2, stats)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        f1, f2, t1, t2, m1, m2, stats = ctx.saved_tensors
        P, hw, C = f1.shape
        dt = dtype_code(f1)
        df1, df2 = torch.empty_like(f1), torch.empty_like(f2)
        ws = torch.empty(lib().gd_cost_volume_kl_workspace_bytes(P, hw, C, dt, 1), dtype=torch.uint8, device=f1.device)
        g = gloss.contiguous().float()
        rc = lib().gd_cost_volume_kl_bwd(ptr(f1), ptr(f2), ptr(t1), ptr(t2), ptr(m1), ptr(m2), P, hw, C, dt, ptr(g),
                                         ptr(stats), ptr(df1), ptr(df2), ptr(ws), stream())
        check(rc, "gd_cost_volume_kl_bwd")
        return df1, df2, None, None, None, None, None


def cost_volume_kl(f1, f2, t1, t2, m1, m2, variant="vggt"):
    """Fused dense cost-volume KL for P pairs.  f1,f2 [P,hw,C] raw student features (f32|bf16);
    t1,t2 [P,hw,hw] teacher maps (f32); m1,m2 [P,hw] bool row masks -> loss [P] (f32)."""
    return _CostVolumeKL.apply(f1, f2, t1, t2, m1, m2, variant)


def attention_fwd(qkv, B, N, H):
    """qkv [B*N, 3*H*64] packed (q|k|v, heads inner) -> o [B*N, H*64], lse [B,H,N] (f32)."""
    _req(qkv.is_contiguous() and qkv.shape == (B * N, 3 * H * 64), "attention_fwd: qkv must be [B*N, 3*H*64]")
    o = torch.empty(B * N, H * 64, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    rc = lib().gd_attention_fwd(ptr(qkv), ptr(o), ptr(lse), B, N, H, 64, 64 ** -0.5, dtype_code(qkv), stream())
    check(rc, "gd_attention_fwd")
    return o, lse


def attention_bwd(qkv, o, dout, lse, B, N, H):
    """-> dqkv [B*N, 3*H*64] (same dtype as qkv)."""
    _req(dout.is_contiguous() and dout.shape == o.shape and dout.dtype == qkv.dtype, "attention_bwd: bad dout")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    rc = lib().gd_attention_bwd(ptr(qkv), ptr(o), ptr(dout), ptr(lse), ptr(dqkv), ptr(delta), B, N, H, 64,
                                64 ** -0.5, dtype_code(qkv), stream())
    check(rc, "gd_attention_bwd")
    return dqkv
