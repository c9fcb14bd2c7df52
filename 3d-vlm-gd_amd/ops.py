"""Thin tensor-level wrappers over the C ABI (include/gd_hip.h).  Caller-owned memory, current
stream, no hidden allocation inside the library: outputs and workspaces are torch.empty here."""
import torch

from . import _lib
from ._lib import check, dtype_code, lib, ptr, stream
from .options import option


def _req(cond, msg):
    if not cond:
        raise _lib.GdHipError(msg)


class GemmProfiler:
    """HIP-event bracket around every gd_gemm_nt launch on the current stream (bench.py's roofline leg)."""

    def __init__(self):
        self.records = []

    def totals(self, keep=None):
        """(FLOPs, ms, launches) over the recorded launches; keep(tag) with tag = (M, N, K, B, epilogue, out dtype, operand
        dtype) selects a subset (e.g. the launches that run on the persistent MFMA kernel, leaving out the HBM-bound N <= 8
        streaming launches)."""
        torch.cuda.synchronize()
        recs = [r for r in self.records if keep is None or keep(r[3])]
        flops = sum(r[2] for r in recs)
        ms = sum(r[0].elapsed_time(r[1]) for r in recs)
        return flops, ms, len(recs)

    def roofline_time(self, keep, peak_flops, peak_bytes):
        """(sum over the kept launches of max(FLOPs / peak_flops, algorithmic bytes / peak_bytes) in ms, measured ms, launches whose bound is the
        byte term, launches).  Algorithmic bytes of a launch: both operands once, C once, every [M, N] epilogue tensor (preact, dact_src,
        residual, the accumulate read) once — what the launch must move if nothing is re-read."""
        torch.cuda.synchronize()
        es = {"float32": 4, "bfloat16": 2, "float16": 2}
        lim = ms = 0.0
        nb = n = 0
        for e0, e1, fl, tag in self.records:
            if keep is not None and not keep(tag):
                continue
            M, N, K, B, epi, odt, adt = tag
            nt = sum(epi.split("a")[0].count(c) for c in "pdr") + (1 if epi.endswith("+") else 0)
            by = B * (M * K + N * K) * es[adt] + B * M * N * es[odt] * (1 + nt)
            tf, tb = fl / peak_flops, by / peak_bytes
            lim += max(tf, tb) * 1e3
            nb += tb > tf
            ms += e0.elapsed_time(e1)
            n += 1
        return lim, ms, nb, n

    def algorithmic_bytes(self, keep=None):
        """(sum of the kept launches' algorithmic bytes — both operands once, C once, every [M, N] epilogue tensor once, as roofline_time counts
        them — and the number of launches)."""
        es = {"float32": 4, "bfloat16": 2, "float16": 2}
        tot = n = 0
        for _, _, _, tag in self.records:
            if keep is not None and not keep(tag):
                continue
            M, N, K, B, epi, odt, adt = tag
            nt = sum(epi.split("a")[0].count(c) for c in "pdr") + (1 if epi.endswith("+") else 0)
            tot += B * (M * K + N * K) * es[adt] + B * M * N * es[odt] * (1 + nt)
            n += 1
        return tot, n

    def by_shape(self):
        """{(M, N, K, B, epilogue tag): (launches, ms, TFLOP/s)} — where the step's GEMM time goes."""
        torch.cuda.synchronize()
        acc = {}
        for r in self.records:
            n, ms, fl = acc.get(r[3], (0, 0.0, 0.0))
            acc[r[3]] = (n + 1, ms + r[0].elapsed_time(r[1]), fl + r[2])
        return {k: (n, ms, fl / ms / 1e9 if ms > 0 else 0.0) for k, (n, ms, fl) in acc.items()}


_PROFILER = None


def set_gemm_profiler(p):
    global _PROFILER
    _PROFILER = p


def time_on_stream(fn, warm=2, iters=5):
    """seconds per call, HIP events on the current stream."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def gemm_nt(a, w, out=None, *, out_dtype=None, alpha=1.0, bias=None, lora_t=None, lora_b=None, preact=None,
            act=0, dact_src=None, dact=0, residual=None, accumulate=False, out_split=False, alpha_dev=None):
    """out[M,N] = epilogue(alpha * a[M,K] @ w[N,K]^T).  a, w: same dtype (f32 | bf16), last dim contiguous.
    Batched when a is 3-D ([B,M,K] x [B,N,K] -> [B,M,N], no epilogue tensors).
    Epilogue order: +bias[N] (f32) -> +lora_t[M,r] @ lora_b[r,N] (f32) -> store preact -> act (1 GELU, 2 ReLU,
    3 GELU with `preact` receiving GELU'(v)) -> *act'(dact_src) (1 dGELU(pre), 2 src>0, 3 v *= src) -> +residual
    -> +out (accumulate).
    out_split: the f32 result leaves as its bf16 operand split [M, 3N] = [hi | lo | hi] (what split3(out, "a") would give: the left
    operand of the next tf32x GEMM) without the f32 tensor ever reaching memory; preact / dact_src stay f32.  split_out_ok() says
    whether a shape is served.
    fp16 operands (tf32h engine, `cast16`): f32 results with f32 epilogue tensors, or out_dtype=torch.float16 — fp16 C (saturated), preact, dact_src;
    alpha_dev: a device scalar multiplied into alpha (the 1/s of an operand scaled by `amax_scale`)."""
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
    if out_split:
        _req(out is None and not batched and a.dtype == torch.bfloat16, "gemm_nt: out_split takes bf16 (split) operands, 2-D, and allocates its output")
        out = torch.empty(M, 3 * N, dtype=torch.bfloat16, device=a.device)
    if out is None:
        odt = out_dtype or a.dtype
        out = torch.empty((B, M, N) if batched else (M, N), dtype=odt, device=a.device)
    _req(out.stride(-1) == 1, "gemm_nt: out must be contiguous along N")
    cdt = F32X3 if out_split else dtype_code(out)
    sdt = torch.float32 if out_split else out.dtype
    for t, name in ((preact, "preact"), (dact_src, "dact_src"), (residual, "residual")):
        if t is not None:
            _req(t.dtype == sdt and t.stride(-1) == 1 and tuple(t.shape) == (M, N), f"gemm_nt: bad {name}")
    for t, name in ((bias, "bias"), (lora_t, "lora_t"), (lora_b, "lora_b")):
        if t is not None:
            _req(t.dtype == torch.float32 and t.is_contiguous(), f"gemm_nt: {name} must be contiguous fp32")
    rt = 0
    if lora_t is not None:
        rt = lora_t.shape[1]
        _req(tuple(lora_t.shape) == (M, rt) and tuple(lora_b.shape) == (rt, N), "gemm_nt: bad lora shapes")
    if _PROFILER is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    tail = (ptr(bias), ptr(lora_t), ptr(lora_b), rt, ptr(preact), preact.stride(0) if preact is not None else 0, int(act),
            ptr(dact_src), dact_src.stride(0) if dact_src is not None else 0, int(dact), ptr(residual),
            residual.stride(0) if residual is not None else 0, 1 if accumulate else 0, stream())
    head = (ptr(a), ptr(w), ptr(out), M, N, K, lda, ldw, out.stride(-2), B, sA, sW, out.stride(0) if batched else 0, dtype_code(a), cdt,
            float(alpha))
    if alpha_dev is not None:
        _req(alpha_dev.is_cuda and alpha_dev.dtype == torch.float32 and alpha_dev.numel() >= 1, "gemm_nt: alpha_dev must be a CUDA fp32 scalar")
        rc = lib().gd_gemm_nt_scaled(*head, ptr(alpha_dev), *tail)
    else:
        rc = lib().gd_gemm_nt(*head, *tail)
    if _PROFILER is not None:
        e1.record()
        tag = "".join(c for c, t in (("b", bias), ("l", lora_t), ("p", preact), ("d", dact_src), ("r", residual)) if t is not None)
        _PROFILER.records.append((e0, e1, 2.0 * M * N * K * B, (M, N, K, B, f"{tag}a{act}{'+' if accumulate else ''}", str(out.dtype)[6:], str(a.dtype)[6:])))
    check(rc, "gd_gemm_nt")
    return out


def split_out_ok(M, N, K3):
    """Shapes gemm_nt(..., out_split=True) serves (the persistent kernel's: gd_gemm_nt, c_dtype GD_F32X3)."""
    return M >= 1024 and N >= 256 and N % 8 == 0 and K3 % 64 == 0


def gemm_nt_copy16(a, w, residual, *, bias=None, alpha=1.0, alpha_dev=None, copy_scale=None):
    """fp16 operands: out [M,N] f32 = alpha * alpha_dev * a.w^T + bias + residual, and copy [M,N] fp16 = sat(out * copy_scale[0]) from the same
    epilogue — the residual-stream result plus the operand the next tf32h product takes (gd_gemm_nt_copy16; copy16_ok() says which shapes)."""
    _req(a.dtype == torch.float16 and w.dtype == torch.float16 and a.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and
         residual.dtype == torch.float32 and residual.stride(1) == 1, "gemm_nt_copy16: fp16 operands, fp32 residual")
    (M, K), N = a.shape, w.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    copy = torch.empty(M, N, dtype=torch.float16, device=a.device)
    check(lib().gd_gemm_nt_copy16(ptr(a), ptr(w), ptr(out), M, N, K, a.stride(0), w.stride(0), N, float(alpha), ptr(alpha_dev), ptr(bias),
                                  ptr(residual), residual.stride(0), ptr(copy), N, ptr(copy_scale), stream()), "gd_gemm_nt_copy16")
    return out, copy


def copy16_ok(M, N, K):
    return M >= 1024 and N >= 256 and N % 8 == 0 and K % 64 == 0


# ---- tf32h range bookkeeping (device side; nothing here synchronises) ------------------------------------------------------------------
# _RANGE: the engine's two range counters (saturated / below-normal-range fp16 gradient operands; FinetuneGD.range_report reads them), or
# None.  _SLOTS: per device, the 256-word max-|x| scratch the amax kernels share (every user leaves it zeroed).  _AMAX: scales that a
# producer computed for the tensor it returned (layernorm_bwd(want_amax=True)), keyed by the tensor's address until its consumer takes it.
_RANGE = None
_SLOTS = {}
_AMAX = {}
GRAD_TARGET = 8.0      # |gradient| * s <= 8: 2^13 of fp16 headroom above the block's incoming gradient, 2^17 of full-precision range below


def set_range_counters(t):
    """t: a zeroed int32 CUDA tensor [128] (or None): every scaled fp16 cast of the tf32h engine adds its saturated (words 0-63) / below-normal-range
    (words 64-127) counts as per-block partial sums; `range_totals(t)` adds them up."""
    global _RANGE
    _RANGE = t


def range_totals(t):
    """[saturated, below_normal] from a range-counter tensor (one host read)."""
    return [int(v) for v in t.view(2, 64).sum(1).tolist()]


def _slots(dev):
    t = _SLOTS.get(dev)
    if t is None:
        t = _SLOTS[dev] = torch.zeros(256, dtype=torch.int32, device=dev)
    return t


def amax_register(t, sc):
    # the entry keeps the tensor alive: its address cannot be handed to another tensor while the scale waits for its consumer, so a hit is
    # never a stale scale (an unconsumed entry pins one gradient tensor until amax_clear(): FinetuneGD.backward's end, prepare_trainables)
    _AMAX[t.data_ptr()] = (sc, t)


def amax_take(t):
    """the scale a producer registered for exactly this tensor (popped), or None."""
    rec = _AMAX.pop(t.data_ptr(), None)
    return rec[0] if rec is not None and rec[1].numel() == t.numel() and rec[1].dtype == t.dtype else None


def amax_clear():
    _AMAX.clear()


# LayerNorm row statistics a block's forward took of its INPUT, offered to whoever else normalises the same tensor (a tapped block output is the
# next block's input: vit._TapFn / kp_gather apply `model.norm` to it from these statistics instead of a LayerNorm pass of their own).  Keyed by the
# tensor's address; one entry per tensor, popped by the taker, cleared with the step's other per-step state.
_LN_STATS = {}


def ln_stats_put(x, mean, rstd, eps):
    _LN_STATS[x.data_ptr()] = (mean, rstd, float(eps), x.numel())


def ln_stats_take(x, eps):
    rec = _LN_STATS.pop(x.data_ptr(), None)
    return (rec[0], rec[1]) if rec is not None and rec[3] == x.numel() and rec[2] == float(eps) and rec[0] is not None else None


def ln_stats_clear():
    _LN_STATS.clear()
    _LN_OUT.clear()


# The NEXT block's LayerNorm 1 of a block output, written by the producer that still held the rows on chip (adapter_fused_h_ln): keyed by the output's
# address until the next block's forward takes it (vit._BlockFn.forward) instead of running its own LayerNorm pass.
_LN_OUT = {}


def ln_out_put(x, y, mean, rstd, w):
    _LN_OUT[x.data_ptr()] = (y, mean, rstd, w.data_ptr(), x.numel())


def ln_out_take(x, w):
    rec = _LN_OUT.pop(x.data_ptr(), None)
    return rec[:3] if rec is not None and rec[3] == w.data_ptr() and rec[4] == x.numel() else None


def cast16(x, scale=1.0, scale_dev=None):
    """f32 [rows, K] (rows may be strided) -> fp16 [rows, K] = sat(x * scale * scale_dev[0]): an operand of the tf32h engine's products (fp16
    carries TF32's 11-bit significand).  Forward activations and weights go in unscaled; gradients with the power of two of `amax_scale`
    (those casts are counted in the range counters, when the engine has registered a pair)."""
    _req(x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1, "cast16: a 2-D fp32 CUDA tensor with contiguous rows")
    rows, K = x.shape
    out = torch.empty(rows, K, dtype=torch.float16, device=x.device)
    rng = _RANGE if (scale_dev is not None and _RANGE is not None and _RANGE.device == x.device) else None
    check(lib().gd_cast_f16_ex(ptr(x), ptr(out), rows, K, x.stride(0), float(scale), ptr(scale_dev), ptr(rng), stream()), "gd_cast_f16_ex")
    return out


def amax_scale(x, target=64.0):
    """-> device fp32 [3] = {s, 1/s, unused}: s the power of two with target/2 < max|x| * s <= target (1 for an all-zero tensor, NaN when x holds
    a non-finite element: every product scaled with it is then NaN, as it would be in fp32).  No host round trip: s goes to
    `cast16(scale_dev=r[0:1])`, 1/s to `gemm_nt(alpha_dev=r[1:2])`."""
    _req(x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1, "amax_scale: a 2-D fp32 CUDA tensor with contiguous rows")
    out = torch.empty(3, dtype=torch.float32, device=x.device)
    check(lib().gd_amax_scale(ptr(x), x.shape[0], x.shape[1], x.stride(0), float(target), ptr(out), ptr(_slots(x.device)), stream()), "gd_amax_scale")
    return out


def split3(x, which):
    """f32 [rows, K] (rows may be strided) -> bf16 [rows, 3K]: the hi / lo planes of the 3-term split product (gd_split3).  which = "a":
    left operand [hi | lo | hi]; "w": right operand [hi | hi | lo].  gemm_nt(split3(a, "a"), split3(w, "w"), out_dtype=torch.float32)
    is a . w^T to ~4e-6 relative (TF32: ~3e-4) on the bf16 matrix cores."""
    _req(x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1, "split3: a 2-D fp32 CUDA tensor with contiguous rows")
    rows, K = x.shape
    out = torch.empty(rows, 3 * K, dtype=torch.bfloat16, device=x.device)
    check(lib().gd_split3(ptr(x), ptr(out), rows, K, x.stride(0), {"a": 0, "w": 1}[which], stream()), "gd_split3")
    return out


def gemm_nt_x3(a, w3, **kw):
    """gemm_nt on an fp32 left operand and a PRE-SPLIT right operand w3 = split3(w, "w") (frozen weights are split once): fp32 output and
    fp32 epilogue tensors, three bf16 MFMA products per term."""
    kw.setdefault("out_dtype", torch.float32)
    return gemm_nt(split3(a, "a"), w3, **kw)


def gemm_tn(y, x, out=None, *, alpha=1.0, alpha_dev=None):
    """out[N,K] (fp32) += alpha * y[M,N]^T @ x[M,K]   (weight gradients; fp32 accumulation).  alpha_dev: a device scalar multiplied into alpha
    (the 1/s of a scaled fp16 gradient operand, ops.amax_scale)."""
    _req(y.is_cuda and x.is_cuda and y.stride(-1) == 1 and x.stride(-1) == 1 and y.shape[:-1] == x.shape[:-1],
         "gemm_tn: bad operands")
    batched = y.dim() == 3
    M, N = y.shape[-2:]
    K = x.shape[-1]
    B = y.shape[0] if batched else 1
    if out is None:
        out = torch.zeros((B, N, K) if batched else (N, K), dtype=torch.float32, device=y.device)
    _req(out.dtype == torch.float32 and out.stride(-1) == 1, "gemm_tn: out must be fp32, contiguous rows")
    args = (ptr(y), ptr(x), ptr(out), M, N, K, y.stride(-2), x.stride(-2), out.stride(-2), B, y.stride(0) if batched else 0,
            x.stride(0) if batched else 0, out.stride(0) if batched else 0, dtype_code(y), dtype_code(x), float(alpha))
    rc = lib().gd_gemm_tn(*args, stream()) if alpha_dev is None else lib().gd_gemm_tn_scaled(*args, ptr(alpha_dev), stream())
    check(rc, "gd_gemm_tn")
    return out


VARIANTS = {"vggt": 0, "mast3r": 1}


def pad_teacher_maps(t):
    """[P, hw, hw] -> [P, hw, ldt] view-compatible copy with ldt = hw rounded up to 32 floats (128-byte rows, zero pad): the
    layout the fast cost-volume path reads (it needs 16-byte rows; whole-line rows keep a tile's 512-byte row segments from
    straddling cache lines that a neighbouring tile — on another XCD — fetches again: PMC FETCH_SIZE 1.37x -> ~1.0x of the teacher
    bytes, profiles/README.md round 3).  Done ONCE per cached pair (TeacherTargetCache), never per step."""
    P, hw, w = t.shape
    ldt = (w + 31) // 32 * 32
    if ldt == w and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 128 == 0:
        return t
    out = torch.zeros(P, hw, ldt, dtype=torch.float32, device=t.device)
    out[:, :, :w] = t
    return out


def cost_volume_teacher_stats(t1, t2):
    """Per teacher row {max(rowsum, 1e-8), W, A, 0} (see gd_cost_volume_teacher_stats): depends on the teacher maps only —
    computed once when a pair's targets are cached.  t1, t2 [P, hw, ldt] -> [P, 2, hw, 4] fp32."""
    P, hw, ldt = t1.shape
    t1, t2 = t1.contiguous().float(), t2.contiguous().float()
    out = torch.empty(P, 2, hw, 4, dtype=torch.float32, device=t1.device)
    check(lib().gd_cost_volume_teacher_stats(ptr(t1), ptr(t2), P, hw, ldt, ptr(out), stream()), "gd_cost_volume_teacher_stats")
    return out


class _CostVolumeKL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, t1, t2, m1, m2, variant, tstats, inv1=None, inv2=None, x3=False, h16=None, kcap=0):
        P, hw, C = f1.shape
        f1, f2 = f1.contiguous(), f2.contiguous()
        t1, t2 = t1.contiguous().float(), t2.contiguous().float()
        ldt = t1.shape[-1]
        _req(t1.shape == (P, hw, ldt) and t2.shape == (P, hw, ldt) and ldt >= hw, "cost_volume_kl: teacher maps must be [P, hw, ldt >= hw]")
        m1 = m1.contiguous().to(torch.uint8)
        m2 = m2.contiguous().to(torch.uint8)
        dt = dtype_code(f1)
        loss = torch.empty(P, dtype=torch.float32, device=f1.device)
        stats = torch.empty(P, 2, hw, 4, dtype=torch.float32, device=f1.device)
        ws = torch.empty(lib().gd_cost_volume_kl_workspace_bytes(P, hw, C, dt, 0), dtype=torch.uint8, device=f1.device)
        if tstats is not None:
            _req(tstats.shape == (P, 2, hw, 4) and tstats.dtype == torch.float32 and tstats.is_contiguous(), "cost_volume_kl: bad tstats")
        if inv1 is not None:
            _req(inv1.shape == (P, hw) and inv2.shape == (P, hw) and inv1.dtype == torch.float32 and inv2.dtype == torch.float32 and
                 inv1.is_contiguous() and inv2.is_contiguous(), "cost_volume_kl: inv_norms must be two contiguous fp32 [P, hw] tensors")
            h16_saved = None
            rows = kcap > 0 and tstats is not None and ldt % 4 == 0      # sparse row masks: the kept-row kernel (gd_cost_volume_kl_fwd_rows)

            ctx.kcap = 0

            def fwd(fa, fb, cc, code):
                if rows and (cc * (2 if code else 4)) % 128 == 0 and cc * (2 if code else 4) >= 384:
                    ctx.kcap = kcap      # the backward takes the kept-row form too
                    wsr = torch.empty(lib().gd_cost_volume_kl_rows_workspace_bytes(P, hw, kcap), dtype=torch.uint8, device=f1.device)
                    return lib().gd_cost_volume_kl_fwd_rows(ptr(fa), ptr(fb), ptr(inv1), ptr(inv2), ptr(t1), ptr(t2), ldt, ptr(tstats), ptr(m1), ptr(m2),
                                                            P, hw, cc, kcap, VARIANTS[variant], code, ptr(loss), ptr(stats), ptr(wsr), stream())
                wsd = torch.empty(lib().gd_cost_volume_kl_workspace_bytes(P, hw, cc, code, 0), dtype=torch.uint8, device=f1.device)
                return lib().gd_cost_volume_kl_fwd_prenorm(ptr(fa), ptr(fb), ptr(inv1), ptr(inv2), ptr(t1), ptr(t2), ldt, ptr(tstats), ptr(m1), ptr(m2),
                                                           P, hw, cc, VARIANTS[variant], code, ptr(loss), ptr(stats), ptr(wsd), stream())
            if x3 == "h" and f1.dtype == torch.float32 and C % 8 == 0:
                # tf32h: S from the fp16 copies of the features (TF32's significand); the backward recomputes S from the SAME copies (kept)
                if h16 is not None:
                    a16, b16 = h16[0].contiguous(), h16[1].contiguous()
                    _req(a16.dtype == torch.float16 and a16.shape == f1.shape and b16.shape == f2.shape, "cost_volume_kl: h16 must be the fp16 copies of f1, f2")
                else:
                    a16, b16 = cast16(f1.view(P * hw, C)), cast16(f2.view(P * hw, C))
                rc = fwd(a16, b16, C, 3)
                h16_saved = (a16, b16)
            elif x3 and f1.dtype == torch.float32 and C % 8 == 0:
                # tf32x: S = f1 . f2^T as three bf16 MFMA products of the (hi, lo) splits on the bf16 tile kernel (K = 3C) instead of the
                # exact-f32 MFMA; the row norms stay those of the f32 rows, and the backward (which recomputes S in f32) reads the same
                # stats layout — logZ from this S agrees with its own to ~1e-6
                a3, b3 = split3(f1.view(P * hw, C), "a"), split3(f2.view(P * hw, C), "w")
                ws = torch.empty(lib().gd_cost_volume_kl_workspace_bytes(P, hw, 3 * C, 1, 0), dtype=torch.uint8, device=f1.device)
                rc = lib().gd_cost_volume_kl_fwd_prenorm(ptr(a3), ptr(b3), ptr(inv1), ptr(inv2), ptr(t1), ptr(t2), ldt, ptr(tstats), ptr(m1),
                                                         ptr(m2), P, hw, 3 * C, VARIANTS[variant], 1, ptr(loss), ptr(stats), ptr(ws), stream())
            else:
                rc = fwd(f1, f2, C, dt)
        else:
            rc = lib().gd_cost_volume_kl_fwd(ptr(f1), ptr(f2), ptr(t1), ptr(t2), ldt, ptr(tstats), ptr(m1), ptr(m2), P, hw, C,
                                             VARIANTS[variant], dt, ptr(loss), ptr(stats), ptr(ws), stream())
        check(rc, "gd_cost_volume_kl_fwd")
        # (the fp16 feature copies are saved tensors like the rest — autograd's lifetime and version checks apply to them; ctx keeps plain ints only)
        if inv1 is None:
            h16_saved, ctx.kcap = None, 0
        ctx.has_h16 = h16_saved is not None
        ctx.save_for_backward(f1, f2, t1, t2, m1, m2, stats, *(h16_saved or ()))
        return loss

    @staticmethod
    def backward(ctx, gloss):
        f1, f2, t1, t2, m1, m2, stats = ctx.saved_tensors[:7]
        h = tuple(ctx.saved_tensors[7:9]) if ctx.has_h16 else None
        P, hw, C = f1.shape
        kcap = ctx.kcap
        if kcap and option("cv_bwd_rows"):      # sparse row masks: G only for the kept rows of each direction (gd_cost_volume_kl_bwd_rows)
            code = 3 if h is not None else dtype_code(f1)
            dfull = torch.empty((2 * P, hw, C), dtype=f1.dtype, device=f1.device)
            ws = torch.empty(lib().gd_cost_volume_kl_bwd_rows_workspace_bytes(P, hw, C, kcap, code), dtype=torch.uint8, device=f1.device)
            g = gloss.contiguous().float()
            rc = lib().gd_cost_volume_kl_bwd_rows(ptr(f1), ptr(f2), ptr(h[0]) if h else None, ptr(h[1]) if h else None, ptr(t1), ptr(t2), t1.shape[-1],
                                                  ptr(m1), ptr(m2), P, hw, C, kcap, code, ptr(g), ptr(stats), ptr(dfull[:P]), ptr(dfull[P:]), ptr(ws), stream())
            check(rc, "gd_cost_volume_kl_bwd_rows")
            return dfull[:P], dfull[P:], None, None, None, None, None, None, None, None, None, None, None
        if h is not None:      # tf32h: fp16 S recompute and G contractions, fp32 gradient through the normalisation
            a16, b16 = h
            dfull = torch.empty((2 * P, hw, C), dtype=torch.float32, device=f1.device)
            ws = torch.empty(lib().gd_cost_volume_kl_bwd_h_workspace_bytes(P, hw, C), dtype=torch.uint8, device=f1.device)
            g = gloss.contiguous().float()
            rc = lib().gd_cost_volume_kl_bwd_h(ptr(f1), ptr(f2), ptr(a16), ptr(b16), ptr(t1), ptr(t2), t1.shape[-1], ptr(m1), ptr(m2), P, hw, C,
                                               ptr(g), ptr(stats), ptr(dfull[:P]), ptr(dfull[P:]), ptr(ws), stream())
            check(rc, "gd_cost_volume_kl_bwd_h")
            return dfull[:P], dfull[P:], None, None, None, None, None, None, None, None, None, None, None
        dt = dtype_code(f1)
        # the two halves of ONE buffer: split_pairs' backward hands it on without a concatenation pass (134 MB at the step's size)
        dfull = torch.empty((2 * P, hw, C), dtype=f1.dtype, device=f1.device)
        df1, df2 = dfull[:P], dfull[P:]
        ws = torch.empty(lib().gd_cost_volume_kl_workspace_bytes(P, hw, C, dt, 1), dtype=torch.uint8, device=f1.device)
        g = gloss.contiguous().float()
        rc = lib().gd_cost_volume_kl_bwd(ptr(f1), ptr(f2), ptr(t1), ptr(t2), t1.shape[-1], ptr(m1), ptr(m2), P, hw, C, dt, ptr(g),
                                         ptr(stats), ptr(df1), ptr(df2), ptr(ws), stream())
        check(rc, "gd_cost_volume_kl_bwd")
        return df1, df2, None, None, None, None, None, None, None, None, None, None, None


def kept_row_capacity(hw, kept_rows_max):
    """Row capacity (a multiple of 128) of the kept-row cost-volume kernels for a caller's bound on the kept rows of any (pair, view), or 0 when the
    dense hw x hw sweep is the better form: no bound given, or the two compacted problems (2 x capacity rows) would cover more than the hw rows."""
    if kept_rows_max is None or int(kept_rows_max) <= 0:
        return 0
    kc = (min(int(kept_rows_max), int(hw)) + 127) // 128 * 128
    return kc if 2 * kc <= hw else 0


def cost_volume_kl(f1, f2, t1, t2, m1, m2, variant="vggt", tstats=None, inv_norms=None, x3=False, h16=None, kept_rows_max=None):
    """Fused dense cost-volume KL for P pairs.  f1,f2 [P,hw,C] raw student features (f32|bf16); t1,t2 [P,hw,ldt] teacher
    maps (f32; ldt = hw, or hw padded to a multiple of 4 by `pad_teacher_maps`: the fast path); m1,m2 [P,hw] bool row
    masks; tstats: `cost_volume_teacher_stats(t1, t2)` computed once per cached pair (None: recomputed here, one more pass
    over the maps); inv_norms = (inv1, inv2), fp32 [P, hw] each: 1 / max(||row||, 1e-12) of the feature rows as stored, when their
    producer already took them (`tap_mean(..., with_norm=True)`) — the op then skips its own pass over the features
    x3 (fp32 features with inv_norms): the forward's similarity matrix as a split-precision bf16 product (tf32x engine)
    kept_rows_max (with inv_norms and tstats): a bound on the number of kept rows of any (pair, view) — e.g. the keypoint count behind keypoint-patch
    masks; when it is below half of hw the forward runs as two compacted row problems (gd_cost_volume_kl_fwd_rows) instead of one hw x hw sweep
    -> loss [P] (f32)."""
    if inv_norms is not None:
        kcap = kept_row_capacity(f1.shape[1], kept_rows_max) if tstats is not None else 0
        return _CostVolumeKL.apply(f1, f2, t1, t2, m1, m2, variant, tstats, inv_norms[0].detach(), inv_norms[1].detach(), x3, h16, kcap)
    return _CostVolumeKL.apply(f1, f2, t1, t2, m1, m2, variant, tstats)


F32X3 = 2      # gd_attention_* dtype code: fp32 tensors, products as three bf16 MFMAs of (hi, lo) splits (include/gd_hip.h GD_F32X3)


def attention_fwd(qkv, B, N, H, x3=False):
    """qkv [B*N, 3*H*64] packed (q|k|v, heads inner) -> o [B*N, H*64], lse [B,H,N] (f32).  x3 (fp32 tensors only): the TF32-class
    split-precision kernels instead of the exact-f32 MFMA ones."""
    _req(qkv.is_contiguous() and qkv.shape == (B * N, 3 * H * 64), "attention_fwd: qkv must be [B*N, 3*H*64]")
    o = torch.empty(B * N, H * 64, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    _req(not x3 or qkv.dtype == torch.float32, "attention_fwd: x3 needs fp32 tensors")
    rc = lib().gd_attention_fwd(ptr(qkv), ptr(o), ptr(lse), B, N, H, 64, 64 ** -0.5, F32X3 if x3 else dtype_code(qkv), stream())
    check(rc, "gd_attention_fwd")
    return o, lse


def attention_bwd(qkv, o, dout, lse, B, N, H, vfirst=False, need_dk=True, x3=False):
    """-> dqkv [B*N, 3*H*64] (same dtype as qkv), column blocks (dq, dk, dv) — or (dq, dv, dk) with vfirst.  need_dk=False: the dK
    columns are left unwritten (bf16: the dK/dV kernel then runs its dV half only)."""
    _req(dout.is_contiguous() and dout.shape == o.shape and dout.dtype == qkv.dtype, "attention_bwd: bad dout")
    dqkv = torch.empty_like(qkv)
    npad = (N + 63) // 64 * 64      # workspace: the dQ kernel leaves [-delta | -lse in log2 units] per query for the dK/dV kernel, rows padded to whole 64-query tiles
    delta = torch.empty(2, B, H, npad, dtype=torch.float32, device=qkv.device)
    rc = lib().gd_attention_bwd(ptr(qkv), ptr(o), ptr(dout), ptr(lse), ptr(dqkv), ptr(delta), B, N, H, 64,
                                64 ** -0.5, F32X3 if x3 else dtype_code(qkv), (1 if vfirst else 0) | (0 if need_dk else 2), stream())
    check(rc, "gd_attention_bwd")
    return dqkv


# ----------------------------------------------------------------------------------------------
# views of the two halves of a [2P, ...] batch (view 1 / view 2 of P pairs)
# ----------------------------------------------------------------------------------------------
class _SplitPairs(torch.autograd.Function):
    """x[:P], x[P:] whose backward is ONE concatenation (autograd's own slice backward zero-fills a full-size tensor per
    half, copies, then adds the two)."""

    @staticmethod
    def forward(ctx, x, P):
        ctx.P, ctx.shape, ctx.dt = P, x.shape, x.dtype
        return x[:P], x[P:]

    @staticmethod
    def backward(ctx, g1, g2):
        P, shp = ctx.P, ctx.shape
        if g1 is None and g2 is None:
            return None, None
        if g1 is not None and g2 is not None:
            base = g1._base
            if (base is not None and base is g2._base and base.is_contiguous() and tuple(base.shape) == tuple(shp)
                    and base.dtype == g1.dtype and g1.is_contiguous() and g2.is_contiguous() and g1.data_ptr() == base.data_ptr()
                    and g2.data_ptr() == base.data_ptr() + g1.numel() * g1.element_size()):
                return base, None              # already the halves of one buffer (e.g. cost_volume_kl's backward)
        dev = (g1 if g1 is not None else g2).device
        if g1 is None:
            g1 = torch.zeros((P,) + tuple(shp[1:]), dtype=g2.dtype, device=dev)
        if g2 is None:
            g2 = torch.zeros((shp[0] - P,) + tuple(shp[1:]), dtype=g1.dtype, device=dev)
        return torch.cat([g1, g2], 0), None


def split_pairs(x, P):
    return _SplitPairs.apply(x, P)


# ----------------------------------------------------------------------------------------------
# bottleneck adapter (utils/model.py:7-25), one fused pass
# ----------------------------------------------------------------------------------------------
def adapter_fused_supported(x, bottleneck):
    return bool(lib().gd_adapter_fused_supported(int(x.shape[-1]), int(bottleneck), dtype_code(x)))


def adapter_fused(x, w1, w2, gate_src=None, save_hidden=True):
    """out = x + gate(x @ w1.T) @ w2.T, hidden = gate(x @ w1.T)   (x [M,D] bf16, w1 [64,D], w2 [D,64], same dtype).
    gate_src None: ReLU (the forward, w1 = down, w2 = up); gate_src [M,64]: keep where gate_src > 0 (the backward-to-input
    with x = dOut, w1 = up^T, w2 = down^T, gate_src = the forward's hidden).  -> (out, hidden or None)."""
    M, D = x.shape
    bott = w1.shape[0]
    _req(x.is_contiguous() and w1.is_contiguous() and w2.is_contiguous() and w1.shape == (bott, D) and
         w2.shape == (D, bott) and w1.dtype == x.dtype and w2.dtype == x.dtype, "adapter_fused: layout")
    _req(gate_src is None or (gate_src.is_contiguous() and gate_src.shape == (M, bott) and gate_src.dtype == x.dtype),
         "adapter_fused: gate_src layout")
    out = torch.empty_like(x)
    hidden = torch.empty(M, bott, dtype=x.dtype, device=x.device) if save_hidden else None
    rc = lib().gd_adapter_fused(ptr(x), ptr(w1), ptr(w2), ptr(gate_src), ptr(hidden), ptr(out), M, D, bott,
                                dtype_code(x), stream())
    check(rc, "gd_adapter_fused")
    return out, hidden


def adapter_fused_h_supported(M, D, bottleneck):
    return bool(option("adapter_h_fused")) and bool(lib().gd_adapter_fused_h_supported(int(D), int(bottleneck), int(M)))


def adapter_fused_h(x32, w1, w2, gate_src=None, in_scale=None, alpha_dev=None, copy_scale=None, want_copy=False):
    """The adapter pass of the tf32h engine in one kernel: x32 [M,D] fp32, w1 [64,D] / w2 [D,64] fp16 ->
    out32 = x32 + alpha * gate(fp16(x32 * in_scale) @ w1.T) @ w2.T (fp32), hidden [M,64] fp16 (the gated first product, still times in_scale),
    and with want_copy fp16(out32 * copy_scale).  gate_src None: ReLU (forward); gate_src [M,64] fp16: keep where gate_src > 0 (backward-to-input:
    x32 = dOut, in_scale = s, alpha = 1/s).  The three scales are device scalars (ops.amax_scale slices) or None = 1."""
    M, D = x32.shape
    bott = w1.shape[0]
    _req(x32.dtype == torch.float32 and x32.is_contiguous() and w1.dtype == torch.float16 and w2.dtype == torch.float16 and w1.is_contiguous() and
         w2.is_contiguous() and w1.shape == (bott, D) and w2.shape == (D, bott), "adapter_fused_h: layout")
    _req(gate_src is None or (gate_src.is_contiguous() and gate_src.shape == (M, bott) and gate_src.dtype == torch.float16), "adapter_fused_h: gate_src layout")
    out = torch.empty_like(x32)
    hidden = torch.empty(M, bott, dtype=torch.float16, device=x32.device)
    copy = torch.empty(M, D, dtype=torch.float16, device=x32.device) if want_copy else None
    check(lib().gd_adapter_fused_h(ptr(x32), ptr(w1), ptr(w2), ptr(gate_src), ptr(hidden), ptr(out), ptr(copy), ptr(in_scale), ptr(alpha_dev),
                                   ptr(copy_scale), M, D, bott, stream()), "gd_adapter_fused_h")
    return out, hidden, copy


def adapter_fused_h_ln(x32, w1, w2, ln_w, ln_b, ln_eps):
    """adapter_fused_h (forward form) that also writes the NEXT block's LayerNorm 1 of its result: -> (out32, hidden fp16, y16 = fp16(LN(out32)), mean, rstd)
    (gd_adapter_fused_h_ln)."""
    M, D = x32.shape
    bott = w1.shape[0]
    _req(x32.dtype == torch.float32 and x32.is_contiguous() and w1.dtype == torch.float16 and w2.dtype == torch.float16 and w1.is_contiguous() and
         w2.is_contiguous() and w1.shape == (bott, D) and w2.shape == (D, bott) and ln_w.dtype == torch.float32 and ln_b.dtype == torch.float32, "adapter_fused_h_ln: layout")
    out = torch.empty_like(x32)
    hidden = torch.empty(M, bott, dtype=torch.float16, device=x32.device)
    y16 = torch.empty(M, D, dtype=torch.float16, device=x32.device)
    mean = torch.empty(M, dtype=torch.float32, device=x32.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x32.device)
    check(lib().gd_adapter_fused_h_ln(ptr(x32), ptr(w1), ptr(w2), ptr(hidden), ptr(out), ptr(ln_w), ptr(ln_b), float(ln_eps), ptr(y16), ptr(mean), ptr(rstd),
                                      M, D, bott, stream()), "gd_adapter_fused_h_ln")
    return out, hidden, y16, mean, rstd


# ----------------------------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, *, save_stats=True, out_dtype=None):
    """x [M,D] (rows may be strided) -> y [M,D], mean [M], rstd [M] (fp32; None when save_stats=False)."""
    M, D = x.shape
    y = torch.empty(M, D, dtype=out_dtype or x.dtype, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    rc = lib().gd_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), M, D, x.stride(0),
                                y.stride(0), float(eps), dtype_code(x), dtype_code(y), stream())
    check(rc, "gd_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, dyscale=1.0, dres2=None, cast_scale=None, dy_scale=None, want_amax=False):
    """dx [M,D] (dtype of x) = LN'(dy * dyscale) (+ dres) (+ dres2).  cast_scale (fp32 tensors; a device scalar, e.g. amax_scale(...)[0:1]):
    also returns fp16(dx * cast_scale) — (dx, dx16) — the tf32h engine's next left operand, from the same pass.
    tf32h extras (fp32 x): dy may be fp16 with dy_scale (a device scalar multiplied in: 1/s of a gradient kept in its scaled domain);
    want_amax: the pass also takes max |dx| and the scale for the tensor's consumer is registered under it (`amax_take`)."""
    M, D = x.shape
    if cast_scale is not None or dy_scale is not None or want_amax or dy.dtype == torch.float16:
        _req(x.is_contiguous() and x.dtype == torch.float32 and dy.dtype in (torch.float32, torch.float16) and dy.stride(-1) == 1 and
             all(t is None or (t.is_contiguous() and t.dtype == x.dtype and t.numel() == x.numel()) for t in (dres, dres2)),
             "layernorm_bwd (tf32h form): fp32 x / dres, fp32 or fp16 dy, contiguous")
        dx = torch.empty_like(x)
        dx16 = torch.empty(M, D, dtype=torch.float16, device=x.device) if cast_scale is not None else None
        slots = _slots(x.device) if want_amax else None
        rng = _RANGE if ((dx16 is not None or dy.dtype == torch.float16) and _RANGE is not None and _RANGE.device == x.device) else None      # (an fp16 dy: its saturated entries are counted here, its producer's epilogue has no counter)
        check(lib().gd_layernorm_bwd_ex(ptr(dy), dtype_code(dy), ptr(dy_scale), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dres2), ptr(dx),
                                        ptr(dx16), ptr(cast_scale), ptr(slots), ptr(rng), M, D, dy.stride(0), x.stride(0), float(dyscale), stream()),
              "gd_layernorm_bwd_ex")
        if want_amax:
            sc = torch.empty(3, dtype=torch.float32, device=x.device)
            check(lib().gd_scale_from_amax(ptr(slots), GRAD_TARGET, ptr(sc), stream()), "gd_scale_from_amax")
            amax_register(dx, sc)
        return (dx, dx16) if cast_scale is not None else dx
    _req(x.is_contiguous() and dy.stride(-1) == 1 and all(t is None or (t.is_contiguous() and t.dtype == x.dtype and
                                                                        t.numel() == x.numel()) for t in (dres, dres2)),
         "layernorm_bwd: layout")
    dx = torch.empty_like(x)
    rc = lib().gd_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dres2), ptr(dx), M, D,
                                dy.stride(0), x.stride(0), float(dyscale), dtype_code(x), dtype_code(dy), stream())
    check(rc, "gd_layernorm_bwd")
    return dx


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        x = x.contiguous().float()
        M, D = x.numel() // x.shape[-1], x.shape[-1]
        y = torch.empty_like(x)
        inv = torch.empty(M, dtype=torch.float32, device=x.device)
        check(lib().gd_l2norm_fwd(ptr(x), ptr(y), ptr(inv), M, D, float(eps), stream()), "gd_l2norm_fwd")
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dy = dy.contiguous().float()
        dx = torch.empty_like(y)
        check(lib().gd_l2norm_bwd(ptr(y), ptr(dy), ptr(inv), ptr(dx), inv.numel(), y.shape[-1], stream()), "gd_l2norm_bwd")
        return dx, None


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, p=2, dim=-1) on fp32 rows, HIP forward/backward."""
    return _L2Norm.apply(x, eps)


# ----------------------------------------------------------------------------------------------
# image prep / tokens / conv glue
# ----------------------------------------------------------------------------------------------
def patch_im2col(img, H, W, P, Kp, mean, std, dtype, stride=None):
    """img [B,3,h,w] fp32 in [0,1] -> col [B*gh*gw, Kp]: resize to (H,W) + Normalize + im2col, zero padded.  stride = (sy, sx)
    of the patch conv (default = P: gh = H/P; a smaller stride gives overlapping patches, gh = 1 + (H - P) // sy)."""
    import ctypes
    img = img.contiguous().float()
    B, _, h, w = img.shape
    sy, sx = (P, P) if stride is None else stride
    col = torch.empty(B * (1 + (H - P) // sy) * (1 + (W - P) // sx), Kp, dtype=dtype, device=img.device)
    m3, s3 = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    check(lib().gd_patch_im2col_strided(ptr(img), ptr(col), B, h, w, H, W, P, sy, sx, Kp, m3, s3, dtype_code(col), stream()),
          "gd_patch_im2col_strided")
    return col


def assemble_tokens(patch, cls, pos, B, Np):
    D = patch.shape[-1]
    out = torch.empty(B * (Np + 1), D, dtype=patch.dtype, device=patch.device)
    check(lib().gd_assemble_tokens(ptr(patch), ptr(cls), ptr(pos), ptr(out), B, Np, D, dtype_code(patch), stream()),
          "gd_assemble_tokens")
    return out


def im2col3x3(x, bstride, B, gh, gw, D):
    """x: tensor whose data_ptr is grid element (b=0,y=0,x=0,c=0); batch stride `bstride` elements."""
    col = torch.empty(B * gh * gw, 9 * D, dtype=x.dtype, device=x.device)
    check(lib().gd_im2col3x3(ptr(x), bstride, ptr(col), B, gh, gw, D, dtype_code(x), stream()), "gd_im2col3x3")
    return col


def col2im3x3(dcol, B, gh, gw, D, prefix=0):
    """-> [B, prefix + gh*gw, D]: gradient of the token grid the 3x3 conv read, prefix-token rows zero."""
    dx = torch.empty(B, prefix + gh * gw, D, dtype=dcol.dtype, device=dcol.device)
    if prefix:
        dx[:, :prefix].zero_()
    check(lib().gd_col2im3x3(ptr(dcol), dx[:, prefix:].data_ptr(), (prefix + gh * gw) * D, B, gh, gw, D, dtype_code(dcol),
                             stream()), "gd_col2im3x3")
    return dx


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def kp_gather_fwd(grids, bstride, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, stride=None, pitch=None):
    """grids: list (1..4) of tensors whose data_ptr is the (b=0, token 0 of the grid) element; pitch = tokens per grid line in
    memory (default gw)."""
    out = torch.empty(B, Nk, D, dtype=torch.float32, device=kp.device)
    check(lib().gd_kp_gather_fwd(_ptr_array(grids), len(grids), bstride, dtype_code(grids[0]), ptr(kp), ptr(out), B,
                                 Nk, gh, gw, D, float(sx), float(sy), img_h, img_w, patch, patch if stride is None else stride,
                                 gw if pitch is None else pitch, stream()), "gd_kp_gather_fwd")
    return out


def kp_gather_fwd_ln(grids, means, rstds, sstride, ln_w, ln_b, bstride, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, stride=None, pitch=None):
    """kp_gather_fwd on RAW grids with the final LayerNorm applied where they are sampled: means / rstds = per-grid fp32 statistics tensors whose
    data_ptr is the statistic of the grid's first token (same offset as the grid pointers), sstride = statistics per image."""
    out = torch.empty(B, Nk, D, dtype=torch.float32, device=kp.device)
    check(lib().gd_kp_gather_fwd_ln(_ptr_array(grids), _ptr_array(means), _ptr_array(rstds), len(grids), bstride, sstride, dtype_code(grids[0]),
                                    ptr(ln_w), ptr(ln_b), ptr(kp), ptr(out), B, Nk, gh, gw, D, float(sx), float(sy), img_h, img_w, patch,
                                    patch if stride is None else stride, gw if pitch is None else pitch, stream()), "gd_kp_gather_fwd_ln")
    return out


def kp_gather_bwd(ngrid, kp, dout, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, prefix=0, stride=None, pitch=None):
    """-> list of ngrid fp32 gradient buffers [B, prefix + gh*pitch, D] (prefix-token rows / separator columns stay zero)."""
    pt = gw if pitch is None else pitch
    dg = [torch.zeros(B, prefix + gh * pt, D, dtype=torch.float32, device=kp.device) for _ in range(ngrid)]
    dout = dout.contiguous().float()
    check(lib().gd_kp_gather_bwd(_ptr_array([t[:, prefix:] for t in dg]), ngrid, (prefix + gh * pt) * D, ptr(kp), ptr(dout), B, Nk, gh, gw, D, float(sx), float(sy),
                                 img_h, img_w, patch, patch if stride is None else stride, pt, stream()), "gd_kp_gather_bwd")
    return dg


def lora_bwd_fused(dqv, t, bt_qv, gbt):
    """dqv [M, K] bf16 view (row stride may exceed K), t [M, 8] f32, bt_qv [8, K] bf16, gbt [8, K] f32 (accumulated in place)
    -> dt [M, 8] f32.  One pass over dqv for both LoRA-backward products (gd_lora_bwd_fused)."""
    M, K = dqv.shape
    dt = torch.empty(M, 8, dtype=torch.float32, device=dqv.device)
    check(lib().gd_lora_bwd_fused(ptr(dqv), dqv.stride(0), ptr(t), ptr(bt_qv), ptr(dt), ptr(gbt), M, K, stream()), "gd_lora_bwd_fused")
    return dt


def skinny_tn_mfma(t, x, out):
    """out [8, K] f32 += t [M, 8]^T . x [M, K] (bf16 rows) on the LDS-slab MFMA kernel (gd_lora_bwd_fused with bt = NULL)."""
    M, K = x.shape
    check(lib().gd_lora_bwd_fused(ptr(x), x.stride(0), ptr(t), None, None, ptr(out), M, K, stream()), "gd_lora_bwd_fused")
    return out


def lora_bwd_fused_h(dqv, t, bt_qv, gbt, t_mul=None, out_mul=None, dt_scaled=False):
    """The fused LoRA backward on fp16 operands (tf32h engine): dqv [M, K] fp16 view, t [M, 8] f32, bt_qv [8, K] fp16 or None (then only
    gbt [8, K] f32 += (t * t_mul)^T . dqv * out_mul), t_mul / out_mul: one-element device tensors (the step's gradient scale s and 1 / s) or None
    -> dt [M, 8] f32 = dqv . bt_qv^T * out_mul (None without bt_qv); dt_scaled: dt without out_mul (still in dqv's scaled domain).
    gd_lora_bwd_fused_scaled."""
    M, K = dqv.shape
    dt = torch.empty(M, 8, dtype=torch.float32, device=dqv.device) if bt_qv is not None else None
    check(lib().gd_lora_bwd_fused_scaled(ptr(dqv), dqv.stride(0), ptr(t), ptr(bt_qv), ptr(dt), ptr(gbt), M, K, dtype_code(dqv), ptr(t_mul), ptr(out_mul),
                                         1 if dt_scaled else 0, stream()), "gd_lora_bwd_fused_scaled")
    return dt


def lora_bwd_fused_h_supported(dqv, t, bt_qv, gbt):
    return (option("lora_fused") and dqv.dtype == torch.float16 and (bt_qv is None or (bt_qv.dtype == torch.float16 and bt_qv.is_contiguous()))
            and t.dtype == torch.float32 and t.shape[1] == 8 and t.is_contiguous() and gbt.is_contiguous() and gbt.dtype == torch.float32
            and dqv.stride(1) == 1 and dqv.shape[1] % 256 == 0 and dqv.shape[1] // 256 in (1, 2, 3, 4, 6, 8) and dqv.stride(0) % 8 == 0
            and dqv.data_ptr() % 16 == 0)


def lora_bwd_fused_supported(dqv, t, bt_qv, gbt):
    return (option("lora_fused") and dqv.dtype == torch.bfloat16 and (bt_qv is None or bt_qv.dtype == torch.bfloat16)
            and t.dtype == torch.float32 and t.shape[1] == 8 and t.is_contiguous() and (bt_qv is None or bt_qv.is_contiguous()) and gbt.is_contiguous()
            and dqv.stride(1) == 1 and dqv.shape[1] % 256 == 0 and dqv.shape[1] // 256 in (1, 2, 3, 4, 6, 8) and dqv.stride(0) % 8 == 0)


def conv_weight_pack(weight, dtype, with_wu=False):
    """refine_conv weight [D, D, 3, 3] fp32 -> (wk [D, 9D] with K order (ky, kx, c), wt [D, 9D] = flipped kernel, K order (kx, ky, n),
    wu [9D, D] = wk transposed or None) in `dtype`, one kernel."""
    D = weight.shape[0]
    w = weight.detach().float().contiguous()
    wk = torch.empty(D, 9 * D, dtype=dtype, device=w.device)
    wt = torch.empty(D, 9 * D, dtype=dtype, device=w.device)
    wu = torch.empty(9 * D, D, dtype=dtype, device=w.device) if with_wu else None
    check(lib().gd_conv_weight_pack(ptr(w), ptr(wk), ptr(wt), ptr(wu), D, dtype_code(wk), stream()), "gd_conv_weight_pack")
    return wk, wt, wu


def kp_patch_bwd_det(U, kp, out_dtype, B, Nk, Nt, gh, gw, D, sx, sy, img_h, img_w, patch):
    """U [B*Nk, 9D] (= dfeat . W) -> the token gradient [B, Nt, D] of kp_patch_gather, deterministic, prefix rows zero."""
    out = torch.empty(B, Nt, D, dtype=out_dtype, device=U.device)
    check(lib().gd_kp_patch_bwd_det(ptr(U), ptr(out), dtype_code(out), Nt * D, Nt - gh * gw, ptr(kp), B, Nk, gh, gw, D, float(sx), float(sy),
                                    img_h, img_w, patch, patch, stream()), "gd_kp_patch_bwd_det")
    return out


def kp_gather_bwd_det(kp, dout, scale, out_dtype, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, prefix=0, stride=None, pitch=None):
    """Deterministic interpolate_features backward (no atomics, no zero-fill, no cast pass): -> [B, prefix + gh*pitch, D] of
    `out_dtype`, every element written (prefix rows / separator columns zero).  None when the shape is outside the kernel's range
    (Nk > 1024, D % 8, D > 1024) or GD_GATHER_DET=0: the caller then takes the atomic scatter."""
    pt = gw if pitch is None else pitch
    if not option("gather_det") or Nk > 1024 or D % 8 != 0 or D > 1024:
        return None
    out = torch.empty(B, prefix + gh * pt, D, dtype=out_dtype, device=kp.device)
    dout = dout.contiguous().float()
    check(lib().gd_kp_gather_bwd_det(ptr(out), dtype_code(out), (prefix + gh * pt) * D, prefix, ptr(kp), ptr(dout), float(scale), B, Nk,
                                     gh, gw, D, float(sx), float(sy), img_h, img_w, patch, patch if stride is None else stride, pt,
                                     stream()), "gd_kp_gather_bwd_det")
    return out


def kp_patch_gather(grid, bstride, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, stride=None, pitch=None, half=False):
    """-> [B*Nk, 9*D] (grid dtype): the bilinear mix of the four neighbours' 3x3 input patches, K order (ky, kx, c) — the GEMM
    operand of refine_conv evaluated at the keypoints only (gd_kp_patch_gather).  `grid` is addressed from its data_ptr.
    half (fp32 grid): the taps leave as fp16, the tf32h engine's operand (gd_kp_patch_gather_h)."""
    if half:
        _req(grid.dtype == torch.float32 and D % 4 == 0, "kp_patch_gather(half=True): an fp32 grid with D % 4 == 0")
        out = torch.empty(B * Nk, 9 * D, dtype=torch.float16, device=grid.device)
        check(lib().gd_kp_patch_gather_h(ptr(grid), bstride, ptr(kp), ptr(out), B, Nk, gh, gw, D, float(sx), float(sy), img_h, img_w, patch,
                                         patch if stride is None else stride, gw if pitch is None else pitch, stream()), "gd_kp_patch_gather_h")
        return out
    out = torch.empty(B * Nk, 9 * D, dtype=grid.dtype, device=grid.device)
    check(lib().gd_kp_patch_gather(ptr(grid), bstride, dtype_code(grid), ptr(kp), ptr(out), B, Nk, gh, gw, D, float(sx), float(sy),
                                   img_h, img_w, patch, patch if stride is None else stride, gw if pitch is None else pitch,
                                   stream()), "gd_kp_patch_gather")
    return out


def stack3_rows(src, B, gh, gw, D, src_bstride, src_row0, src_pitch, dtype):
    """-> buf [B*gh*(gw+1) + 2, 3*D] of `dtype` (see gd_stack3_rows); `src` is addressed from its data_ptr."""
    buf = torch.empty(B * gh * (gw + 1) + 2, 3 * D, dtype=dtype, device=src.device)
    check(lib().gd_stack3_rows(ptr(src), ptr(buf), B, gh, gw, D, src_bstride, src_row0, src_pitch, dtype_code(src), dtype_code(buf),
                               stream()), "gd_stack3_rows")
    return buf


def conv_view(buf, rows, D):
    """The overlapping-row GEMM operand of a stack3 buffer: A[r][k] = buf_flat[r*3D + k], [rows, 9D] with row stride 3D."""
    return torch.as_strided(buf, (rows, 9 * D), (3 * D, 1))


def unpitch_tokens(src, B, gh, gw, D, prefix):
    """pitched [B*gh*(gw+1), D] -> [B, prefix + gh*gw, D], prefix rows zero."""
    out = torch.empty(B, prefix + gh * gw, D, dtype=src.dtype, device=src.device)
    check(lib().gd_unpitch_tokens(ptr(src), ptr(out), B, gh, gw, D, prefix, dtype_code(src), stream()), "gd_unpitch_tokens")
    return out


class _TapMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prefix, with_norm, *grids):
        B, Nt, D = grids[0].shape
        gs = [g.contiguous() for g in grids]
        out = torch.empty(B, Nt - prefix, D, dtype=gs[0].dtype, device=gs[0].device)
        ctx.meta = (prefix, len(gs), B, Nt, D)
        ctx.set_materialize_grads(False)      # (the inverse norms / the fp16 copy are not differentiable: no zero-filled [B, hw, D] "gradient" for them)
        if with_norm == 2:      # + the fp16 copy of the rows (tf32h engine: the cost-volume products' operands)
            _req(out.dtype == torch.float32 and D % 8 == 0, "tap_mean(with_norm=2): fp32 taps, D % 8 == 0")
            inv = torch.empty(B, Nt - prefix, dtype=torch.float32, device=gs[0].device)
            o16 = torch.empty(B, Nt - prefix, D, dtype=torch.float16, device=gs[0].device)
            check(lib().gd_tap_mean_norm_fwd_h(_ptr_array(gs), len(gs), Nt * D, prefix, ptr(out), ptr(o16), ptr(inv), B, Nt - prefix, D, stream()),
                  "gd_tap_mean_norm_fwd_h")
            ctx.mark_non_differentiable(inv, o16)
            return out, inv, o16
        if with_norm:
            inv = torch.empty(B, Nt - prefix, dtype=torch.float32, device=gs[0].device)
            check(lib().gd_tap_mean_norm_fwd(_ptr_array(gs), len(gs), Nt * D, prefix, ptr(out), ptr(inv), B, Nt - prefix, D,
                                             dtype_code(out), stream()), "gd_tap_mean_norm_fwd")
            ctx.mark_non_differentiable(inv)
            return out, inv
        check(lib().gd_tap_mean_fwd(_ptr_array(gs), len(gs), Nt * D, prefix, ptr(out), B, Nt - prefix, D,
                                    dtype_code(out), stream()), "gd_tap_mean_fwd")
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        prefix, ng, B, Nt, D = ctx.meta
        if dout is None:
            return (None, None) + (None,) * ng
        dout = dout.contiguous()
        # the ng gradients are identical (dout / ng): one buffer, handed to every grid
        dg = torch.empty(B, Nt, D, dtype=dout.dtype, device=dout.device)
        check(lib().gd_tap_mean_bwd(_ptr_array([dg]), 1, prefix, ptr(dout), B, Nt - prefix, D, 1.0 / ng, dtype_code(dout),
                                    stream()), "gd_tap_mean_bwd")
        return (None, None) + (dg,) * ng


def tap_mean(grids, prefix=1, with_norm=False):
    """mean of 1..4 tap outputs [B, prefix+hw, D], prefix tokens dropped -> contiguous [B, hw, D]; with_norm: -> (mean, inv_norm
    [B, hw] fp32 = 1 / max(||row||, 1e-12) of the rows as stored, not differentiable: an input of cost_volume_kl(inv_norms=...));
    with_norm=2 (fp32 taps): -> (mean, inv_norm, fp16 copy of mean) — cost_volume_kl(x3="h", h16=...)."""
    return _TapMean.apply(prefix, 2 if with_norm == 2 else bool(with_norm), *grids)


def kp_depth(depth, kp):
    """depth [B,H,W] f32, kp [B,Nk,2] (x,y) -> [B,Nk] 3x3 mean depth (extract_kp_depth)."""
    B, H, W = depth.shape
    Nk = kp.shape[1]
    out = torch.empty(B, Nk, dtype=torch.float32, device=kp.device)
    dm, kq = depth.contiguous().float(), kp.contiguous().float()      # keep both alive across the launch
    check(lib().gd_kp_depth(ptr(dm), ptr(kq), ptr(out), B, Nk, H, W, stream()), "gd_kp_depth")
    return out


def patch_mask(kp, H, W, P):
    """kp [B,Nk,2] -> uint8 [B, (H//P)*(W//P)] (get_patch_mask_from_kp_tensor)."""
    B, Nk, _ = kp.shape
    mask = torch.zeros(B, (H // P) * (W // P), dtype=torch.uint8, device=kp.device)
    kq = kp.contiguous().float()
    check(lib().gd_patch_mask(ptr(kq), ptr(mask), B, Nk, H, W, P, stream()), "gd_patch_mask")
    return mask


# ----------------------------------------------------------------------------------------------
# sparse losses
# ----------------------------------------------------------------------------------------------
_CONST = {}


def _const(kind, n, value, device):
    """small constant device vectors (all-ones scales, full keypoint counts), built once per (kind, length, value, device)."""
    key = (kind, int(n), value, device)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full((int(n),), value, dtype=torch.int32 if kind == "i" else torch.float32, device=device)
    return t


class _SmoothAP(torch.autograd.Function):
    """desc [2P, N0, C]: the unit descriptors of view 1 (first P) and view 2 (last P) as the extractor returns them — one buffer in, one gradient
    buffer out (no per-view slices, concatenations or per-tensor padding passes)."""

    @staticmethod
    def forward(ctx, desc, pts1, pts2, counts, variant, thr, temp):
        P2, N0, C = desc.shape
        P = P2 // 2
        N = (N0 + 7) // 8 * 8                                      # gemm_tn wants multiples of 8: zero-pad keypoints
        dev = desc.device
        if counts is None:
            counts = _const("i", P, N0, dev)
        desc = desc.float()
        if N == N0:
            d12 = desc.contiguous()
            pp = torch.stack([pts1, pts2]).float().contiguous()
        else:      # ONE zero-filled buffer for the padded descriptors and points of both views
            buf = torch.zeros(P2 * N * (C + 4), dtype=torch.float32, device=dev)
            d12 = buf[:P2 * N * C].view(P2, N, C)
            pp = buf[P2 * N * C:P2 * N * (C + 3)].view(2, P, N, 3)
            d12[:, :N0] = desc
            pp[0, :, :N0] = pts1
            pp[1, :, :N0] = pts2
        d1, d2 = d12[:P], d12[P:]
        sim = gemm_nt(d1, d2)                                     # [P,N,N] fp32, exact-f32 MFMA
        loss = torch.empty(P, dtype=torch.float32, device=dev)
        dsim = torch.empty_like(sim)
        rows = torch.empty(P, N, 2, dtype=torch.float32, device=dev)
        if variant == "me":      # every pair closer than thr[0] is a positive (src/finetune_timm_me.py:191-220)
            rc = lib().gd_smooth_ap_me(ptr(sim), ptr(pp[0]), ptr(pp[1]), ptr(counts), P, N, float(thr[0]), float(thr[1]),
                                       float(temp), ptr(loss), ptr(dsim), ptr(rows), stream())
            check(rc, "gd_smooth_ap_me")
        else:
            rc = lib().gd_smooth_ap(ptr(sim), ptr(pp[0]), ptr(pp[1]), ptr(counts), P, N, VARIANTS[variant], float(thr),
                                    float(temp), ptr(loss), ptr(dsim), ptr(rows), stream())
            check(rc, "gd_smooth_ap")
        ctx.save_for_backward(d12, dsim)
        ctx.n0 = N0
        return loss

    @staticmethod
    def backward(ctx, g):
        d12, dsim = ctx.saved_tensors
        P, N, _ = dsim.shape
        d1, d2 = d12[:P], d12[P:]
        gs, gst = torch.empty_like(dsim), torch.empty_like(dsim)
        check(lib().gd_scale_and_transpose(ptr(dsim), ptr(g.contiguous().float()), ptr(gs), ptr(gst), P, N, N, stream()), "gd_scale_and_transpose")
        dd = torch.zeros_like(d12)                                 # (gemm_tn accumulates)
        gemm_tn(gst, d2, out=dd[:P])                              # [P,N,C] = (dsim g) d2
        gemm_tn(gs, d1, out=dd[P:])                               # [P,N,C] = (dsim g)^T d1
        return dd[:, :ctx.n0], None, None, None, None, None, None


def smooth_ap_pairs(desc, P, pts3d_1, pts3d_2, counts=None, variant="vggt", thres3d_neg=0.1, temp=0.01, thres3d_pos=5e-3):
    """smooth_ap on the view-major batch the extractor returns: desc [2P, N, C] (view 1 of all pairs, then view 2) -> loss [P]."""
    _req(desc.shape[0] == 2 * P, "smooth_ap_pairs: desc must be [2P, N, C]")
    thr = (thres3d_pos, thres3d_neg) if variant == "me" else thres3d_neg
    return _SmoothAP.apply(desc, pts3d_1, pts3d_2, counts, variant, thr, temp)


def smooth_ap(desc1, desc2, pts3d_1, pts3d_2, counts=None, variant="vggt", thres3d_neg=0.1, temp=0.01, thres3d_pos=5e-3):
    """desc [P,N,C] unit descriptors, pts3d [P,N,3], counts int32 [P] (valid keypoints per pair) -> loss [P].
    variant "vggt" | "mast3r": positives on the diagonal; "me": every pair closer than thres3d_pos (finetune_timm_me.py)."""
    return smooth_ap_pairs(torch.cat([desc1, desc2], 0), desc1.shape[0], pts3d_1, pts3d_2, counts, variant, thres3d_neg, temp, thres3d_pos)


HEAD_KEYS = ("w1", "b1", "ln_w", "ln_b", "w2", "b2")


class _DepthLosses(torch.autograd.Function):
    """depth L1 + intra-view ranking on keypoint features [P,2,N,D] with the DepthAwareFeatureFusion head.
    The loss kernels are fused forward+backward: they emit, per keypoint set, the loss AND its gradients for a
    unit upstream gradient; backward() scales and adds them in one pass (gd_depth_bwd_combine) and contracts."""

    @staticmethod
    def forward(ctx, feats, d1, d2, counts, thr, w1, b1, ln_w, ln_b, w2, b2):
        vm = feats.dim() == 3                  # [2P,N,D] view-major (view 1 of all pairs, then view 2): no big transpose —
        dev = feats.device                     # the head's first layer is row-wise, so only its 128-wide output is re-ordered
        f = feats.contiguous().float()
        if vm:
            P, N, D = feats.shape[0] // 2, feats.shape[1], feats.shape[2]
            u = gemm_nt(f.view(2 * P * N, D), w1.contiguous()).view(2, P, N, 128).transpose(0, 1).contiguous()
        else:
            P, _, N, D = feats.shape
            u = gemm_nt(f.view(P * 2 * N, D), w1.contiguous())         # [P*2*N,128] = W1 f
        dv1, dv2 = d1.contiguous().float(), d2.contiguous().float()
        depth = torch.stack([dv1, dv2], 1)                         # [P,2,N]
        cnt2 = counts.repeat_interleave(2).contiguous() if counts is not None else None
        hp = [t.contiguous().float() for t in (b1, ln_w, ln_b, w2.view(-1), b2)]
        rank = torch.empty(2 * P, dtype=torch.float32, device=dev)
        du_r = torch.empty(2 * P, N, 128, dtype=torch.float32, device=dev)
        hg_r = torch.empty(2 * P, 516, dtype=torch.float32, device=dev)
        ws = torch.empty(lib().gd_pair_rank_workspace_bytes(2 * P), dtype=torch.uint8, device=dev)
        check(lib().gd_pair_rank(ptr(u), ptr(depth), ptr(cnt2), None, 2 * P, N, float(thr), *[ptr(t) for t in hp],
                                 ptr(rank), ptr(du_r), None, ptr(hg_r), ptr(ws), stream()), "gd_pair_rank")
        l1 = torch.empty(P, dtype=torch.float32, device=dev)
        du_l = torch.empty(P, 2, N, 128, dtype=torch.float32, device=dev)
        hg_l = torch.empty(P, 516, dtype=torch.float32, device=dev)
        check(lib().gd_depth_l1(ptr(u), ptr(dv1), ptr(dv2), ptr(counts), None, P, N, *[ptr(t) for t in hp],
                                ptr(l1), ptr(du_l), None, ptr(hg_l), None, stream()), "gd_depth_l1")
        ctx.save_for_backward(f, w1, du_r, du_l, hg_r, hg_l)
        ctx.dims = (P, N, D, vm)
        return l1, rank.view(P, 2).mean(1)

    @staticmethod
    def backward(ctx, g_l1, g_intra):
        f, w1, du_r, du_l, hg_r, hg_l = ctx.saved_tensors
        P, N, D, vm = ctx.dims
        dev = f.device
        du = torch.empty(P * 2 * N, 128, dtype=torch.float32, device=dev)                # rows in the order of f
        hg = torch.empty(516, dtype=torch.float32, device=dev)
        check(lib().gd_depth_bwd_combine(ptr(du_r), ptr(du_l), ptr(hg_r), ptr(hg_l), ptr(g_l1.contiguous().float()), ptr(g_intra.contiguous().float()),
                                         P, N, 1 if vm else 0, ptr(du), ptr(hg), stream()), "gd_depth_bwd_combine")
        df = gemm_nt(du, w1.t().contiguous()).view(f.shape)        # du . W1
        dw1 = gemm_tn(du, f.view(P * 2 * N, D))                    # du^T f  [128, D]
        return (df, None, None, None, None, dw1, hg[0:128], hg[128:256], hg[256:384], hg[384:512].view(1, 128), hg[512:513])


def depth_losses(feats, depth_1, depth_2, head, counts=None, depth_threshold=0.05):
    """feats [P,2,N,D] keypoint features of both views — or [2P,N,D] view-major as the extractor returns them —, depth_k
    [P,N]; head: dict w1,b1,ln_w,ln_b,w2,b2.  -> (depth_l1 [P], intra_rank [P])  (calculate_depth_loss tail)."""
    return _DepthLosses.apply(feats, depth_1, depth_2, counts, depth_threshold, head["w1"], head["b1"], head["ln_w"],
                              head["ln_b"], head["w2"], head["b2"])


class _DepthHead(torch.autograd.Function):
    """Eager DepthAwareFeatureFusion rows: tanh(fusion_layer(f)) for f [M, D] -> [M] (utils/model.py:101-127)."""

    @staticmethod
    def forward(ctx, feats, w1, b1, ln_w, ln_b, w2, b2):
        shp = feats.shape
        f = feats.reshape(-1, shp[-1]).contiguous().float()
        M = f.shape[0]
        w1c = w1.detach().float().contiguous()
        u = gemm_nt(f, w1c)                                                   # [M, 128]
        hp = [t.detach().contiguous().float() for t in (b1, ln_w, ln_b, w2.reshape(-1), b2)]
        out = torch.empty(M, dtype=torch.float32, device=f.device)
        check(lib().gd_depth_head_fwd(ptr(u), M, *[ptr(t) for t in hp], ptr(out), stream()), "gd_depth_head_fwd")
        ctx.save_for_backward(f, w1c, u, *hp)
        ctx.shp = shp
        return out.view(shp[:-1])

    @staticmethod
    def backward(ctx, dout):
        f, w1c, u, *hp = ctx.saved_tensors
        M = f.shape[0]
        dz = dout.reshape(-1).contiguous().float()
        du = torch.empty_like(u)
        hg = torch.zeros(516, dtype=torch.float32, device=f.device)
        check(lib().gd_depth_head_bwd(ptr(u), ptr(dz), M, *[ptr(t) for t in hp], ptr(du), ptr(hg), stream()), "gd_depth_head_bwd")
        df = gemm_nt(du, w1c.t().contiguous()).view(ctx.shp)
        dw1 = gemm_tn(du, f)
        return (df, dw1, hg[0:128].clone(), hg[128:256].clone(), hg[256:384].clone(), hg[384:512].view(1, 128).clone(),
                hg[512:513].clone())


def depth_head(feats, head):
    """feats [..., D] -> tanh(w2 . GELU(LN(W1 f + b1)) + b2) [...]; head: dict w1,b1,ln_w,ln_b,w2,b2."""
    return _DepthHead.apply(feats, head["w1"], head["b1"], head["ln_w"], head["ln_b"], head["w2"], head["b2"])


class _PairRank(torch.autograd.Function):
    """pairwise_logistic_ranking_loss over B keypoint sets, the reference's GLOBAL mean over the valid pairs of all sets
    (utils/losses.py:36-40): per-set means from gd_pair_rank, re-weighted by the per-set valid-pair counts."""

    @staticmethod
    def forward(ctx, feats, depths, thr, w1, b1, ln_w, ln_b, w2, b2):
        B, N, D = feats.shape
        dev = feats.device
        f = feats.contiguous().float()
        w1c = w1.detach().float().contiguous()
        u = gemm_nt(f.view(B * N, D), w1c).view(B, N, 128)
        hp = [t.detach().contiguous().float() for t in (b1, ln_w, ln_b, w2.reshape(-1), b2)]
        ones = torch.ones(B, dtype=torch.float32, device=dev)
        loss = torch.empty(B, dtype=torch.float32, device=dev)
        du = torch.empty(B, N, 128, dtype=torch.float32, device=dev)
        hg = torch.empty(B, 516, dtype=torch.float32, device=dev)
        nbytes = lib().gd_pair_rank_workspace_bytes(B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dd = depths.contiguous().float()
        check(lib().gd_pair_rank(ptr(u), ptr(dd), None, ptr(ones), B, N, float(thr), *[ptr(t) for t in hp], ptr(loss),
                                 ptr(du), None, ptr(hg), ptr(ws), stream()), "gd_pair_rank")
        cnt = ws[(B * 517) * 4:(B * 518) * 4].view(torch.int32).float()         # workspace layout: hg [B,516] | loss_sum [B] | pair_cnt [B]
        wgt = cnt / cnt.sum().clamp_min(1.0)
        ctx.save_for_backward(f, w1c, du, hg, wgt)
        return (loss * wgt).sum()

    @staticmethod
    def backward(ctx, g):
        f, w1c, du, hg, wgt = ctx.saved_tensors
        B, N, D = f.shape
        sc = (g.float() * wgt)
        du = (du * sc.view(B, 1, 1)).view(B * N, 128)
        hgs = (hg * sc[:, None]).sum(0)
        df = gemm_nt(du, w1c.t().contiguous()).view(B, N, D)
        dw1 = gemm_tn(du, f.view(B * N, D))
        return (df, None, None, dw1, hgs[0:128].clone(), hgs[128:256].clone(), hgs[256:384].clone(),
                hgs[384:512].view(1, 128).clone(), hgs[512:513].clone())


def pair_rank_loss(feats, depths, head, depth_threshold=0.0):
    """feats [B,N,D], depths [B,N] -> scalar: mean of log(1 + exp(-sign(d_j - d_i) head(f_j - f_i))) over the pairs with
    |d_j - d_i| > depth_threshold (0 when there is none)."""
    return _PairRank.apply(feats, depths, depth_threshold, head["w1"], head["b1"], head["ln_w"], head["ln_b"], head["w2"],
                           head["b2"])


class _LossCombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ap, depth, intra, kl, counts, weights):
        P = ap.shape[0]
        ts = [t.contiguous().float() for t in (ap, depth, intra, kl)]
        w4 = (ctypes_float_array(weights))
        loss = torch.empty((), dtype=torch.float32, device=ap.device)
        terms = torch.empty(4, P, dtype=torch.float32, device=ap.device)
        cnt = counts.contiguous().to(device=ap.device, dtype=torch.int32) if counts is not None else None
        check(lib().gd_loss_combine_fwd(*[ptr(t) for t in ts], w4, ptr(cnt), P, ptr(loss), ptr(terms), stream()), "gd_loss_combine_fwd")
        ctx.cfg = (tuple(float(w) for w in weights), cnt, P)
        ctx.mark_non_differentiable(terms)
        ctx.set_materialize_grads(False)
        return loss, terms

    @staticmethod
    def backward(ctx, g, _unused=None):
        weights, cnt, P = ctx.cfg
        if g is None:
            return None, None, None, None, None, None
        grads = torch.empty(4, P, dtype=torch.float32, device=g.device)
        check(lib().gd_loss_combine_bwd(ptr(g.contiguous().float()), ctypes_float_array(weights), ptr(cnt), P, ptr(grads), stream()), "gd_loss_combine_bwd")
        return grads[0], grads[1], grads[2], grads[3], None, None


def loss_combine(ap, depth, intra, kl, weights, counts=None):
    """The step's scalar loss = mean over pairs of (w_ap ap + w_depth depth + w_intra intra + w_kl kl), a pair with counts[p] == 0 contributing a
    constant zero (src/finetune_timm_vggt.py:599-616, src/finetune_timm_mast3r.py:604-607, 650-653) -> (loss [], terms [4, P] masked the same way,
    not differentiable).  One launch each way instead of ~20 elementwise / reduce launches on [P]-vectors."""
    return _LossCombine.apply(ap, depth, intra, kl, counts, tuple(weights))


def ctypes_float_array(vals):
    import ctypes
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


# ----------------------------------------------------------------------------------------------
# stand-alone forms of the reference's loss helpers (compat.py); the step uses the fused ops above
# ----------------------------------------------------------------------------------------------
class _SigmoidTemp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, temp):
        xc = x.contiguous().float()
        y = torch.empty_like(xc)
        check(lib().gd_sigmoid_temp(ptr(xc), None, ptr(y), xc.numel(), float(temp), stream()), "gd_sigmoid_temp")
        ctx.save_for_backward(xc)
        ctx.temp = float(temp)
        return y

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        dyc = dy.contiguous().float()
        dx = torch.empty_like(xc)
        check(lib().gd_sigmoid_temp(ptr(xc), ptr(dyc), ptr(dx), xc.numel(), ctx.temp, stream()), "gd_sigmoid_temp")
        return dx, None


def sigmoid_temp(x, temp=1.0):
    return _SigmoidTemp.apply(x, temp)


class _MaskedPatchCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, m1, m2, eps, use_softmax, temperature):
        B, R, C = cost.shape
        c = cost.contiguous().float()
        m1c = m1.contiguous().to(torch.uint8)
        m2c = m2.contiguous().to(torch.uint8) if m2 is not None else None
        _req(m1c.numel() == R and (m2c is None or m2c.numel() == C), "masked_patch_cost: mask sizes")
        y = torch.empty_like(c)
        check(lib().gd_masked_patch_cost_fwd(ptr(c), ptr(m1c), ptr(m2c), ptr(y), B, R, C, float(eps), int(bool(use_softmax)),
                                             float(temperature), stream()), "gd_masked_patch_cost_fwd")
        ctx.save_for_backward(c, y, m1c, m2c)
        ctx.cfg = (float(eps), int(bool(use_softmax)), float(temperature))
        return y

    @staticmethod
    def backward(ctx, dy):
        c, y, m1c, m2c = ctx.saved_tensors
        B, R, C = c.shape
        dyc = dy.contiguous().float()
        dc = torch.empty_like(c)
        eps, sm, temp = ctx.cfg
        check(lib().gd_masked_patch_cost_bwd(ptr(c), ptr(y), ptr(dyc), ptr(m1c), ptr(m2c), ptr(dc), B, R, C, eps, sm, temp,
                                             stream()), "gd_masked_patch_cost_bwd")
        return dc, None, None, None, None, None


def masked_patch_cost(cost, m1, m2=None, eps=1e-8, use_softmax=False, temperature=1.0):
    return _MaskedPatchCost.apply(cost, m1, m2, eps, use_softmax, temperature)


class _KLMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, p, eps):
        _req(t.shape == p.shape, "kl_divergence_map: shapes differ")
        tc, pc = t.contiguous().float(), p.contiguous().float()
        cols = tc.shape[-1]
        rows = tc.numel() // cols
        loss = torch.empty(1, dtype=torch.float32, device=tc.device)
        ws = torch.empty(rows, dtype=torch.float32, device=tc.device)
        check(lib().gd_kl_divergence_map_fwd(ptr(tc), ptr(pc), rows, cols, float(eps), ptr(loss), ptr(ws), stream()),
              "gd_kl_divergence_map_fwd")
        ctx.save_for_backward(tc, pc)
        ctx.eps = float(eps)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        tc, pc = ctx.saved_tensors
        cols = tc.shape[-1]
        rows = tc.numel() // cols
        gt, gp = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dt = torch.empty_like(tc) if gt else None
        dp = torch.empty_like(pc) if gp else None
        gg = g.reshape(1).contiguous().float()
        check(lib().gd_kl_divergence_map_bwd(ptr(tc), ptr(pc), ptr(gg), rows, cols, ctx.eps, ptr(dt), ptr(dp), stream()),
              "gd_kl_divergence_map_bwd")
        return dt, dp, None


def kl_divergence_map(t, p, eps=1e-8):
    return _KLMap.apply(t, p, eps)


# ----------------------------------------------------------------------------------------------
# optimiser
# ----------------------------------------------------------------------------------------------
def clip_adamw_step(params, grads, exp_avg, exp_avg_sq, step, lr=1e-5, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-8,
                    max_norm=1.0, grad_scale=1.0, ranges=None):
    """In place on the flat fp32 buffers; returns the pre-clip global grad norm (device scalar).  `ranges` = [(a, b), ...]
    restricts the UPDATE (moments, decay, step) to those element ranges — parameters outside them are left untouched, as
    torch.optim.AdamW leaves a parameter whose .grad is None; the norm is always taken over the whole buffer (their
    gradient slice is zero)."""
    norm = torch.empty(1, dtype=torch.float32, device=params.device)
    ws = torch.empty(lib().gd_adamw_workspace_bytes(), dtype=torch.uint8, device=params.device)
    n = params.numel()
    if not ranges or (len(ranges) == 1 and tuple(ranges[0]) == (0, n)):
        check(lib().gd_clip_adamw_step(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), n, int(step),
                                       lr, weight_decay, betas[0], betas[1], eps, max_norm, grad_scale, ptr(norm), ptr(ws),
                                       stream()), "gd_clip_adamw_step")
        return norm
    check(lib().gd_clip_adamw_ranges(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), n, int(step), lr, weight_decay,
                                     betas[0], betas[1], eps, max_norm, grad_scale, ptr(norm), ptr(ws),
                                     (ctypes_long_array([x for ab in ranges for x in ab])), len(ranges), stream()),
          "gd_clip_adamw_ranges")
    return norm


def ctypes_long_array(vals):
    import ctypes
    return (ctypes.c_long * len(vals))(*[int(v) for v in vals])


def cast(x, dtype, scale=1.0):
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(lib().gd_cast(ptr(x), ptr(out), x.numel(), float(scale), dtype_code(x), dtype_code(out), stream()), "gd_cast")
    return out
