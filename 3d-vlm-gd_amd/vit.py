"""Student ViT on HIP kernels behind the timm `VisionTransformer` attribute surface.

What the reference touches on the timm model keeps working: `model.blocks` (indexable, sliceable,
item-assignable), `blk.attn.qkv` (`nn.Linear`, replaceable by `_LoRA_qkv`), blocks replaceable by
`BlockWithAdapter`, `patch_embed.patch_size` / `.proj`, `norm`, `num_prefix_tokens`,
`_intermediate_layers(x, n)`, `forward_features(x)`, `parameters()` (src/finetune_timm_vggt.py:106-162,
276, 317-320).  Parameter names follow timm / the in-tree DINOv2 ViT (vggt/layers/vision_transformer.py),
so those state_dicts load.

Each transformer block — including a LoRA-wrapped qkv and an Adapter wrapper discovered on the module
list — runs as ONE autograd.Function over the C ABI: LN -> QKV GEMM (+bias, + rank-r LoRA epilogue) ->
flash attention -> proj GEMM (+residual) -> LN -> fc1 GEMM (+GELU) -> fc2 GEMM (+residual) -> adapter.
The backbone is frozen: the backward computes dX only (pre-transposed frozen weights, LayerScale folded
into proj/fc2 at plan time) plus the LoRA / adapter weight gradients, and blocks that see no
grad-requiring input run forward-only.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .options import option
from .model import Adapter, BlockWithAdapter, _LoRA_qkv  # noqa: F401


def _dt(name):
    # "tf32x" / "tf32h": fp32 storage and arithmetic everywhere except the matrix products, which take their operands as 3-term bf16 splits
    # (ops.split3: ~4e-6 products) resp. as fp16 (ops.cast16: TF32's 11-bit significand, one MFMA per term)
    return {"f32": torch.float32, "bf16": torch.bfloat16, "tf32x": torch.float32, "tf32h": torch.float32, torch.float32: torch.float32,
            torch.bfloat16: torch.bfloat16}[name]


def _opa(x, fmt, sc=None):
    """left operand of a product in the operand format `fmt`: "x3" -> [hi | lo | hi] bf16 planes; "h" -> fp16 (times the power-of-two scale
    sc[0] of ops.amax_scale for a gradient tensor: the consuming GEMM then takes alpha_dev = sc[1:2])."""
    return ops.split3(x, "a") if fmt == "x3" else ops.cast16(x, scale_dev=None if sc is None else sc[0:1])


def _opw(w, fmt):
    """right operand (a weight): "x3" -> [hi | hi | lo] bf16 planes; "h" -> fp16."""
    return ops.split3(w, "w") if fmt == "x3" else ops.cast16(w)


def _mm(x, plan, key, xs=None, sc=None, **kw):
    """x . W^T with the plan's frozen weight `key`: one GEMM in the f32 / bf16 engines; in the tf32x / tf32h engines the weight was put into
    the operand format once (plan['x3'] = "x3" | "h") and the activation goes in on the way (xs: its formatted copy, when the caller already
    has it) — fp32 accumulation / output / epilogue.  sc (tf32h, gradients): the ops.amax_scale triple the left operand was / is scaled with."""
    fmt = plan.get("x3")
    if fmt:
        if not kw.get("out_split") and kw.get("out_dtype") is None:
            kw["out_dtype"] = torch.float32
        if fmt == "h" and sc is not None and kw.get("out_dtype") != torch.float16:      # an fp16 result stays in the scaled domain
            kw["alpha_dev"] = sc[1:2]
        return ops.gemm_nt(xs if xs is not None else _opa(x, fmt, sc), plan[key], **kw)
    return ops.gemm_nt(x, plan[key], **kw)


class GDAttention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=True, proj_bias=True):
        super().__init__()
        assert dim % num_heads == 0 and dim // num_heads == 64, "HIP attention is built for head_dim 64"
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)


class GDMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)


class GDLayerScale(nn.Module):
    def __init__(self, dim, init_values):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class GDBlock(nn.Module):
    """Parameter container of one pre-LN block; executed by `run_block`."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, init_values=None, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = GDAttention(dim, num_heads)
        self.ls1 = GDLayerScale(dim, init_values) if init_values else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = GDMlp(dim, int(dim * mlp_ratio))
        self.ls2 = GDLayerScale(dim, init_values) if init_values else nn.Identity()
        self._plan = None
        self.split3 = False          # tf32x / tf32h engines (set by GDViT to "x3" / "h"): frozen weights are kept in that operand format

    def forward(self, x):
        return run_block(self, x)

    # ---- frozen-weight plan: casted / folded / pre-transposed copies, built once per dtype ----
    def plan(self, dtype):
        x3 = (("x3" if self.split3 is True else self.split3) or "") if dtype == torch.float32 else ""
        if self._plan is not None and self._plan["dtype"] == dtype and self._plan["x3"] == x3:
            return self._plan
        base = self.attn.qkv.qkv if hasattr(self.attn.qkv, "linear_a_q") else self.attn.qkv      # any _LoRA_qkv-shaped wrapper
        dev = base.weight.device
        D = base.in_features

        def f32(t):
            return t.detach().float().contiguous()

        def zeros(n):
            return torch.zeros(n, dtype=torch.float32, device=dev)

        g1 = f32(self.ls1.gamma) if isinstance(self.ls1, GDLayerScale) else None
        g2 = f32(self.ls2.gamma) if isinstance(self.ls2, GDLayerScale) else None
        wproj, bproj = f32(self.attn.proj.weight), f32(self.attn.proj.bias) if self.attn.proj.bias is not None else zeros(D)
        wfc2, bfc2 = f32(self.mlp.fc2.weight), f32(self.mlp.fc2.bias) if self.mlp.fc2.bias is not None else zeros(D)
        if g1 is not None:
            wproj, bproj = wproj * g1[:, None], bproj * g1
        if g2 is not None:
            wfc2, bfc2 = wfc2 * g2[:, None], bfc2 * g2
        wqkv, w1 = f32(base.weight), f32(self.mlp.fc1.weight)

        def both(w):
            if x3:      # [N, 3K] bf16 planes [hi | hi | lo] of W and of W^T (ops.split3), or their fp16 casts
                return _opw(w.contiguous(), x3), _opw(w.t().contiguous(), x3)
            return w.to(dtype).contiguous(), w.t().to(dtype).contiguous()

        p = {"dtype": dtype, "x3": x3, "D": D, "H": self.attn.num_heads, "eps1": self.norm1.eps, "eps2": self.norm2.eps,
             "ln1_w": f32(self.norm1.weight), "ln1_b": f32(self.norm1.bias),
             "ln2_w": f32(self.norm2.weight), "ln2_b": f32(self.norm2.bias),
             "bqkv": f32(base.bias) if base.bias is not None else zeros(3 * D), "bproj": bproj.contiguous(),
             "b1": f32(self.mlp.fc1.bias) if self.mlp.fc1.bias is not None else zeros(w1.shape[0]),
             "b2": bfc2.contiguous()}
        p["wqkv"], p["wqkv_t"] = both(wqkv)
        # the attention backward hands back (dq, dv, dk): W^T with its K blocks in that order for dX = dqkv . W
        wqvk_t = torch.cat([wqkv[:D], wqkv[2 * D:], wqkv[D:2 * D]], 0).t().contiguous()
        p["wqkv_t_qvk"] = _opw(wqvk_t, x3) if x3 else wqvk_t.to(dtype).contiguous()
        p["wproj"], p["wproj_t"] = both(wproj)
        p["w1"], p["w1_t"] = both(w1)
        p["w2"], p["w2_t"] = both(wfc2)
        self._plan = p
        return p


def _unwrap(blk):
    """-> (GDBlock, lora module or None, adapter module or None)."""
    adapter = None
    if isinstance(blk, BlockWithAdapter) or (hasattr(blk, "block") and hasattr(blk, "adapter")):
        adapter, blk = blk.adapter, blk.block
    if not isinstance(blk, GDBlock):
        raise ops._lib.GdHipError(f"cannot fuse block of type {type(blk).__name__}")
    q = blk.attn.qkv
    lora = q if hasattr(q, "linear_a_q") else None
    return blk, lora, adapter


class BlockGradGate:
    """When may a block's slices of the flat gradient buffer be exchanged?  When the LAST backward node the block owes this step has run — not the
    first.  geometry="reference" runs two `forward_all` passes per step through the same blocks (the 80 x 80 keypoint grid and the cost grid:
    FinetuneGD.training_step), so every block has two backward nodes per `loss.backward()` and both accumulate into the same slices (`gemm_tn(out=)`
    is +=): an all-reduce started after the first would race with the second's accumulation and exchange its share twice.  `_BlockFn.forward`
    calls `armed()` for every node it records, `_BlockFn.backward` calls `node_done()`; the hook (dp.OverlappedGradReducer._block_done) fires at zero.
    A node whose backward never runs (its output did not reach the loss) leaves the count above zero: the hook then never fires and the slices
    travel with the reducer's late ranges after the backward (`remaining_late`) — late, never wrong."""

    def __init__(self, hook, block_index, spans):
        self.hook, self.block_index, self.spans, self.pending, self.fired = hook, block_index, spans, 0, False

    def armed(self):
        if self.fired:
            raise RuntimeError("BlockGradGate: a forward through a block whose gradient slices were already handed to the exchange this step")
        self.pending += 1

    def node_done(self):
        """-> True when this was the block's last outstanding backward node (the hook has then been called)."""
        self.pending -= 1
        if self.pending == 0 and not self.fired:
            self.fired = True
            self.hook(self.block_index, self.spans)
            return True
        return False


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, plan, B, Nt, a_q, b_q, a_v, b_v, down, up):
        T, D, H = plan["dtype"], plan["D"], plan["H"]
        M = B * Nt
        x = x.contiguous()
        need = x.requires_grad or any(t is not None and t.requires_grad for t in (a_q, b_q, a_v, b_v, down, up))
        h16 = {"out_dtype": torch.float16} if plan["x3"] == "h" else {}      # tf32h: LN(x) only ever feeds matrix products — written as their fp16 operand
        pre1 = ops.ln_out_take(x, plan["ln1_w"]) if plan["x3"] == "h" else None     # the block above left this block's LayerNorm 1 (its adapter kernel: options.adapter_ln)
        if pre1 is not None:
            y1, mean1, rstd1 = pre1
        else:
            y1, mean1, rstd1 = ops.layernorm_fwd(x, plan["ln1_w"], plan["ln1_b"], plan["eps1"], save_stats=need, **h16)
        if need and plan.get("offer_ln_stats"):      # this block's input is a tap: its row statistics serve the tap's `model.norm` too (_TapFn)
            ops.ln_stats_put(x, mean1, rstd1, plan["eps1"])
        t = at = bt = None
        tw = plan.get("tw")           # per-step pack of the trainable weights (GDViT.prepare_trainables), or None
        if a_q is not None:
            r = a_q.shape[0]
            if tw is not None:
                at, bt, at_T = tw["at"], tw["bt"], tw["at_T"]
            else:
                at = torch.cat([a_q, a_v], 0).detach()                              # [2r, D] fp32
                bt = torch.zeros(2 * r, 3 * D, dtype=torch.float32, device=x.device)
                bt[:r, :D] = b_q.detach().t()
                bt[r:, 2 * D:] = b_v.detach().t()
                at_T = at.to(T).contiguous()
        fmt = plan["x3"]
        y1s = (y1 if fmt == "h" else _opa(y1, fmt)) if fmt else None    # tf32x / tf32h: ONE formatted copy of LN1(x) feeds the LoRA-A and the QKV GEMM
        if a_q is not None:
            if plan["x3"]:     # [M, 2r] on the streaming N <= 8 bf16 kernel over the 3K-wide operands (the fp32 tile kernel spends a 128-wide tile on 8 columns)
                t = ops.gemm_nt(y1s, tw["at_w3"] if tw is not None and "at_w3" in tw else _opw(at.contiguous(), fmt), out_dtype=torch.float32)
            else:
                t = ops.gemm_nt(y1, at_T, out_dtype=torch.float32)  # [M, 2r]
        if fmt == "h":     # tf32h: q / k / v, the attention output and their gradients live as fp16 — they are operands of matrix products only
            qkv = _mm(y1, plan, "wqkv", xs=y1s, bias=plan["bqkv"], lora_t=t, lora_b=bt, out_dtype=torch.float16)
            o, lse = ops.attention_fwd(qkv, B, Nt, H)
            x1 = _mm(None, plan, "wproj", xs=o, bias=plan["bproj"], residual=x)
        else:
            qkv = _mm(y1, plan, "wqkv", xs=y1s, bias=plan["bqkv"], lora_t=t, lora_b=bt)
            o, lse = ops.attention_fwd(qkv, B, Nt, H, x3=bool(fmt))
            x1 = _mm(o, plan, "wproj", bias=plan["bproj"], residual=x)
        y2, mean2, rstd2 = ops.layernorm_fwd(x1, plan["ln2_w"], plan["ln2_b"], plan["eps2"], save_stats=need, **h16)
        # tf32x: fc1 writes GELU(.) directly as the split left operand of fc2 (no f32 [M, 4D] round trip + split pass)
        # (tf32h: as fp16, and the stored GELU'(.) — a factor of an elementwise product in the backward — as fp16 too)
        hs = bool(fmt) and ops.split_out_ok(M, plan["w1"].shape[0], plan["w1"].shape[1])
        pre = torch.empty(M, plan["w1"].shape[0], dtype=torch.float16 if (fmt == "h" and hs) else T, device=x.device) if need else None
        hkw = {} if not hs else {"out_split": True} if fmt == "x3" else {"out_dtype": torch.float16}
        h = _mm(None if fmt == "h" else y2, plan, "w1", xs=y2 if fmt == "h" else None, bias=plan["b1"], act=3, preact=pre, **hkw)   # pre <- GELU'(fc1 output): all the backward needs
        x2a = None
        ad_h = fmt == "h" and down is not None and ops.adapter_fused_h_supported(M, D, down.shape[0])      # the fused fp16-operand adapter kernel
        if fmt == "h" and hs and down is not None and not ad_h and ops.copy16_ok(M, D, plan["w2"].shape[1]):
            # tf32h: fc2 writes the residual-stream result and, from the same epilogue, the fp16 copy the adapter's down projection takes
            x2, x2a = ops.gemm_nt_copy16(h, plan["w2"], x1, bias=plan["b2"])
        else:
            x2 = _mm(None if hs else h, plan, "w2", xs=h if hs else None, bias=plan["b2"], residual=x1)
        out, hd, hd16 = x2, None, None
        if down is not None:
            down_T = tw["down_T"] if tw is not None else down.detach().to(T).contiguous()
            up_T = tw["up_T"] if tw is not None else up.detach().to(T).contiguous()
            if ops.adapter_fused_supported(x2, down.shape[0]):
                out, hd = ops.adapter_fused(x2, down_T, up_T, save_hidden=need)
            elif ad_h:      # tf32h: one pass — x2 read once (rounded to fp16 in flight), both products on the fp16 MFMA, fp32 residual add
                w_dn = tw["down_w3"] if tw is not None and "down_w3" in tw else _opw(down_T, fmt)
                w_up = tw["up_w3"] if tw is not None and "up_w3" in tw else _opw(up_T, fmt)
                nl = plan.get("next_ln")
                if nl is not None:      # ... and the NEXT block's LayerNorm 1 of the result, from the rows the kernel still holds (that block takes it from the registry)
                    out, hd16, y1n, mean_n, rstd_n = ops.adapter_fused_h_ln(x2, w_dn, w_up, *nl)
                    ops.ln_out_put(out, y1n, mean_n, rstd_n, nl[0])
                else:
                    out, hd16, _ = ops.adapter_fused_h(x2, w_dn, w_up)
                hd = hd16
            elif plan["x3"]:        # tf32x: both projections as split-precision products on the bf16 kernels (the fp32 tile kernel: 2 x 370 us)
                w_dn = tw["down_w3"] if tw is not None and "down_w3" in tw else _opw(down_T, fmt)
                w_up = tw["up_w3"] if tw is not None and "up_w3" in tw else _opw(up_T, fmt)
                if x2a is None:
                    x2a = _opa(x2, fmt)
                hd = ops.gemm_nt(x2a, w_dn, act=2, out_dtype=torch.float32)
                hda = _opa(hd, fmt)
                out = ops.gemm_nt(hda, w_up, residual=x2, out_dtype=torch.float32)
                if fmt == "h" and need:      # the backward contracts these two again (adapter weight gradients): keep the fp16 copies, not x2
                    x2, hd16 = x2a, hda
            else:
                hd = ops.gemm_nt(x2, down_T, act=2)
                out = ops.gemm_nt(hd, up_T, residual=x2)
        if need:
            ctx.plan, ctx.dims, ctx.tw = plan, (B, Nt), tw
            ctx.gate = tw.get("on_grads") if tw is not None else None      # (data parallelism: this node is one the block owes before its slices may travel)
            if ctx.gate is not None:
                ctx.gate.armed()
            ctx.has_lora, ctx.has_ad = a_q is not None, down is not None
            ctx.hd16 = hd16
            ctx.save_for_backward(x, mean1, rstd1, y1, t, at, bt, qkv, o, lse, x1, mean2, rstd2, pre, x2, hd,
                                  down, up)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x, mean1, rstd1, y1, t, at, bt, qkv, o, lse, x1, mean2, rstd2, pre, x2, hd, down, up) = ctx.saved_tensors
        plan, (B, Nt) = ctx.plan, ctx.dims
        T, D, H = plan["dtype"], plan["D"], plan["H"]
        dout = dout.contiguous().to(T)
        g_down = g_up = g_aq = g_bq = g_av = g_bv = None
        dx2, dx2a = dout, None
        tw = ctx.tw
        fmt = plan["x3"]
        # tf32h: ONE power-of-two scale per BLOCK, from the incoming gradient's maximum (on the device), carries every gradient operand of
        # the block into fp16's range: |dout| * s <= 8 leaves 2^13 of headroom above and 2^17 of full-precision range below.  The maximum
        # comes for free: the LayerNorm backward that PRODUCED dout (the block above, or a tap's) took it on its way out and registered the
        # scale under the tensor (ops.layernorm_bwd(want_amax=True)); only a gradient that arrives from elsewhere (the loss side of the top
        # block of each backward graph) costs a pass of its own.  A non-finite dout makes s NaN: every gradient of the block is then NaN.
        sc = None
        if fmt == "h":
            sc = ops.amax_take(dout)
            if sc is None:
                sc = ops.amax_scale(dout.view(-1, D), ops.GRAD_TARGET)
        # the four weight-gradient accumulators of the block come out of ONE zero-filled buffer
        bott = down.shape[0] if ctx.has_ad else 0
        r2 = at.shape[0] if ctx.has_lora else 0
        direct = tw is not None and "g_at" in tw       # weight gradients accumulate straight into the flat gradient buffer
        if direct:
            z_up, z_down, z_bt, z_at = tw["g_up"], tw["g_down"], tw["g_bt"], tw["g_at"]
        else:
            zb = torch.zeros(2 * bott * D + r2 * 3 * D, dtype=torch.float32, device=dout.device)
            z_up, z_down = zb[:D * bott].view(D, bott), zb[D * bott:2 * D * bott].view(bott, D)
            z_bt = zb[2 * D * bott:2 * D * bott + r2 * 2 * D].view(r2, 2 * D)     # t^T [dq | dv]
            z_at = zb[2 * D * bott + r2 * 2 * D:].view(r2, D)
        if ctx.has_ad:
            up_tT = tw["up_tT"] if tw is not None else up.detach().t().to(T).contiguous()
            down_tT = tw["down_tT"] if tw is not None else down.detach().t().to(T).contiguous()
            ad_h = fmt == "h" and hd.dtype == torch.float16 and ops.adapter_fused_h_supported(dout.shape[0], D, bott)
            if ops.adapter_fused_supported(dout, bott):
                dx2, dhp = ops.adapter_fused(dout, up_tT, down_tT, gate_src=hd)                   # dX and d(hidden) [M, 64]
            elif ad_h:
                # dX = dOut + ((dOut s . up) * [h > 0]) . down / s in one pass; d(hidden) leaves as fp16 (times s), dX also as the scaled fp16
                # operand of the fc2 backward; the two weight gradients contract the fp32 dOut / x2 against the fp16 hidden tiles
                w_ut = tw["up_tw3"] if tw is not None and "up_tw3" in tw else _opw(up_tT, fmt)
                w_dt = tw["down_tw3"] if tw is not None and "down_tw3" in tw else _opw(down_tT, fmt)
                dx2, dhpa, dx2a = ops.adapter_fused_h(dout.view(-1, D), w_ut, w_dt, gate_src=hd, in_scale=sc[0:1], alpha_dev=sc[1:2], copy_scale=sc[0:1],
                                                      want_copy=True)
                g_up = ops.gemm_tn(dout.view(-1, D), hd, out=z_up, alpha_dev=sc[1:2])             # [D, 64]  (dOut rounded to fp16 under s inside the kernel)
                g_down = ops.gemm_tn(dhpa, x2, out=z_down, alpha_dev=sc[1:2])                     # [64, D]
            elif plan["x3"]:
                w_ut = tw["up_tw3"] if tw is not None and "up_tw3" in tw else _opw(up_tT, fmt)
                w_dt = tw["down_tw3"] if tw is not None and "down_tw3" in tw else _opw(down_tT, fmt)
                ad = None if sc is None else sc[1:2]
                douta = _opa(dout.view(-1, D), fmt, sc)
                dhp = ops.gemm_nt(douta, w_ut, dact_src=hd, dact=2, out_dtype=torch.float32, alpha_dev=ad)
                dhpa = _opa(dhp, fmt, sc)
                if fmt == "h" and ops.copy16_ok(dhpa.shape[0], D, dhpa.shape[1]):      # dx2 and its scaled fp16 copy (the fc2 backward's operand) at once
                    dx2, dx2a = ops.gemm_nt_copy16(dhpa, w_dt, dout.view(-1, D), alpha_dev=ad, copy_scale=sc[0:1])
                else:
                    dx2 = ops.gemm_nt(dhpa, w_dt, residual=dout.view(-1, D), out_dtype=torch.float32, alpha_dev=ad)
            else:
                dhp = ops.gemm_nt(dout, up_tT, dact_src=hd, dact=2)                               # [M, 64]
                dx2 = ops.gemm_nt(dhp, down_tT, residual=dout)
            if ad_h:
                pass      # (weight gradients taken above)
            elif fmt == "h" and ctx.hd16 is not None and x2.dtype == torch.float16:
                # the weight gradients on the fp16 MFMA kernel from the operands at hand: (dout s)^T hd and (dhp s)^T x2, times 1/s on the device
                g_up = ops.gemm_tn(douta, ctx.hd16, out=z_up, alpha_dev=sc[1:2])                  # [D, 64]
                g_down = ops.gemm_tn(dhpa, x2, out=z_down, alpha_dev=sc[1:2])                     # [64, D]
                del douta, dhpa
            else:
                g_up = ops.gemm_tn(dout, hd, out=z_up)                                            # [D, 64]
                g_down = ops.gemm_tn(dhp, x2, out=z_down)                                         # [64, D]
        hs = bool(fmt) and ops.split_out_ok(dx2.shape[0], plan["w2_t"].shape[0], plan["w2_t"].shape[1]) and (fmt != "h" or pre.dtype == torch.float16)
        hkw = {} if not hs else {"out_split": True} if fmt == "x3" else {"out_dtype": torch.float16}      # (tf32h: fp16, still times s)
        dpre = _mm(dx2, plan, "w2_t", xs=dx2a, sc=sc, dact_src=pre, dact=3, **hkw)                # [M, 4D] (x stored GELU')
        # tf32h: the two dX products that feed a LayerNorm backward leave as fp16 IN THE SCALED DOMAIN (half the C bytes written and read
        # again; the LayerNorm backward multiplies 1/s back in and does its arithmetic in fp32)
        h16dy = fmt == "h" and hs and bool(option("h16_dy"))
        dy2 = _mm(None if hs else dpre, plan, "w1_t", xs=dpre if hs else None, sc=sc, **({"out_dtype": torch.float16} if h16dy else {}))
        del dpre
        if fmt == "h":      # the LN backward also writes fp16(dx1 * s), the left operand of the projection's backward; do leaves as fp16, times s
            dx1, dx1h = ops.layernorm_bwd(dy2, x1, plan["ln2_w"], mean2, rstd2, dres=dx2, cast_scale=sc[0:1], dy_scale=sc[1:2] if h16dy else None)
            do = _mm(None, plan, "wproj_t", xs=dx1h, sc=sc, out_dtype=torch.float16)
            del dx1h
        else:
            dx1 = ops.layernorm_bwd(dy2, x1, plan["ln2_w"], mean2, rstd2, dres=dx2)
            do = _mm(dx1, plan, "wproj_t", sc=sc)
        # gradient columns come back as (dq, dv, dk): the q / v LoRA factors only ever touch the first two thirds
        # the first trainable block: no gradient flows below it, dK has no consumer (the LoRA factors contract dq and dv only)
        dqkv = ops.attention_bwd(qkv, o, do, lse, B, Nt, H, vfirst=True, need_dk=bool(ctx.needs_input_grad[0]), x3=fmt == "x3")
        if ctx.has_lora:
            r = at.shape[0] // 2
            bt_T = tw["bt_T"] if tw is not None else bt.to(T).contiguous()
            bt_qv = tw["bt_qv"] if tw is not None else torch.cat([bt_T[:, :D], bt_T[:, 2 * D:]], 1).contiguous()   # [2r, 2D]
            dqv = dqkv[:, :2 * D]
            gat = None
            dt_is_scaled = False
            if fmt == "h" and sc is not None and y1.dtype == torch.float16 and ops.lora_bwd_fused_h_supported(dqv, t, None, z_bt):
                # tf32h: both LoRA-backward products in one pass over the fp16 (dq, dv) block (it carries the step's scale s: results times 1 / s),
                # then the LoRA-A gradient dt^T . LN(x) on the same kernel with dt going in under s
                # (dt stays in the scaled domain with h16dy: both of its consumers — the LoRA-A gradient and the rank update of the dX GEMM —
                #  take it under s)
                bt16 = tw["bt_qv_w3"] if tw is not None and "bt_qv_w3" in tw and tw["bt_qv_w3"].dtype == torch.float16 else ops.cast16(bt_qv.float().contiguous())
                dt = ops.lora_bwd_fused_h(dqv, t, bt16, z_bt, out_mul=sc[1:2], dt_scaled=h16dy)
                gbt = z_bt
                if ops.lora_bwd_fused_h_supported(y1, dt, None, z_at):
                    ops.lora_bwd_fused_h(y1, dt, None, z_at, t_mul=None if h16dy else sc[0:1], out_mul=sc[1:2])
                else:       # (D not a multiple of 256, e.g. ViT-S: the streaming N = 8 kernel, fp32 dt against the fp16 LN(x))
                    ops.gemm_tn(dt, y1, out=z_at, alpha_dev=sc[1:2] if h16dy else None)
                gat = z_at
                dqkv_s = dqkv
                dt_is_scaled = h16dy
            elif fmt:
                ad = None if sc is None else sc[1:2]
                if ctx.needs_input_grad[0]:
                    # tf32x: dt = dqv . Bt^T on the split of the WHOLE dqkv row (the dX GEMM below needs that split anyway) against
                    # [Bt_q | Bt_v | 0] — the dk third contributes zeros; bf16 tile kernel instead of the fp32 one
                    dqkv_s = dqkv if fmt == "h" else _opa(dqkv, fmt, sc)           # (tf32h: already fp16, already times s)
                    btz = torch.zeros(bt_qv.shape[0], 3 * D, dtype=torch.float32, device=dqkv.device)
                    btz[:, :2 * D] = bt_qv
                    dt = ops.gemm_nt(dqkv_s, _opw(btz, fmt), out_dtype=torch.float32, alpha_dev=ad)      # [M, 2r]
                else:      # first trainable block: the dk third of dqkv was never written — split the (dq, dv) view only
                    dt = ops.gemm_nt(dqv if fmt == "h" else _opa(dqv, fmt, sc), _opw(bt_qv.contiguous(), fmt), out_dtype=torch.float32,
                                     alpha_dev=ad)
                if fmt == "h":       # dqv carries the block's scale s: contract into a scratch and add it unscaled
                    gs = ops.gemm_tn(t, dqv)
                    z_bt.add_(gs * sc[1])
                    gbt = z_bt
                else:
                    gbt = ops.gemm_tn(t, dqv, out=z_bt)                                           # [2r, 2D]
            elif ops.lora_bwd_fused_supported(dqv, t, bt_qv, z_bt):
                dt = ops.lora_bwd_fused(dqv, t, bt_qv, z_bt)                                      # both products, one pass over dqv
                gbt = z_bt
            else:
                dt = ops.gemm_nt(dqv, bt_qv, out_dtype=torch.float32)                             # [M, 2r], streams 2/3 of dqkv
                gbt = ops.gemm_tn(t, dqv, out=z_bt)                                               # [2r, 2D]
            if gat is not None:
                pass                                                                              # (tf32h: taken above)
            else:
                if ops.lora_bwd_fused_supported(y1, dt, None, z_at):
                    gat = ops.skinny_tn_mfma(dt, y1, z_at)                                    # [2r, D] on the same slab kernel
                else:
                    gat = ops.gemm_tn(dt, y1, out=z_at)                                       # [2r, D]
            g_bq, g_bv = gbt[:r, :D].t(), gbt[r:, D:].t()          # strided views: the gradient gather copies them anyway
            g_aq, g_av = gat[:r], gat[r:]
        if direct:       # already in the flat buffer (LoRA-B: GDViT.finish_trainable_grads transposes the stash once per step)
            g_aq = g_bq = g_av = g_bv = g_down = g_up = None
            if ctx.gate is not None:      # the block's LAST backward node of the step: its LoRA-A / adapter slices of the flat gradient buffer are final,
                ctx.gate.node_done()      # their exchange may start under the blocks below (BlockGradGate)
        if not ctx.needs_input_grad[0]:   # first trainable block: nothing below it learns, skip dX (one GEMM + one LN backward)
            return None, None, None, None, g_aq, g_bq, g_av, g_bv, g_down, g_up
        if ctx.has_lora:
            # (fp16 dy1: dqkv_s . W + dt_s . At, everything under s — the LoRA rank update needs dt in the same domain as the main product)
            h16dy1 = h16dy and dt_is_scaled
            dy1 = _mm(dqkv, plan, "wqkv_t_qvk", xs=dqkv_s if fmt else None, sc=sc, lora_t=dt, lora_b=at.contiguous(),   # dqkv.W + dt.At
                      **({"out_dtype": torch.float16} if h16dy1 else {}))
        else:
            h16dy1 = h16dy and dqkv.dtype == torch.float16
            dy1 = _mm(dqkv, plan, "wqkv_t_qvk", xs=dqkv if dqkv.dtype == torch.float16 else None, sc=sc, **({"out_dtype": torch.float16} if h16dy1 else {}))
        if fmt == "h":      # max |dx| rides along: the block below takes its gradient scale from it without a pass of its own
            dx = ops.layernorm_bwd(dy1, x, plan["ln1_w"], mean1, rstd1, dres=dx1, dy_scale=sc[1:2] if dy1.dtype == torch.float16 else None, want_amax=True)
        else:
            dx = ops.layernorm_bwd(dy1, x, plan["ln1_w"], mean1, rstd1, dres=dx1)
        return dx, None, None, None, g_aq, g_bq, g_av, g_bv, g_down, g_up


def run_block(blk, x, offer_ln_stats=False, next_block=None):
    """x [B, Nt, D] (engine dtype) -> block output, fused with its LoRA / adapter wrappers.  offer_ln_stats: the row statistics the block's first
    LayerNorm takes of x are left in ops' registry for the tap that normalises the same tensor (forward_all).  next_block (forward_all): the block that
    will consume the output — with options.adapter_ln the fused adapter kernel of the tf32h engine writes that block's LayerNorm 1 as well."""
    inner, lora, adapter = _unwrap(blk)
    B, Nt, D = x.shape
    plan = inner.plan(x.dtype)
    a_q = b_q = a_v = b_v = down = up = None
    if lora is not None:
        if lora.linear_a_k is not None or lora.linear_a_v is None:
            raise ops._lib.GdHipError("the fused LoRA epilogue implements the reference's q+v configuration")
        a_q, b_q = lora.linear_a_q.weight, lora.linear_b_q.weight
        a_v, b_v = lora.linear_a_v.weight, lora.linear_b_v.weight
        if 2 * a_q.shape[0] > 8:
            raise ops._lib.GdHipError("LoRA rank > 4 not supported by the fused epilogue")
    if adapter is not None:
        down, up = adapter.down.weight, adapter.up.weight
    tw = getattr(inner, "_tw", None)
    if tw is not None and tw.get("dtype") == x.dtype:
        plan = dict(plan, tw=tw)
    if offer_ln_stats:
        plan = dict(plan, offer_ln_stats=True)
    if next_block is not None and adapter is not None and plan["x3"] == "h" and bool(option("adapter_ln")):
        pn = _unwrap(next_block)[0].plan(x.dtype)
        plan = dict(plan, next_ln=(pn["ln1_w"], pn["ln1_b"], pn["eps1"]))
    out = _BlockFn.apply(x.reshape(B * Nt, D), plan, B, Nt, a_q, b_q, a_v, b_v, down, up)
    return out.view(B, Nt, D)


class _LNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
        y, mean, rstd = ops.layernorm_fwd(x2, wf, bf, eps, save_stats=x.requires_grad)
        if x.requires_grad:
            ctx.save_for_backward(x2, wf, mean, rstd)
            ctx.shp = shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, wf, mean, rstd = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        if dy2.dtype not in (torch.float32, x2.dtype):
            dy2 = dy2.float()
        if x2.dtype == torch.float32 and dy2.dtype != torch.float32:
            dy2 = dy2.float()
        return ops.layernorm_bwd(dy2, x2, wf, mean, rstd).view(ctx.shp), None, None, None


class _TapFn(torch.autograd.Function):
    """A tapped block output x feeds three consumers in the distillation step: the next block, `model.norm` (keypoint
    features) and the un-normed tap mean (cost-volume features).  Left to autograd, the three gradients are summed with
    two extra full-tensor passes per tap; here x goes out as two aliases + its norm, and the backward folds the two
    pass-through gradients into the LayerNorm backward kernel as residuals (one pass)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, out_dtype=None, amax=False, defer=None):
        shp = x.shape
        ctx.amax = bool(amax)      # tf32h: the summed gradient is the block below's dout — take its maximum on the way out
        x2 = x.reshape(-1, shp[-1]).contiguous()
        wf, bf = w.detach().float().contiguous(), b.detach().float().contiguous()
        if defer is not None:
            # Round 5: the normed tap is NOT written.  The third output is one more alias of x; `defer` (filled by forward_all once the NEXT block's
            # LayerNorm has taken the row statistics of this very tensor) tells its one consumer, kp_gather, to apply the norm where it samples
            # (gd_kp_gather_fwd_ln), and this node's backward where to find the statistics.  4 x 94 us of ln_fwd_kernel<float, float> per step.
            defer.update(x2=x2, w=wf, b=bf, eps=float(eps))
            ctx.defer = defer
            ctx.save_for_backward(x2, wf)
            ctx.shp = shp
            return x.view_as(x), x.view_as(x), x.view_as(x)
        ctx.defer = None
        y, mean, rstd = ops.layernorm_fwd(x2, wf, bf, eps, save_stats=x.requires_grad, out_dtype=out_dtype)
        if x.requires_grad:
            ctx.save_for_backward(x2, wf, mean, rstd)
            ctx.shp = shp
        return x.view_as(x), x.view_as(x), y.view(shp)

    @staticmethod
    def backward(ctx, g_next, g_raw, g_norm):
        if ctx.defer is not None:
            x2, wf = ctx.saved_tensors
            mean, rstd = deferred_stats(ctx.defer)
        else:
            x2, wf, mean, rstd = ctx.saved_tensors
        if g_next is not None:
            ops.amax_take(g_next)      # the next block's LN1 backward registered a scale for this tensor; its consumer here (ln_bwd's dres) takes fp32: drop the entry, it pins the tensor
        res = [g.reshape(x2.shape).contiguous().to(x2.dtype) for g in (g_next, g_raw) if g is not None]
        if g_norm is None:
            dx = None if not res else (res[0] if len(res) == 1 else res[0] + res[1])
        else:
            dy = g_norm.reshape(x2.shape).contiguous()
            if dy.dtype not in (torch.float32, x2.dtype) or (x2.dtype == torch.float32 and dy.dtype != torch.float32):
                dy = dy.float()
            dx = ops.layernorm_bwd(dy, x2, wf, mean, rstd, dres=res[0] if res else None,
                                   dres2=res[1] if len(res) > 1 else None, want_amax=ctx.amax and x2.dtype == torch.float32)
        return (dx.view(ctx.shp) if dx is not None else None), None, None, None, None, None, None


def deferred_stats(defer):
    """(mean, rstd) of a deferred tap norm: what the next block's LayerNorm left in the registry, or — when that block kept none (it ran without
    saving for a backward) — one statistics pass now."""
    if "mean" not in defer:
        _, defer["mean"], defer["rstd"] = ops.layernorm_fwd(defer["x2"], defer["w"], defer["b"], defer["eps"], save_stats=True)
    return defer["mean"], defer["rstd"]


class _DeferredNormFn(torch.autograd.Function):
    """`model.norm` of a deferred tap, materialised for a consumer that cannot apply it itself.  The tap's own node (_TapFn) expects the gradient with
    respect to the NORMED tensor on its third output: the backward here hands the incoming gradient through unchanged."""

    @staticmethod
    def forward(ctx, t, rec):
        return ops.layernorm_fwd(rec["x2"], rec["w"], rec["b"], rec["eps"], save_stats=False)[0].view(t.shape)

    @staticmethod
    def backward(ctx, dy):
        return dy, None


class DeferredTapNorm:
    """`model.norm(tap)` that has NOT been computed: what `forward_all(norm_taps="deferred")` hands out in place of a normed tap.  Deliberately not a
    tensor — nothing can slice, cast or stack it and silently get un-normalised features.  Its one cheap consumer is `kp_gather`, which applies the
    norm at the sampled rows (gd_kp_gather_fwd_ln) when every grid of the call is deferred; anybody else calls `.materialize()` for the normed tensor
    (one LayerNorm pass; its gradient reaches the tap's node as the gradient of the NORMED tap)."""
    __slots__ = ("raw", "rec")

    def __init__(self, raw, rec):
        self.raw, self.rec = raw, rec      # raw: the tap's third alias out of _TapFn (it carries the normed tap's gradient); rec: the deferred-norm record

    shape = property(lambda self: self.raw.shape)
    dtype = property(lambda self: self.raw.dtype)
    device = property(lambda self: self.raw.device)

    def materialize(self):
        return _DeferredNormFn.apply(self.raw, self.rec)


class GDLayerNorm(nn.LayerNorm):
    """`model.norm` / `norm_pre`: frozen affine, HIP forward and backward-to-input."""

    def forward(self, x):
        return _LNFn.apply(x, self.weight, self.bias, self.eps)


class GDPatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim, bias=True):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)


VIT_PRESETS = {
    # name: (dim, depth, heads)
    "vit_tiny_test": (64, 8, 1), "vit_small": (384, 12, 6), "vit_base": (768, 12, 12), "vit_large": (1024, 24, 16),
}


class GDViT(nn.Module):
    """Parametric pre-LN ViT (patch, dim, depth, heads, LayerScale, pre_norm, LN eps, pos-embed resampling) covering
    the reference's CLIP ViT-B/16 and BASELINE.json's DINOv2 /14 variants."""

    def __init__(self, img_size=518, patch_size=14, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 init_values=None, pre_norm=False, ln_eps=1e-6, pos_interp="dinov2", mean=(0.485, 0.456, 0.406),
                 std=(0.229, 0.224, 0.225), dtype="bf16"):
        super().__init__()
        self.embed_dim = self.num_features = embed_dim
        self.num_prefix_tokens = 1
        self.patch_embed = GDPatchEmbed(patch_size, 3, embed_dim, bias=not pre_norm)
        g = img_size // patch_size
        self.native_grid = g
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, g * g + 1, embed_dim))
        self.norm_pre = GDLayerNorm(embed_dim, eps=ln_eps) if pre_norm else nn.Identity()
        self.blocks = nn.ModuleList([GDBlock(embed_dim, num_heads, mlp_ratio, init_values, ln_eps) for _ in range(depth)])
        self.norm = GDLayerNorm(embed_dim, eps=ln_eps)
        self.pos_interp = pos_interp
        self.mean, self.std = tuple(mean), tuple(std)
        self.dtype = _dt(dtype)
        # tf32x: TF32-class GEMMs on the bf16 matrix cores (gfx950 has no TF32 MFMA; the reference's MASt3R path computes its matmuls in
        # TF32, dust3r/croco/models/croco.py:12): fp32 everywhere, the eight big frozen-weight GEMMs of a block as 3-term bf16 splits
        self.opfmt = {"tf32x": "x3", "tf32h": "h"}.get(dtype, "")
        self.gemm_split3 = bool(self.opfmt)
        for blk in self.blocks:
            blk.split3 = self.opfmt
        self._pos_cache = {}
        self._pe_plan = None
        # dtype of the final-normed tap grids that feed the keypoint features (None = the engine dtype).  torch.float32 keeps
        # LN(tap) unrounded: with near-identical tokens (deep random-init backbones) the depth-ranking gradient is a sum of
        # feature DIFFERENCES, and a bf16 rounding of the normed values is amplified by |feature| / |difference|
        self.tap_norm_dtype = None
        self.init_weights()
        # the per-dtype plans cache cast / folded / transposed copies of the frozen weights: drop them whenever the weights
        # can have changed under them (load_state_dict; .to() / .float() / .cuda() go through _apply below)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_plans())

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_plans()
        return out

    def init_weights(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    m.weight.copy_(torch.nn.init.trunc_normal_(torch.empty_like(m.weight), std=0.02, generator=g))
                    if m.bias is not None:
                        m.bias.zero_()
            self.pos_embed.copy_(torch.nn.init.trunc_normal_(torch.empty_like(self.pos_embed), std=0.02, generator=g))
            self.cls_token.copy_(torch.randn(self.cls_token.shape, generator=g) * 1e-6)
            w = self.patch_embed.proj.weight
            w.copy_(torch.nn.init.trunc_normal_(torch.empty_like(w), std=0.02, generator=g))
            if self.patch_embed.proj.bias is not None:
                self.patch_embed.proj.bias.zero_()

    def prepare_trainables(self, flat=None):
        """Per-step pack of every adapted block's trainable weights in the engine's layouts (LoRA A stacked [2r, D] and its
        cast, LoRA B scattered into the [2r, 3D] epilogue operand and its cast, adapter weights and their transposes
        cast).  Valid until the weights change: FinetuneGD.training_step builds it and `release_trainables` drops it.
        flat = FinetuneGD's flat-buffer record (configure_optimizers) with the spans of the LoRA-A / LoRA-B / adapter
        tensors: the packs are then built from VIEWS of the flat fp32 parameter buffer (the tensors sit there back to back in
        exactly the stacked order: ~10 kernels per step instead of ~25 stacks / cats), and the blocks' backward accumulates
        the LoRA / adapter weight gradients straight into the matching views of the flat GRADIENT buffer — no per-block
        zero-filled scratch, no 48-tensor gather, no strided LoRA-B copies (`finish_trainable_grads` transposes them once)."""
        T = self.dtype
        lo = []
        self._direct = None
        ops.amax_clear()        # (tf32h: scales registered for gradients of an earlier step that nobody consumed)
        for blk in self.blocks:
            inner, lora, adapter = _unwrap(blk)
            inner._tw = None
            if lora is not None and lora.linear_a_k is None and lora.linear_a_v is not None and adapter is not None:
                lo.append((inner, lora, adapter))
        if not lo:
            return
        L = len(lo)
        r, D = lo[0][1].linear_a_q.weight.shape
        bott = lo[0][2].down.weight.shape[0]
        use_flat = flat is not None and flat.get("spans") is not None and flat["spans"]["L"] == L
        extra = [{} for _ in range(L)]
        with torch.no_grad():
            if use_flat:
                sp, fp, fg = flat["spans"], flat["p"], flat["g"]
                at = fp[sp["A"]:sp["A"] + L * 2 * r * D].view(L, 2 * r, D)
                Bv = fp[sp["B"]:sp["B"] + L * 2 * D * r].view(L, 2, D, r)
                Ad = fp[sp["ad"]:sp["ad"] + L * 2 * bott * D].view(L, 2, bott * D)
                bt = getattr(self, "_bt_buf", None)
                if bt is None or bt.shape != (L, 2 * r, 3 * D) or bt.device != fp.device:
                    bt = self._bt_buf = torch.zeros(L, 2 * r, 3 * D, dtype=torch.float32, device=fp.device)   # the k block stays zero
                bt[:, :r, :D] = Bv[:, 0].transpose(1, 2)
                bt[:, r:, 2 * D:] = Bv[:, 1].transpose(1, 2)
                down, up = Ad[:, 0].view(L, bott, D), Ad[:, 1].view(L, D, bott)
                gA = fg[sp["A"]:sp["A"] + L * 2 * r * D].view(L, 2 * r, D)
                gAd = fg[sp["ad"]:sp["ad"] + L * 2 * bott * D].view(L, 2, bott * D)
                gbt = torch.zeros(L, 2 * r, 2 * D, dtype=torch.float32, device=fp.device)       # t^T [dq | dv] stash
                for i in range(L):
                    extra[i] = {"g_at": gA[i], "g_bt": gbt[i], "g_down": gAd[i, 0].view(bott, D), "g_up": gAd[i, 1].view(D, bott)}
                hook = getattr(self, "block_grad_hook", None)      # data parallelism (dp.OverlappedGradReducer): block i's slices are final when its backward returns
                if hook is not None:
                    nA, nAd = 2 * r * D, 2 * bott * D
                    for i in range(L):
                        extra[i]["on_grads"] = BlockGradGate(hook, i, [(sp["A"] + i * nA, sp["A"] + (i + 1) * nA), (sp["ad"] + i * nAd, sp["ad"] + (i + 1) * nAd)])
                self._direct = {"gbt": gbt, "gB": fg[sp["B"]:sp["B"] + L * 2 * D * r].view(L, 2, D, r), "r": r, "D": D}
            else:
                at = torch.stack([torch.cat([l.linear_a_q.weight, l.linear_a_v.weight], 0) for _, l, _ in lo]).float()   # [L, 2r, D]
                bt = torch.zeros(L, 2 * r, 3 * D, dtype=torch.float32, device=at.device)
                bt[:, :r, :D] = torch.stack([l.linear_b_q.weight for _, l, _ in lo]).transpose(1, 2)
                bt[:, r:, 2 * D:] = torch.stack([l.linear_b_v.weight for _, l, _ in lo]).transpose(1, 2)
                down = torch.stack([a.down.weight for _, _, a in lo])        # [L, 64, D]
                up = torch.stack([a.up.weight for _, _, a in lo])            # [L, D, 64]
            at_T, bt_T, down_T, up_T = at.to(T), bt.to(T), down.to(T), up.to(T)
            bt_qv = torch.cat([bt_T[:, :, :D], bt_T[:, :, 2 * D:]], 2).contiguous()      # [L, 2r, 2D]: the (dq, dv) column order
            down_tT, up_tT = down_T.transpose(1, 2).contiguous(), up_T.transpose(1, 2).contiguous()
            if getattr(self, "gemm_split3", False):      # tf32x / tf32h: the adapter weights as right-hand operands (four small passes per step)
                w3 = lambda w: _opw(w.reshape(-1, w.shape[-1]).contiguous(), self.opfmt).view(L, w.shape[1], -1)
                for i, pack in enumerate(zip(w3(down_T), w3(up_T), w3(down_tT), w3(up_tT))):
                    extra[i].update(zip(("down_w3", "up_w3", "down_tw3", "up_tw3"), pack))
                # ... and the LoRA factors every block would otherwise format on its own (two launches instead of 2 L: A for the forward's rank
                # projection, the (dq, dv) columns of B for the backward's)
                for i, w in enumerate(w3(at.float())):
                    extra[i]["at_w3"] = w
                if self.opfmt == "h":      # (only the fp16-operand LoRA backward reads a formatted B: the tf32x one takes the plain tensors)
                    for i, w in enumerate(w3(bt_qv.float())):
                        extra[i]["bt_qv_w3"] = w
        for i, (inner, _, _) in enumerate(lo):
            inner._tw = {"dtype": T, "at": at[i], "bt": bt[i], "at_T": at_T[i], "bt_T": bt_T[i], "bt_qv": bt_qv[i], "down_T": down_T[i],
                         "up_T": up_T[i], "down_tT": down_tT[i], "up_tT": up_tT[i], **extra[i]}

    def finish_trainable_grads(self):
        """After the backward of a step prepared with `prepare_trainables(flat)`: the LoRA-B gradients were accumulated as
        t^T [dq | dv] ([2r, 2D] per block); transpose the q and v blocks into the flat buffer's [D, r] tensors (2 kernels)."""
        d = getattr(self, "_direct", None)
        if d is None:
            return
        r, D = d["r"], d["D"]
        with torch.no_grad():
            L = d["gbt"].shape[0]
            # the (q, q) and (v, v) blocks of the [L, (q | v) r, (q | v) D] stash = the diagonal over the two (q | v) axes -> [L, r, D, 2]: ONE add
            diag = torch.diagonal(d["gbt"].view(L, 2, r, 2, D), dim1=1, dim2=3)
            d["gB"] += diag.permute(0, 3, 2, 1)
        self._direct = None

    def release_trainables(self):
        for blk in self.blocks:
            _unwrap(blk)[0]._tw = None

    def invalidate_plans(self):
        self._pos_cache, self._pe_plan = {}, None
        for b in getattr(self, "blocks", ()):
            inner = b.block if hasattr(b, "block") and hasattr(b, "adapter") else b
            if isinstance(inner, GDBlock):
                inner._plan = None
                inner._tw = None

    # ---- frozen prologue pieces ----
    def _pos(self, gh, gw):
        key = (gh, gw)
        if key not in self._pos_cache:
            pe = self.pos_embed.detach().float()
            m = self.native_grid
            if (gh, gw) != (m, m):
                grid = pe[:, 1:].reshape(1, m, m, -1).permute(0, 3, 1, 2)
                if self.pos_interp == "dinov2":   # vggt/layers/vision_transformer.py:181-213
                    grid = F.interpolate(grid, mode="bicubic", antialias=False,
                                         scale_factor=(float(gh + 0.1) / m, float(gw + 0.1) / m))
                else:                             # timm resample_abs_pos_embed (bicubic, antialias)
                    grid = F.interpolate(grid, size=(gh, gw), mode="bicubic", antialias=True, align_corners=False)
                assert grid.shape[-2:] == (gh, gw)
                pe = torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], 1)
            self._pos_cache[key] = pe[0].contiguous()
        return self._pos_cache[key]

    def _pos_strided(self, gh, gw, H, W):
        """Position table for the overlapping-patch mode.  A foreign `interpolate_pos_encoding` bound onto the model (what
        src/evaluate_timm.py:268-269 does with `_fix_pos_enc`) is honoured: it is called as the DINO ViT calls it — (tokens,
        w, h) with w = the image's FIRST spatial extent — on a shape-only stand-in for the token tensor.  Without one the
        table is the same formula (`_fix_pos_enc` == the dinov2 resampling with the strided grid: bicubic, +0.1 offset)."""
        fn = getattr(self, "interpolate_pos_encoding", None)
        if fn is None:
            return self._pos(gh, gw)
        key = ("foreign", gh, gw, H, W)
        if key not in self._pos_cache:
            with torch.no_grad():
                stand_in = torch.empty(1, gh * gw + 1, self.embed_dim, device=self.pos_embed.device, dtype=self.pos_embed.dtype)
                pe = fn(stand_in, H, W)
            if tuple(pe.shape) != (1, gh * gw + 1, self.embed_dim):
                raise ops._lib.GdHipError(f"interpolate_pos_encoding returned {tuple(pe.shape)}, expected {(1, gh * gw + 1, self.embed_dim)}")
            self._pos_cache[key] = pe.detach().float()[0].contiguous()
        return self._pos_cache[key]

    def _patch_plan(self):
        if self._pe_plan is None or self._pe_plan["dtype"] != self.dtype:
            w = self.patch_embed.proj.weight.detach().float()
            D, K = w.shape[0], w[0].numel()
            Kp = (K + 63) // 64 * 64
            wp = torch.zeros(D, Kp, dtype=torch.float32, device=w.device)
            wp[:, :K] = w.reshape(D, K)
            b = self.patch_embed.proj.bias
            fmt = getattr(self, "opfmt", "") if self.dtype == torch.float32 else ""
            self._pe_plan = {"dtype": self.dtype, "w": _opw(wp.contiguous(), fmt) if fmt else wp.to(self.dtype).contiguous(), "Kp": Kp, "fmt": fmt,
                             "b": b.detach().float().contiguous() if b is not None else None,
                             "cls": self.cls_token.detach().float().reshape(-1).contiguous()}
        return self._pe_plan

    def embed(self, img, size=None):
        """img [B,3,h,w] fp32 in [0,1] (NOT normalised: Normalize and the optional bilinear resize to `size`
        are fused into the im2col kernel) -> tokens [B, 1+gh*gw, D]."""
        P = self.patch_embed.patch_size[0]
        st = self.patch_embed.proj.stride
        st = (st, st) if isinstance(st, int) else tuple(int(v) for v in st)
        B, _, h, w = img.shape
        H, W = size if size is not None else (h, w)
        pp = self._patch_plan()
        cdt = torch.float16 if pp["fmt"] == "h" else self.dtype      # tf32h: im2col writes the projection's fp16 operand itself (the patch conv is frozen)
        if st == (P, P):
            assert H % P == 0 and W % P == 0, f"image size {(H, W)} not a multiple of patch {P}"
            gh, gw = H // P, W // P
            col = ops.patch_im2col(img, H, W, P, pp["Kp"], self.mean, self.std, cdt)
            pos = self._pos(gh, gw)
        else:
            # src/evaluate_timm.py:262-279: the tracking evaluation sets `patch_embed.proj.stride = (s, s)` (s = patch / 2) on the
            # live model for dense features and swaps in `_fix_pos_enc`'s resampler (utils/functions.py:169-196).  Same conv
            # weights on overlapping windows, 1 + (H - P) // s positions per axis (the patch conv is frozen: nothing else changes).
            if st[0] <= 0 or st[1] <= 0 or H < P or W < P:
                raise ops._lib.GdHipError(f"patch_embed.proj.stride = {st} with image {(H, W)} and patch {P}")
            gh, gw = 1 + (H - P) // st[0], 1 + (W - P) // st[1]
            col = ops.patch_im2col(img, H, W, P, pp["Kp"], self.mean, self.std, cdt, stride=st)
            pos = self._pos_strided(gh, gw, H, W)
        if pp["fmt"]:       # tf32x / tf32h: the patch projection on formatted operands too (the exact-f32 MFMA spends 0.7 ms on it)
            tok = ops.gemm_nt(col if pp["fmt"] == "h" else _opa(col, pp["fmt"]), pp["w"], bias=pp["b"], out_dtype=torch.float32)
        else:
            tok = ops.gemm_nt(col, pp["w"], bias=pp["b"])
        x = ops.assemble_tokens(tok, pp["cls"], pos, B, gh * gw).view(B, gh * gw + 1, -1)
        if not isinstance(self.norm_pre, nn.Identity):
            x = self.norm_pre(x)
        return x

    # ---- timm surface ----
    def _intermediate_layers(self, x, n=1, size=None):
        """timm VisionTransformer._intermediate_layers: outputs of the blocks listed in n (or the last n).
        x is the image batch in [0,1]; see `embed`."""
        take = set(range(len(self.blocks) - n, len(self.blocks)) if isinstance(n, int) else n)
        x = self.embed(x, size)
        outs = []
        for i, blk in enumerate(self.blocks):
            x = run_block(blk, x)      # not blk(x): a foreign wrapper class (utils/model.py's own) is fused by shape, never called
            if i in take:
                outs.append(x)
        return outs

    def forward_features(self, x, size=None):
        x = self.embed(x, size)
        for blk in self.blocks:
            x = run_block(blk, x)      # not blk(x): a foreign wrapper class (utils/model.py's own) is fused by shape, never called
        return self.norm(x)

    def forward_all(self, x, taps, size=None, norm_taps=False):
        """One pass: (tap outputs in `taps` order, last block output) — the shared-forward mode.  With norm_taps the
        final norm of every tap is taken on the way (`_TapFn`: one fused gradient sum per tap in the backward) and the
        result is (taps, last, normed taps): with norm_taps=True the third list holds the normed TENSORS.
        norm_taps="deferred" (FinetuneGD's step; GD_TAP_NORM_FUSED, default on): a tap that has a block above it is not normalised here — its entry
        of the third list is a `DeferredTapNorm` (not a tensor: `kp_gather` applies the norm where it samples, `.materialize()` gives the tensor);
        entries that cannot be deferred (the last block's tap, autograd off, `tap_norm_dtype` set) are normed tensors as with True."""
        ops.ln_stats_clear()      # (both registries are keyed by tensor address: nothing of an earlier pass — an exception mid-forward, a skipped block — may be taken)
        x = self.embed(x, size)
        outs, normed = {}, {}
        pending = None            # (tapped tensor, its deferred-norm record): the next block's LayerNorm fills in the statistics
        # deferred tap norms: fp32 / bf16 token rows of 16-byte multiples, autograd on (the consumer is the step's kp_gather)
        defer_ok = (norm_taps == "deferred" and bool(option("tap_norm_fused")) and torch.is_grad_enabled() and self.tap_norm_dtype is None
                    and (self.embed_dim * (2 if self.dtype == torch.bfloat16 else 4)) % 16 == 0)
        for i, blk in enumerate(self.blocks):
            x_in = x
            x = run_block(blk, x, offer_ln_stats=pending is not None, next_block=self.blocks[i + 1] if i + 1 < len(self.blocks) else None)      # not blk(x): a foreign wrapper class (utils/model.py's own) is fused by shape, never called
            if pending is not None:
                st = ops.ln_stats_take(x_in.reshape(-1, x_in.shape[-1]), pending["eps"])
                if st is not None:
                    pending["mean"], pending["rstd"] = st
                pending = None
            if i in taps:
                if norm_taps and i + 1 < len(self.blocks):
                    rec = {} if (defer_ok and x.requires_grad) else None
                    x, outs[i], normed[i] = _TapFn.apply(x, self.norm.weight, self.norm.bias, self.norm.eps, self.tap_norm_dtype, self.opfmt == "h", rec)
                    if rec is not None:
                        inner = _unwrap(self.blocks[i + 1])[0]
                        if inner.norm1.eps == self.norm.eps:
                            pending = rec
                        normed[i] = DeferredTapNorm(normed[i], rec)
                else:
                    outs[i] = x
                    if norm_taps:
                        normed[i] = self.norm(x)
        if norm_taps:
            return [outs[i] for i in taps], x, [normed[i] for i in taps]
        return [outs[i] for i in taps], x

    def forward(self, x):
        return self.forward_features(x)


def create_vit(name="vit_base", patch_size=14, img_size=518, **kw):
    dim, depth, heads = VIT_PRESETS[name]
    return GDViT(img_size=img_size, patch_size=patch_size, embed_dim=dim, depth=depth, num_heads=heads, **kw)


# ------------------------------------------------------------------------------------------------
# refine_conv on token grids and keypoint sampling (autograd over the C ABI)
# ------------------------------------------------------------------------------------------------
class _Conv3x3Fn(torch.autograd.Function):
    """refine_conv through im2col (the materialising form: a [M, 9D] column buffer each way).  Kept for `GD_CONV_STACKED=0` and
    as the reference point of tests; the default is _Conv3x3StackedFn."""

    @staticmethod
    def forward(ctx, tok, weight, bias, gh, gw):
        """tok [B, 1+gh*gw, D] (prefix token skipped) -> [B, gh*gw, D] fp32."""
        B, Nt, D = tok.shape
        tok = tok.contiguous()
        T = tok.dtype
        col = ops.im2col3x3(tok[:, Nt - gh * gw:], Nt * D, B, gh, gw, D)
        wk = weight.detach().permute(0, 2, 3, 1).reshape(D, 9 * D).to(T).contiguous()
        out = ops.gemm_nt(col, wk, bias=bias.detach().float().contiguous(), out_dtype=torch.float32)
        ctx.save_for_backward(col, wk)
        ctx.dims = (B, Nt, D, gh, gw)
        return out.view(B, gh * gw, D)

    @staticmethod
    def backward(ctx, dy):
        col, wk = ctx.saved_tensors
        B, Nt, D, gh, gw = ctx.dims
        T = col.dtype
        dyf = dy.reshape(B * gh * gw, D).contiguous().float()
        dyt = dyf if T == torch.float32 else ops.cast(dyf, T)
        gw_ = ops.gemm_tn(dyt, col)                                         # [Dout, 9*Din]
        gweight = gw_.view(D, 3, 3, D).permute(0, 3, 1, 2).contiguous()
        gbias = dyf.sum(0)
        dcol = ops.gemm_nt(dyt, wk.t().contiguous())                        # [M, 9D]
        dtok = ops.col2im3x3(dcol, B, gh, gw, D, prefix=Nt - gh * gw)     # straight into the [B, Nt, D] token gradient
        return dtok, gweight, gbias, None, None


class _Conv3x3StackedFn(torch.autograd.Function):
    """refine_conv as ONE GEMM over an overlapping-row view of a 3-row stacked token buffer (gd_stack3_rows): no im2col /
    col2im, a third of the bytes.  The output lives on the separator-column grid: [B, gh*(gw+1), D] fp32, pitch gw + 1 (the
    separator positions hold finite garbage and receive zero gradient: kp_gather reads with the pitch and never touches them)."""

    @staticmethod
    def forward(ctx, tok, weight, bias, gh, gw):
        B, Nt, D = tok.shape
        tok = tok.contiguous()
        T = tok.dtype
        rows = B * gh * (gw + 1)
        buf = ops.stack3_rows(tok, B, gh, gw, D, Nt * D, (Nt - gh * gw) * D, gw, T)
        # K index = (dx, dy, c): weight[n, c, ky, kx] -> [n, kx, ky, c]
        wk = weight.detach().permute(0, 3, 2, 1).reshape(D, 9 * D).to(T).contiguous()
        out = ops.gemm_nt(ops.conv_view(buf, rows, D), wk, bias=bias.detach().float().contiguous(), out_dtype=torch.float32)
        ctx.save_for_backward(buf, weight)
        ctx.dims = (B, Nt, D, gh, gw)
        return out.view(B, gh * (gw + 1), D)

    @staticmethod
    def backward(ctx, dy):
        buf, weight = ctx.saved_tensors
        B, Nt, D, gh, gw = ctx.dims
        T = buf.dtype
        rows = B * gh * (gw + 1)
        dyf = dy.reshape(rows, D).contiguous().float()               # zero at the separator positions (nothing reads them)
        dyt = dyf if T == torch.float32 else ops.cast(dyf, T)
        gk = ops.gemm_tn(dyt, ops.conv_view(buf, rows, D))            # [Dout, (kx, ky, Din)]
        gweight = gk.view(D, 3, 3, D).permute(0, 3, 2, 1).contiguous()
        gbias = dyf.sum(0)
        # transposed conv = the same view of a stacked dY against the flipped kernel:
        # dX[r][ci] = sum_{dx, dy, n} dY[r + dy*pitch + dx][n] * W[n, ci, 1 - dy, 1 - dx]
        sbuf = ops.stack3_rows(dyf, B, gh, gw, D, gh * (gw + 1) * D, 0, gw + 1, T)
        wt = weight.detach().flip(2, 3).permute(1, 3, 2, 0).reshape(D, 9 * D).to(T).contiguous()      # [ci, (kx, ky, n)]
        dxp = ops.gemm_nt(ops.conv_view(sbuf, rows, D), wt)            # [rows, D] on the pitched grid
        dtok = ops.unpitch_tokens(dxp, B, gh, gw, D, Nt - gh * gw)
        return dtok, gweight, gbias, None, None


class _ConvAtKpFn(torch.autograd.Function):
    """get_feature's `interpolate_features(refine_conv(grid))(kp)` (src/finetune_timm_vggt.py:319-325) with the two linear maps
    swapped: the 3x3 input patches of a keypoint's four neighbours are mixed with the bilinear weights FIRST
    (gd_kp_patch_gather -> [B*Nk, 9D]) and the convolution is one GEMM over B*Nk rows instead of B*gh*gw (4.6x fewer at 518^2 with
    300 keypoints, 21x in the reference geometry); the weight gradient contracts the same B*Nk rows, the bias gradient is the
    column sum of the keypoint gradient (bilinear weights sum to one).  The input gradient still goes through the dense transposed
    convolution (scatter to the grid, stacked-row GEMM against the flipped kernel): a 9-tap scatter of [B*Nk, 9D] would be 2 GB of
    float atomics."""

    @staticmethod
    def forward(ctx, tok, weight, bias, kp, geom, x3=False):
        gh, gw, sx, sy, img_h, img_w, patch = geom
        B, Nt, D = tok.shape
        tok = tok.contiguous()
        kp = kp.contiguous().float()
        Nk = kp.shape[1]
        T = tok.dtype
        sparse_dx = tok.requires_grad and Nk <= 1024 and D % 8 == 0 and D <= 1024 and bool(option("conv_dx_at_kp"))
        x3 = ("x3" if x3 is True else x3) if (x3 and T == torch.float32 and sparse_dx) else ""   # tf32x / tf32h: the K = 9D GEMMs on formatted operands
        colp = ops.kp_patch_gather(tok[:, Nt - gh * gw:], Nt * D, kp, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, half=x3 == "h")
        # [n, (ky, kx, c)], the flipped [ci, (kx, ky, n)] of the dense backward, [(ky, kx, c), n] of the backward at the keypoints
        wk, wt, wu = ops.conv_weight_pack(weight, T, with_wu=sparse_dx)
        if x3 == "h":         # fp16 operands; the gathered patches are fp16 only, written so by the gather (the weight gradient contracts them again)
            out = ops.gemm_nt(colp, ops.cast16(wk), bias=bias.detach().float().contiguous(), out_dtype=torch.float32)
        elif x3:
            out = ops.gemm_nt_x3(colp, ops.split3(wk, "w"), bias=bias.detach().float().contiguous())
        else:
            out = ops.gemm_nt(colp, wk, bias=bias.detach().float().contiguous(), out_dtype=torch.float32)
        ctx.save_for_backward(colp, wu if sparse_dx else wt, kp)
        ctx.sparse_dx, ctx.x3 = sparse_dx, x3
        ctx.meta = (geom, B, Nt, D, Nk)
        return out.view(B, Nk, D)

    @staticmethod
    def backward(ctx, dfeat):
        colp, wt, kp = ctx.saved_tensors
        (gh, gw, sx, sy, img_h, img_w, patch), B, Nt, D, Nk = ctx.meta
        T = torch.float32 if ctx.x3 == "h" else colp.dtype
        dfe = dfeat.reshape(B * Nk, D).contiguous().float()
        if ctx.x3 == "h":     # gradient operand under its own power-of-two scale (ops.amax_scale), undone on the device
            sc = ops.amax_scale(dfe, 8.0)
            dft = ops.cast16(dfe, scale_dev=sc[0:1])
            gk = ops.gemm_tn(dft, colp, alpha_dev=sc[1:2])
        else:
            dft = dfe if T == torch.float32 else ops.cast(dfe, T)
            gk = ops.gemm_tn(dft, colp)                                    # [Dout, (ky, kx, Din)]
        gweight = gk.view(D, 3, 3, D).permute(0, 3, 1, 2).contiguous()
        gbias = dfe.sum(0)
        dtok = None
        if ctx.needs_input_grad[0] and ctx.sparse_dx:
            # dcol = dfeat . W over the B*Nk keypoint rows, then every token gathers its contributions (no atomics, no dense GEMM)
            if ctx.x3 == "h":
                U = ops.gemm_nt(dft, ops.cast16(wt), out_dtype=torch.float32, alpha_dev=sc[1:2])
            else:
                U = ops.gemm_nt_x3(dft, ops.split3(wt, "w")) if ctx.x3 else ops.gemm_nt(dft, wt, out_dtype=T)   # (`wt` holds wu here) [B*Nk, 9D]
            dtok = ops.kp_patch_bwd_det(U, kp, T, B, Nk, Nt, gh, gw, D, sx, sy, img_h, img_w, patch)
        elif ctx.needs_input_grad[0]:
            rows = B * gh * (gw + 1)
            dy = ops.kp_gather_bwd_det(kp, dfe.view(B, Nk, D), 1.0, T, B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, pitch=gw + 1)
            if dy is None:
                dy = ops.kp_gather_bwd(1, kp, dfe.view(B, Nk, D), B, Nk, gh, gw, D, sx, sy, img_h, img_w, patch, pitch=gw + 1)[0]
            sbuf = ops.stack3_rows(dy, B, gh, gw, D, gh * (gw + 1) * D, 0, gw + 1, T)
            dxp = ops.gemm_nt(ops.conv_view(sbuf, rows, D), wt)        # [rows, D] on the pitched grid
            dtok = ops.unpitch_tokens(dxp, B, gh, gw, D, Nt - gh * gw)
        return dtok, gweight, gbias, None, None, None


def conv3x3_at_keypoints(tok, weight, bias, kp, gh, gw, sx, sy, img_h, img_w, patch, x3=False):
    """refine_conv (3x3, padding 1) of the token grid of tok [B, prefix + gh*gw, D], bilinearly sampled at kp [B, Nk, 2] (pixels)
    -> [B, Nk, D] fp32, or None when the layout does not allow it (rows that are not 16-byte multiples; GD_CONV_AT_KP=0)."""
    es = 2 if tok.dtype == torch.bfloat16 else 4
    if not option("conv_at_kp") or (tok.shape[-1] * es) % 16 != 0:
        return None
    return _ConvAtKpFn.apply(tok, weight, bias, kp, (gh, gw, float(sx), float(sy), int(img_h), int(img_w), int(patch)), x3)


def conv3x3_tokens(tok, weight, bias, gh, gw):
    """refine_conv (3x3, padding 1) on the token grid of tok [B, prefix + gh*gw, D].  -> (fmap, pitch): fmap [B, gh*pitch, D] fp32
    with pitch = gw + 1 (separator-column layout, the default) or gw (im2col path: GD_CONV_STACKED=0, or D rows that are not
    multiples of 16 bytes)."""
    es = 2 if tok.dtype == torch.bfloat16 else 4
    if option("conv_stacked") and (tok.shape[-1] * es) % 16 == 0:
        return _Conv3x3StackedFn.apply(tok, weight, bias, gh, gw), gw + 1
    return _Conv3x3Fn.apply(tok, weight, bias, gh, gw), gw


class _GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kp, geom, ctx_ln, *grids):
        gh, gw, sx, sy, img_h, img_w, patch, stride, pitch = geom
        B, Ng, D = grids[0].shape
        prefix = Ng - gh * pitch
        gs = [g.contiguous() for g in grids]
        kp = kp.contiguous().float()
        recs = ctx_ln
        if recs is not None:      # deferred tap norms: raw grids + the next block's row statistics, normalised where they are sampled
            sts = [deferred_stats(r) for r in recs]
            out = ops.kp_gather_fwd_ln([g[:, prefix:] for g in gs], [m.view(B, Ng)[:, prefix:] for m, _ in sts], [r.view(B, Ng)[:, prefix:] for _, r in sts], Ng,
                                       recs[0]["w"], recs[0]["b"], Ng * D, kp, B, kp.shape[1], gh, gw, D, sx, sy, img_h, img_w, patch, stride=stride, pitch=pitch)
        else:
            out = ops.kp_gather_fwd([g[:, prefix:] for g in gs], Ng * D, kp, B, kp.shape[1], gh, gw, D, sx, sy, img_h,
                                    img_w, patch, stride=stride, pitch=pitch)
        ctx.save_for_backward(kp)
        ctx.meta = (geom, B, Ng, D, prefix, len(gs), gs[0].dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        (kp,) = ctx.saved_tensors
        (gh, gw, sx, sy, img_h, img_w, patch, stride, pitch), B, Ng, D, prefix, ng, T = ctx.meta
        # every grid of the mean receives the SAME gradient (w * dout / ng): scatter it once and hand the one buffer, cast
        # once to the grids' dtype, to all of them (four zero-filled fp32 grids + four scatters + four casts otherwise)
        dg = ops.kp_gather_bwd_det(kp, dout, 1.0 / ng, T, B, kp.shape[1], gh, gw, D, sx, sy, img_h, img_w, patch, prefix=prefix,
                                   stride=stride, pitch=pitch)       # one pass: no zero-fill, no atomics, no cast
        if dg is None:
            dg = ops.kp_gather_bwd(1, kp, dout * (1.0 / ng), B, kp.shape[1], gh, gw, D, sx, sy, img_h, img_w, patch, prefix=prefix,
                                   stride=stride, pitch=pitch)[0]
            if dg.dtype != T:
                dg = dg.to(T)
        return (None, None, None) + (dg,) * ng


def kp_gather(grids, kp, gh, gw, sx, sy, img_h, img_w, patch, stride=None, pitch=None):
    """interpolate_features on token-major grids [B, prefix+gh*pitch, D] (mean over the list) -> [B, Nk, D] fp32; pitch = tokens
    per grid line in memory (default gw; gw + 1 for conv3x3_tokens' separator-column output)."""
    recs = None
    if any(isinstance(g, DeferredTapNorm) for g in grids):
        if all(isinstance(g, DeferredTapNorm) for g in grids) and len(grids) <= 4 and grids[0].dtype in (torch.float32, torch.bfloat16):
            recs, grids = tuple(g.rec for g in grids), [g.raw for g in grids]
        else:      # mixed / unsupported: materialise the deferred norms (not the step's path)
            grids = [g.materialize() if isinstance(g, DeferredTapNorm) else g for g in grids]
    return _GatherFn.apply(kp, (gh, gw, float(sx), float(sy), int(img_h), int(img_w), int(patch),
                                int(patch if stride is None else stride), int(gw if pitch is None else pitch)), recs, *grids)
