"""Host-side mirror of the reference's trainable add-ons (utils/model.py): same class names, constructor
arguments, attribute names and state_dict keys, so the reference's wrapping code
(src/finetune_timm_vggt.py:134-162) and checkpoints apply unchanged.

These modules are parameter containers + markers.  The fused engine (vit.py) discovers them on the block
list and folds them into HIP kernels (LoRA as a rank-r epilogue of the QKV GEMM, the adapter as one fused
pass); calling them directly ("eager") on CUDA tensors also routes through the HIP kernels.
There is no CPU fallback.
"""
import torch
import torch.nn as nn

from . import ops


class _LinearFn(torch.autograd.Function):
    """y = x W^T (+ b) through gd_gemm_nt; grads for x, W, b (used only when an add-on is called stand-alone)."""

    @staticmethod
    def forward(ctx, x, w, b):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        wc = w.to(x2.dtype).contiguous()
        y = ops.gemm_nt(x2, wc, bias=b.float().contiguous() if b is not None else None)
        ctx.save_for_backward(x2, wc)
        ctx.has_b, ctx.shp = b is not None, shp
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, wc = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).to(x2.dtype).contiguous()
        dx = ops.gemm_nt(dy2, wc.t().contiguous()).view(ctx.shp)
        pad = (-dy2.shape[1]) % 8
        if pad:  # gemm_tn wants the output-row count in multiples of 8
            dy2 = torch.cat([dy2, dy2.new_zeros(dy2.shape[0], pad)], 1)
        dw = ops.gemm_tn(dy2, x2)[:dy.shape[-1]]
        db = dy.reshape(-1, dy.shape[-1]).float().sum(0) if ctx.has_b else None
        return dx, dw, db


def hip_linear(x, weight, bias=None):
    return _LinearFn.apply(x, weight, bias)


class Adapter(nn.Module):
    """utils/model.py:7-15 — bottleneck adapter: up(relu(down(x))), both without bias."""

    def __init__(self, dim, bottleneck_dim):
        super().__init__()
        self.down = nn.Linear(dim, bottleneck_dim, bias=False)
        self.relu = nn.ReLU()
        self.up = nn.Linear(bottleneck_dim, dim, bias=False)

    def forward(self, x):
        return hip_linear(torch.relu(hip_linear(x, self.down.weight)), self.up.weight)


class BlockWithAdapter(nn.Module):
    """utils/model.py:17-25 — out = block(x); out + adapter(out).  The engine fuses it when `block` is a GDBlock."""

    def __init__(self, block, adapter):
        super().__init__()
        self.block = block
        self.adapter = adapter

    def forward(self, x):
        from .vit import GDBlock, run_block
        if isinstance(self.block, GDBlock):
            return run_block(self, x)
        out = self.block(x)
        return out + self.adapter(out)


class _LoRA_qkv(nn.Module):
    """utils/model.py:27-71 — LoRA on the q and v slices of a fused qkv Linear."""

    def __init__(self, qkv, linear_a_q, linear_b_q, linear_a_v=None, linear_b_v=None, linear_a_k=None,
                 linear_b_k=None):
        super().__init__()
        self.qkv = qkv
        self.linear_a_q = linear_a_q
        self.linear_b_q = linear_b_q
        self.linear_a_v = linear_a_v
        self.linear_b_v = linear_b_v
        self.linear_a_k = linear_a_k
        self.linear_b_k = linear_b_k
        self.dim = qkv.in_features
        self.w_identity = torch.eye(qkv.in_features)

    @property
    def in_features(self):
        return self.qkv.in_features

    def forward(self, x):
        qkv = hip_linear(x, self.qkv.weight, self.qkv.bias)
        D = self.dim
        parts = [qkv[..., :D] + hip_linear(hip_linear(x, self.linear_a_q.weight), self.linear_b_q.weight)]
        mid = qkv[..., D:2 * D]
        if self.linear_a_k is not None and self.linear_b_k is not None:
            mid = mid + hip_linear(hip_linear(x, self.linear_a_k.weight), self.linear_b_k.weight)
        parts.append(mid)
        last = qkv[..., 2 * D:]
        if self.linear_a_v is not None and self.linear_b_v is not None:
            last = last + hip_linear(hip_linear(x, self.linear_a_v.weight), self.linear_b_v.weight)
        parts.append(last)
        return torch.cat(parts, dim=-1)


class DepthAwareFeatureFusion(nn.Module):
    """utils/model.py:88-127 — the reference's layout (depth_attention is never used by the losses but is part of the
    checkpoint).  The fused losses evaluate fusion_layer inside gd_pair_rank / gd_depth_l1 (`head_params()` hands them
    the tensors); `forward` is the eager form on the same HIP head kernels."""

    def __init__(self, input_dim, hidden_dim=128, use_tanh=True):
        super().__init__()
        assert hidden_dim == 128 and use_tanh, "the HIP depth head is specialised for hidden_dim=128 + tanh"
        self.use_tanh = use_tanh
        self.depth_attention = nn.Sequential(nn.Linear(1, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, input_dim),
                                             nn.Sigmoid())
        self.fusion_layer = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.GELU(),
                                          nn.Linear(hidden_dim, 1))

    def head_params(self):
        fl = self.fusion_layer
        return {"w1": fl[0].weight, "b1": fl[0].bias, "ln_w": fl[1].weight, "ln_b": fl[1].bias,
                "w2": fl[3].weight, "b2": fl[3].bias}

    def forward(self, features, depths=None):
        """utils/model.py:101-127, eager: features [..., D] (+ depths [...]) -> tanh(fusion_layer(.)) [...].  The first
        Linear runs on gd_gemm_nt, LayerNorm + GELU + the 128 -> 1 Linear + tanh in gd_depth_head_{fwd,bwd}.  The
        `depths` branch (never taken by the reference's losses) modulates the features by depth_attention first."""
        if depths is not None:
            da = self.depth_attention
            e = torch.nn.functional.gelu(depths.unsqueeze(-1).float() * da[0].weight[:, 0] + da[0].bias)   # Linear(1, 128): outer product
            features = features.float() * torch.sigmoid(hip_linear(e, da[2].weight, da[2].bias))
        return ops.depth_head(features, self.head_params())
