"""HIP twin of the reference's cuRoPE2D (dust3r/croco/models/curope/curope2d.py:12-40): same class names, same
calling convention (tokens [B, heads, N, D], positions int64 [B, N, 2]), in place, backward = the same kernel with -F0.
Unlike the reference there is no pure-PyTorch fallback (dust3r/croco/models/pos_embed.py:106-110): a missing
libgd_hip.so raises."""
import torch

from ._lib import check, dtype_code, lib, ptr, stream, GdHipError


def rope_2d(tokens, positions, base, fwd):
    """tokens: [B, N, H, D] view (H*D contiguous per token), modified in place — the reference's rope_2d signature
    and checks (curope.cpp:49-69)."""
    if tokens.dim() != 4:
        raise GdHipError("tokens must have 4 dimensions")
    if positions.dim() != 3:
        raise GdHipError("positions must have 3 dimensions")
    if tokens.size(0) != positions.size(0):
        raise GdHipError("batch size differs between tokens & positions")
    if tokens.size(1) != positions.size(1):
        raise GdHipError("seq_length differs between tokens & positions")
    if positions.size(2) != 2:
        raise GdHipError("positions.shape[2] must be equal to 2")
    if not (tokens.is_cuda and positions.is_cuda):
        raise GdHipError("tokens and positions must be on the GPU (no CPU fallback)")
    B, N, H, D = tokens.shape
    if tokens.stride(3) != 1 or tokens.stride(2) != D or tokens.stride(0) != N * tokens.stride(1):
        raise GdHipError("tokens must be a [B,N,H,D] view with contiguous (H,D) per token")
    pos = positions.contiguous().long()
    check(lib().gd_rope_2d(ptr(tokens), ptr(pos), B, N, H, D, tokens.stride(1), float(base), float(fwd),
                           dtype_code(tokens), stream()), "gd_rope_2d")


class cuRoPE2D_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, positions, base, F0=1):
        ctx.save_for_backward(positions)
        ctx.saved_base, ctx.saved_F0 = base, F0
        rope_2d(tokens, positions, base, F0)
        ctx.mark_dirty(tokens)
        return tokens

    @staticmethod
    def backward(ctx, grad_res):
        positions, base, F0 = ctx.saved_tensors[0], ctx.saved_base, ctx.saved_F0
        grad_res = grad_res.contiguous() if grad_res.stride(3) != 1 else grad_res
        rope_2d(grad_res, positions, base, -F0)
        return grad_res, None, None, None


class cuRoPE2D(torch.nn.Module):
    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base, self.F0 = freq, F0

    def forward(self, tokens, positions):
        """tokens [B, heads, N, D] stored as [B, N, heads, D] underneath (as in the reference, which passes
        tokens.transpose(1,2) to the kernel)."""
        cuRoPE2D_func.apply(tokens.transpose(1, 2), positions, self.base, self.F0)
        return tokens
