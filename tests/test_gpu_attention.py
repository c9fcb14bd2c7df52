"""GPU parity: flash attention fwd/bwd (C ABI) vs explicit softmax attention in fp64."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ref(qkv, B, N, H, dout):
    x = qkv.double().reshape(B, N, 3, H, 64).requires_grad_(True)
    q, k, v = x.permute(2, 0, 3, 1, 4).unbind(0)
    s = (q * 64 ** -0.5) @ k.transpose(-1, -2)
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * N, H * 64)
    lse = torch.logsumexp(s, -1)
    o.backward(dout.double())
    return o.detach(), lse.detach(), x.grad.reshape(B * N, 3 * H * 64)


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 1e-5, 2e-5), (torch.bfloat16, 2e-2, 3e-2)])
@pytest.mark.parametrize("B,N,H", [(1, 17, 1), (2, 64, 2), (1, 257, 3), (2, 1370, 2), (1, 128, 12)])
def test_attention(dtype, tol, gtol, B, N, H):
    from gd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(N)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g, device="cuda").to(dtype)
    dout = torch.randn(B * N, H * 64, generator=g, device="cuda").to(dtype)
    o, lse = ops.attention_fwd(qkv, B, N, H)
    dqkv = ops.attention_bwd(qkv, o, dout, lse, B, N, H)
    ro, rl, rg = _ref(qkv, B, N, H, dout)
    assert rel_err(o, ro) < tol
    assert rel_err(lse, rl) < 1e-5 if dtype == torch.float32 else rel_err(lse, rl) < 1e-2
    assert rel_err(dqkv, rg) < gtol
    # grad_order 1: the same gradients with the dK and dV column blocks exchanged -> (dq, dv, dk)
    d2 = ops.attention_bwd(qkv, o, dout, lse, B, N, H, vfirst=True)
    D = H * 64
    assert torch.equal(d2[:, :D], dqkv[:, :D]) and torch.equal(d2[:, D:2 * D], dqkv[:, 2 * D:]) and \
        torch.equal(d2[:, 2 * D:], dqkv[:, D:2 * D])
