"""GPU parity: in-place 2-D RoPE (gd_rope_2d) against the fixture generated from the reference's own RoPE2D
(dust3r/croco/models/pos_embed.py) and the oracle."""
import pytest
import torch

import gd_oracle as O
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def test_rope2d_golden_and_inverse():
    from gd_amd.rope import cuRoPE2D, rope_2d
    g = load_golden("g13_rope2d")
    tok_bhnd = g["tokens_bhnd"]                                   # [B,H,N,D]
    stored = tok_bhnd.transpose(1, 2).contiguous().cuda()         # memory layout [B,N,H,D]
    view_bhnd = stored.transpose(1, 2)                            # what the reference's attention holds
    out = cuRoPE2D(g["base"], 1.0)(view_bhnd, g["positions"].cuda())
    assert rel_err(out, g["out_bhnd"]) < 1e-5
    rope_2d(stored, g["positions"].cuda(), g["base"], -1.0)       # inverse rotation restores the input
    assert rel_err(stored.transpose(1, 2), tok_bhnd) < 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_rope2d_teacher_shapes_and_backward(dtype, tol):
    from gd_amd.rope import cuRoPE2D_func
    B, N, H, D = 2, 768, 12, 64                                   # MASt3R decoder: 24x32 tokens, 12 heads of 64
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, N, H, D, generator=gen, device="cuda").to(dtype)
    pos = torch.stack([torch.arange(N, device="cuda") // 32, torch.arange(N, device="cuda") % 32], -1)[None].expand(B, -1, -1).contiguous()
    ref = O.rope_2d(x.float().cpu(), pos.cpu(), 100.0, 1.0)
    xin = x.clone().requires_grad_(True)
    y = cuRoPE2D_func.apply(xin.clone(), pos, 100.0, 1.0)
    assert rel_err(y.float(), ref) < tol
    w = torch.randn(B, N, H, D, generator=gen, device="cuda").to(dtype)
    # backward = rotation by -theta of the incoming gradient
    gref = O.rope_2d(w.float().cpu(), pos.cpu(), 100.0, -1.0)
    gout = cuRoPE2D_func.backward(type("C", (), {"saved_tensors": (pos,), "saved_base": 100.0, "saved_F0": 1.0})(), w.clone())[0]
    assert rel_err(gout.float(), gref) < tol


def test_rope2d_argument_checks():
    from gd_amd.rope import rope_2d
    from gd_amd._lib import GdHipError
    x = torch.zeros(2, 5, 3, 16, device="cuda")
    with pytest.raises(GdHipError):
        rope_2d(x, torch.zeros(2, 4, 2, dtype=torch.long, device="cuda"), 100.0, 1.0)   # seq length differs
    with pytest.raises(GdHipError):
        rope_2d(x[0], torch.zeros(2, 5, 2, dtype=torch.long, device="cuda"), 100.0, 1.0)  # 3-D tokens
