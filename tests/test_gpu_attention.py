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
# bf16 dK/dV: N % 256 in 1..128 takes the four-wave 128-key blocks, everything else (200, 512) the eight-wave 256-key blocks
@pytest.mark.parametrize("B,N,H", [(1, 17, 1), (2, 64, 2), (1, 257, 3), (2, 1370, 2), (1, 128, 12), (2, 200, 2), (1, 512, 3)])
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
    # need_dk=False (the first trainable block): dq and dv exactly as before, the dK columns are don't-care
    d3 = ops.attention_bwd(qkv, o, dout, lse, B, N, H, vfirst=True, need_dk=False)
    assert torch.equal(d3[:, :2 * D], d2[:, :2 * D])


# fp16 (the tf32h engine's kernels, clamp-free conversions: p <= 2^8 by the lagged reference point, |dS| <= |dP - delta|): the same four cases,
# finite outputs, and 8x tighter bounds than bf16 (11 significant bits against 8)
@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-5, 1e-4), (torch.bfloat16, 2e-2, 4e-2), (torch.float16, 3e-3, 6e-3)])
@pytest.mark.parametrize("case", ["peaked", "rising", "all_negative", "outlier_token"])
def test_attention_reference_point_moves(dtype, tol, gtol, case):
    """The forward keeps a LAGGED softmax reference point (raised only when a score exceeds it by 2^8): logits with a
    large spread, row maxima that keep rising from key tile to key tile (every tile takes the rescale branch), rows whose
    scores are all far below zero, and one outlier key far above the rest."""
    from gd_amd import ops
    B, N, H = 2, 333, 2
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, N, 3, H, 64, generator=g, device="cuda")
    if case == "peaked":
        x[:, :, 0] *= 6.0                                     # logits std ~ 6
    elif case == "rising":
        ramp = torch.linspace(0.0, 40.0, N, device="cuda")    # key j adds ~ramp[j] to every query's score
        x[:, :, 0, :, 0] = 8.0
        x[:, :, 1, :, 0] = ramp[None, :, None]
    elif case == "all_negative":
        x[:, :, 0, :, 1] = 8.0
        x[:, :, 1, :, 1] = -25.0                              # every score ~ -25 (-36 in log2 units)
    else:
        x[:, 200, 1] *= 12.0                                  # one key with a huge norm, in the 4th tile
    qkv = x.reshape(B * N, 3 * H * 64).to(dtype)
    dout = torch.randn(B * N, H * 64, generator=g, device="cuda").to(dtype)
    o, lse = ops.attention_fwd(qkv, B, N, H)
    dqkv = ops.attention_bwd(qkv, o, dout, lse, B, N, H)
    ro, rl, rg = _ref(qkv, B, N, H, dout)
    assert torch.isfinite(o.float()).all() and torch.isfinite(dqkv.float()).all()
    assert rel_err(o, ro) < (tol if not (dtype == torch.float16 and case in ("rising", "all_negative")) else 4 * tol)
    # bf16: q * scale * log2(e) is rounded to bf16 once more, a relative 2^-9 on every score; in "rising" / "all_negative" the
    # scores are ~40-60 (log2 units) and built from ONE coordinate that is the same for every query, so that rounding is
    # coherent instead of averaging out: p is off by a few percent there (a 0.2 % temperature error).  f32 is exact.
    coherent = dtype != torch.float32 and case in ("rising", "all_negative")
    # (fp16: the outlier key's scores are ~35 in log2 units: 2^-11 of that is 0.012 in lse)
    lse_tol = {torch.float32: (1e-4, 1e-4), torch.bfloat16: (8e-2, 0.15), torch.float16: (2e-2 if case == "outlier_token" else 1e-2, 2e-2)}[dtype][int(coherent)]
    assert float((lse.double().cpu() - rl.cpu()).abs().max()) < lse_tol
    assert rel_err(dqkv, rg) < ((0.2 if dtype == torch.bfloat16 else 0.03) if coherent else gtol)


@pytest.mark.parametrize("B,N,H", [(1, 17, 1), (2, 64, 2), (1, 257, 3), (2, 1370, 2), (2, 200, 2)])
def test_attention_split_precision(B, N, H):
    """The x3 instantiation of the attention kernels (dtype code GD_F32X3: fp32 tensors, every MFMA product as three bf16 MFMAs of
    (hi, lo) operand splits): output / lse / gradients within 3e-5 of the fp64 reference — TF32-class, 1000x inside plain bf16 —
    and the same column-order / dV-only contracts as the other dtypes."""
    from gd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(N + 1)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g, device="cuda")
    dout = torch.randn(B * N, H * 64, generator=g, device="cuda")
    o, lse = ops.attention_fwd(qkv, B, N, H, x3=True)
    dqkv = ops.attention_bwd(qkv, o, dout, lse, B, N, H, x3=True)
    ro, rl, rg = _ref(qkv, B, N, H, dout)
    fro = lambda a, b: float((a.double().cpu() - b.cpu()).norm() / b.cpu().norm())
    assert fro(o, ro) < 3e-5 and rel_err(lse, rl) < 1e-5 and fro(dqkv, rg) < 5e-5, (fro(o, ro), rel_err(lse, rl), fro(dqkv, rg))
    d2 = ops.attention_bwd(qkv, o, dout, lse, B, N, H, vfirst=True, x3=True)
    D = H * 64
    assert torch.equal(d2[:, :D], dqkv[:, :D]) and torch.equal(d2[:, D:2 * D], dqkv[:, 2 * D:]) and torch.equal(d2[:, 2 * D:], dqkv[:, D:2 * D])
    d3 = ops.attention_bwd(qkv, o, dout, lse, B, N, H, vfirst=True, need_dk=False, x3=True)
    assert torch.equal(d3[:, :2 * D], d2[:, :2 * D])


def test_attention_fp16_operands():
    """dtype GD_F16 (tf32h engine): the bf16 kernels' layouts on fp16 q / k / v / p — TF32's significand.  Output, lse and gradients against
    the fp64 reference of the same fp16-rounded inputs: an order of magnitude inside the bf16 kernels' error; a gradient-sized dout goes in
    times a power of two and the gradients come back with it (linear in dout); nothing overflows to inf."""
    from gd_amd import ops
    B, N, H = 2, 1370, 12
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv32 = torch.randn(B * N, 3 * H * 64, generator=g, device="cuda")
    do32 = torch.randn(B * N, H * 64, generator=g, device="cuda") * 1e-6
    s = 2.0 ** 24
    qkv, do = qkv32.half(), (do32 * s).half()
    o, lse = ops.attention_fwd(qkv, B, N, H)
    dqkv = ops.attention_bwd(qkv, o, do, lse, B, N, H)
    assert o.dtype == torch.float16 and dqkv.dtype == torch.float16 and bool(torch.isfinite(dqkv.float()).all())
    q, k, v = [t.reshape(B, N, H, 64).permute(0, 2, 1, 3).double().requires_grad_(True) for t in qkv.float().view(B * N, 3, H * 64).unbind(1)]
    sc = (q @ k.transpose(-1, -2)) * 64 ** -0.5
    ref = torch.softmax(sc, -1) @ v
    ref_o = ref.permute(0, 2, 1, 3).reshape(B * N, H * 64)
    ref_o.backward(do.double() / s)
    ref_lse = torch.logsumexp(sc, -1)
    assert rel_err(o, ref_o.detach()) < 1e-3 and rel_err(lse, ref_lse.detach()) < 1e-4
    got = dqkv.double().view(B * N, 3, H, 64) / s
    for i, t in enumerate((q, k, v)):
        assert rel_err(got[:, i], t.grad.permute(0, 2, 1, 3).reshape(B * N, H, 64)) < 2e-3, i
    ob, _ = ops.attention_fwd(qkv32.bfloat16(), B, N, H)
    assert rel_err(o, ref_o.detach()) < 0.25 * rel_err(ob, ref_o.detach())


@pytest.mark.parametrize("dtype,tol,gtol", [(torch.float32, 2e-5, 5e-5), (torch.bfloat16, 2e-2, 4e-2), (torch.float16, 3e-3, 6e-3)])
@pytest.mark.parametrize("N", [4801, 6401])
def test_attention_reference_geometry_token_counts(dtype, tol, gtol, N):
    """The reference's own keypoint-feature geometry (target_res 640 / downsample 8: 60 x 80 + 1 = 4801 tokens on 4:3 inputs, 80 x 80 + 1 = 6401 on
    square ones; src/finetune_timm_mast3r.py:145,251-256) through all three kernels of every engine dtype: long key sweeps
    (100 tiles), the eight-wave dK/dV form, a partial last tile, 12 heads."""
    from gd_amd import ops
    B, H = 1, 12
    g = torch.Generator(device="cuda").manual_seed(N)
    qkv = torch.randn(B * N, 3 * H * 64, generator=g, device="cuda").to(dtype)
    dout = torch.randn(B * N, H * 64, generator=g, device="cuda").to(dtype)
    o, lse = ops.attention_fwd(qkv, B, N, H)
    dqkv = ops.attention_bwd(qkv, o, dout, lse, B, N, H)
    ro, rl, rg = _ref(qkv, B, N, H, dout)
    assert bool(torch.isfinite(dqkv.float()).all())
    assert rel_err(o, ro) < tol and rel_err(lse, rl) < (1e-5 if dtype == torch.float32 else 1e-2) and rel_err(dqkv, rg) < gtol


