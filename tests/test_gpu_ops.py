"""GPU parity: normalisation, image prep, conv glue, keypoint gather, sparse losses and the optimiser
(all through the C ABI) against the CPU oracle / reference-generated fixtures."""
import pytest
import torch
import torch.nn.functional as F

import gd_oracle as O
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("M,D", [(5, 64), (1370, 768), (300, 1024), (33, 384)])
def test_layernorm(dtype, tol, M, D):
    from gd_amd import ops
    x = (torch.randn(M, D, generator=_g(1), device="cuda") * 2 + 0.5).to(dtype)
    gm = torch.randn(D, generator=_g(2), device="cuda")
    bt = torch.randn(D, generator=_g(3), device="cuda")
    dy = torch.randn(M, D, generator=_g(4), device="cuda").to(dtype)
    dres = torch.randn(M, D, generator=_g(5), device="cuda").to(dtype)
    y, mean, rstd = ops.layernorm_fwd(x, gm, bt, 1e-6)
    xr = x.double().requires_grad_(True)
    yr = F.layer_norm(xr, (D,), gm.double(), bt.double(), 1e-6)
    assert rel_err(y, yr) < tol
    yr.backward(dy.double())
    dx = ops.layernorm_bwd(dy, x, gm, mean, rstd, dres=dres)
    assert rel_err(dx, xr.grad + dres.double()) < tol * 3
    # fp32 dy against bf16 x, with a scale
    dx2 = ops.layernorm_bwd(dy.float(), x, gm, mean, rstd, dyscale=0.25)
    assert rel_err(dx2, 0.25 * xr.grad) < tol * 3


def test_layernorm_fp16_operand_outputs():
    """tf32h engine: LayerNorm writes its result as the fp16 operand directly (y_dtype GD_F16), and the fp32 backward also emits
    fp16(dx * s) with s a device scalar (gd_layernorm_bwd_cast) — both bit-identical to the fp32 kernel followed by gd_cast_f16."""
    from gd_amd import ops
    M, D = 700, 768
    x = torch.randn(M, D, generator=_g(11), device="cuda") * 2 + 0.5
    gm, bt = torch.randn(D, generator=_g(12), device="cuda"), torch.randn(D, generator=_g(13), device="cuda")
    dy = torch.randn(M, D, generator=_g(14), device="cuda") * 1e-6
    dres = torch.randn(M, D, generator=_g(15), device="cuda") * 1e-6
    y32, mean, rstd = ops.layernorm_fwd(x, gm, bt, 1e-6)
    y16, m2, r2 = ops.layernorm_fwd(x, gm, bt, 1e-6, out_dtype=torch.float16)
    assert y16.dtype == torch.float16 and torch.equal(y16, ops.cast16(y32)) and torch.equal(mean, m2) and torch.equal(rstd, r2)
    sc = ops.amax_scale(dy, 8.0)
    dx, dx16 = ops.layernorm_bwd(dy, x, gm, mean, rstd, dres=dres, cast_scale=sc[0:1])
    ref = ops.layernorm_bwd(dy, x, gm, mean, rstd, dres=dres)
    assert torch.equal(dx, ref) and dx16.dtype == torch.float16
    want = ops.cast16(ref, scale_dev=sc[0:1])
    assert float((dx16 != want).float().mean()) < 0.01 and rel_err(dx16.float(), want.float()) < 1e-3


def test_tap_mean_with_fp16_copy():
    """ops.tap_mean(with_norm=2) on fp32 taps: the mean rows, their inverse norms and the fp16 copy of the rows (the cost-volume products' operands in
    the tf32h engine) from one pass — the copy equals gd_cast_f16 of the rows."""
    from gd_amd import ops
    taps = [torch.randn(4, 1 + 300, 256, generator=_g(20 + i), device="cuda").requires_grad_(True) for i in range(4)]
    f, inv, f16 = ops.tap_mean(taps, prefix=1, with_norm=2)
    f0, inv0 = ops.tap_mean(taps, prefix=1, with_norm=True)
    assert torch.equal(f, f0) and torch.equal(inv, inv0) and f16.dtype == torch.float16 and not f16.requires_grad
    assert torch.equal(f16.view(-1, 256), ops.cast16(f.detach().view(-1, 256)))
    f.sum().backward()
    assert taps[0].grad is not None


def test_l2norm():
    from gd_amd import ops
    x = torch.randn(2, 37, 96, generator=_g(6), device="cuda").requires_grad_(True)
    w = torch.randn(2, 37, 96, generator=_g(7), device="cuda")
    y = ops.l2_normalize(x)
    (y * w).sum().backward()
    xr = x.detach().double().requires_grad_(True)
    yr = F.normalize(xr, dim=-1)
    (yr * w.double()).sum().backward()
    assert rel_err(y, yr) < 1e-6 and rel_err(x.grad, xr.grad) < 1e-5


# patch 14 (BASELINE's DINOv2-style students) and patch 16 (the reference's own CLIP ViT-B/16: config/*.yaml:2, src/finetune_timm_mast3r.py:68-70):
# identity and real bilinear resizes, the engine's three operand dtypes (fp16 = the tf32h engine's patch operand), K padded to 128 / 64 multiples
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2), (torch.float16, 1e-3)])
@pytest.mark.parametrize("P,hw_in,HW", [(14, (56, 70), (56, 70)), (14, (40, 52), (56, 84)), (16, (64, 80), (64, 80)), (16, (48, 64), (96, 128))])
def test_patch_im2col_and_tokens(dtype, tol, P, hw_in, HW):
    from gd_amd import ops
    B, D = 2, 32
    mean, std = (0.48, 0.45, 0.40), (0.26, 0.25, 0.27)
    img = torch.rand(B, 3, *hw_in, generator=_g(8), device="cuda")
    Kp = 640 if P == 14 else 768
    col = ops.patch_im2col(img, HW[0], HW[1], P, Kp, mean, std, dtype)
    ref = O.normalize_image(O.resize_bilinear(img.cpu(), HW), mean, std)
    refcol = F.unfold(ref, P, stride=P).transpose(1, 2).reshape(-1, 3 * P * P)
    assert rel_err(col[:, :3 * P * P], refcol) < tol
    assert Kp == 3 * P * P or float(col[:, 3 * P * P:].abs().max()) == 0.0        # (patch 16: K = 768 needs no padding)
    Np = (HW[0] // P) * (HW[1] // P)
    tdt = torch.float32 if dtype == torch.float16 else dtype       # (tf32h engine: the patch projection's fp32 output is what gets assembled)
    patch = torch.randn(B * Np, D, generator=_g(9), device="cuda").to(tdt)
    cls = torch.randn(D, generator=_g(10), device="cuda")
    pos = torch.randn(Np + 1, D, generator=_g(11), device="cuda")
    tok = ops.assemble_tokens(patch, cls, pos, B, Np).view(B, Np + 1, D)
    want = torch.cat([cls.expand(B, 1, D), patch.float().view(B, Np, D)], 1) + pos
    assert rel_err(tok, want) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
def test_conv3x3_via_im2col(dtype, tol):
    from gd_amd import ops
    B, gh, gw, D = 2, 5, 7, 64
    tok = torch.randn(B, 1 + gh * gw, D, generator=_g(12), device="cuda").to(dtype)   # with a prefix token
    w = torch.randn(D, D, 3, 3, generator=_g(13), device="cuda") * 0.05
    bias = torch.randn(D, generator=_g(14), device="cuda")
    col = ops.im2col3x3(tok[:, 1:], (1 + gh * gw) * D, B, gh, gw, D)
    wk = w.permute(0, 2, 3, 1).reshape(D, 9 * D).contiguous().to(dtype)           # [Dout, (ky,kx), Din]
    out = ops.gemm_nt(col, wk, bias=bias, out_dtype=torch.float32)
    x = tok[:, 1:].double().reshape(B, gh, gw, D).permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv2d(x, w.double(), bias.double(), padding=1)
    assert rel_err(out.view(B, gh, gw, D).permute(0, 3, 1, 2), ref) < tol
    dy = torch.randn(B * gh * gw, D, generator=_g(15), device="cuda").to(dtype)
    ref.backward(dy.double().view(B, gh, gw, D).permute(0, 3, 1, 2))
    dcol = ops.gemm_nt(dy, wk.t().contiguous())                                    # [M, 9D]
    dx = ops.col2im3x3(dcol, B, gh, gw, D)
    assert rel_err(dx.view(B, gh, gw, D).permute(0, 3, 1, 2), x.grad) < tol * 2


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("B,gh,gw,D,prefix", [(2, 5, 7, 64, 1), (3, 37, 37, 128, 1), (1, 4, 9, 32, 0)])
def test_conv3x3_stacked_implicit_gemm(dtype, tol, B, gh, gw, D, prefix):
    """refine_conv as one GEMM over the overlapping-row view of the 3-row stacked buffer (no im2col): forward, input gradient,
    weight and bias gradients against F.conv2d in fp64; the output sits on the separator-column grid and is sampled through
    kp_gather with the pitch.  Also: identical to the im2col path."""
    from gd_amd.vit import conv3x3_tokens, kp_gather, _Conv3x3Fn
    tok = torch.randn(B, prefix + gh * gw, D, generator=_g(12), device="cuda").to(dtype).requires_grad_(True)
    w = (torch.randn(D, D, 3, 3, generator=_g(13), device="cuda") * 0.05).requires_grad_(True)
    bias = torch.randn(D, generator=_g(14), device="cuda").requires_grad_(True)
    fmap, pitch = conv3x3_tokens(tok, w, bias, gh, gw)
    assert pitch == gw + 1 and fmap.shape == (B, gh * pitch, D)
    grid = fmap.view(B, gh, pitch, D)[:, :, :gw]
    x = tok.detach().cpu()[:, prefix:].double().reshape(B, gh, gw, D).permute(0, 3, 1, 2).requires_grad_(True)
    wr, br = w.detach().cpu().double().requires_grad_(True), bias.detach().cpu().double().requires_grad_(True)
    ref = F.conv2d(x, wr, br, padding=1)
    assert rel_err(grid.permute(0, 3, 1, 2), ref) < tol
    old = _Conv3x3Fn.apply(tok.detach(), w.detach(), bias.detach(), gh, gw).view(B, gh, gw, D)
    assert rel_err(grid, old) < 1e-5      # the same products, summed in another order
    # gradient: through keypoint sampling on the pitched grid (what the step does) + a dense weight on the real positions
    P = 14
    kp = torch.rand(B, 11, 2, generator=_g(15), device="cuda") * torch.tensor([gw * P - 1.0, gh * P - 1.0], device="cuda")
    samp = kp_gather([fmap], kp, gh, gw, 1.0, 1.0, gh * P, gw * P, P, pitch=pitch)
    wd = torch.randn(B, gh, gw, D, generator=_g(16), device="cuda")
    ws = torch.randn(samp.shape, generator=_g(17), device="cuda")
    ((grid * wd).sum() + (samp * ws).sum()).backward()
    rs = O.interpolate_features(ref, kp.cpu().double(), gh * P, gw * P, False, P, P).permute(0, 2, 1)
    assert rel_err(samp, rs) < tol
    ((ref.permute(0, 2, 3, 1) * wd.cpu().double()).sum() + (rs * ws.cpu().double()).sum()).backward()
    assert rel_err(tok.grad[:, prefix:].reshape(B, gh, gw, D).permute(0, 3, 1, 2), x.grad) < 2 * tol
    if prefix:
        assert float(tok.grad[:, :prefix].abs().max()) == 0.0
    assert rel_err(w.grad, wr.grad) < 2 * tol and rel_err(bias.grad, br.grad) < 2 * tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("B,gh,gw,D,prefix,sxy", [(2, 5, 7, 64, 1, (1.0, 1.0)), (3, 37, 37, 128, 1, (1.0, 1.0)), (1, 4, 9, 32, 0, (1.7, 2.1))])
def test_conv3x3_at_keypoints(dtype, tol, B, gh, gw, D, prefix, sxy):
    """get_feature's conv + bilinear sample with the two linear maps swapped (gd_kp_patch_gather + a GEMM over B*Nk rows):
    forward and all three gradients against F.conv2d + the oracle's interpolate_features in fp64; keypoints on the border, outside
    the image and the -1 padding of ragged batches included."""
    from gd_amd.vit import conv3x3_at_keypoints
    tok = torch.randn(B, prefix + gh * gw, D, generator=_g(12), device="cuda").to(dtype).requires_grad_(True)
    w = (torch.randn(D, D, 3, 3, generator=_g(13), device="cuda") * 0.05).requires_grad_(True)
    bias = torch.randn(D, generator=_g(14), device="cuda").requires_grad_(True)
    P = 14
    sx, sy = sxy
    kp = torch.rand(B, 13, 2, generator=_g(15), device="cuda") * torch.tensor([gw * P / sx - 1.0, gh * P / sy - 1.0], device="cuda")
    kp[:, 0] = 0.0                                                   # corner
    kp[:, 1] = torch.tensor([gw * P / sx - 1.0, gh * P / sy - 1.0])   # opposite corner (clamped neighbours)
    kp[:, 2] = torch.tensor([gw * P / sx + 40.0, -25.0])             # outside the image: border clamp
    kp[:, 3] = -1.0                                                  # ragged-batch padding
    samp = conv3x3_at_keypoints(tok, w, bias, kp, gh, gw, sx, sy, gh * P, gw * P, P)
    assert samp is not None and samp.shape == (B, 13, D) and samp.dtype == torch.float32
    x = tok.detach().cpu()[:, prefix:].double().reshape(B, gh, gw, D).permute(0, 3, 1, 2).requires_grad_(True)
    wr, br = w.detach().cpu().double().requires_grad_(True), bias.detach().cpu().double().requires_grad_(True)
    ref = F.conv2d(x, wr, br, padding=1)
    kps = kp.cpu().double() * torch.tensor([sx, sy], dtype=torch.float64)
    rs = O.interpolate_features(ref, kps, gh * P, gw * P, False, P, P).permute(0, 2, 1)
    assert rel_err(samp, rs) < tol
    ws = torch.randn(samp.shape, generator=_g(17), device="cuda")
    (samp * ws).sum().backward()
    (rs * ws.cpu().double()).sum().backward()
    assert rel_err(tok.grad[:, prefix:].reshape(B, gh, gw, D).permute(0, 3, 1, 2), x.grad) < 2 * tol
    if prefix:
        assert float(tok.grad[:, :prefix].abs().max()) == 0.0
    assert rel_err(w.grad, wr.grad) < 2 * tol and rel_err(bias.grad, br.grad) < 2 * tol


@pytest.mark.parametrize("P", [14, 16])
def test_kp_gather_golden(P):
    from gd_amd import ops
    g = load_golden(f"g02_interp_p{P}")
    desc = g["desc"].cuda()                                   # [1,C,ph,pw]
    _, C, ph, pw = desc.shape
    grid = desc.permute(0, 2, 3, 1).reshape(1, ph * pw, C).contiguous()
    kp = g["pts"].cuda()
    out = ops.kp_gather_fwd([grid], ph * pw * C, kp, 1, kp.shape[1], ph, pw, C, 1.0, 1.0, g["h"], g["w"], P)
    assert rel_err(out[0].t(), g["out"][0]) < 1e-5
    dg = ops.kp_gather_bwd(1, kp, g["gout"][0].t().contiguous().cuda()[None], 1, kp.shape[1], ph, pw, C, 1.0, 1.0,
                           g["h"], g["w"], P)[0]
    assert rel_err(dg.view(ph, pw, C).permute(2, 0, 1), g["gdesc"][0]) < 1e-5
    # the deterministic (no-atomics) backward: same fixture, every element written
    dd = ops.kp_gather_bwd_det(kp, g["gout"][0].t().contiguous().cuda()[None], 1.0, torch.float32, 1, kp.shape[1], ph, pw, C, 1.0, 1.0,
                               g["h"], g["w"], P)
    assert dd is not None and rel_err(dd.view(ph, pw, C).permute(2, 0, 1), g["gdesc"][0]) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,Nk,gh,gw,D,prefix,pitch", [(3, 300, 37, 37, 768, 1, 37), (2, 50, 5, 7, 64, 0, 8), (2, 700, 12, 9, 128, 1, 9)])
def test_kp_gather_bwd_deterministic(dtype, B, Nk, gh, gw, D, prefix, pitch):
    """gd_kp_gather_bwd_det against the atomic scatter: same sums (fp32 order differs), prefix rows / separator columns zero,
    bit-identical from run to run; keypoints on the border, outside, padded with -1, and many on one grid line."""
    from gd_amd import ops
    P = 14
    kp = torch.rand(B, Nk, 2, generator=_g(31), device="cuda") * torch.tensor([gw * P - 1.0, gh * P - 1.0], device="cuda")
    kp[:, 0] = 0.0
    kp[:, 1] = torch.tensor([gw * P - 1.0, gh * P - 1.0])
    kp[:, 2] = torch.tensor([gw * P + 30.0, -12.0])
    kp[:, 3] = -1.0
    kp[:, 4:Nk // 2, 1] = 3.0 * P                                   # half of the keypoints on one grid line
    dout = torch.randn(B, Nk, D, generator=_g(32), device="cuda")
    ref = ops.kp_gather_bwd(1, kp, dout * 0.25, B, Nk, gh, gw, D, 1.0, 1.0, gh * P, gw * P, P, prefix=prefix, pitch=pitch)[0]
    a = ops.kp_gather_bwd_det(kp, dout, 0.25, dtype, B, Nk, gh, gw, D, 1.0, 1.0, gh * P, gw * P, P, prefix=prefix, pitch=pitch)
    b = ops.kp_gather_bwd_det(kp, dout, 0.25, dtype, B, Nk, gh, gw, D, 1.0, 1.0, gh * P, gw * P, P, prefix=prefix, pitch=pitch)
    assert a.dtype == dtype and a.shape == ref.shape and torch.equal(a, b)
    assert rel_err(a.float(), ref) < (1e-5 if dtype == torch.float32 else 6e-3)
    if prefix:
        assert float(a[:, :prefix].abs().max()) == 0.0
    if pitch > gw:
        assert float(a[:, prefix:].view(B, gh, pitch, D)[:, :, gw:].abs().max()) == 0.0


def test_kp_gather_multi_grid_mean():
    from gd_amd import ops
    B, gh, gw, D, Nk, P = 2, 6, 5, 48, 11, 14
    toks = [torch.randn(B, 1 + gh * gw, D, generator=_g(20 + t), device="cuda").bfloat16() for t in range(4)]
    kp = torch.rand(B, Nk, 2, generator=_g(30), device="cuda") * torch.tensor([gw * P * 0.5, gh * P * 0.5], device="cuda")
    out = ops.kp_gather_fwd([t[:, 1:] for t in toks], (1 + gh * gw) * D, kp, B, Nk, gh, gw, D, 2.0, 2.0, gh * P, gw * P, P)
    ref = 0
    for t in toks:
        grid = t[:, 1:].float().cpu().reshape(B, gh, gw, D).permute(0, 3, 1, 2)
        ref = ref + O.interpolate_features(grid, kp.cpu() * 2.0, gh * P, gw * P, False, P, P).permute(0, 2, 1)
    assert rel_err(out, ref / 4) < 1e-5


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-6)])
def test_kp_gather_with_the_final_layernorm_applied_at_the_samples(dtype, tol):
    """gd_kp_gather_fwd_ln (round 5): the mean over four RAW tap grids of bilinear samples of LayerNorm(grid) — get_intermediate_feature's
    `self.model.norm(feat)` then interpolate_features, src/finetune_timm_vggt.py:262-288 — from per-token statistics, against the gather of the
    materialised norms (fp64 reference on the same rounded inputs); keypoints on the border and outside; a prefix token; rows with a large mean."""
    from gd_amd import ops
    B, gh, gw, D, Nk, P, pre = 3, 7, 9, 256, 40, 14, 1
    Ng = pre + gh * gw
    toks = [(torch.randn(B, Ng, D, generator=_g(40 + t), device="cuda") * (1 + t) + 3.0 * t).to(dtype) for t in range(4)]
    w = 1 + 0.2 * torch.randn(D, generator=_g(50), device="cuda")
    b = 0.3 * torch.randn(D, generator=_g(51), device="cuda")
    kp = torch.rand(B, Nk, 2, generator=_g(52), device="cuda") * torch.tensor([gw * P - 1.0, gh * P - 1.0], device="cuda")
    kp[:, 0] = 0.0
    kp[:, 1] = torch.tensor([gw * P - 1.0, gh * P - 1.0])
    kp[:, 2] = torch.tensor([gw * P + 30.0, -12.0])
    sts = [ops.layernorm_fwd(t.view(-1, D), w, b, 1e-6, save_stats=True)[1:] for t in toks]
    out = ops.kp_gather_fwd_ln([t[:, pre:] for t in toks], [m.view(B, Ng)[:, pre:] for m, _ in sts], [r.view(B, Ng)[:, pre:] for _, r in sts], Ng, w, b,
                               Ng * D, kp, B, Nk, gh, gw, D, 1.0, 1.0, gh * P, gw * P, P)
    ref = 0
    for t in toks:
        n = F.layer_norm(t.double().cpu(), (D,), w.double().cpu(), b.double().cpu(), 1e-6)
        grid = n[:, pre:].reshape(B, gh, gw, D).permute(0, 3, 1, 2)
        ref = ref + O.interpolate_features(grid, kp.cpu().double(), gh * P, gw * P, False, P, P).permute(0, 2, 1)
    assert rel_err(out, ref / 4) < 5e-6


@pytest.mark.parametrize("taps", [(3, 4, 5, 6), (4, 5, 6, 7)])
def test_deferred_tap_norm_equals_the_materialised_one(taps):
    """vit.forward_all(norm_taps="deferred") with GD_TAP_NORM_FUSED (default): the taps' `model.norm` is not written — the keypoint gather applies it from
    the row statistics the NEXT block's LayerNorm took of the same tensor (gd_kp_gather_fwd_ln), and _TapFn's backward uses those statistics.  Same
    features and the same gradients as the materialising path (option off), f32 and tf32h engines.  Taps (3, 4, 5, 6) of the 8-block test ViT are all
    deferred (the fused gather kernel); with (4, 5, 6, 7) the last tap has no next block: the deferred ones are materialised for the mixed gather
    (_DeferredNormFn: the gradient still reaches _TapFn as the gradient of the NORMED tap).
    The PUBLIC contract (round 6): norm_taps=True returns normed TENSORS whatever the option says; a deferred entry is a vit.DeferredTapNorm — not a
    tensor, so nothing can slice / cast / stack it and silently get un-normalised features — whose .materialize() equals the normed tensor."""
    from gd_amd.finetune import FinetuneGD
    from gd_amd.options import set_option
    from gd_amd.vit import DeferredTapNorm, kp_gather
    for dt, tol in (("f32", 2e-5), ("tf32h", 2e-5)):
        res = {}
        for fused in (1, 0):
            old = set_option("tap_norm_fused", fused)
            try:
                torch.manual_seed(0)
                eng = FinetuneGD(r=4, backbone="vit_tiny_test", patch_size=14, img_size=56, variant="mast3r", geometry="shared", dtype=dt,
                                 lora_b_std=0.05, vit_kwargs=dict(init_values=1.0), teacher_patch=14, adapter_start_idx=2).cuda()
                eng.model.prepare_trainables(None)
                rgb = torch.rand(2, 3, 56, 70, generator=_g(60), device="cuda")
                kp = torch.rand(2, 9, 2, generator=_g(61), device="cuda") * torch.tensor([69.0, 55.0], device="cuda")
                if fused:      # the documented contract of norm_taps=True: tensors, equal to model.norm(tap)
                    with torch.no_grad():
                        raw_t, _, normed_t = eng.model.forward_all(rgb, taps, size=(56, 70), norm_taps=True)
                    assert all(isinstance(t, torch.Tensor) for t in normed_t)
                    for r_, n_ in zip(raw_t, normed_t):
                        assert rel_err(n_, eng.model.norm(r_)) < 1e-6
                raw, x, normed = eng.model.forward_all(rgb, taps, size=(56, 70), norm_taps="deferred")
                flags = [isinstance(t, DeferredTapNorm) for t in normed]
                assert flags == ([i + 1 < 8 for i in taps] if fused else [False] * 4)
                if fused:
                    with torch.no_grad():
                        assert rel_err(normed[0].materialize(), eng.model.norm(raw[0])) < 1e-6
                feat = kp_gather(normed, kp, 4, 5, 1.0, 1.0, 56, 70, 14)
                cw = torch.randn(feat.shape, generator=_g(62), device="cuda")
                cr = torch.randn(raw[0].shape, generator=_g(63), device="cuda")
                ((feat * cw).sum() + sum((t.float() * cr).sum() for t in raw) + 0.1 * (x.float() * cr).sum()).backward()
                res[fused] = (feat.detach().clone(), [q.grad.detach().clone() for q in eng.trainable_parameters() if q.grad is not None])
                eng.model.release_trainables()
                eng.clear_cache()
            finally:
                set_option("tap_norm_fused", old)
        assert rel_err(res[1][0], res[0][0]) < tol
        assert len(res[1][1]) == len(res[0][1]) > 0
        for a, c in zip(res[1][1], res[0][1]):
            assert rel_err(a, c) < 50 * tol


def test_kp_depth_and_patch_mask_golden():
    from gd_amd import ops
    g = load_golden("g09_kp_depth")
    out = ops.kp_depth(g["depth"][None].cuda(), g["kp"].cuda())
    assert rel_err(out, g["out"]) < 1e-6
    g = load_golden("g03_patch_mask")
    m = ops.patch_mask(g["kp"][None].cuda(), g["H"], g["W"], g["patch"])
    assert torch.equal(m[0].bool().cpu(), g["mask"])


@pytest.mark.parametrize("variant", ["vggt", "mast3r"])
def test_smooth_ap_golden(variant):
    from gd_amd import ops
    g = load_golden(f"g08_match_{variant}")
    d1 = g["desc1"].cuda().requires_grad_(True)
    d2 = g["desc2"].cuda().requires_grad_(True)
    loss = ops.smooth_ap(d1, d2, g["pts3d_1"].cuda(), g["pts3d_2"].cuda(), None, variant)
    assert abs(loss.item() - g["loss"]) < 2e-5
    loss.sum().backward()
    assert rel_err(d1.grad, g["gdesc1"]) < 2e-3 and rel_err(d2.grad, g["gdesc2"]) < 2e-3


@pytest.mark.parametrize("variant", ["vggt", "mast3r"])
@pytest.mark.parametrize("N", [300, 296])      # 296: a multiple of 8 — the op then takes its operands without the zero-padded copy
def test_smooth_ap_batched_ragged(variant, N):
    from gd_amd import ops
    P, C = 3, 128
    r1 = torch.randn(P, N, C, generator=_g(40), device="cuda")
    d1 = F.normalize(r1, dim=-1)
    d2 = F.normalize(d1 + 0.04 * torch.randn(P, N, C, generator=_g(41), device="cuda"), dim=-1)
    p1 = torch.rand(P, N, 3, generator=_g(42), device="cuda") * 2
    p2 = p1 + 0.02 * torch.randn(P, N, 3, generator=_g(43), device="cuda")
    counts = torch.tensor([N, 123, 1], dtype=torch.int32, device="cuda")
    a, b = d1.clone().requires_grad_(True), d2.clone().requires_grad_(True)
    loss = ops.smooth_ap(a, b, p1, p2, counts, variant)
    (loss * torch.tensor([1.0, 2.0, 0.5], device="cuda")).sum().backward()
    for p, w in zip(range(P), (1.0, 2.0, 0.5)):
        n = int(counts[p])
        x = d1[p:p + 1, :n].double().cpu().requires_grad_(True)
        y = d2[p:p + 1, :n].double().cpu().requires_grad_(True)
        ref = O.smooth_ap_loss(x, y, p1[p:p + 1, :n].double().cpu(), p2[p:p + 1, :n].double().cpu(), variant)
        ref.backward()
        assert abs(loss[p].item() - ref.item()) < 2e-4 * max(1.0, abs(ref.item()))
        gscale = max(1e-12, float(x.grad.abs().max()))
        assert float((a.grad[p, :n].cpu().double() - w * x.grad[0]).abs().max()) < 2e-2 * w * gscale
        assert float(a.grad[p, n:].abs().max()) == 0.0 if n < N else True


def _head(g, dev="cuda"):
    return {k: g["hp_" + k].to(dev).clone().requires_grad_(True) for k in ("w1", "b1", "ln_w", "ln_b", "w2", "b2")}


def test_depth_losses_golden():
    from gd_amd import ops
    g = load_golden("g11_depth_loss")
    head = _head(g)
    feats = torch.stack([g["kf1"][0], g["kf2"][0]], 0)[None].cuda().requires_grad_(True)     # [1,2,N,D]
    d1 = ops.kp_depth(g["depth_1"][None].cuda(), g["kp_1"].cuda())
    d2 = ops.kp_depth(g["depth_2"][None].cuda(), g["kp_2"].cuda())
    l1, intra = ops.depth_losses(feats, d1, d2, head)
    assert abs(l1.item() - g["depth_loss"]) < 1e-5 and abs(intra.item() - g["intra_loss"]) < 1e-5
    (l1 + intra).sum().backward()
    assert rel_err(feats.grad[0, 0], g["g_kf1"][0]) < 1e-3 and rel_err(feats.grad[0, 1], g["g_kf2"][0]) < 1e-3
    assert rel_err(head["w1"].grad, g["g_w1"]) < 1e-3
    assert rel_err(head["w2"].grad, g["g_w2"]) < 1e-3
    assert rel_err(head["ln_w"].grad, g["g_ln_w"]) < 1e-3


def test_depth_losses_on_a_side_stream():
    """The C ABI takes the caller's stream: the same op on a non-default stream (producers and consumers of u / du on that
    stream only) gives the same numbers — gd_depth_l1 once dropped its stream argument and ran on stream 0."""
    from gd_amd import ops
    g = load_golden("g11_depth_loss")
    N, D = g["kf1"].shape[1], g["kf1"].shape[2]
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        head = _head(g)
        big = torch.randn(4096, 4096, device="cuda")
        for _ in range(4):
            big = big @ big * 1e-2                 # keeps the side stream busy ahead of the op's inputs
        scale = (big.sum() * 0).nan_to_num() + 1.0       # = 1, but only once the matmuls are done
        feats = (torch.stack([g["kf1"][0], g["kf2"][0]], 0)[None].cuda() * scale).requires_grad_(True)
        d1 = ops.kp_depth(g["depth_1"][None].cuda(), g["kp_1"].cuda())
        d2 = ops.kp_depth(g["depth_2"][None].cuda(), g["kp_2"].cuda())
        l1, intra = ops.depth_losses(feats, d1, d2, head)
        (l1 + intra).sum().backward()
    s.synchronize()
    assert abs(l1.item() - g["depth_loss"]) < 1e-5 and abs(intra.item() - g["intra_loss"]) < 1e-5
    assert rel_err(feats.grad[0, 0], g["g_kf1"][0]) < 1e-3 and rel_err(head["w2"].grad, g["g_w2"]) < 1e-3


def test_ranking_golden_and_batched():
    from gd_amd import ops
    g = load_golden("g07_ranking")
    head = _head(g)
    f = g["feats"][0].cuda()
    feats = torch.stack([f, f], 0)[None].clone().requires_grad_(True)
    d = g["depths"].cuda()
    l1, intra = ops.depth_losses(feats, d, d, head, depth_threshold=g["thr"])
    assert abs(intra.item() - g["loss"]) < 1e-5
    intra.sum().backward()
    # both views carry the same set: each gets half the fixture's gradient
    assert rel_err(feats.grad[0, 0], 0.5 * g["gfeats"][0]) < 1e-3
    for k in ("w1", "b1", "ln_w", "ln_b", "w2", "b2"):
        assert rel_err(head[k].grad, g["g_" + k]) < 2e-3, k
    # ragged batch vs oracle
    P, N, D = 2, 40, 64
    hp = {k: v.detach().cpu() for k, v in _head(load_golden("g11_depth_loss"), "cpu").items()}
    hp["w1"] = torch.randn(128, D, generator=torch.Generator().manual_seed(1)) * 0.1
    hd = {k: v.cuda().clone().requires_grad_(True) for k, v in hp.items()}
    ft = torch.randn(P, 2, N, D, generator=_g(50), device="cuda").requires_grad_(True)
    dd1 = torch.rand(P, N, generator=_g(51), device="cuda") * 3
    dd2 = torch.rand(P, N, generator=_g(52), device="cuda") * 3
    counts = torch.tensor([40, 17], dtype=torch.int32, device="cuda")
    l1, intra = ops.depth_losses(ft, dd1, dd2, hd, counts=counts)
    (l1 * 0.7 + intra).sum().backward()
    tot = 0
    hpr = {k: v.double().requires_grad_(True) for k, v in hp.items()}
    fr = ft.detach().double().cpu().requires_grad_(True)
    for p in range(P):
        n = int(counts[p])
        a, b = O.depth_losses(hpr, fr[p, 0:1, :n], fr[p, 1:2, :n], dd1[p:p + 1, :n].double().cpu(), dd2[p:p + 1, :n].double().cpu())
        assert abs(l1[p].item() - a.item()) < 1e-5 and abs(intra[p].item() - b.item()) < 1e-5
        tot = tot + 0.7 * a + b
    tot.backward()
    assert rel_err(ft.grad, fr.grad) < 1e-3
    for k in hd:
        assert rel_err(hd[k].grad, hpr[k].grad) < 2e-3, k


def test_clip_adamw_matches_oracle():
    from gd_amd import ops
    n = 100_003
    p = torch.randn(n, generator=_g(60), device="cuda")
    g = torch.randn(n, generator=_g(61), device="cuda") * 0.01
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    pc, gc = p.cpu().clone(), g.cpu().clone()
    st = [(torch.zeros(n), torch.zeros(n))]
    for step in (1, 2, 3):
        norm = ops.clip_adamw_step(p, g, m, v, step, grad_scale=0.5)
        rn = O.clip_and_adamw([pc], [gc * 0.5], st, step)
        assert abs(norm.item() - rn.item()) < 1e-5 * rn.item()
    assert rel_err(p, pc) < 1e-6
    # a real torch AdamW + clip_grad_norm_ agrees as well
    q = torch.nn.Parameter(torch.randn(1000, generator=torch.Generator().manual_seed(3)))
    opt = torch.optim.AdamW([q], lr=1e-5, weight_decay=1e-4)
    p2 = q.detach().cuda().clone()
    m2, v2 = torch.zeros_like(p2), torch.zeros_like(p2)
    for step in (1, 2):
        gr = torch.randn(1000, generator=torch.Generator().manual_seed(10 + step)) * 3
        q.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_([q], 1.0)
        opt.step()
        ops.clip_adamw_step(p2, gr.cuda(), m2, v2, step)
    assert rel_err(p2, q.detach()) < 1e-6


def test_smooth_ap_me_variant_golden_and_ragged():
    """gd_smooth_ap_me: reference fixture (several positives in one row, rows without any), then a ragged batch vs the oracle."""
    from gd_amd import ops
    g = load_golden("g08_match_me")
    d1 = g["desc1"].cuda().requires_grad_(True)
    d2 = g["desc2"].cuda().requires_grad_(True)
    loss = ops.smooth_ap(d1, d2, g["pts3d_1"].cuda(), g["pts3d_2"].cuda(), variant="me")
    assert abs(loss.item() - float(g["loss"])) < 2e-5
    loss.sum().backward()
    assert rel_err(d1.grad, g["gdesc1"]) < 2e-3 and rel_err(d2.grad, g["gdesc2"]) < 2e-3
    # ragged batch: pair 1 uses 31 of 50 keypoints, pair 2 has no positive at all
    P, N, C = 3, 50, 24
    gen = torch.Generator().manual_seed(5)
    # clustered descriptors: similarities next to the positives' so the temp-0.01 sigmoids are not all saturated
    cen = torch.nn.functional.normalize(torch.randn(4, C, generator=gen), dim=-1)
    a = torch.nn.functional.normalize(cen[torch.randint(0, 4, (P, N), generator=gen)] + 0.03 * torch.randn(P, N, C, generator=gen), dim=-1)
    b = torch.nn.functional.normalize(a + 0.03 * torch.randn(P, N, C, generator=gen), dim=-1)
    p1 = torch.rand(P, N, 3, generator=gen)
    p2 = p1 + 1e-3 * torch.randn(P, N, 3, generator=gen)
    p2[2] = p1[2] + 0.5
    counts = torch.tensor([50, 31, 50], dtype=torch.int32)
    ag, bg = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    out = ops.smooth_ap(ag, bg, p1.cuda(), p2.cuda(), counts=counts.cuda(), variant="me")
    out.sum().backward()
    for q in range(P):
        n = int(counts[q])
        ar, br = a[q:q + 1, :n].double().requires_grad_(True), b[q:q + 1, :n].double().requires_grad_(True)
        if q == 2:
            assert out[q].item() == 0.0 and float(ag.grad[q].abs().max()) == 0.0
            continue
        ref = O.smooth_ap_loss_me(ar, br, p1[q:q + 1, :n].double(), p2[q:q + 1, :n].double())
        ref.backward()
        assert abs(out[q].item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item())), q
        assert float(ar.grad.abs().max()) > 1e-4      # the comparison below is informative
        assert rel_err(ag.grad[q, :n], ar.grad[0]) < 2e-3 and rel_err(bg.grad[q, :n], br.grad[0]) < 2e-3, q
        assert float(ag.grad[q, n:].abs().max()) == 0.0 if n < N else True


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_tap_fn_and_split_pairs_gradients(dtype, tol):
    """vit._TapFn (a tapped block output handed to three consumers, their gradients folded into one LayerNorm-backward
    launch through dres / dres2) and ops.split_pairs (one-cat backward) against plain autograd on the same graph."""
    from gd_amd import ops
    from gd_amd.vit import _TapFn
    B, Nt, D = 4, 37, 128
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, Nt, D, device="cuda", generator=g).to(dtype).requires_grad_(True)
    w = (1 + 0.1 * torch.randn(D, device="cuda", generator=g)).requires_grad_(False)
    b = (0.1 * torch.randn(D, device="cuda", generator=g)).requires_grad_(False)
    c1, c2, c3 = (torch.randn(B, Nt, D, device="cuda", generator=g).to(dtype) for _ in range(3))
    nxt, raw, normed = _TapFn.apply(x, w, b, 1e-6)
    assert torch.equal(nxt, x) and torch.equal(raw, x)
    a1, a2 = ops.split_pairs(raw, B // 2)
    loss = (nxt.float() * c1.float()).sum() + (normed.float() * c2.float()).sum() + \
           (a1.float() * c3[:B // 2].float()).sum() + 2.0 * (a2.float() * c3[B // 2:].float()).sum()
    loss.backward()
    got = x.grad.double()
    xr = x.detach().double().requires_grad_(True)
    nr = torch.nn.functional.layer_norm(xr, (D,), w.double(), b.double(), 1e-6)
    ref = (xr * c1.double()).sum() + (nr * c2.double()).sum() + (xr[:B // 2] * c3[:B // 2].double()).sum() + \
          2.0 * (xr[B // 2:] * c3[B // 2:].double()).sum()
    ref.backward()
    assert rel_err(normed, nr.detach()) < tol and rel_err(got, xr.grad) < tol
    # only some consumers used: the missing gradients are None inside the backward
    x2 = x.detach().clone().requires_grad_(True)
    n2, r2, m2 = _TapFn.apply(x2, w, b, 1e-6)
    (n2.float() * c1.float()).sum().backward()
    assert rel_err(x2.grad, c1.double()) < tol


def test_flat_allreduce_cabi_single_rank():
    """gd_comm_* / gd_flat_allreduce (RCCL bound at run time): a one-rank communicator leaves the buffer unchanged under
    both algorithms; argument errors are reported through gd_last_error (multi-rank runs need one GPU per rank)."""
    from gd_amd import dp
    from gd_amd._lib import GdHipError
    from gd_amd._lib import lib
    comm = dp.RcclComm(0, 1)
    assert 20000 <= lib().gd_comm_rccl_version() < 30000       # the copy torch carries, 2.x ABI (enum values / id size)
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    y = x.clone()
    comm.all_reduce_(y, algo=0)
    comm.all_reduce_(y, algo=1)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    with pytest.raises(GdHipError):
        comm.all_reduce_(y, algo=2)
    red = dp.DirectGradReducer(y, comm)
    red.start()
    assert red.finish() == 1.0
    comm.close()


def test_fp16_operand_outputs_of_the_gather_and_im2col_kernels_equal_their_f32_forms_rounded():
    """tf32h engine: gd_kp_patch_gather writes the refine-conv GEMM's left operand as fp16 directly (`half=True`), gd_patch_im2col /
    gd_patch_im2col_strided the patch projection's (dtype fp16) — the f32 form of the same launch rounded once, no separate cast pass."""
    from gd_amd import ops
    B, gh, gw, D, P = 2, 9, 11, 64, 14
    grid = torch.randn(B, 1 + gh * gw, D, generator=_g(31), device="cuda")
    kp = torch.rand(B, 17, 2, generator=_g(32), device="cuda") * torch.tensor([gw * P - 1.0, gh * P - 1.0], device="cuda")
    kp[:, 0] = 0.0
    kp[:, 1] = -1.0
    a32 = ops.kp_patch_gather(grid[:, 1:], (1 + gh * gw) * D, kp, B, 17, gh, gw, D, 1.0, 1.0, gh * P, gw * P, P)
    a16 = ops.kp_patch_gather(grid[:, 1:], (1 + gh * gw) * D, kp, B, 17, gh, gw, D, 1.0, 1.0, gh * P, gw * P, P, half=True)
    assert a32.dtype == torch.float32 and a16.dtype == torch.float16 and a16.shape == a32.shape
    # (the fp16 form mixes the four neighbours with the products rounded in a different order: equal to one fp16 ulp, not bit for bit)
    assert float((a16.float() - a32).abs().max()) <= 2.0 ** -10 * float(a32.abs().max()) and float((a16 != a32.half()).float().mean()) < 0.05
    mean, std = (0.48, 0.45, 0.40), (0.26, 0.25, 0.27)
    img = torch.rand(B, 3, 56, 70, generator=_g(33), device="cuda")
    for stride in (None, (7, 7)):
        c32 = ops.patch_im2col(img, 56, 70, P, 640, mean, std, torch.float32, stride=stride)
        c16 = ops.patch_im2col(img, 56, 70, P, 640, mean, std, torch.float16, stride=stride)
        assert c16.dtype == torch.float16 and c16.shape == c32.shape and torch.equal(c16, c32.half()), stride


def test_loss_combine_matches_the_weighted_sum_and_mean():
    """ops.loss_combine (gd_loss_combine_fwd / _bwd) = mean over pairs of w_ap ap + w_depth depth + w_intra intra + w_kl kl with the pairs whose keypoint
    count is zero contributing a constant zero (src/finetune_timm_vggt.py:599-616, src/finetune_timm_mast3r.py:604-607): value, reported terms and the
    gradient of every term against the torch expression, with and without counts, a zero weight included."""
    from gd_amd import ops
    P = 37
    g = _g(70)
    ts = [torch.rand(P, generator=g, device="cuda").requires_grad_(True) for _ in range(4)]
    w = (1.0, 0.0, 0.7, 2.5)
    for counts in (None, torch.tensor([0 if i % 5 == 2 else 7 for i in range(P)], dtype=torch.int32, device="cuda")):
        for t in ts:
            t.grad = None
        loss, terms = ops.loss_combine(*ts, w, counts)
        (3.0 * loss).backward()
        got = [t.grad.clone() for t in ts]
        keep = torch.ones(P, device="cuda") if counts is None else (counts > 0).float()
        rs = [t.detach().clone().requires_grad_(True) for t in ts]
        ref = (keep * sum(wi * r for wi, r in zip(w, rs))).mean()
        (3.0 * ref).backward()
        assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
        assert terms.shape == (4, P) and not terms.requires_grad
        for i in range(4):
            assert torch.allclose(terms[i], keep * ts[i].detach(), rtol=0, atol=0)
            assert torch.allclose(got[i], rs[i].grad, rtol=1e-6, atol=1e-9)
