"""The tf32h engine's RANGE contract at its edges (fp16 operands have 5 exponent bits where TF32 has 8): per-block power-of-two gradient
scales taken on the device, non-finite gradients that stay non-finite, and the two counters that report what fell outside the contract.

The contract: inside one block's backward every gradient operand is cast as fp16(x * s) with s the power of two that puts the
block's incoming gradient maximum into (4, 8] — 2^13 of headroom above it, full 11-bit precision down to 2^-17 of it, decreasing precision
(fp16 subnormals) down to 2^-27, zero below.  TF32 keeps 11 bits over 2^±127; what the narrower window costs is bounded by the magnitude of
what falls out of it, relative to the block's largest gradient entry."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _tf32(x):
    return ((x.clone().view(torch.int32) + 0x1000) & ~0x1FFF).view(torch.float32)


def test_amax_scale_is_a_power_of_two_and_a_non_finite_gradient_poisons_it():
    """gd_amax_scale: target/2 < max|x| s <= target with s a power of two; a NaN or an Inf anywhere in x makes s and 1/s NaN (fmaxf-style
    reductions drop NaNs: the f32 / bf16 engines would carry the NaN into every weight gradient, and so must this one); all-zero -> 1."""
    from gd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(3000, 768, generator=g, device="cuda") * 3e-6
    sc = ops.amax_scale(x, 8.0)
    s = float(sc[0])
    assert 4.0 < float(x.abs().max()) * s <= 8.0 and float(sc[1]) * s == 1.0 and float(torch.log2(sc[0])) == round(float(torch.log2(sc[0])))
    for bad in (float("nan"), float("inf"), -float("inf")):
        y = x.clone()
        y[1234, 77] = bad
        sb = ops.amax_scale(y, 8.0)
        assert bool(torch.isnan(sb[0])) and bool(torch.isnan(sb[1])), bad
    assert float(ops.amax_scale(torch.zeros(64, 64, device="cuda"), 8.0)[0]) == 1.0
    # the shared slot buffer is left zeroed by every call: a second tensor is not contaminated by the first one's maximum
    small = torch.full((256, 64), 1e-9, device="cuda")
    assert 4.0 < 1e-9 * float(ops.amax_scale(small, 8.0)[0]) <= 8.0


def test_scaled_casts_count_what_left_fp16s_range():
    """gd_cast_f16_ex's two counters against the same classification done in torch: results beyond +-65504 (saturated, not Inf) and non-zero
    inputs below fp16's normal range 2^-14 after the scale (subnormal or flushed)."""
    from gd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(2048, 768, generator=g, device="cuda")
    x[:, :100] *= 1e-7        # after s = 2: ~2e-7, far below 6.1e-5
    x[::7, 300] = 5e4         # after s = 2: 1e5 > 65504
    x[5, 5] = 0.0
    s = torch.tensor([2.0], device="cuda")
    cnt = torch.zeros(128, dtype=torch.int32, device="cuda")
    ops.set_range_counters(cnt)
    try:
        h = ops.cast16(x, scale_dev=s)
    finally:
        ops.set_range_counters(None)
    v = (x * 2.0).abs()
    assert ops.range_totals(cnt) == [int((v > 65504.0).sum()), int(((v < 2.0 ** -14) & (v != 0)).sum())]
    assert bool(torch.isfinite(h.float()).all()) and float(h.float().abs().max()) == 65504.0
    # forward operands (no device scale) are not counted
    cnt.zero_()
    ops.set_range_counters(cnt)
    try:
        ops.cast16(x)
    finally:
        ops.set_range_counters(None)
    assert ops.range_totals(cnt) == [0, 0]


@pytest.mark.parametrize("dy16", [False, True])
def test_layernorm_backward_with_the_device_side_extras(dy16):
    """gd_layernorm_bwd_ex: dx against the fp64 LayerNorm backward; dy as fp16 under a scale that the pass undoes; the fp16 copy of dx under the
    block's scale; and max |dx| taken on the way out — the scale registered for dx equals gd_amax_scale of dx (the next block's, for free)."""
    from gd_amd import ops
    M, D = 4100, 768
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(M, D, generator=g, device="cuda") * 2 + 0.3
    gamma = 1 + 0.1 * torch.randn(D, generator=g, device="cuda")
    dy = torch.randn(M, D, generator=g, device="cuda") * 1e-6
    dres = torch.randn(M, D, generator=g, device="cuda") * 1e-6
    beta = torch.zeros(D, device="cuda")
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-6)
    sc = ops.amax_scale(dy, ops.GRAD_TARGET)
    dyin = ops.cast16(dy, scale_dev=sc[0:1]) if dy16 else dy
    dx, dx16 = ops.layernorm_bwd(dyin, x, gamma, mean, rstd, dres=dres, cast_scale=sc[0:1], dy_scale=sc[1:2] if dy16 else None, want_amax=True)
    xd = x.double().requires_grad_(True)
    dy_eff = (dyin.double() / float(sc[0])) if dy16 else dy.double()
    torch.nn.functional.layer_norm(xd, (D,), gamma.double(), None, 1e-6).backward(dy_eff)
    ref = xd.grad + dres.double()
    assert rel_err(dx, ref) < 2e-5
    assert rel_err(dx16.double() / float(sc[0]), ref) < 1e-3
    got = ops.amax_take(dx)
    assert got is not None and ops.amax_take(dx) is None                       # popped
    want = ops.amax_scale(dx, ops.GRAD_TARGET)
    assert float(got[0]) == float(want[0]) and float(got[1]) == float(want[1])
    # a NaN in dy reaches dx and the registered scale
    dyn = dy.clone()
    dyn[17, 5] = float("nan")
    dxn = ops.layernorm_bwd(dyn, x, gamma, mean, rstd, dres=dres, want_amax=True)
    assert bool(torch.isnan(dxn[17]).all()) and bool(torch.isnan(ops.amax_take(dxn)[0]))


class _GradGain(torch.autograd.Function):
    """identity forward, gradient times `gain` backward: stands in for a stack whose gradient grows from block to block"""

    @staticmethod
    def forward(ctx, x, gain):
        ctx.gain = gain
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.gain, None


def _stack_grads(dtype, gain, seed=0):
    from gd_amd.finetune import FinetuneGD
    from gd_amd.vit import run_block
    torch.manual_seed(seed)
    eng = FinetuneGD(r=4, backbone="vit_base", patch_size=14, img_size=224, variant="vggt", geometry="shared", dtype=dtype, teacher_patch=14,
                     lora_b_std=1e-2, vit_kwargs=dict(init_values=1.0)).cuda()
    g = torch.Generator(device="cuda").manual_seed(11)
    img = torch.rand(4, 3, 224, 224, generator=g, device="cuda")           # 4 x 257 tokens = 1028 rows: the fused tf32h kernels' shapes
    w = torch.randn(4, 257, 768, generator=g, device="cuda") * 1e-3
    eng.model.prepare_trainables(None)
    x = eng.model.embed(img)
    for blk in eng.model.blocks:
        x = run_block(blk, x)
        x = _GradGain.apply(x, gain)
    (x.float() * w).sum().backward()
    eng.model.release_trainables()
    ps = [l.weight for l in eng.w_As] + [l.weight for l in eng.w_Bs] + list(eng.adapters.parameters())
    return [p.grad.detach().double().cpu() for p in ps], eng


def test_gradient_that_grows_2p14_down_the_stack_keeps_tf32_class_products():
    """Eight trainable blocks with the gradient multiplied by 4 between consecutive blocks: 2^14 from block 11's incoming gradient to block 4's —
    beyond the 2^13 of fp16 headroom a single per-step scale leaves (silent saturation at +-65504).  With the scale taken per block (from the
    maximum the producing LayerNorm backward measures on its way out) every block's LoRA / adapter gradient stays within a few percent of the f32
    engine's tensor by tensor (2 % on the whole vector), the lowest block no worse than the top one, nothing saturates, and the counters say so."""
    from gd_amd import ops
    ref, _ = _stack_grads("f32", 4.0)
    cnt = torch.zeros(128, dtype=torch.int32, device="cuda")
    ops.set_range_counters(cnt)
    try:
        got, _ = _stack_grads("tf32h", 4.0)
    finally:
        ops.set_range_counters(None)
    errs = []
    for a, b in zip(got, ref):
        assert bool(torch.isfinite(a).all())
        errs.append(float((a - b).norm() / b.norm()) if float(b.norm()) > 0 else 0.0)
    print("per-tensor relative errors (A x16, B x16, adapters x16):", [round(e, 4) for e in errs])
    # per TENSOR (48 of them, the small LoRA factors included): a few percent at worst — block 4, whose incoming gradient is
    # 2^14 times block 11's, carries the accumulated rounding noise of the blocks above it and nothing worse; the whole trainable vector within 2 %
    assert max(errs) < 4e-2, errs
    va, vb = torch.cat([a.reshape(-1) for a in got]), torch.cat([b.reshape(-1) for b in ref])
    assert float((va - vb).norm() / vb.norm()) < 2e-2
    # block 4's LoRA-A against block 11's: rounding noise accumulates over the seven blocks in between (measured 1.1 % against 0.4 %), it does not
    # explode — a saturating gradient would be off by O(1)
    assert sum(errs[0:2]) / 2 < 4.0 * max(sum(errs[14:16]) / 2, 5e-3)
    assert ops.range_totals(cnt)[0] == 0, ops.range_totals(cnt)
    # the norms really span the range: the lowest trainable block's LoRA-A gradient is > 2^11 times the top block's
    assert float(ref[0].norm()) > 2.0 ** 11 * float(ref[14].norm()), (float(ref[0].norm()), float(ref[14].norm()))


def test_gradient_bulk_2p20_under_one_outlier_the_contract_and_its_counter():
    """A gradient whose bulk sits 2^-20 under one outlier (the reference's smooth-AP at temperature 0.01 produces such sparse gradients): under
    the block's scale the bulk lands at 2^-17...2^-18 — fp16 SUBNORMALS, 6-7 significant bits instead of 11.  What the contract promises and this
    test holds: (1) the product's error measured against the WHOLE result (the norm the weight gradients and the next block see) is within
    1.05 x emulated TF32, for dX (row-wise products) and for a weight gradient (a contraction over the rows); (2) the bulk rows alone are NOT
    TF32-class (their relative error is the subnormals' 2^-7, asserted so that the statement stays honest); (3) the engine's below-normal
    counter reports every such operand, so a run can tell."""
    from gd_amd import ops
    M, N, K = 2048, 768, 768
    g = torch.Generator(device="cuda").manual_seed(5)
    d = torch.randn(M, K, generator=g, device="cuda") * 2.0 ** -22        # |bulk| ~ 2^-22 ... 2^-20
    d[7, 13] = 1.0
    w = torch.randn(N, K, generator=g, device="cuda") * 0.05
    x = torch.randn(M, 64, generator=g, device="cuda").half()
    cnt = torch.zeros(128, dtype=torch.int32, device="cuda")
    ops.set_range_counters(cnt)
    try:
        sc = ops.amax_scale(d, ops.GRAD_TARGET)
        dh = ops.cast16(d, scale_dev=sc[0:1])
    finally:
        ops.set_range_counters(None)
    assert float(sc[0]) == 8.0
    got = ops.gemm_nt(dh, ops.cast16(w), out_dtype=torch.float32, alpha_dev=sc[1:2])
    ref = d.double() @ w.double().t()
    t32 = _tf32(d).double() @ _tf32(w).double().t()
    fro = lambda a, r: float((a.double() - r).norm() / r.norm())
    assert fro(got, ref) < 1.05 * fro(t32, ref)                                            # (1) dX, whole result
    gw = ops.gemm_tn(dh, x, alpha_dev=sc[1:2])
    gw_ref = d.double().t() @ x.double()
    gw_t32 = _tf32(d).double().t() @ x.double()
    assert fro(gw, gw_ref) < 1.05 * max(fro(gw_t32, gw_ref), 2e-4)                         # (1) weight gradient
    bulk = torch.ones(M, dtype=torch.bool, device="cuda")
    bulk[7] = False
    e_bulk = float((got[bulk].double() - ref[bulk]).norm() / ref[bulk].norm())
    assert 1e-3 < e_bulk < 5e-2, e_bulk                                                     # (2) subnormal operands: 6-7 bits
    lost = int(((d * 8.0).abs() < 2.0 ** -14).sum())
    assert ops.range_totals(cnt) == [0, lost] and lost > 0.99 * M * K                       # (3) and the counter says so


def test_over_capacity_kept_rows_give_a_nan_gradient_norm_in_the_fp16_operand_engine():
    """The kept-row cost volume poisons a pair whose mask keeps more rows than the caller's bound with NaN (never a silently truncated sum).  In
    the tf32h engine that NaN must survive the fp16 operand casts (v_med3 saturation maps NaN to a finite value): the block's gradient scale
    becomes NaN and with it every weight gradient and the clip norm, as in the f32 engine."""
    from gd_amd import ops
    from gd_amd.finetune import FinetuneGD
    from gd_testutil import synthetic_batch
    for dtype in ("f32", "tf32h"):
        torch.manual_seed(0)
        eng = FinetuneGD(r=4, backbone="vit_base", patch_size=14, img_size=518, variant="mast3r", geometry="shared", dtype=dtype, teacher_patch=14,
                         lora_b_std=1e-3, vit_kwargs=dict(init_values=1.0)).cuda()
        eng.configure_optimizers()
        batch = synthetic_batch(1, 518, 518, 300, 37 * 37, "cuda", seed=7, teacher_patch=14)
        # the engine derives the bound from the keypoint tensor's width: hand it keypoints that mark MORE distinct patches than a narrower
        # tensor promises by calling the loss directly with a too-small bound
        rgbs = torch.cat([batch["rgb_1"], batch["rgb_2"]], 0)
        eng.model.prepare_trainables(eng._flat)
        eng._fuse_taps = True
        fc = eng.get_feature_cost(rgbs, with_norm=True)
        f, inv, f16 = fc if len(fc) == 3 else (fc[0], fc[1], None)
        m1 = ops.patch_mask(batch["kp_1"], 518, 518, 14)
        m2 = ops.patch_mask(batch["kp_2"], 518, 518, 14)
        assert int(m1.sum()) > 128
        f1, f2 = ops.split_pairs(f, 1)
        ts = batch.get("cost_tstats")
        if ts is None:
            c1, c2 = ops.pad_teacher_maps(batch["cost_1"]), ops.pad_teacher_maps(batch["cost_2"])
            ts = ops.cost_volume_teacher_stats(c1, c2)
        else:
            c1, c2 = batch["cost_1"], batch["cost_2"]
        kl = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, "mast3r", tstats=ts, inv_norms=(inv[:1], inv[1:]), x3=eng.model.opfmt,
                                h16=None if f16 is None else (f16[:1], f16[1:]), kept_rows_max=100)          # capacity 128 < kept rows
        assert bool(torch.isnan(kl).all())
        eng.backward(kl.mean())
        eng.model.release_trainables()
        eng.clear_cache()
        norm = eng.optimizer_step()
        assert bool(torch.isnan(norm)), (dtype, float(norm))


def test_fp16_conversions_saturate_infinities_and_propagate_nan():
    """every fp32 -> fp16 conversion of the tf32h engine saturates at +-65504 (finite overflow and +-inf alike) and PROPAGATES NaN: the clamp is
    gfx950's IEEE-754-2019 minimum / maximum on the converted halves (csrc/gd_common.h f16_sat2), not a median-of-three — v_med3_f32 returns
    -65504 for a NaN and would hide a poisoned activation.  Checked on the stand-alone cast, on an fp16-C GEMM epilogue (persistent and tile
    kernels) and on the LayerNorm forward's fp16 output."""
    from gd_amd import ops
    nan, inf = float("nan"), float("inf")
    x = torch.tensor([[nan, inf, -inf, 7e4, -7e4, 65519.0, 1.0, -0.0]] * 2, device="cuda")
    h = ops.cast16(x)
    assert torch.isnan(h[:, 0]).all() and torch.equal(h[:, 1:6], torch.tensor([[65504.0, -65504.0, 65504.0, -65504.0, 65504.0]] * 2, device="cuda").half())
    assert h[0, 6] == 1.0 and h[0, 7] == 0.0
    g = torch.Generator(device="cuda").manual_seed(5)
    for M, N, K in [(2048, 512, 256), (300, 128, 64)]:           # persistent 256 x 256 kernel / 128 x 128 tile kernel
        a = torch.randn(M, K, device="cuda", generator=g).half()
        w = torch.randn(N, K, device="cuda", generator=g).half()
        a[3, 5] = nan
        o = ops.gemm_nt(a, w, out_dtype=torch.float16)
        assert torch.isnan(o[3]).all() and torch.isfinite(o[:3]).all() and torch.isfinite(o[4:]).all(), (M, N, K)
        big = ops.gemm_nt(a[8:] * 0 + 200.0, w * 0 + 200.0, out_dtype=torch.float16)      # 200 * 200 * K > 65504: saturates, does not become inf
        assert torch.equal(big, torch.full_like(big, 65504.0))
    xs = torch.randn(64, 768, device="cuda", generator=g)
    xs[7, 100] = nan
    y, _, _ = ops.layernorm_fwd(xs, torch.ones(768, device="cuda"), torch.zeros(768, device="cuda"), 1e-6, out_dtype=torch.float16)
    assert torch.isnan(y[7]).all() and torch.isfinite(y[:7]).all() and torch.isfinite(y[8:]).all()
