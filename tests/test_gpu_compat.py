"""The reference-signature surface (gd_amd/compat.py) against the reference-generated fixtures G1/G2/G4/G7/G9/G3
and the oracle, the eager DepthAwareFeatureFusion.forward, duck-typed wrapper discovery with foreign classes, the
`patch_embed.proj.stride` override (G21), and plan invalidation when frozen weights are reloaded."""
import pytest
import torch
import torch.nn as nn

import gd_oracle as O
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def test_sigmoid_golden_and_grad():
    from gd_amd.compat import sigmoid
    g = load_golden("g01_sigmoid")
    x = g["x"].cuda().requires_grad_(True)
    y = sigmoid(x, temp=g["temp"])
    assert rel_err(y, g["y"]) < 1e-6
    y.sum().backward()
    xr = g["x"].double().requires_grad_(True)
    O.sigmoid_t(xr, g["temp"]).sum().backward()
    assert rel_err(x.grad, xr.grad) < 1e-5
    big = torch.tensor([-3.0, 3.0], device="cuda", requires_grad=True)     # |x / temp| > 50: clamped, zero gradient
    yb = sigmoid(big, temp=0.01)
    yb.sum().backward()
    assert torch.equal(big.grad, torch.zeros_like(big.grad)) and yb[0] < 1e-20 and yb[1] == 1.0


@pytest.mark.parametrize("P", [14, 16])
def test_interpolate_features_golden(P):
    from gd_amd.compat import interpolate_features
    g = load_golden(f"g02_interp_p{P}")
    desc = g["desc"].cuda().requires_grad_(True)
    out = interpolate_features(desc, g["pts"].cuda(), h=g["h"], w=g["w"], normalize=False, patch_size=P, stride=P)
    assert out.shape == g["out"].shape and rel_err(out, g["out"]) < 1e-5
    (out * g["gout"].cuda()).sum().backward()
    assert rel_err(desc.grad, g["gdesc"]) < 1e-5
    outn = interpolate_features(g["desc"].cuda(), g["pts"].cuda(), h=g["h"], w=g["w"], normalize=True, patch_size=P, stride=P)
    assert rel_err(outn, g["out_norm"]) < 1e-5


def test_masked_patch_cost_and_kl_golden():
    from gd_amd.compat import get_masked_patch_cost, kl_divergence_map
    g = load_golden("g04_masked_cost")
    cost, rm = g["cost"].cuda(), g["row_mask"].cuda().bool()
    a = get_masked_patch_cost(cost, rm)
    b = get_masked_patch_cost(cost * 4 - 2, rm, use_softmax=True, temperature=0.7)      # the fixture's softmax input (tools/make_golden.py)
    assert rel_err(a, g["renorm"]) < 1e-6 and rel_err(b, g["softmax_t07"]) < 1e-6
    kl = kl_divergence_map(g["kl_t"].cuda(), g["kl_p"].cuda())
    assert abs(kl.item() - g["kl"]) < 1e-6 * max(1.0, abs(g["kl"]))


@pytest.mark.parametrize("use_softmax", [False, True])
@pytest.mark.parametrize("col_mask", [False, True])
def test_masked_patch_cost_kl_gradients_vs_oracle(use_softmax, col_mask):
    """cost -> get_masked_patch_cost -> kl_divergence_map, gradients wrt the student cost AND the teacher map, hw = 1369
    (4-byte aligned rows), including all-masked rows, entries below eps and a column mask."""
    from gd_amd.compat import get_masked_patch_cost, kl_divergence_map
    gen = torch.Generator().manual_seed(3)
    B, R, C = 2, 137, 1369
    cost = torch.rand(B, R, C, generator=gen)
    cost[0, 5] = 0.0                      # a row whose sum hits the eps clamp
    t = torch.softmax(3 * torch.randn(B, R, C, generator=gen), -1)
    t[1, 7, :50] = 0.0                    # teacher entries below eps
    m1 = torch.rand(R, generator=gen) > 0.3
    m2 = (torch.rand(C, generator=gen) > 0.2) if col_mask else None
    cg, tg = cost.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    p = get_masked_patch_cost(cg, m1.cuda(), m2.cuda() if col_mask else None, use_softmax=use_softmax, temperature=0.5)
    loss = kl_divergence_map(tg, p)
    loss.backward()
    cr, trf = cost.double().requires_grad_(True), t.double().requires_grad_(True)
    keep = m1[:, None] & (m2[None, :] if col_mask else torch.ones(1, C, dtype=torch.bool))
    mc = torch.where(keep[None], cr, torch.zeros_like(cr))
    pr = torch.softmax(mc / 0.5, -1) if use_softmax else mc / mc.sum(-1, keepdim=True).clamp_min(1e-8)
    lr = O.kl_divergence_map(trf, pr)
    lr.backward()
    assert abs(loss.item() - lr.item()) < 1e-5 * abs(lr.item())
    assert rel_err(cg.grad, cr.grad) < 2e-4 and rel_err(tg.grad, trf.grad) < 1e-4


class _RefStyleHead(nn.Module):
    """A depth head defined OUTSIDE gd_amd with the layout of utils/model.py:88-99 (what a reference caller passes)."""

    def __init__(self, D):
        super().__init__()
        self.depth_attention = nn.Sequential(nn.Linear(1, 128), nn.GELU(), nn.Linear(128, D), nn.Sigmoid())
        self.fusion_layer = nn.Sequential(nn.Linear(D, 128), nn.LayerNorm(128), nn.GELU(), nn.Linear(128, 1))


def test_pairwise_logistic_ranking_loss_golden_and_batched():
    from gd_amd.compat import pairwise_logistic_ranking_loss
    g = load_golden("g07_ranking")
    D = g["feats"].shape[-1]
    head = _RefStyleHead(D).cuda()
    fl = head.fusion_layer
    with torch.no_grad():
        for t, k in ((fl[0].weight, "w1"), (fl[0].bias, "b1"), (fl[1].weight, "ln_w"), (fl[1].bias, "ln_b"),
                     (fl[3].weight, "w2"), (fl[3].bias, "b2")):
            t.copy_(g["hp_" + k])
    feats = g["feats"].cuda().requires_grad_(True)
    loss = pairwise_logistic_ranking_loss(head, feats, g["depths"].cuda(), depth_threshold=g["thr"])
    assert abs(loss.item() - g["loss"]) < 1e-5
    loss.backward()
    assert rel_err(feats.grad, g["gfeats"]) < 1e-3
    for t, k in ((fl[0].weight, "w1"), (fl[0].bias, "b1"), (fl[1].weight, "ln_w"), (fl[1].bias, "ln_b"), (fl[3].weight, "w2"),
                 (fl[3].bias, "b2")):
        assert rel_err(t.grad, g["g_" + k]) < 2e-3, k
    # B = 3 sets: the reference's mean runs over the valid pairs of ALL sets (utils/losses.py:36-40)
    gen = torch.Generator().manual_seed(11)
    f3 = torch.randn(3, 29, D, generator=gen)
    d3 = torch.rand(3, 29, generator=gen) * torch.tensor([3.0, 0.2, 1.0])[:, None]     # set 1: few pairs pass the threshold
    hp = {k: g["hp_" + k].double() for k in ("w1", "b1", "ln_w", "ln_b", "w2", "b2")}
    fr = f3.double().requires_grad_(True)
    diff = fr.unsqueeze(1) - fr.unsqueeze(2)                     # [B, i, j, D] = f_j - f_i
    s = O.depth_head(diff.reshape(3, -1, D), hp).view(3, 29, 29)
    dd = d3.double().unsqueeze(1) - d3.double().unsqueeze(2)
    valid = dd.abs() > 0.05
    ref = torch.log1p(torch.exp(-torch.sign(dd) * s))[valid].mean()
    ref.backward()
    fg = f3.cuda().requires_grad_(True)
    for q in head.parameters():
        q.grad = None
    out = pairwise_logistic_ranking_loss(head, fg, d3.cuda(), depth_threshold=0.05)
    out.backward()
    assert abs(out.item() - ref.item()) < 1e-5 and rel_err(fg.grad, fr.grad) < 1e-3
    # no valid pair at all -> 0 (utils/losses.py:37-38)
    z = pairwise_logistic_ranking_loss(head, fg.detach(), torch.ones(3, 29, device="cuda"), depth_threshold=0.05)
    assert z.item() == 0.0


def test_extract_kp_depth_and_patch_mask_compat():
    from gd_amd.compat import extract_kp_depth, get_patch_mask_from_kp_tensor
    g = load_golden("g09_kp_depth")
    out = extract_kp_depth(g["depth"].cuda(), g["kp"].cuda())
    assert rel_err(out, g["out"]) < 1e-6
    m = load_golden("g03_patch_mask")
    mask = get_patch_mask_from_kp_tensor(m["kp"].cuda(), m["H"], m["W"], m["patch"])
    assert torch.equal(mask.cpu(), m["mask"].bool())


def test_depth_aware_feature_fusion_eager_forward():
    """DepthAwareFeatureFusion.forward (utils/model.py:101-127) on HIP, both branches, values and every gradient."""
    from gd_amd.model import DepthAwareFeatureFusion
    torch.manual_seed(3)
    D = 96
    head = DepthAwareFeatureFusion(input_dim=D).cuda()
    with torch.no_grad():
        for q in head.parameters():
            q.add_(0.05 * torch.randn_like(q))
    x = torch.randn(2, 37, D, device="cuda", requires_grad=True)
    w = torch.randn(2, 37, device="cuda")
    out = head(x)
    assert out.shape == (2, 37)
    (out * w).sum().backward()
    hp = {k: v.detach().double().cpu().requires_grad_(True) for k, v in head.head_params().items()}
    xr = x.detach().double().cpu().requires_grad_(True)
    ref = O.depth_head(xr, hp)
    (ref * w.double().cpu()).sum().backward()
    assert rel_err(out, ref) < 1e-5 and rel_err(x.grad, xr.grad) < 1e-4
    for k, v in head.head_params().items():
        assert rel_err(v.grad, hp[k].grad) < 1e-4, k
    # depths branch: features * depth_attention(depth) first
    d = torch.rand(2, 37, device="cuda") * 4
    out2 = head(x.detach(), d)
    da = head.depth_attention
    att = torch.sigmoid(torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(
        d.cpu().double().unsqueeze(-1), da[0].weight.detach().cpu().double(), da[0].bias.detach().cpu().double())),
        da[2].weight.detach().cpu().double(), da[2].bias.detach().cpu().double()))
    ref2 = O.depth_head(x.detach().cpu().double() * att, {k: v.detach() for k, v in hp.items()})
    assert rel_err(out2, ref2) < 1e-4


# ---- wrappers defined OUTSIDE gd_amd, named and laid out like utils/model.py's: the engine must discover them by shape ----
class _LoRA_qkv(nn.Module):
    def __init__(self, qkv, linear_a_q, linear_b_q, linear_a_v=None, linear_b_v=None, linear_a_k=None, linear_b_k=None):
        super().__init__()
        self.qkv, self.linear_a_q, self.linear_b_q = qkv, linear_a_q, linear_b_q
        self.linear_a_v, self.linear_b_v, self.linear_a_k, self.linear_b_k = linear_a_v, linear_b_v, linear_a_k, linear_b_k
        self.dim = qkv.in_features


class Adapter(nn.Module):
    def __init__(self, dim, bottleneck_dim):
        super().__init__()
        self.down = nn.Linear(dim, bottleneck_dim, bias=False)
        self.relu = nn.ReLU()
        self.up = nn.Linear(bottleneck_dim, dim, bias=False)


class BlockWithAdapter(nn.Module):
    def __init__(self, block, adapter):
        super().__init__()
        self.block, self.adapter = block, adapter

    def forward(self, x):
        raise AssertionError("the engine must fuse this wrapper, not call it")


def test_duck_typed_wrapper_discovery_with_foreign_classes():
    """Wrap a GDViT exactly as src/finetune_timm_vggt.py:134-162 does, but with wrapper classes that are NOT gd_amd.model's:
    the taps and the LoRA / adapter gradients still match the oracle."""
    from gd_amd.vit import create_vit
    from gd_testutil import oracle_params
    torch.manual_seed(0)
    model = create_vit("vit_tiny_test", patch_size=14, img_size=56, dtype="f32", init_values=1.0)
    for p_ in model.parameters():
        p_.requires_grad = False
    r, D = 4, model.embed_dim
    for i, blk in enumerate(model.blocks[4:]):
        q = blk.attn.qkv
        assert q.in_features == D
        mk = lambda a, b: nn.Linear(a, b, bias=False)
        aq, bq, av, bv = mk(D, r), mk(r, D), mk(D, r), mk(r, D)
        for t in (bq, bv):
            nn.init.normal_(t.weight, std=0.05)
        blk.attn.qkv = _LoRA_qkv(q, aq, bq, av, bv)
        model.blocks[4 + i] = BlockWithAdapter(blk, Adapter(D, 64))
    model = model.cuda()

    class _Eng:                                    # the attributes gd_testutil.oracle_params reads
        pass
    eng = _Eng()
    eng.model, eng.patch_size, eng.variant, eng.resize_patch_size, eng.geometry = model, 14, "vggt", 14, "shared"
    eng.target_res, eng.downsample_factor = 640, 8
    eng.refine_conv = nn.Conv2d(D, D, 3, padding=1)
    from gd_amd.model import DepthAwareFeatureFusion
    eng.depth_diff_head = DepthAwareFeatureFusion(D)
    p, tr, _, _, cfg = oracle_params(eng)
    img = torch.rand(2, 3, 56, 70, generator=torch.Generator().manual_seed(1))
    taps = model._intermediate_layers(img.cuda(), n=[5, 7])
    wt = [torch.randn(t.shape, generator=torch.Generator().manual_seed(2 + i)) for i, t in enumerate(taps)]
    sum((t * w.cuda()).sum() for t, w in zip(taps, wt)).backward()
    for d in tr.values():
        for blk in d.values():
            for k in blk:
                blk[k] = blk[k].double().requires_grad_(True)
    pd = {k: v.double() for k, v in p.items()}
    rt, _ = O.vit_forward(O.normalize_image(img.double(), cfg["mean"], cfg["std"]), pd, cfg, tr, taps=(5, 7))
    for a, b in zip(taps, rt):
        assert rel_err(a, b) < 2e-5
    sum((t * w.double()).sum() for t, w in zip(rt, wt)).backward()
    for i in (4, 5, 7):
        q = model.blocks[i].block.attn.qkv
        assert rel_err(q.linear_a_q.weight.grad, tr["lora"][i]["a_q"].grad) < 2e-4
        assert rel_err(q.linear_b_v.weight.grad, tr["lora"][i]["b_v"].grad) < 2e-4
        assert rel_err(model.blocks[i].adapter.up.weight.grad, tr["adapter"][i]["up"].grad) < 2e-4


def _g21_model(dtype):
    from gd_amd.vit import GDViT
    g = load_golden("g21_stride_override")
    model = GDViT(img_size=56, patch_size=14, embed_dim=64, depth=2, num_heads=1, init_values=1.0, dtype=dtype)
    missing, unexpected = model.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return g, model.cuda()


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-5), ("bf16", 2e-2)])
def test_patch_embed_stride_override_g21(dtype, tol):
    """src/evaluate_timm.py:262-272 mutates `patch_embed.proj.stride` on the live model (overlapping patches for dense
    tracking features) and binds `_fix_pos_enc`'s resampler as `interpolate_pos_encoding`.  Fixture G21 = that recipe on the
    reference's in-tree ViT.  The engine must honour the stride (a) with its built-in position resampling and (b) with a
    foreign `interpolate_pos_encoding` bound onto the model exactly as the reference binds it."""
    import math
    import types
    import torch.nn.functional as F
    g, model = _g21_model(dtype)

    def foreign_fix_pos_enc(patch_size, stride_hw):       # a `_fix_pos_enc`-shaped override written for this test
        def interpolate_pos_encoding(self, x, w, h):
            npatch, N = x.shape[1] - 1, self.pos_embed.shape[1] - 1
            if npatch == N and w == h:
                return self.pos_embed
            w0, h0 = 1 + (w - patch_size) // stride_hw[1], 1 + (h - patch_size) // stride_hw[0]
            assert w0 * h0 == npatch
            m, dim = int(math.sqrt(N)), x.shape[-1]
            pp = F.interpolate(self.pos_embed[:, 1:].reshape(1, m, m, dim).permute(0, 3, 1, 2), mode="bicubic", align_corners=False,
                               scale_factor=((w0 + 0.1) / m, (h0 + 0.1) / m), recompute_scale_factor=False)
            return torch.cat((self.pos_embed[:, :1], pp.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)
        return interpolate_pos_encoding

    for tag in ("sq", "rect"):
        img = g[f"{tag}.img"].cuda()
        if hasattr(model, "interpolate_pos_encoding"):
            del model.interpolate_pos_encoding
        model.patch_embed.proj.stride = (14, 14)
        base = model.forward_features(img)
        assert base.shape[1] == 1 + (img.shape[-2] // 14) * (img.shape[-1] // 14)
        model.patch_embed.proj.stride = (7, 7)
        with torch.no_grad():
            out = model.forward_features(img)
        assert out.shape == g[f"{tag}.xnorm"].shape
        assert rel_err(out, g[f"{tag}.xnorm"]) < tol, (tag, "built-in")
        model.interpolate_pos_encoding = types.MethodType(foreign_fix_pos_enc(14, (7, 7)), model)
        model.invalidate_plans()
        with torch.no_grad():
            out2 = model.forward_features(img)
        assert rel_err(out2, g[f"{tag}.xnorm"]) < tol, (tag, "foreign override")
        gh, gw = 1 + (img.shape[-2] - 14) // 7, 1 + (img.shape[-1] - 14) // 7
        assert rel_err(model._pos_strided(gh, gw, img.shape[-2], img.shape[-1]), g[f"{tag}.pos"][0]) < 1e-6


def test_patch_embed_bad_stride_fails_loudly():
    from gd_amd._lib import GdHipError
    from gd_amd.vit import create_vit
    model = create_vit("vit_tiny_test", patch_size=14, img_size=56, dtype="f32").cuda()
    img = torch.rand(1, 3, 56, 56, device="cuda")
    model.patch_embed.proj.stride = (0, 7)
    with pytest.raises(GdHipError):
        model.forward_features(img)
    model.patch_embed.proj.stride = (14, 14)
    model.forward_features(img)


def test_plans_follow_reloaded_frozen_weights():
    """A forward builds cast / folded / transposed copies of the frozen weights; load_state_dict afterwards must invalidate them."""
    from gd_amd.vit import create_vit
    torch.manual_seed(0)
    a = create_vit("vit_tiny_test", patch_size=14, img_size=56, dtype="f32", init_values=1.0).cuda()
    b = create_vit("vit_tiny_test", patch_size=14, img_size=56, dtype="f32", init_values=1.0).cuda()
    with torch.no_grad():
        for q in b.parameters():
            q.add_(0.02 * torch.randn_like(q))
    img = torch.rand(1, 3, 56, 70, device="cuda")
    ya, yb = a.forward_features(img), b.forward_features(img)
    assert rel_err(ya, yb) > 1e-3
    a.load_state_dict(b.state_dict())
    assert torch.equal(a.forward_features(img), yb)
    a.float()                                              # nn.Module._apply route: plans dropped as well
    assert torch.equal(a.forward_features(img), yb)
