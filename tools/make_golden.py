"""Generate tests/golden/*.npz from the REFERENCE's own code and pin oracle/ against it.

Runs ONLY in the build container (needs /root/reference, read-only).  For each fixture
it (1) calls the reference function (imported through tools/ref_import.py), (2) calls the
oracle/ restatement on the same seeded inputs and asserts agreement, (3) stores inputs
and the reference's outputs as a small .npz.  The fixtures are data; no reference source
travels.  Usage:  python tools/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_import as R  # noqa: E402
import gd_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **conv)
    sz = os.path.getsize(os.path.join(OUT, name + ".npz"))
    print(f"  wrote {name}.npz ({sz / 1024:.1f} KiB)")


def close(a, b, tol=1e-5, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item() / max(1e-12, b.abs().max().item())
    assert err <= tol, f"oracle != reference for {what}: rel err {err:.3e}"
    return err


def g(seed):
    return torch.Generator().manual_seed(seed)


L, Fn, M = R.ref_utils()
MAST3R, VGGT, ME = R.ref_modules()

# ---------------------------------------------------------------- G1 sigmoid
x = torch.linspace(-1.2, 1.2, 97)
y = Fn.sigmoid(x, temp=0.01)
close(O.sigmoid_t(x, 0.01), y, 1e-7, "sigmoid")
save("g01_sigmoid", x=x, y=y, temp=0.01)

# ---------------------------------------------------------------- G2 interpolate_features
for P in (14, 16):
    C, ph, pw, N = 8, 5, 7, 19
    desc = torch.randn(1, C, ph, pw, generator=g(2), requires_grad=True)
    pts = torch.rand(1, N, 2, generator=g(3)) * torch.tensor([pw * P + 20.0, ph * P + 20.0]) - 10.0
    out = Fn.interpolate_features(desc, pts, ph * P, pw * P, normalize=False, patch_size=P, stride=P)
    w = torch.randn(out.shape, generator=g(4))
    (out * w).sum().backward()
    o2 = O.interpolate_features(desc.detach(), pts, ph * P, pw * P, False, P, P)
    close(o2, out, 1e-6, "interpolate_features")
    outn = Fn.interpolate_features(desc, pts, ph * P, pw * P, normalize=True, patch_size=P, stride=P)
    close(O.interpolate_features(desc.detach(), pts, ph * P, pw * P, True, P, P), outn, 1e-6, "interp norm")
    save(f"g02_interp_p{P}", desc=desc, pts=pts, h=ph * P, w=pw * P, patch=P, out=out, out_norm=outn,
         gout=w, gdesc=desc.grad)

# ---------------------------------------------------------------- G3 patch mask
kp = torch.tensor([[0., 0.], [15.9, 3.], [16., 16.], [-1., 5.], [111.9, 95.9], [112., 10.], [50., 96.], [63., 47.]])
m = Fn.get_patch_mask_from_kp_tensor(kp, 96, 112, 16)
assert torch.equal(m, O.patch_mask_from_kp(kp, 96, 112, 16))
save("g03_patch_mask", kp=kp, H=96, W=112, patch=16, mask=m)

# ---------------------------------------------------------------- G4/G5 masked cost + KL
hw = 48
cost = torch.rand(1, hw, hw, generator=g(5))
rm = torch.rand(hw, generator=g(6)) > 0.3
a = Fn.get_masked_patch_cost(cost, rm)
b = Fn.get_masked_patch_cost(cost * 4 - 2, rm, use_softmax=True, temperature=0.7)
close(O.masked_patch_cost(cost, rm), a, 1e-6, "masked cost")
close(O.masked_patch_cost(cost * 4 - 2, rm, use_softmax=True, temperature=0.7), b, 1e-6, "masked softmax")
t = a.clone()
t[0, 3, :5] = 1e-10
kl = L.kl_divergence_map(t, b)
close(O.kl_divergence_map(t, b), kl, 1e-6, "kl")
save("g04_masked_cost", cost=cost, row_mask=rm, renorm=a, softmax_t07=b, kl_t=t, kl_p=b, kl=kl)


# ---------------------------------------------------------------- G6 cost loss, both variants
class _CostSelf:
    def __init__(self, feats, patch, rps=None):
        self.feats = list(feats)
        self.patch_size = patch
        self.resize_patch_size = rps

    def get_feature_cost(self, rgbs, normalize=True, resize=True):
        return self.feats.pop(0)


def cost_fixture(name, variant, ph, pw, C, P, seed):
    hw_ = ph * pw
    f1 = torch.randn(1, ph, pw, C, generator=g(seed), requires_grad=True)
    f2 = torch.randn(1, ph, pw, C, generator=g(seed + 1), requires_grad=True)
    t1 = torch.softmax(3 * torch.randn(1, hw_, hw_, generator=g(seed + 2)), -1)
    t2 = torch.softmax(3 * torch.randn(1, hw_, hw_, generator=g(seed + 3)), -1)
    H, W = ph * P, pw * P
    rgb = torch.zeros(1, 3, H, W)
    if variant == "mast3r":
        N = 11
        kp1 = torch.stack([torch.randint(0, W, (N,), generator=g(seed + 4)),
                           torch.randint(0, H, (N,), generator=g(seed + 5))], -1).float()[None]
        kp2 = torch.stack([torch.randint(0, W, (N,), generator=g(seed + 6)),
                           torch.randint(0, H, (N,), generator=g(seed + 7))], -1).float()[None]
        fs = _CostSelf([f1, f2], P)
        loss = MAST3R.FinetuneMASt3RTIMM.calculate_cost_loss(fs, rgb, rgb, kp1, kp2, t1[0], t2[0], 0)
        m1 = O.patch_mask_from_kp(kp1[0], H, W, P)
        m2 = O.patch_mask_from_kp(kp2[0], H, W, P)
        extra = dict(kp_1=kp1, kp_2=kp2)
    else:
        pm1 = torch.rand(H, W, generator=g(seed + 4)) > 0.3
        pm2 = torch.rand(H, W, generator=g(seed + 5)) > 0.3
        fs = _CostSelf([f1, f2], P, rps=P)
        loss = VGGT.FinetuneVGGTTIMM.calculate_cost_loss(fs, rgb, rgb, t1, t2, mask_1=pm1, mask_2=pm2)
        m1 = F.interpolate(pm1[None, None].float(), size=(ph, pw), mode="nearest").bool().view(-1)
        m2 = F.interpolate(pm2[None, None].float(), size=(ph, pw), mode="nearest").bool().view(-1)
        extra = dict(pixel_mask_1=pm1, pixel_mask_2=pm2)
    loss.backward()
    o = O.cost_volume_kl(f1.detach().view(1, hw_, C), f2.detach().view(1, hw_, C), t1, t2, m1, m2, variant)
    e = close(o, loss, 1e-5, f"cost loss {variant}")
    print(f"  cost loss {variant}: ref {loss.item():.6f} oracle rel err {e:.2e}")
    save(name, f1=f1.detach().view(hw_, C), f2=f2.detach().view(hw_, C), t1=t1[0], t2=t2[0], m1=m1, m2=m2,
         loss=loss, g1=f1.grad.view(hw_, C), g2=f2.grad.view(hw_, C), ph=ph, pw=pw, patch=P, **extra)


cost_fixture("g06_cost_vggt", "vggt", 6, 8, 32, 14, 60)
cost_fixture("g06_cost_mast3r", "mast3r", 6, 8, 32, 16, 70)


# ---------------------------------------------------------------- G7 ranking loss + head
def head_params(head):
    fl = head.fusion_layer
    return {"w1": fl[0].weight.detach().clone(), "b1": fl[0].bias.detach().clone(),
            "ln_w": fl[1].weight.detach().clone(), "ln_b": fl[1].bias.detach().clone(),
            "w2": fl[3].weight.detach().clone(), "b2": fl[3].bias.detach().clone()}


torch.manual_seed(7)
N, D = 17, 32
head = M.DepthAwareFeatureFusion(input_dim=D, use_tanh=True)
with torch.no_grad():
    head.fusion_layer[1].weight.add_(0.1 * torch.randn(128))
    head.fusion_layer[1].bias.add_(0.1 * torch.randn(128))
feats = torch.randn(1, N, D, requires_grad=True)
depths = torch.rand(1, N) * 2 + 0.5
depths[0, 3] = depths[0, 4] + 0.01  # an invalid (below-threshold) pair
loss = L.pairwise_logistic_ranking_loss(head, feats, depths, depth_threshold=0.05)
loss.backward()
hp = head_params(head)
close(O.pairwise_ranking_loss(hp, feats.detach(), depths, 0.05), loss, 1e-5, "ranking loss")
hg = {k: v for k, v in zip(["w1", "b1", "ln_w", "ln_b", "w2", "b2"],
                           [head.fusion_layer[0].weight.grad, head.fusion_layer[0].bias.grad,
                            head.fusion_layer[1].weight.grad, head.fusion_layer[1].bias.grad,
                            head.fusion_layer[3].weight.grad, head.fusion_layer[3].bias.grad])}
save("g07_ranking", feats=feats, depths=depths, loss=loss, gfeats=feats.grad, thr=0.05,
     **{"hp_" + k: v for k, v in hp.items()}, **{"g_" + k: v for k, v in hg.items()})


# ---------------------------------------------------------------- G8 matching loss x3
class _MatchSelf:
    def __init__(self, descs):
        self.descs = list(descs)
        self.device = torch.device("cpu")
        self.thres3d_neg = 0.1
        self.thres3d_pos = 5e-3

    def get_feature(self, rgb, kp, normalize=True):
        return self.descs.pop(0)


def match_fixture(name, variant, seed):
    N, C, Hh, Ww = 23, 16, 12, 14
    # clustered descriptors: within-cluster similarities sit next to the positives' (~0.99), so the temp-0.01
    # sigmoids are NOT all saturated and loss / gradients are informative
    centers = F.normalize(torch.randn(6, C, generator=g(seed + 9)), dim=-1)
    assign = torch.randint(0, 6, (N,), generator=g(seed + 10))
    r1 = (centers[assign] + 0.03 * torch.randn(N, C, generator=g(seed)))[None].requires_grad_(True)
    r2 = torch.randn(1, N, C, generator=g(seed + 1), requires_grad=True)
    d1 = F.normalize(r1, dim=-1)
    d2 = F.normalize(d1.detach() + 0.03 * r2, dim=-1)
    pm1 = torch.rand(Hh, Ww, 3, generator=g(seed + 2)) * 2
    kp1 = torch.stack([torch.randint(0, Ww, (N,), generator=g(seed + 3)),
                       torch.randint(0, Hh, (N,), generator=g(seed + 4))], -1).float()[None]
    kp2 = torch.stack([torch.randint(0, Ww, (N,), generator=g(seed + 5)),
                       torch.randint(0, Hh, (N,), generator=g(seed + 6))], -1).float()[None]
    pm2 = torch.rand(Hh, Ww, 3, generator=g(seed + 7)) * 2
    # make the matched 3-D points close: write pm1's points (plus noise) into pm2 at kp2
    p1 = pm1[kp1[0, :, 1].long(), kp1[0, :, 0].long()]
    pm2[kp2[0, :, 1].long(), kp2[0, :, 0].long()] = p1 + 0.03 * torch.randn(N, 3, generator=g(seed + 8))
    fs = _MatchSelf([d1, d2])
    if variant == "vggt":
        loss = VGGT.FinetuneVGGTTIMM.calculate_matching_loss(fs, None, None, kp1, kp2, pm1, pm2)
    else:
        loss = MAST3R.FinetuneMASt3RTIMM.calculate_matching_loss(fs, None, None, kp1, kp2, pm1, pm2)
    loss.backward()
    pts1 = pm1[kp1[..., 1].long(), kp1[..., 0].long()]
    pts2 = pm2[kp2[..., 1].long(), kp2[..., 0].long()]
    o = O.smooth_ap_loss(d1.detach(), d2.detach(), pts1, pts2, variant)
    close(o, loss, 1e-5, f"matching loss {variant}")
    # gradient wrt the normalised descriptors (what the fused kernel consumes)
    d1n = d1.detach().clone().requires_grad_(True)
    d2n = d2.detach().clone().requires_grad_(True)
    O.smooth_ap_loss(d1n, d2n, pts1, pts2, variant).backward()
    print(f"  matching loss {variant}: {loss.item():.6f}  max|grad| {d1n.grad.abs().max().item():.3e}")
    assert loss.item() > 0.05 and d1n.grad.abs().max().item() > 1e-4
    save(name, desc1=d1.detach(), desc2=d2.detach(), pts3d_1=pts1, pts3d_2=pts2, loss=loss,
         gdesc1=d1n.grad, gdesc2=d2n.grad, graw1=r1.grad)


match_fixture("g08_match_vggt", "vggt", 80)
match_fixture("g08_match_mast3r", "mast3r", 90)

# ---------------------------------------------------------------- G9 extract_kp_depth
depth = torch.rand(9, 13, generator=g(9)) * 5 + 0.5
kp = torch.stack([torch.randint(0, 13, (10,), generator=g(10)), torch.randint(0, 9, (10,), generator=g(11))], -1).float()[None]
kp[0, 0] = torch.tensor([0., 0.])
kp[0, 1] = torch.tensor([12., 8.])
dk = Fn.extract_kp_depth(depth, kp)
close(O.extract_kp_depth(depth, kp), dk, 1e-6, "extract_kp_depth")
save("g09_kp_depth", depth=depth, kp=kp, out=dk)

# ---------------------------------------------------------------- G10 LoRA / Adapter modules
torch.manual_seed(10)
D, r, Ntok = 24, 4, 9
qkv = nn.Linear(D, 3 * D)
aq, bq, av, bv = nn.Linear(D, r, bias=False), nn.Linear(r, D, bias=False), nn.Linear(D, r, bias=False), nn.Linear(r, D, bias=False)
lq = M._LoRA_qkv(qkv, aq, bq, av, bv)
x = torch.randn(2, Ntok, D, requires_grad=True)
y = lq(x)
wy = torch.randn(y.shape)
(y * wy).sum().backward()
lo = {"a_q": aq.weight.detach(), "b_q": bq.weight.detach(), "a_v": av.weight.detach(), "b_v": bv.weight.detach()}
close(O.lora_qkv(x.detach(), qkv.weight.detach(), qkv.bias.detach(), lo), y, 1e-6, "lora qkv")
ad = M.Adapter(D, 6)
blk = M.BlockWithAdapter(nn.Identity(), ad)
x2 = torch.randn(2, Ntok, D, requires_grad=True)
y2 = blk(x2)
wy2 = torch.randn(y2.shape)
(y2 * wy2).sum().backward()
close(O.adapter(x2.detach(), {"down": ad.down.weight.detach(), "up": ad.up.weight.detach()}), y2, 1e-6, "adapter")
save("g10_lora_adapter", x=x, qkv_w=qkv.weight, qkv_b=qkv.bias, a_q=aq.weight, b_q=bq.weight, a_v=av.weight,
     b_v=bv.weight, y=y, gy=wy, gx=x.grad, g_a_q=aq.weight.grad, g_b_q=bq.weight.grad, g_a_v=av.weight.grad,
     g_b_v=bv.weight.grad, ad_x=x2, ad_down=ad.down.weight, ad_up=ad.up.weight, ad_y=y2, ad_gy=wy2,
     ad_gx=x2.grad, ad_g_down=ad.down.weight.grad, ad_g_up=ad.up.weight.grad)


# ---------------------------------------------------------------- G11 depth loss (unbound)
class _DepthSelf:
    def __init__(self, feats, head):
        self.feats = list(feats)
        self.depth_diff_head = head
        self.device = torch.device("cpu")

    def get_intermediate_feature(self, rgb, pts=None, n=None, reshape=True, return_class_token=False, normalize=True):
        return self.feats.pop(0)


torch.manual_seed(11)
N, D = 13, 32
head = M.DepthAwareFeatureFusion(input_dim=D)
kf1 = torch.randn(1, N, D, requires_grad=True)
kf2 = torch.randn(1, N, D, requires_grad=True)
dm1 = torch.rand(10, 12) * 4 + 0.5
dm2 = torch.rand(10, 12) * 4 + 0.5
kp1 = torch.stack([torch.randint(0, 12, (N,)), torch.randint(0, 10, (N,))], -1).float()[None]
kp2 = torch.stack([torch.randint(0, 12, (N,)), torch.randint(0, 10, (N,))], -1).float()[None]
dl, il = MAST3R.FinetuneMASt3RTIMM.calculate_depth_loss(_DepthSelf([kf1, kf2], head), dm1, dm2, None, None, kp1, kp2)
dl_v, il_v = VGGT.FinetuneVGGTTIMM.calculate_depth_loss(
    _DepthSelf([kf1, kf2], head), {"depth_pred_1": dm1, "depth_pred_2": dm2}, None, None, kp1, kp2)
assert torch.equal(dl, dl_v) and torch.equal(il, il_v)
(dl + il).backward()
hp = head_params(head)
o_dl, o_il = O.depth_losses(hp, kf1.detach(), kf2.detach(), O.extract_kp_depth(dm1, kp1), O.extract_kp_depth(dm2, kp2))
close(o_dl, dl, 1e-5, "depth l1")
close(o_il, il, 1e-5, "intra depth")
save("g11_depth_loss", kf1=kf1, kf2=kf2, depth_1=dm1, depth_2=dm2, kp_1=kp1, kp_2=kp2, depth_loss=dl, intra_loss=il,
     g_kf1=kf1.grad, g_kf2=kf2.grad, **{"hp_" + k: v for k, v in hp.items()},
     g_w1=head.fusion_layer[0].weight.grad, g_w2=head.fusion_layer[3].weight.grad,
     g_ln_w=head.fusion_layer[1].weight.grad)

# ---------------------------------------------------------------- G12 in-tree DINOv2 ViT + LoRA/adapters
vt = R.ref_vit()
torch.manual_seed(12)
CFG = dict(patch=14, dim=64, depth=6, heads=4, ln_eps=1e-6, pos_interp="dinov2")
ref_vit = vt.DinoVisionTransformer(img_size=56, patch_size=14, embed_dim=64, depth=6, num_heads=4, mlp_ratio=4,
                                   init_values=1.0, block_chunks=0,
                                   block_fn=vt.partial(vt.Block, attn_class=vt.MemEffAttention)).eval()
with torch.no_grad():
    for n_, q in ref_vit.named_parameters():
        if "norm" in n_ or "gamma" in n_ or "bias" in n_:
            q.add_(0.1 * torch.randn_like(q))
    ref_vit.cls_token.copy_(0.02 * torch.randn_like(ref_vit.cls_token))
for q in ref_vit.parameters():
    q.requires_grad = False
base_sd = {k: v.detach().clone() for k, v in ref_vit.state_dict().items() if k != "mask_token"}
w_as, w_bs, adapters = [], [], []
for bi in range(4, 6):
    blk = ref_vit.blocks[bi]
    lin = blk.attn.qkv
    a_q, b_q = nn.Linear(64, 4, bias=False), nn.Linear(4, 64, bias=False)
    a_v, b_v = nn.Linear(64, 4, bias=False), nn.Linear(4, 64, bias=False)
    nn.init.normal_(b_q.weight, std=0.05)
    nn.init.normal_(b_v.weight, std=0.05)
    w_as += [a_q, a_v]
    w_bs += [b_q, b_v]
    blk.attn.qkv = M._LoRA_qkv(lin, a_q, b_q, a_v, b_v)
    adp = M.Adapter(64, 8)
    adapters.append(adp)
    ref_vit.blocks[bi] = M.BlockWithAdapter(blk, adp)
trainable = {"lora": {}, "adapter": {}}
for j, bi in enumerate(range(4, 6)):
    trainable["lora"][bi] = {"a_q": w_as[2 * j].weight.detach().clone(), "b_q": w_bs[2 * j].weight.detach().clone(),
                             "a_v": w_as[2 * j + 1].weight.detach().clone(), "b_v": w_bs[2 * j + 1].weight.detach().clone()}
    trainable["adapter"][bi] = {"down": adapters[j].down.weight.detach().clone(), "up": adapters[j].up.weight.detach().clone()}
for size in (56, 70):
    img = torch.randn(2, 3, size, size, generator=g(120 + size))
    taps_ref = ref_vit._get_intermediate_layers_not_chunked(img, [4, 5])
    xn = ref_vit.norm(taps_ref[-1])
    wt = [torch.randn(t.shape, generator=g(121 + i)) for i, t in enumerate(taps_ref)]
    for q in [*(l.weight for l in w_as), *(l.weight for l in w_bs), *(p_ for a_ in adapters for p_ in a_.parameters())]:
        q.grad = None
    sum((t * w_).sum() for t, w_ in zip(taps_ref, wt)).backward()
    taps_o, xo = O.vit_forward(img, base_sd, CFG, trainable, taps=(4, 5))
    for a_, b_ in zip(taps_o, taps_ref):
        close(a_, b_, 2e-5, f"vit tap ({size})")
    close(O.final_norm(xo, base_sd, CFG), xn, 2e-5, "vit final norm")
    grads = {}
    for j, bi in enumerate(range(4, 6)):
        grads[f"g_a_q_{bi}"] = w_as[2 * j].weight.grad
        grads[f"g_b_q_{bi}"] = w_bs[2 * j].weight.grad
        grads[f"g_a_v_{bi}"] = w_as[2 * j + 1].weight.grad
        grads[f"g_b_v_{bi}"] = w_bs[2 * j + 1].weight.grad
        grads[f"g_down_{bi}"] = adapters[j].down.weight.grad
        grads[f"g_up_{bi}"] = adapters[j].up.weight.grad
    tr_flat = {}
    for bi in (4, 5):
        for k, v in trainable["lora"][bi].items():
            tr_flat[f"lora_{bi}_{k}"] = v
        for k, v in trainable["adapter"][bi].items():
            tr_flat[f"adapter_{bi}_{k}"] = v
    save(f"g12_vit_{size}", img=img, tap4=taps_ref[0], tap5=taps_ref[1], xnorm=xn, wt4=wt[0], wt5=wt[1],
         **{"sd." + k: v for k, v in base_sd.items()}, **tr_flat, **grads)

# ---------------------------------------------------------------- G13 rope_2d (reference CPU fallback class)
install_ok = True
try:
    from models.pos_embed import RoPE2D  # dust3r/croco/models/pos_embed.py:106-158 (pure-torch fallback)
    B, N, H, D = 2, 12, 3, 16
    tok = torch.randn(B, H, N, D, generator=g(13))
    pos = torch.stack([torch.randint(0, 5, (B, N), generator=g(14)), torch.randint(0, 7, (B, N), generator=g(15))], -1)
    out = RoPE2D(100.0, 1.0)(tok, pos)  # (B,H,N,D)
    o = O.rope_2d(tok.transpose(1, 2).contiguous(), pos, 100.0, 1.0).transpose(1, 2)
    close(o, out, 1e-5, "rope2d")
    back = O.rope_2d(o.transpose(1, 2).contiguous(), pos, 100.0, -1.0).transpose(1, 2)
    close(back, tok, 1e-5, "rope2d inverse")
    save("g13_rope2d", tokens_bhnd=tok, positions=pos, out_bhnd=out, base=100.0)
except Exception as e:  # pragma: no cover
    print("  [skip] rope2d fixture:", type(e).__name__, e)

print("all golden fixtures written; oracle pinned against the reference")

# ---------------------------------------------------------------- G15 teacher glue (VGGT side) + depth rasteriser
from vggt.utils.geometry import unproject_depth_map_to_point_map  # noqa: E402

torch.manual_seed(15)
S, Hh, Ww = 2, 18, 22


def rand_cam(seed):
    gg = g(seed)
    A = torch.randn(3, 3, generator=gg)
    Rm, _ = torch.linalg.qr(A)
    if torch.det(Rm) < 0:
        Rm[:, 0] = -Rm[:, 0]
    Rm = torch.matrix_exp(0.15 * (A - A.t()))          # a small rotation
    t = 0.2 * torch.randn(3, generator=gg)
    E = torch.cat([Rm, t[:, None]], 1)
    Kc = torch.tensor([[20.0, 0, Ww / 2], [0, 21.0, Hh / 2], [0, 0, 1]])
    return E, Kc


E1, K1 = rand_cam(151)
E2, K2 = rand_cam(152)
depth = 1.0 + 2 * torch.rand(S, Hh, Ww, generator=g(153))
pm = torch.from_numpy(unproject_depth_map_to_point_map(depth.unsqueeze(-1).numpy(), torch.stack([E1, E2]).numpy(),
                                                       torch.stack([K1, K2]).numpy())).float()
close(O.unproject_depth(depth, torch.stack([E1, E2]), torch.stack([K1, K2])), pm, 1e-5, "unproject")
m1, m2 = Fn.get_coview_masks(pm[0], pm[1], K1, E1, K2, E2, (Hh, Ww))
o1, o2 = O.coview_masks(pm[0], pm[1], K1, E1, K2, E2, (Hh, Ww))
assert torch.equal(m1, o1) and torch.equal(m2, o2) and 0 < int(m1.sum()) < Hh * Ww
conf = 1 + torch.rand(Hh, Ww, generator=g(154))
kps = Fn.sample_keypoints_nms(m1, conf, N=400, min_distance=2)      # M <= N: no RNG involved
assert torch.equal(kps, O.nms_keypoints(m1, conf, 400, 2))
pts = torch.randn(500, 3, generator=g(155)) * torch.tensor([1.0, 1.0, 0.5]) + torch.tensor([0, 0, 2.0])
pts[:40, 2] = -1.0
dimg = Fn.point_cloud_to_depth(pts, K1, Ww, Hh, torch.device("cpu"))
close(O.point_cloud_to_depth(pts, K1, Ww, Hh), dimg, 1e-6, "point_cloud_to_depth")
kpf = torch.stack([torch.randint(0, Ww, (30,), generator=g(156)), torch.randint(0, Hh, (30,), generator=g(157))], -1).float()[None]
cmask = torch.rand(Hh, Ww, generator=g(158)) > 0.4
fk, fi = Fn.filter_kp_by_conf(kpf, cmask)
ok_, oi_ = O.filter_kp_by_conf(kpf, cmask)
assert torch.equal(fk, ok_) and torch.equal(fi, oi_)
save("g15_teacher_glue", depth=depth, E1=E1, K1=K1, E2=E2, K2=K2, point_maps=pm, mask_1=m1, mask_2=m2, conf=conf,
     nms_kps=kps, nms_min_distance=2, pc_points=pts, pc_depth=dimg, fk_kp=kpf, fk_mask=cmask, fk_idx=fi)

# ---------------------------------------------------------------- G16 fast_reciprocal_NNs (MASt3R side)
from mast3r.fast_nn import fast_reciprocal_NNs  # noqa: E402

Hd, Wd, Dd = 24, 32, 24
base = F.normalize(torch.randn(Hd, Wd, Dd, generator=g(160)), dim=-1)
shift = torch.roll(base, shifts=(1, -2), dims=(0, 1))
d1 = base
d2 = F.normalize(shift + 0.15 * torch.randn(Hd, Wd, Dd, generator=g(161)), dim=-1)
xy1, xy2 = fast_reciprocal_NNs(d1, d2, subsample_or_initxy1=4, device="cpu", dist="dot", block_size=2 ** 13)
o1, o2 = O.reciprocal_nns(d1, d2, subsample=4)
assert np.array_equal(np.asarray(xy1), o1.numpy()) and np.array_equal(np.asarray(xy2), o2.numpy()), "reciprocal NNs"
print(f"  reciprocal NNs: {len(xy1)} matches from {(Hd // 4) * (Wd // 4)} seeds")
c1 = torch.rand(Hd, Wd, generator=g(162))
c2 = torch.rand(Hd, Wd, generator=g(163))
k1f, k2f = O.mast3r_keypoint_filter(torch.as_tensor(np.asarray(xy1).copy()), torch.as_tensor(np.asarray(xy2).copy()), c1, c2)
save("g16_reciprocal_nns", desc1=d1, desc2=d2, subsample=4, xy1=np.asarray(xy1).copy(), xy2=np.asarray(xy2).copy(),
     conf1=c1, conf2=c2, kp1_filtered=k1f[0], kp2_filtered=k2f[0])
print("teacher-glue fixtures written")
