"""G18: ONE FULL OPTIMISATION STEP written by the reference's own code — build container only.

`FinetuneVGGTTIMM.training_step` (src/finetune_timm_vggt.py:577-639) and `FinetuneMASt3RTIMM`'s loss methods are called
UNBOUND on a fake self whose student is the reference's in-tree DINOv2 ViT (vggt/layers/vision_transformer.py) wrapped with
the reference's `_LoRA_qkv` / `Adapter` / `BlockWithAdapter` (utils/model.py) exactly as finetune_timm_vggt.py:134-162 does;
the teacher (`extract_vggt_features`, `sample_keypoints`) is replaced by fixture targets.  Two pairs (= two DDP ranks of the
reference, mean-reduced gradients: src/main.py:147-151), then `torch.nn.utils.clip_grad_norm_(…, 1.0)` (Lightning's
gradient_clip_val, src/main.py:153) and the optimiser of the reference's own `configure_optimizers` (AdamW lr 1e-5, wd 1e-4).

Writes tests/golden/g18_full_step_{vggt,mast3r}.npz: inputs, every weight before the step, the per-pair loss terms, the clip
norm, and every trainable tensor AFTER the step.  Asserts that oracle/gd_oracle.py reproduces all of it (fp32, <= 2e-5).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import ref_import as R  # noqa: E402
import gd_oracle as O  # noqa: E402

R.install()
L, Fn, M = R.ref_utils()
MA, VG, _ = R.ref_modules()
vt = R.ref_vit()
OUT = os.path.join(HERE, "..", "tests", "golden")
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
DIM, DEPTH, HEADS, PATCH, NATIVE = 64, 8, 1, 14, 56
H, W, N, P = 70, 84, 14, 2          # teacher-resolution images (5 x 6 teacher tokens), keypoints, pairs
TARGET_RES, DOWN = 112, 8           # keypoint features on a 11 x 14 token grid (the reference's 640 / 8, scaled down)


def g(seed):
    return torch.Generator().manual_seed(seed)


def close(a, b, tol, what):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    assert err <= tol, f"{what}: rel err {err:.3e} > {tol}"
    return err


class TimmShim(nn.Module):
    """The attribute surface the reference reads from its timm model, on the in-tree DINOv2 ViT."""

    def __init__(self, vit):
        super().__init__()
        self.vit = vit
        self.blocks = vit.blocks
        self.norm = vit.norm
        self.num_prefix_tokens = 1

    def _intermediate_layers(self, x, n):
        return self.vit._get_intermediate_layers_not_chunked(x, n)

    def forward_features(self, x):
        x = self.vit.prepare_tokens_with_masks(x)
        for blk in self.vit.blocks:
            x = blk(x)
        return self.vit.norm(x)


def build(variant, seed):
    torch.manual_seed(seed)
    vit = vt.DinoVisionTransformer(img_size=NATIVE, patch_size=PATCH, embed_dim=DIM, depth=DEPTH, num_heads=HEADS, mlp_ratio=4,
                                   init_values=1.0, block_chunks=0,
                                   block_fn=vt.partial(vt.Block, attn_class=vt.MemEffAttention)).eval()
    with torch.no_grad():
        for n_, q in vit.named_parameters():
            if "norm" in n_ or "gamma" in n_ or "bias" in n_:
                q.add_(0.1 * torch.randn_like(q))
        vit.cls_token.copy_(0.02 * torch.randn_like(vit.cls_token))
    for q in vit.parameters():
        q.requires_grad = False
    base_sd = {k: v.detach().clone() for k, v in vit.state_dict().items() if k != "mask_token"}
    fake = types.SimpleNamespace()
    fake.w_As, fake.w_Bs, fake.adapters = [], [], nn.ModuleList()
    # src/finetune_timm_vggt.py:134-162 (same loop in finetune_timm_mast3r.py:118-141)
    for blk_idx in range(4, DEPTH):
        blk = vit.blocks[blk_idx]
        lin = blk.attn.qkv
        a_q, b_q = nn.Linear(DIM, 4, bias=False), nn.Linear(4, DIM, bias=False)
        a_v, b_v = nn.Linear(DIM, 4, bias=False), nn.Linear(4, DIM, bias=False)
        nn.init.normal_(b_q.weight, std=0.05)          # small non-zero B (zero-init hides bugs)
        nn.init.normal_(b_v.weight, std=0.05)
        fake.w_As += [a_q, a_v]
        fake.w_Bs += [b_q, b_v]
        blk.attn.qkv = M._LoRA_qkv(lin, a_q, b_q, a_v, b_v)
        adp = M.Adapter(dim=DIM, bottleneck_dim=16)
        vit.blocks[blk_idx] = M.BlockWithAdapter(blk, adp)
        fake.adapters.append(adp)
    fake.model = TimmShim(vit)
    mean, std = torch.tensor(MEAN).view(1, 3, 1, 1), torch.tensor(STD).view(1, 3, 1, 1)
    fake.input_transform = lambda x: (x - mean) / std
    fake.refine_conv = nn.Conv2d(DIM, DIM, kernel_size=3, stride=1, padding=1)
    fake.depth_diff_head = M.DepthAwareFeatureFusion(input_dim=DIM, use_tanh=True)
    fake.patch_size, fake.resize_patch_size = PATCH, PATCH
    fake.target_res, fake.downsample_factor = TARGET_RES, DOWN
    fake.thres3d_neg = 0.1
    fake.device = torch.device("cpu")
    fake.log = lambda *a, **k: None
    if variant == "vggt":
        C = VG.FinetuneVGGTTIMM
        fake.ap_loss_weight = fake.depth_loss_weight = fake.intra_depth_loss_weight = fake.kl_loss_weight = 1.0
    else:
        C = MA.FinetuneMASt3RTIMM
        fake.ap_loss_weight, fake.depth_loss_weight, fake.intra_depth_loss_weight, fake.kl_loss_weight = 1.0, 0.0, 1.0, 1.0
    for name in ("calculate_depth_loss", "calculate_cost_loss", "calculate_matching_loss", "get_intermediate_feature",
                 "get_feature", "get_feature_cost", "configure_optimizers"):
        setattr(fake, name, types.MethodType(getattr(C, name), fake))
    fake.training_step = types.MethodType(C.training_step, fake)
    return fake, base_sd


def targets(seed):
    """Synthetic teacher targets of SURVEY 8d for one pair (ragged keypoint count)."""
    gg = g(seed)
    n = N - (seed % 5)
    rgb_1, rgb_2 = torch.rand(1, 3, H, W, generator=gg), torch.rand(1, 3, H, W, generator=gg)
    kp1 = torch.stack([torch.randint(3, W - 3, (n,), generator=gg), torch.randint(3, H - 3, (n,), generator=gg)], -1).float()
    kp2 = (kp1 + torch.round(torch.randn(n, 2, generator=gg) * 4)).clamp(min=3)
    kp2[:, 0].clamp_(max=W - 4)
    kp2[:, 1].clamp_(max=H - 4)
    hw = (H // PATCH) * (W // PATCH)
    pm1 = torch.rand(H, W, 3, generator=gg) * 2
    pm2 = pm1 + 0.05 * torch.randn(H, W, 3, generator=gg)
    return dict(rgb_1=rgb_1, rgb_2=rgb_2, kp_1=kp1[None], kp_2=kp2[None], pm_1=pm1, pm_2=pm2,
                depth_1=0.5 + 5 * torch.rand(H, W, generator=gg), depth_2=0.5 + 5 * torch.rand(H, W, generator=gg),
                cost_1=torch.softmax(3 * torch.randn(1, hw, hw, generator=gg), -1),
                cost_2=torch.softmax(3 * torch.randn(1, hw, hw, generator=gg), -1),
                mask_1=torch.rand(H, W, generator=gg) < 0.7, mask_2=torch.rand(H, W, generator=gg) < 0.7)


def reference_pair_loss(fake, variant, t):
    """-> (loss, terms) of one pair by the reference's own step code."""
    rec = {}
    if variant == "vggt":
        feats = {"image_shape": (H, W), "cost_1": t["cost_1"], "cost_2": t["cost_2"], "point_map_view_1": t["pm_1"],
                 "point_map_view_2": t["pm_2"], "depth_pred_1": t["depth_1"], "depth_pred_2": t["depth_2"]}
        fake.extract_vggt_features = lambda rgb, batch_idx=None: feats
        fake.sample_keypoints = lambda f, num_keypoints=300, min_distance=5: (t["kp_1"], t["kp_2"], None, t["mask_1"], t["mask_2"])
        fake.log = lambda k, v, **kw: rec.__setitem__(k, float(v))
        if hasattr(fake, "batch_metrics"):
            del fake.batch_metrics
        loss = fake.training_step({"rgb_1": t["rgb_1"], "rgb_2": t["rgb_2"], "rgb_vggt": None}, 0)
        return loss, {"ap": rec["ap_loss"], "depth": rec["depth_loss"], "intra": rec["intra_depth_loss"], "kl": rec["kl_loss"]}
    # FinetuneMASt3RTIMM.training_step (src/finetune_timm_mast3r.py:592-681) with the two teacher calls faked
    feats = {"pts3d_1": t["pm_1"], "pts3d_2_from_1": t["pm_2"], "pts3d_2": t["pm_2"], "cost_1": t["cost_1"][0], "cost_2": t["cost_2"][0]}
    fake.extract_mast3r_features = lambda a, b: feats
    fake.filter_and_match_keypoints = lambda f, r1, r2: (t["kp_1"], t["kp_2"], r1, r2, W, H)
    fake.log = lambda k, v, **kw: rec.__setitem__(k, float(v))
    if hasattr(fake, "batch_metrics"):
        del fake.batch_metrics
    loss = fake.training_step({"rgb_1": t["rgb_1"], "rgb_2": t["rgb_2"], "rgb_mast3r_1": None, "rgb_mast3r_2": None,
                               "intrinsic": torch.eye(3)[None], "depth_1": t["depth_1"][None], "depth_2": t["depth_2"][None]}, 0)
    return loss, {"ap": rec["ap_loss"], "depth": rec["depth_loss"], "intra": rec["intra_depth_loss"], "kl": rec["kl_loss"]}


def run(variant, seed):
    fake, base_sd = build(variant, seed)
    ts = [targets(seed * 10 + q) for q in range(P)]
    opt = fake.configure_optimizers()
    params = [q for grp in opt.param_groups for q in grp["params"]]
    before = [q.detach().clone() for q in params]
    opt.zero_grad()
    terms_all, total = [], 0.0
    for t in ts:
        loss, terms = reference_pair_loss(fake, variant, t)
        (loss / P).backward()                          # DDP: mean over ranks (src/main.py:147-151)
        terms_all.append(terms)
        total += float(loss) / P
    grads = [q.grad.detach().clone() if q.grad is not None else torch.zeros_like(q) for q in params]      # before the clip scales them
    norm = torch.nn.utils.clip_grad_norm_([q for q in params if q.grad is not None], 1.0)   # Lightning gradient_clip_val=1.0
    opt.step()
    after = [q.detach().clone() for q in params]

    # ---- the oracle must reproduce it ------------------------------------------------------------------------------
    cfg = dict(patch=PATCH, dim=DIM, depth=DEPTH, heads=HEADS, ln_eps=1e-6, pos_interp="dinov2", pre_norm=False, mean=MEAN,
               std=STD, variant=variant, teacher_patch=PATCH, geometry="reference", target_res=TARGET_RES, downsample_factor=DOWN)
    nb = DEPTH - 4
    it = iter(before)
    A = [next(it) for _ in range(2 * nb)]
    B = [next(it) for _ in range(2 * nb)]
    refine = {"weight": next(it), "bias": next(it)}
    head_list = [next(it) for _ in range(len(list(fake.depth_diff_head.parameters())))]
    ad = [next(it) for _ in range(2 * nb)]
    leaves = []

    def leaf(x):
        x = x.clone().requires_grad_(True)
        leaves.append(x)
        return x
    tr = {"lora": {}, "adapter": {}}
    for j in range(nb):
        tr["lora"][4 + j] = {"a_q": leaf(A[2 * j]), "a_v": leaf(A[2 * j + 1])}
    for j in range(nb):
        tr["lora"][4 + j].update({"b_q": leaf(B[2 * j]), "b_v": leaf(B[2 * j + 1])})
    refine = {"weight": leaf(refine["weight"]), "bias": leaf(refine["bias"])}
    hl = [leaf(x) for x in head_list]       # depth_attention (4 tensors, unused) then fusion_layer: Linear, LayerNorm, Linear
    hp = {"w1": hl[4], "b1": hl[5], "ln_w": hl[6], "ln_b": hl[7], "w2": hl[8], "b2": hl[9]}
    for j in range(nb):
        names = [n_ for n_, _ in fake.adapters[j].named_parameters()]
        assert names == ["down.weight", "up.weight"], names
        tr["adapter"][4 + j] = {"down": leaf(ad[2 * j]), "up": leaf(ad[2 * j + 1])}
    weights = {"ap": fake.ap_loss_weight, "depth": fake.depth_loss_weight, "intra": fake.intra_depth_loss_weight, "kl": fake.kl_loss_weight}
    o_total = 0
    for t, ref_terms in zip(ts, terms_all):
        tp = PATCH
        one = {"rgb_1": t["rgb_1"], "rgb_2": t["rgb_2"], "kp_1": t["kp_1"], "kp_2": t["kp_2"], "depth_1": t["depth_1"],
               "depth_2": t["depth_2"], "cost_1": t["cost_1"], "cost_2": t["cost_2"],
               "pts3d_1": t["pm_1"][t["kp_1"][0, :, 1].long(), t["kp_1"][0, :, 0].long()][None],
               "pts3d_2": t["pm_2"][t["kp_2"][0, :, 1].long(), t["kp_2"][0, :, 0].long()][None],
               "mask_patch_1": torch.nn.functional.interpolate(t["mask_1"][None, None].float(), size=(H // tp, W // tp), mode="nearest").bool().view(-1),
               "mask_patch_2": torch.nn.functional.interpolate(t["mask_2"][None, None].float(), size=(H // tp, W // tp), mode="nearest").bool().view(-1)}
        terms = O.pair_losses(one, base_sd, cfg, tr, refine, hp)
        for k in ("ap", "depth", "intra", "kl"):
            close(terms[k], ref_terms[k], 2e-5, f"{variant} term {k}")
        o_total = o_total + O.total_loss(terms, weights) / P
    close(o_total, total, 2e-5, f"{variant} loss")
    o_total.backward()
    og = [x.grad if x.grad is not None else torch.zeros_like(x) for x in leaves]
    for i, (a, b) in enumerate(zip(og, grads)):
        if b.abs().max() > 0:
            close(a, b, 5e-4, f"{variant} grad {i}")
    op = [x.detach().clone() for x in leaves]
    st = [(torch.zeros_like(x), torch.zeros_like(x)) for x in op]
    # torch.optim.AdamW skips parameters without .grad (depth_attention): no decay, no moments
    live = [i for i, q in enumerate(params) if q.grad is not None]
    onorm = O.clip_and_adamw([op[i] for i in live], [og[i] for i in live], [st[i] for i in live], 1)
    close(onorm, norm, 2e-5, f"{variant} clip norm")
    for i, (a, b, b0) in enumerate(zip(op, after, before)):
        # the UPDATE (lr-sized), not the weight.  Frobenius: where |g| ~ eps = 1e-8 the AdamW quotient g / (|g| + eps) turns
        # fp32 rounding noise of the gradient into percent-level differences of single elements
        da, db = (a - b0).double(), (b - b0).double()
        err = ((da - db).norm() / db.norm().clamp_min(1e-30)).item()
        assert err < 2e-3, f"{variant} update {i}: {err:.3e}"
    print(f"[g18 {variant}] loss {total:.6f} terms {terms_all} clip norm {float(norm):.5f}: oracle agrees")

    arrs = {"sd." + k: v for k, v in base_sd.items()}
    for i, (b0, a1, g1) in enumerate(zip(before, after, grads)):
        arrs[f"before_{i:03d}"], arrs[f"after_{i:03d}"], arrs[f"grad_{i:03d}"] = b0, a1, g1
    for q, t in enumerate(ts):
        for k, v in t.items():
            arrs[f"pair{q}.{k}"] = v
        for k, v in terms_all[q].items():
            arrs[f"pair{q}.term_{k}"] = torch.tensor(v)
    arrs["loss"], arrs["clip_norm"] = torch.tensor(total), norm.detach()
    arrs["n_params"], arrs["n_live"] = torch.tensor(len(params)), torch.tensor(live)
    arrs["cfg_target_res"], arrs["cfg_downsample_factor"] = torch.tensor(TARGET_RES), torch.tensor(DOWN)
    arrs["cfg_bottleneck"] = torch.tensor(16)
    np.savez_compressed(os.path.join(OUT, f"g18_full_step_{variant}.npz"),
                        **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})


if __name__ == "__main__":
    run("vggt", 18)
    run("mast3r", 19)
