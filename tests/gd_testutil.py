"""Shared helpers for the GPU step tests: export an engine's weights into the oracle's dict layout and
build seeded synthetic teacher targets."""
import torch


def oracle_params(engine):
    """FinetuneGD -> (base param dict, trainable dict, refine dict, head dict, cfg) for oracle/gd_oracle.py (CPU fp32)."""
    m = engine.model
    p = {"cls_token": m.cls_token, "pos_embed": m.pos_embed, "patch_embed.proj.weight": m.patch_embed.proj.weight,
         "norm.weight": m.norm.weight, "norm.bias": m.norm.bias}
    if m.patch_embed.proj.bias is not None:
        p["patch_embed.proj.bias"] = m.patch_embed.proj.bias
    if hasattr(m.norm_pre, "weight"):
        p["norm_pre.weight"], p["norm_pre.bias"] = m.norm_pre.weight, m.norm_pre.bias
    tr = {"lora": {}, "adapter": {}}
    for i, blk in enumerate(m.blocks):
        inner = blk.block if hasattr(blk, "adapter") else blk
        if hasattr(blk, "adapter"):
            tr["adapter"][i] = {"down": blk.adapter.down.weight, "up": blk.adapter.up.weight}
        q = inner.attn.qkv
        if hasattr(q, "linear_a_q"):
            tr["lora"][i] = {"a_q": q.linear_a_q.weight, "b_q": q.linear_b_q.weight, "a_v": q.linear_a_v.weight,
                             "b_v": q.linear_b_v.weight}
            q = q.qkv
        pre = f"blocks.{i}."
        p[pre + "norm1.weight"], p[pre + "norm1.bias"] = inner.norm1.weight, inner.norm1.bias
        p[pre + "norm2.weight"], p[pre + "norm2.bias"] = inner.norm2.weight, inner.norm2.bias
        p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"] = q.weight, q.bias
        p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"] = inner.attn.proj.weight, inner.attn.proj.bias
        p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"] = inner.mlp.fc1.weight, inner.mlp.fc1.bias
        p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"] = inner.mlp.fc2.weight, inner.mlp.fc2.bias
        if hasattr(inner.ls1, "gamma"):
            p[pre + "ls1.gamma"], p[pre + "ls2.gamma"] = inner.ls1.gamma, inner.ls2.gamma

    def cpu(t):
        return t.detach().float().cpu().clone()

    p = {k: cpu(v) for k, v in p.items()}
    tr = {a: {i: {k: cpu(v) for k, v in d.items()} for i, d in dd.items()} for a, dd in tr.items()}
    refine = {"weight": cpu(engine.refine_conv.weight), "bias": cpu(engine.refine_conv.bias)}
    head = {k: cpu(v) for k, v in engine.depth_diff_head.head_params().items()} if engine.depth_diff_head is not None else {}
    inner0 = m.blocks[0].block if hasattr(m.blocks[0], "adapter") else m.blocks[0]
    cfg = dict(patch=engine.patch_size, dim=m.embed_dim, depth=len(m.blocks), heads=inner0.attn.num_heads,
               ln_eps=inner0.norm1.eps, pos_interp=m.pos_interp, pre_norm=hasattr(m.norm_pre, "weight"),
               mean=m.mean, std=m.std, variant=engine.variant, teacher_patch=engine.resize_patch_size,
               geometry=engine.geometry, target_res=engine.target_res, downsample_factor=engine.downsample_factor)
    return p, tr, refine, head, cfg


def trainable_list(tr, refine, head, order_blocks):
    """Oracle tensors in the engine's configure_optimizers order: A's (q,v per block), B's, refine, head, adapters."""
    out = []
    for i in order_blocks:
        out += [tr["lora"][i]["a_q"], tr["lora"][i]["a_v"]]
    for i in order_blocks:
        out += [tr["lora"][i]["b_q"], tr["lora"][i]["b_v"]]
    out += [refine["weight"], refine["bias"]]
    return out, head, [tr["adapter"][i] for i in order_blocks]


def synthetic_batch(P, h, w, N, hw, device, seed=0, teacher_patch=14, counts=None):
    """Seeded synthetic pair batch + teacher targets."""
    g = torch.Generator().manual_seed(seed)
    b = {"rgb_1": torch.rand(P, 3, h, w, generator=g), "rgb_2": torch.rand(P, 3, h, w, generator=g)}
    kp1 = torch.stack([torch.randint(3, w - 3, (P, N), generator=g), torch.randint(3, h - 3, (P, N), generator=g)], -1).float()
    kp2 = (kp1 + torch.round(torch.randn(P, N, 2, generator=g) * 4)).clamp(min=3)
    kp2[..., 0].clamp_(max=w - 4)
    kp2[..., 1].clamp_(max=h - 4)
    b["kp_1"], b["kp_2"] = kp1, kp2
    b["pts3d_1"] = torch.rand(P, N, 3, generator=g) * 2
    b["pts3d_2"] = b["pts3d_1"] + 0.01 * torch.randn(P, N, 3, generator=g)
    b["depth_1"] = 0.5 + 5 * torch.rand(P, h, w, generator=g)
    b["depth_2"] = 0.5 + 5 * torch.rand(P, h, w, generator=g)
    b["cost_1"] = torch.softmax(3 * torch.randn(P, hw, hw, generator=g), -1)
    b["cost_2"] = torch.softmax(3 * torch.randn(P, hw, hw, generator=g), -1)
    b["mask_1"] = torch.rand(P, h, w, generator=g) < 0.7
    b["mask_2"] = torch.rand(P, h, w, generator=g) < 0.7
    if counts is not None:
        b["counts"] = torch.tensor(counts, dtype=torch.int32)
        for p, n in enumerate(counts):
            b["kp_1"][p, n:] = -1.0
            b["kp_2"][p, n:] = -1.0
    b = {k: v.to(device) for k, v in b.items()}
    rgb = torch.cat([b["rgb_1"], b["rgb_2"]], 0)          # collated as the two halves of one buffer, as a loader would hand them:
    b["rgb_1"], b["rgb_2"] = rgb[:P], rgb[P:]             # the step then needs no concatenation copy (finetune._pair_batch)
    return b


# ------------------------------------------------------------------------------------------------
# G18: one full optimisation step written by the reference (tools/make_golden_g18.py)
# ------------------------------------------------------------------------------------------------
G18_DEPTH, G18_DIM, G18_PATCH = 8, 64, 14


def g18_unpack(g):
    """fixture dict -> (base state dict, tensors before / after / grads in configure_optimizers order, per-pair targets)."""
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    n = int(g["n_params"])
    before = [g[f"before_{i:03d}"] for i in range(n)]
    after = [g[f"after_{i:03d}"] for i in range(n)]
    grads = [g[f"grad_{i:03d}"] for i in range(n)]
    pairs = []
    q = 0
    while f"pair{q}.rgb_1" in g:
        pairs.append({k.split(".", 1)[1]: v for k, v in g.items() if k.startswith(f"pair{q}.")})
        q += 1
    return sd, before, after, grads, pairs


def g18_oracle_inputs(before, variant, g):
    """Tensors in optimiser order -> the oracle's (trainable dict, refine, head params, leaves, cfg)."""
    nb = G18_DEPTH - 4
    leaves = [t.clone().requires_grad_(True) for t in before]
    it = iter(leaves)
    A = [next(it) for _ in range(2 * nb)]
    B = [next(it) for _ in range(2 * nb)]
    refine = {"weight": next(it), "bias": next(it)}
    hl = [next(it) for _ in range(10)]       # depth_attention (4 tensors, never used) then Linear, LayerNorm, Linear
    hp = {"w1": hl[4], "b1": hl[5], "ln_w": hl[6], "ln_b": hl[7], "w2": hl[8], "b2": hl[9]}
    ad = [next(it) for _ in range(2 * nb)]
    tr = {"lora": {}, "adapter": {}}
    for j in range(nb):
        tr["lora"][4 + j] = {"a_q": A[2 * j], "a_v": A[2 * j + 1], "b_q": B[2 * j], "b_v": B[2 * j + 1]}
        tr["adapter"][4 + j] = {"down": ad[2 * j], "up": ad[2 * j + 1]}
    cfg = dict(patch=G18_PATCH, dim=G18_DIM, depth=G18_DEPTH, heads=1, ln_eps=1e-6, pos_interp="dinov2", pre_norm=False,
               mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), variant=variant, teacher_patch=G18_PATCH,
               geometry="reference", target_res=int(g["cfg_target_res"]), downsample_factor=int(g["cfg_downsample_factor"]))
    return tr, refine, hp, leaves, cfg


def g18_pair_batch(t, patch=G18_PATCH):
    """One pair of the fixture in the oracle's pair_losses layout."""
    import torch.nn.functional as F
    H, W = t["rgb_1"].shape[-2:]
    gather = lambda pm, kp: pm[kp[0, :, 1].long(), kp[0, :, 0].long()][None]
    down = lambda m: F.interpolate(m[None, None].float(), size=(H // patch, W // patch), mode="nearest").bool().view(-1)
    return {"rgb_1": t["rgb_1"], "rgb_2": t["rgb_2"], "kp_1": t["kp_1"], "kp_2": t["kp_2"], "depth_1": t["depth_1"],
            "depth_2": t["depth_2"], "cost_1": t["cost_1"], "cost_2": t["cost_2"], "pts3d_1": gather(t["pm_1"], t["kp_1"]),
            "pts3d_2": gather(t["pm_2"], t["kp_2"]), "mask_patch_1": down(t["mask_1"]), "mask_patch_2": down(t["mask_2"])}
