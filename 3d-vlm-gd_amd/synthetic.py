"""Synthetic workload and weight export for benchmarks and tests (no reference counterpart: the reference trains on datasets).

`synthetic_batch` builds a seeded batch of image pairs with teacher targets of the shapes the trainers consume;
`export_params` hands an engine's weights out as plain CPU fp32 dicts (what a checker — tests/, bench.py's parity leg — feeds
to its own reference implementation).  Nothing here touches oracle/."""
import torch


def export_params(engine):
    """FinetuneGD -> (base param dict, trainable dict, refine dict, head dict, cfg) (CPU fp32 copies; the layout oracle/gd_oracle.py reads)."""
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
