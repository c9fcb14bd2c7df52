"""Shared helpers for the GPU step tests: export an engine's weights into the oracle's dict layout and
build seeded synthetic teacher targets."""
import gd_amd  # noqa: F401
import torch

from gd_amd.synthetic import export_params as oracle_params, synthetic_batch  # noqa: E402,F401  (live in the package: bench.py uses them too)


def trainable_list(tr, refine, head, order_blocks):
    """Oracle tensors in the engine's configure_optimizers order: A's (q,v per block), B's, refine, head, adapters."""
    out = []
    for i in order_blocks:
        out += [tr["lora"][i]["a_q"], tr["lora"][i]["a_v"]]
    for i in order_blocks:
        out += [tr["lora"][i]["b_q"], tr["lora"][i]["b_v"]]
    out += [refine["weight"], refine["bias"]]
    return out, head, [tr["adapter"][i] for i in order_blocks]


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
