"""G21 (eval feature API): the reference's stride-override recipe for dense tracking features
(src/evaluate_timm.py:262-272) run on the reference's in-tree DINOv2 ViT — build container only.

    model.patch_embed.proj.stride = (s, s)                      # s = patch / 2: overlapping patches
    model.interpolate_pos_encoding = types.MethodType(_fix_pos_enc(patch, (s, s)), model)
    model.forward_features(imagenet_norm(img))

Writes tests/golden/g21_stride_override.npz: the frozen weights, the image, the position table `_fix_pos_enc` produced
and the normed tokens; asserts that oracle/gd_oracle.py (`vit_tokens` with cfg['patch_stride'], `fix_pos_enc`) reproduces them."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import ref_import as R  # noqa: E402
import gd_oracle as O  # noqa: E402

L, Fn, M = R.ref_utils()
vt = R.ref_vit()
OUT = os.path.join(HERE, "..", "tests", "golden")

torch.manual_seed(21)
P, S = 14, 7
CFG = dict(patch=P, dim=64, depth=2, heads=1, ln_eps=1e-6, pos_interp="dinov2", patch_stride=(S, S))
model = vt.DinoVisionTransformer(img_size=56, patch_size=P, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4, init_values=1.0,
                                 block_chunks=0, block_fn=vt.partial(vt.Block, attn_class=vt.MemEffAttention)).eval()
with torch.no_grad():
    for n_, q in model.named_parameters():
        if "norm" in n_ or "gamma" in n_ or "bias" in n_:
            q.add_(0.1 * torch.randn_like(q))
    model.cls_token.copy_(0.02 * torch.randn_like(model.cls_token))
    model.pos_embed.copy_(0.1 * torch.randn_like(model.pos_embed))
sd = {k: v.detach().clone() for k, v in model.state_dict().items() if k != "mask_token"}

arrs = {"sd." + k: v.numpy() for k, v in sd.items()}
mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
for tag, (h, w) in {"sq": (56, 56), "rect": (56, 70)}.items():
    img = torch.rand(2, 3, h, w, generator=torch.Generator().manual_seed(210 + w))
    nimg = O.normalize_image(img, mean, std)
    # the recipe, verbatim in effect (src/evaluate_timm.py:262-272)
    stride_pair = torch.nn.modules.utils._pair(S)
    model.patch_embed.proj.stride = stride_pair
    model.interpolate_pos_encoding = types.MethodType(Fn._fix_pos_enc(P, stride_pair), model)
    with torch.no_grad():
        out = model.forward_features(nimg)
        xn = torch.cat([out["x_norm_clstoken"][:, None], out["x_norm_patchtokens"]], 1)
        gh, gw = 1 + (h - P) // S, 1 + (w - P) // S
        pos = model.interpolate_pos_encoding(torch.empty(1, gh * gw + 1, 64), h, w)
        # oracle
        x = O.vit_tokens(nimg, sd, CFG)
        for i in range(CFG["depth"]):
            x = O.vit_block(x, sd, i, CFG, None, None)
        xo = O.final_norm(x, sd, CFG)
        po = O.fix_pos_enc(sd["pos_embed"], P, (S, S), gh * gw, h, w)
    e1 = ((xo - xn).norm() / xn.norm()).item()
    e2 = ((po - pos).norm() / pos.norm()).item()
    print(tag, tuple(xn.shape), "oracle vs reference: tokens", e1, "pos", e2)
    assert xn.shape == (2, gh * gw + 1, 64) and e1 < 2e-5 and e2 < 1e-6
    arrs[f"{tag}.img"], arrs[f"{tag}.xnorm"], arrs[f"{tag}.pos"] = img.numpy(), xn.numpy(), pos.numpy()
np.savez_compressed(os.path.join(OUT, "g21_stride_override.npz"), **arrs)
print(os.path.getsize(os.path.join(OUT, "g21_stride_override.npz")) / 1e6, "MB")
