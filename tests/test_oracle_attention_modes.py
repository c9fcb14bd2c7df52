"""CPU: the oracle's three ways of evaluating a block's attention — explicit softmax(q k^T / sqrt d) v (what the golden fixtures pin), checkpointed
query-row blocks (`attention_chunk`: fp64 at thousands of tokens) and torch's fused CPU kernel (`attention_sdpa`: the fp32 oracle of the 4 801 /
6 401-token GPU cases) — are one function: same block output, same gradients with respect to the input and the LoRA / adapter tensors."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _block(mode, dtype):
    import gd_oracle as O
    g = torch.Generator().manual_seed(3)
    D, H, N, B, r = 128, 2, 197, 2, 4
    rn = lambda *s: torch.randn(*s, generator=g, dtype=dtype)
    p = {"blocks.0.norm1.weight": 1 + 0.1 * rn(D), "blocks.0.norm1.bias": 0.1 * rn(D), "blocks.0.norm2.weight": 1 + 0.1 * rn(D),
         "blocks.0.norm2.bias": 0.1 * rn(D), "blocks.0.attn.qkv.weight": 0.2 * rn(3 * D, D), "blocks.0.attn.qkv.bias": 0.1 * rn(3 * D),
         "blocks.0.attn.proj.weight": 0.1 * rn(D, D), "blocks.0.attn.proj.bias": 0.1 * rn(D), "blocks.0.mlp.fc1.weight": 0.1 * rn(4 * D, D),
         "blocks.0.mlp.fc1.bias": 0.1 * rn(4 * D), "blocks.0.mlp.fc2.weight": 0.1 * rn(D, 4 * D), "blocks.0.mlp.fc2.bias": 0.1 * rn(D),
         "blocks.0.ls1.gamma": 1 + 0.1 * rn(D), "blocks.0.ls2.gamma": 1 + 0.1 * rn(D)}
    lora = {"a_q": (0.1 * rn(r, D)).requires_grad_(True), "b_q": (0.1 * rn(D, r)).requires_grad_(True),
            "a_v": (0.1 * rn(r, D)).requires_grad_(True), "b_v": (0.1 * rn(D, r)).requires_grad_(True)}
    ad = {"down": (0.1 * rn(16, D)).requires_grad_(True), "up": (0.1 * rn(D, 16)).requires_grad_(True)}
    x = (2.0 * rn(B, N, D)).requires_grad_(True)
    cfg = {"heads": H, "ln_eps": 1e-6}
    if mode == "chunk":
        cfg["attention_chunk"] = 50          # ragged last block (197 = 3 x 50 + 47)
    if mode == "sdpa":
        cfg["attention_sdpa"] = True
    out = O.vit_block(x, p, 0, cfg, lora=lora, ad=ad)
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(9), dtype=dtype)
    (out * w).sum().backward()
    return [out.detach(), x.grad] + [t.grad for t in lora.values()] + [t.grad for t in ad.values()]


def test_oracle_attention_modes_agree():
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        ref = _block("plain", dtype)
        for mode in ("chunk", "sdpa"):
            got = _block(mode, dtype)
            for a, b in zip(got, ref):
                assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), (mode, dtype)
