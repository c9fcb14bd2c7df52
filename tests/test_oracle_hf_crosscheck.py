"""CPU: an INDEPENDENT cross-check of the oracle's CLIP-style (pre-norm) ViT restatement — SURVEY 8c(ii).

The reference's student is timm's `vit_base_patch16_clip_384.laion2b_ft_in12k_in1k` (src/finetune_timm_mast3r.py:68-70):
timm 0.9.10 is not vendored, so the oracle's `vit_forward(pre_norm=True, ln_eps=1e-5, no LayerScale, no patch-conv bias)` is a
restatement from memory.  HuggingFace `transformers.CLIPVisionModel` implements the same architecture from another code
base: with the same random weights (hidden_act="gelu": the laion2b OpenCLIP towers use the exact GELU) every block output,
and the final-normed sequence, must agree.  The engine's pre-norm mode is pinned on the oracle in tests/test_gpu_step.py."""
import pytest
import torch

import gd_oracle as O

transformers = pytest.importorskip("transformers")


def test_oracle_prenorm_vit_equals_hf_clip_vision_tower():
    from transformers import CLIPVisionConfig, CLIPVisionModel
    torch.manual_seed(0)
    D, L, heads, P, img = 128, 4, 2, 16, 64
    cfg_hf = CLIPVisionConfig(hidden_size=D, intermediate_size=4 * D, num_hidden_layers=L, num_attention_heads=heads, image_size=img,
                              patch_size=P, hidden_act="gelu", layer_norm_eps=1e-5, attention_dropout=0.0)
    m = CLIPVisionModel(cfg_hf).eval().double()
    with torch.no_grad():                         # de-trivialise LayerNorm affines and biases
        for n_, q in m.named_parameters():
            if "layer_norm" in n_ or "layrnorm" in n_ or n_.endswith("bias"):
                q.add_(0.1 * torch.randn_like(q))
    vm = getattr(m, "vision_model", m)            # transformers 5.x: the tower's modules sit on CLIPVisionModel itself
    sd = {k: v.detach() for k, v in vm.state_dict().items()}
    p = {"patch_embed.proj.weight": sd["embeddings.patch_embedding.weight"], "cls_token": sd["embeddings.class_embedding"].reshape(1, 1, D),
         "pos_embed": sd["embeddings.position_embedding.weight"][None], "norm_pre.weight": sd["pre_layrnorm.weight"],
         "norm_pre.bias": sd["pre_layrnorm.bias"], "norm.weight": sd["post_layernorm.weight"], "norm.bias": sd["post_layernorm.bias"]}
    for i in range(L):
        h, o = f"encoder.layers.{i}.", f"blocks.{i}."
        p[o + "norm1.weight"], p[o + "norm1.bias"] = sd[h + "layer_norm1.weight"], sd[h + "layer_norm1.bias"]
        p[o + "norm2.weight"], p[o + "norm2.bias"] = sd[h + "layer_norm2.weight"], sd[h + "layer_norm2.bias"]
        p[o + "attn.qkv.weight"] = torch.cat([sd[h + f"self_attn.{t}_proj.weight"] for t in "qkv"], 0)
        p[o + "attn.qkv.bias"] = torch.cat([sd[h + f"self_attn.{t}_proj.bias"] for t in "qkv"], 0)
        p[o + "attn.proj.weight"], p[o + "attn.proj.bias"] = sd[h + "self_attn.out_proj.weight"], sd[h + "self_attn.out_proj.bias"]
        p[o + "mlp.fc1.weight"], p[o + "mlp.fc1.bias"] = sd[h + "mlp.fc1.weight"], sd[h + "mlp.fc1.bias"]
        p[o + "mlp.fc2.weight"], p[o + "mlp.fc2.bias"] = sd[h + "mlp.fc2.weight"], sd[h + "mlp.fc2.bias"]
    cfg = dict(patch=P, dim=D, depth=L, heads=heads, ln_eps=1e-5, pos_interp="timm", pre_norm=True)
    x = torch.randn(2, 3, img, img, dtype=torch.float64)          # already-normalised pixels
    with torch.no_grad():
        out = m(pixel_values=x, output_hidden_states=True)
        taps, last = O.vit_forward(x, p, cfg, None, taps=tuple(range(L)))
    assert len(out.hidden_states) == L + 1
    for i in range(L):
        assert float((taps[i] - out.hidden_states[i + 1]).abs().max()) < 1e-9, i
    assert float((last - out.last_hidden_state).abs().max()) < 1e-9
    assert float((O.final_norm(last, p, cfg)[:, 0] - out.pooler_output).abs().max()) < 1e-9      # post_layernorm on the class token
