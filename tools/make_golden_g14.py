"""G14: the VGGT teacher's cross-view attention maps (vggt/layers/attention.py:51-85, `return_attn=True`) from the
REFERENCE's own Attention class, pinned against oracle.cross_view_attention_maps and written to tests/golden/.
Build container only (needs /root/reference); same conventions as tools/make_golden.py.  Usage: python tools/make_golden_g14.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import as R  # noqa: E402

R.install()   # stub modules + /root/reference on sys.path
import gd_oracle as O  # noqa: E402

from vggt.layers.attention import Attention  # noqa: E402

torch.manual_seed(0)
B, H, d, n, prefix = 2, 3, 64, 37, 5
N = 2 * (n + prefix)
attn = Attention(dim=H * d, num_heads=H)
g = torch.Generator().manual_seed(140)
q = torch.randn(B, H, N, d, generator=g) * 0.7
k = torch.randn(B, H, N, d, generator=g) * 0.7
v = torch.randn(B, H, N, d, generator=g)
temperature = 0.7
with torch.no_grad():
    _, maps = attn.custom_scaled_dot_product_attention(q, k, v, return_attn=True, temperature=temperature)   # [2B, H, n, n]
ref = maps.mean(dim=1)
got = O.cross_view_attention_maps(q, k, attn.scale, temperature, prefix)
err = (got - ref).abs().max().item()
assert err < 1e-6, err
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g14_cross_view_attn.npz"), q=q.numpy(), k=k.numpy(), scale=attn.scale,
                    temperature=temperature, prefix=prefix, maps=ref.numpy())
print(f"wrote g14_cross_view_attn.npz: maps {tuple(ref.shape)}, oracle max abs err {err:.2e}")
