"""G17: MASt3R teacher -> distillation target `tgt_attn_map` (dust3r/dust3r/model.py:346-366): the reference's own
`AsymmetricCroCo3DStereo.forward` called unbound with a fake self whose encoder / decoder / heads are stubs returning the
fixture's per-layer cross-attention score maps, pinned against oracle.mast3r_tgt_attn_map.  Build container only.
Usage: python tools/make_golden_g17.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import as R  # noqa: E402
import gd_oracle as O  # noqa: E402

R.install()
from dust3r.model import AsymmetricCroCo3DStereo  # noqa: E402

B, H, N, L = 2, 3, 20, 3
g = torch.Generator().manual_seed(170)
tgt = [torch.randn(B, H, N, N, generator=g) * 2 for _ in range(L)]      # CrossAttention's `attn_map`: raw scaled scores
src = [torch.randn(B, H, N, N, generator=g) * 2 for _ in range(L)]


class _Self:
    reciprocity = True
    temperature = 3.0
    count = 0

    def _encode_symmetrized(self, v1, v2):
        f = torch.zeros(B, N, 8)
        return (None, None), (f, f), (None, None), (f, f)

    def _decoder(self, f1, p1, f2, p2):
        return [(f1,), (f2,)], [t.clone() for t in tgt], [s.clone() for s in src]

    def _downstream_head(self, num, toks, shape):
        return {"pts3d": torch.zeros(1)}


res1, res2 = AsymmetricCroCo3DStereo.forward(_Self(), {"img": None}, {"img": None})
ref = res2["tgt_attn_map"]
got = O.mast3r_tgt_attn_map(tgt, src, 3.0)
err = (got - ref).abs().max().item()
assert err < 1e-6, err
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g17_mast3r_tgt_attn_map.npz"), tgt=torch.stack(tgt).numpy(),
                    src=torch.stack(src).numpy(), temperature=3.0, out=ref.numpy())
print(f"wrote g17_mast3r_tgt_attn_map.npz: out {tuple(ref.shape)}, oracle max abs err {err:.2e}")
