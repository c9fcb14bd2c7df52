import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gd_amd  # noqa
from gd_amd import ops
L = gd_amd._lib.lib()
L.gd_debug_set(b"attn_mfma32", 0)
B, N, H = 2, 333, 2
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B, N, 3, H, 64, generator=g, device="cuda")
x[:, :, 0] *= 6.0
qkv = x.reshape(B * N, 3 * H * 64).half()
o, lse = ops.attention_fwd(qkv, B, N, H)
bad = ~torch.isfinite(lse)
xh = qkv.reshape(B, N, 3, H, 64)
c2 = 0.125 * 1.4426950408889634
b_, h_ = 0, 1
idx = bad[b_, h_].nonzero().reshape(-1).tolist()
good = (~bad[b_, h_]).nonzero().reshape(-1).tolist()
k = xh[b_, :, 1, h_].float()
for n_ in idx[:4] + good[:2]:
    q = xh[b_, n_, 0, h_]
    qs = (q.float() * c2).half().float()
    sc = k @ qs          # [N] fp32
    print("n", n_, "bad" if n_ in idx else "good", "q finite", bool(torch.isfinite(q.float()).all()), "qs absmax", float(qs.abs().max()), "score min/max", float(sc.min()), float(sc.max()),
          "q raw bits any inf/nan pattern", bool(((q.view(torch.int16) & 0x7c00) == 0x7c00).any()))
    base = n_ - n_ % 16
    print("    bad flags of the 16-query tile", [int(bad[b_, h_, i]) if i < N else -1 for i in range(base, base + 16)])
print("bad per wave of 32 queries (b0,h1):", [int(bad[b_, h_, i:i + 32].sum()) for i in range(0, N, 32)])
n_ = idx[0]
x2 = xh.clone()
x2[b_, :, 0, h_] = xh[b_, n_, 0, h_]
o2, lse2 = ops.attention_fwd(x2.reshape(B * N, -1).contiguous(), B, N, H)
print("all queries = bad row", n_, ": non-finite lse count", int((~torch.isfinite(lse2[b_, h_])).sum()), "of", N, " lse[0..8]", lse2[b_, h_, :8].tolist())
