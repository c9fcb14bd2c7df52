import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gd_amd  # noqa
from gd_amd import ops

B, N, H = 2, 333, 2
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B, N, 3, H, 64, generator=g, device="cuda")
x[:, :, 0] *= 6.0
qkv = x.reshape(B * N, 3 * H * 64).half()
xd = qkv.double().reshape(B, N, 3, H, 64)
q, k, v = xd.permute(2, 0, 3, 1, 4).unbind(0)
s = (q * 64 ** -0.5) @ k.transpose(-1, -2) * 1.4426950408889634     # [B,H,N,N] log2 units
for m32 in (0, 1):
    gd_amd._lib.lib().gd_debug_set(b"attn_mfma32", m32)
    o, lse = ops.attention_fwd(qkv, B, N, H)
    bad = ~torch.isfinite(lse)          # [B,H,N]
    print("m32", m32, "bad rows per (b,h):", bad.sum(-1).tolist())
    smax = s.max(-1).values
    t0 = s[..., :64].max(-1).values
    # per-tile maxima in processing order for block xb (rot = 2*xb % 5)
    for b_ in range(B):
        for h_ in range(H):
            idx = bad[b_, h_].nonzero().reshape(-1).tolist()
            if not idx:
                continue
            print(" (b,h)", b_, h_, "bad n:", idx[:40])
            for n_ in idx[:5]:
                tiles = [float(s[b_, h_, n_, 64 * i:64 * i + 64].max()) for i in range(6)]
                print("    n", n_, "tile maxima", [round(t, 1) for t in tiles], "abs max |q|", float(q[b_, h_, n_].abs().max()), "max |k|", float(k[b_, h_].abs().max()))
    good = ~bad
    print("   raw max: bad rows min/mean/max", float(smax[bad].min()) if bad.any() else None, float(smax[bad].mean()) if bad.any() else None,
          float(smax[bad].max()) if bad.any() else None, " good rows max", float(smax[good].max()))
    print("   rows with raw max > 16:", int((smax > 16).sum()), "of", smax.numel(), "; bad:", int(bad.sum()))
