import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gd_amd  # noqa
from gd_amd import ops
L = gd_amd._lib.lib()
for val in (1e-3, 1e-4, 3e-5, 1e-5, 1e-6, 1e-7):
    a = torch.full((1024, 64), val, device="cuda").half()
    w = torch.ones(256, 64, device="cuda").half()
    out = ops.gemm_nt(a, w, out_dtype=torch.float32)
    print(f"gemm fp16 a={val:g} (fp16 {float(a[0,0]):.3e}) x ones, K=64: got {float(out[0,0]):.6e} want {64*float(a[0,0]):.6e}")
a = torch.full((1024, 64), 1e-6, device="cuda").half(); a[:, 0] = 1.0
out = ops.gemm_nt(a, torch.ones(256, 64, device="cuda").half(), out_dtype=torch.float32)
print("gemm mixed:", float(out[0, 0]), "want", 1.0 + 63 * float(a[0, 1]))
B, N, H = 1, 64, 1
for gap in (5.0, 10.0, 14.5, 16.0, 20.0, 23.0, 26.0, 40.0):
    x = torch.zeros(B, N, 3, H, 64, device="cuda")
    x[0, :, 0, 0, 0] = 8.0
    x[0, :, 1, 0, 0] = -gap / 1.4426950408889634
    x[0, 0, 1, 0, 0] = 0.0
    x[0, :, 2, 0, :] = 1.0
    qkv = x.reshape(B * N, -1).half()
    for m32 in (0, 1):
        L.gd_debug_set(b"attn_mfma32", m32)
        o, lse = ops.attention_fwd(qkv, B, N, H)
        want = math.log(1 + 63 * 2.0 ** (-gap))
        print(f"gap {gap:5.1f} log2 m32={m32}: lse[5] = {float(lse[0,0,5]):.6e} (exact {want:.6e})  o = {float(o[5,0]):.4f}")
