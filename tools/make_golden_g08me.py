"""G8 (ME variant): the matching loss of src/finetune_timm_me.py:191-220 from the REFERENCE's own training_step (called
unbound with a fake self whose get_feature returns the fixture descriptors), pinned against oracle.smooth_ap_loss_me.
Build container only.  Usage: python tools/make_golden_g08me.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import as R  # noqa: E402
import gd_oracle as O  # noqa: E402

_, _, ME = R.ref_modules()


class _Self:
    thresh3d_pos = 5e-3
    thres3d_neg = 0.1

    def __init__(self, descs):
        self.descs = list(descs)

    def get_feature(self, rgb, kp, normalize=True):
        return self.descs.pop(0)

    def log(self, *a, **k):
        pass


g = lambda s: torch.Generator().manual_seed(s)
N, C = 29, 16
centers = F.normalize(torch.randn(5, C, generator=g(1)), dim=-1)
assign = torch.randint(0, 5, (N,), generator=g(2))
d1 = F.normalize(centers[assign] + 0.03 * torch.randn(N, C, generator=g(3)), dim=-1)[None]
d2 = F.normalize(d1 + 0.03 * torch.randn(1, N, C, generator=g(4)), dim=-1)
pts1 = torch.rand(1, N, 3, generator=g(5)) * 2
pts2 = pts1 + 0.2 * torch.randn(1, N, 3, generator=g(6))           # far by default
pts2[0, :20] = pts1[0, :20] + 1e-3 * torch.randn(20, 3, generator=g(7))    # 20 one-to-one positives
pts2[0, 20] = pts1[0, 3] + 1e-3                                    # row 3 gets a second positive (column 20)
pts2[0, 21] = pts1[0, 3] - 1e-3                                    # ... and a third
d1r = d1.clone().requires_grad_(True)
d2r = d2.clone().requires_grad_(True)
loss = ME.FinetuneTIMM.training_step(_Self([d1r, d2r]), {"rgb_1": None, "pts2d_1": None, "pts3d_1": pts1, "rgb_2": None,
                                                          "pts2d_2": None, "pts3d_2": pts2}, 0)
loss.backward()
o1 = d1.clone().requires_grad_(True)
o2 = d2.clone().requires_grad_(True)
lo = O.smooth_ap_loss_me(o1, o2, pts1, pts2)
lo.backward()
npos = int((torch.cdist(pts1, pts2) < 5e-3).sum())
err = abs(lo.item() - loss.item())
gerr = (o1.grad - d1r.grad).abs().max().item()
assert err < 1e-6 and gerr < 1e-6, (err, gerr)
assert loss.item() > 0.02 and d1r.grad.abs().max().item() > 1e-4
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g08_match_me.npz"), desc1=d1.numpy(), desc2=d2.numpy(), pts3d_1=pts1.numpy(),
                    pts3d_2=pts2.numpy(), loss=loss.item(), gdesc1=d1r.grad.numpy(), gdesc2=d2r.grad.numpy(), npos=npos)
print(f"wrote g08_match_me.npz: loss {loss.item():.6f}, {npos} positives, oracle err {err:.1e} / grad {gerr:.1e}")
