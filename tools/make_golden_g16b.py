"""G16b: the MASt3R keypoint pipeline end to end — `FinetuneMASt3RTIMM.filter_and_match_keypoints`
(src/finetune_timm_mast3r.py:392-469) called UNBOUND on a fake self, i.e. the reference's own reciprocal-NN matching
(subsample 16, dist='dot'), 3-px border filter and union-of-percentile confidence filter, not the oracle's restatement of them.
The oracle (O.reciprocal_nns + O.mast3r_keypoint_filter) is asserted equal.  Build container only.
Usage: python tools/make_golden_g16b.py"""
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
from src.finetune_timm_mast3r import FinetuneMASt3RTIMM  # noqa: E402

H, W, D = 64, 96, 24
g = torch.Generator().manual_seed(1600)
base = torch.randn(H, W, D, generator=g)
d1 = base.half().float()                                                      # fp16-exact values: a compact fixture
d2 = (torch.roll(base, shifts=(-6, -6), dims=(0, 1)) + 0.2 * torch.randn(H, W, D, generator=g)).half().float()
c1 = (0.05 + torch.rand(H, W, generator=g)).half().float()
c2 = (0.05 + torch.rand(H, W, generator=g)).half().float()
# make every branch of the filters fire: matches of the first seed row / column land within 3 px of the border of view 2
# (border filter); two interior matches get the lowest confidence in BOTH views (dropped), a third in view 1 only (kept: the
# reference keeps the UNION of the two confidence filters, src/finetune_timm_mast3r.py:456)
m1, m2 = O.reciprocal_nns(d1, d2, subsample=16)
inner = [i for i in range(len(m1)) if 3 <= m2[i, 0] < W - 3 and 3 <= m2[i, 1] < H - 3 and 3 <= m1[i, 0] < W - 3 and 3 <= m1[i, 1] < H - 3]
assert len(inner) >= 6 and len(inner) < len(m1)
for i in inner[:2]:
    c1[m1[i, 1], m1[i, 0]] = 0.0
    c2[m2[i, 1], m2[i, 0]] = 0.0
c1[m1[inner[2], 1], m1[inner[2], 0]] = 0.0


class _Self:
    device = torch.device("cpu")
    min_conf_thr = 10


feats = {"view_1": {"true_shape": [[H, W]]}, "view_2": {"true_shape": [[H, W]]}, "desc_1": d1, "desc_2": d2, "conf_1": c1, "conf_2": c2}
rgb = torch.rand(1, 3, H, W, generator=g)
kp1, kp2, r1, r2, w, h = FinetuneMASt3RTIMM.filter_and_match_keypoints(_Self(), feats, rgb, rgb)
assert (w, h) == (W, H) and kp1.shape == kp2.shape and kp1.shape[1] > 0
o1, o2 = O.reciprocal_nns(d1, d2, subsample=16)
k1, k2 = O.mast3r_keypoint_filter(o1, o2, c1, c2)
assert torch.equal(k1, kp1) and torch.equal(k2, kp2), "oracle != reference filter_and_match_keypoints"
assert kp1.shape[1] == len(inner) - 2, (kp1.shape, len(inner), len(o1))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g16b_filter_and_match.npz"), desc1=d1.half().numpy(), desc2=d2.half().numpy(),
                    conf1=c1.half().numpy(), conf2=c2.half().numpy(), kp1=kp1[0].numpy(), kp2=kp2[0].numpy())
print(f"wrote g16b_filter_and_match.npz: {kp1.shape[1]} keypoints of {(H // 16) * (W // 16)} seeds survive "
      f"({len(o1)} reciprocal matches before the filters)")
