"""G19: a checkpoint written by the REFERENCE — `FinetuneMASt3RTIMM.on_save_checkpoint` (src/finetune_timm_mast3r.py:172-190)
called unbound on a fake self that holds the reference's own `Adapter` / `DepthAwareFeatureFusion` (utils/model.py) and LoRA
Linear lists — flattened to an .npz (key path -> array).  tests/test_checkpoint_compat.py loads it through
`FinetuneGD.on_load_checkpoint` and checks that `FinetuneGD.on_save_checkpoint` writes the same key layout, and this script asserts
the reverse direction here: the reference's `on_load_checkpoint` (unbound) accepts a checkpoint written by FinetuneGD.
Build container only.  Usage: python tools/make_golden_g19.py"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import as R  # noqa: E402

R.install()
from src.finetune_timm_mast3r import FinetuneMASt3RTIMM  # noqa: E402
from utils.model import Adapter, DepthAwareFeatureFusion  # noqa: E402

D, r, nblk = 64, 4, 4                      # vit_tiny_test width, 8 blocks, adapters on blocks[4:]
torch.manual_seed(190)


def fake():
    class _Self:
        pass
    s = _Self()
    s.w_As = [nn.Linear(D, r, bias=False) for _ in range(2 * nblk)]
    s.w_Bs = [nn.Linear(r, D, bias=False) for _ in range(2 * nblk)]
    for l in s.w_Bs:
        nn.init.normal_(l.weight, std=0.05)
    s.refine_conv = nn.Conv2d(D, D, 3, padding=1)
    s.depth_diff_head = DepthAwareFeatureFusion(input_dim=D, use_tanh=True)
    s.adapters = nn.ModuleList([Adapter(dim=D, bottleneck_dim=64) for _ in range(nblk)])
    return s


ref = fake()
ck = {}
FinetuneMASt3RTIMM.on_save_checkpoint(ref, ck)
flat = {}


def walk(prefix, v):
    if isinstance(v, dict):
        for k, x in v.items():
            walk(f"{prefix}/{k}" if prefix else str(k), x)
    else:
        flat[prefix] = v.detach().numpy()


walk("", ck)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g19_reference_checkpoint.npz"), **flat)
print(f"wrote g19_reference_checkpoint.npz: {len(flat)} tensors, top-level keys {sorted(ck)[:6]} ...")

# reverse direction: a FinetuneGD checkpoint through the reference's on_load_checkpoint
import gd_amd  # noqa: E402,F401
from gd_amd.finetune import FinetuneGD  # noqa: E402
eng = FinetuneGD(r=r, backbone="vit_tiny_test", patch_size=14, img_size=56, variant="mast3r", dtype="f32", lora_b_std=0.05)
mine = eng.on_save_checkpoint({})
tgt = fake()
FinetuneMASt3RTIMM.on_load_checkpoint(tgt, mine)
for a, b in zip(tgt.w_As + tgt.w_Bs, eng.w_As + eng.w_Bs):
    assert torch.equal(a.weight, b.weight)
for a, b in zip(tgt.adapters, eng.adapters):
    assert torch.equal(a.down.weight, b.down.weight) and torch.equal(a.up.weight, b.up.weight)
assert all(torch.equal(x, y) for x, y in zip(tgt.depth_diff_head.state_dict().values(), eng.depth_diff_head.state_dict().values()))
assert torch.equal(tgt.refine_conv.weight, eng.refine_conv.weight) and tgt.loaded
print("reference on_load_checkpoint accepted a FinetuneGD checkpoint (extra keys gd_optimizer_state / epoch ignored)")
