"""Short fixed-batch training run of the bench-shaped engine: the loss must fall and every term must stay finite
(sanity check of the fused / atomic kernels over many consecutive optimiser steps; not a benchmark)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gd_amd  # noqa: F401
from gd_amd.finetune import FinetuneGD
from gd_testutil import synthetic_batch


def main(steps=40, P=8):
    dev = torch.device("cuda", 0)
    img, N, patch = 518, 300, 14
    hw = (img // patch) ** 2
    eng = FinetuneGD(r=4, backbone="vit_base", patch_size=patch, img_size=img, variant="mast3r", geometry="shared",
                     dtype="bf16", teacher_patch=patch, lora_b_std=1e-3, vit_kwargs=dict(init_values=1.0)).to(dev)
    eng.configure_optimizers(lr=1e-4)
    batch = synthetic_batch(P, img, img, N, hw, dev, seed=7, teacher_patch=patch)
    first = last = None
    for s in range(steps):
        loss, terms = eng.training_step(batch)
        eng.backward(loss)
        gn = eng.optimizer_step()
        v = float(loss)
        assert all(torch.isfinite(t).all() for t in terms.values()) and v == v, f"non-finite at step {s}"
        first = v if first is None else first
        last = v
        if s % 5 == 0 or s == steps - 1:
            print(f"step {s:3d} loss {v:.4f} grad-norm {float(gn):.3f} " + " ".join(f"{k} {float(t.mean()):.4f}" for k, t in terms.items()))
    print(f"first {first:.4f} last {last:.4f}")
    assert last < first, "loss did not decrease"


if __name__ == "__main__":
    main()
