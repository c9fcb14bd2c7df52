"""Short fixed-batch training run of the bench-shaped engine: the loss must fall and every term must stay finite
(sanity check of the fused / atomic kernels over many consecutive optimiser steps; not a benchmark)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gd_amd  # noqa: F401
from gd_amd.finetune import FinetuneGD
from gd_testutil import synthetic_batch


def compare(steps=40, P=8):
    """bf16 and f32 engines from the same initial weights on the same fixed batch: the two loss trajectories side by side
    (training-level parity of the throughput mode against the reference-precision mode)."""
    dev = torch.device("cuda", 0)
    img, N, patch = 518, 300, 14
    hw = (img // patch) ** 2
    engs = {}
    for dt in ("f32", "bf16"):
        torch.manual_seed(0)
        e = FinetuneGD(r=4, backbone="vit_base", patch_size=patch, img_size=img, variant="mast3r", geometry="shared",
                       dtype=dt, teacher_patch=patch, lora_b_std=1e-3, vit_kwargs=dict(init_values=1.0)).to(dev)
        e.configure_optimizers(lr=1e-4)
        engs[dt] = e
    with torch.no_grad():        # identical weights (the constructor's init is seeded, but make it explicit)
        for a, b in zip(engs["f32"].parameters(), engs["bf16"].parameters()):
            b.copy_(a)
    batch = synthetic_batch(P, img, img, N, hw, dev, seed=7, teacher_patch=patch)
    worst = 0.0
    for s in range(steps):
        row = {}
        for dt, e in engs.items():
            loss, terms = e.training_step(batch)
            e.backward(loss)
            e.optimizer_step()
            row[dt] = float(loss)
        rel = abs(row["bf16"] - row["f32"]) / abs(row["f32"])
        worst = max(worst, rel)
        if s % 5 == 0 or s == steps - 1:
            print(f"step {s:3d}  f32 {row['f32']:.5f}  bf16 {row['bf16']:.5f}  rel diff {rel:.2e}")
    print(f"worst relative difference of the loss over {steps} steps: {worst:.2e}")
    assert worst < 5e-3


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
    if "compare" in sys.argv[1:]:
        compare()
    else:
        main()
