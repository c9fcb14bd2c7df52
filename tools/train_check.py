"""Short fixed-batch training run of the bench-shaped engine: the loss must fall and every term must stay finite
(sanity check of the fused / atomic kernels over many consecutive optimiser steps; not a benchmark)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gd_amd  # noqa: F401
from gd_amd.finetune import FinetuneGD
from gd_testutil import synthetic_batch


def compare(steps=40, P=8, others=("bf16",), lr=1e-4):
    """the f32 engine and the faster engines from the same initial weights on the same fixed batch: the loss trajectories side by side
    (training-level parity of the throughput / TF32-class modes against the reference-precision mode).  For tf32h the fp16 range counters
    of the whole run are printed (saturated must stay 0 while the gradients shrink with the loss)."""
    dev = torch.device("cuda", 0)
    img, N, patch = 518, 300, 14
    hw = (img // patch) ** 2
    engs = {}
    for dt in ("f32",) + tuple(others):
        torch.manual_seed(0)
        e = FinetuneGD(r=4, backbone="vit_base", patch_size=patch, img_size=img, variant="mast3r", geometry="shared",
                       dtype=dt, teacher_patch=patch, lora_b_std=1e-3, vit_kwargs=dict(init_values=1.0)).to(dev)
        e.configure_optimizers(lr=lr)
        engs[dt] = e
    with torch.no_grad():        # identical weights (the constructor's init is seeded, but make it explicit)
        for dt in others:
            for a, b in zip(engs["f32"].parameters(), engs[dt].parameters()):
                b.copy_(a)
    batch = synthetic_batch(P, img, img, N, hw, dev, seed=7, teacher_patch=patch)
    worst = {dt: 0.0 for dt in others}
    for s in range(steps):
        row = {}
        for dt, e in engs.items():
            loss, terms = e.training_step(batch)
            e.backward(loss)
            e.optimizer_step()
            row[dt] = float(loss)
        for dt in others:
            worst[dt] = max(worst[dt], abs(row[dt] - row["f32"]) / abs(row["f32"]))
        if s % 5 == 0 or s == steps - 1:
            print(f"step {s:3d}  f32 {row['f32']:.5f}  " + "  ".join(f"{dt} {row[dt]:.5f} (rel {abs(row[dt] - row['f32']) / abs(row['f32']):.1e})" for dt in others), flush=True)
    for dt in others:
        print(f"{dt}: worst relative difference of the loss over {steps} steps (lr {lr:g}): {worst[dt]:.2e}")
    if "tf32h" in engs:
        print("tf32h fp16 range counters over the run:", engs["tf32h"].range_report())
    assert all(w < 5e-3 for w in worst.values())


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
    if "compare3" in sys.argv[1:]:          # python tools/train_check.py compare3 [steps] [lr]
        extra = [a for a in sys.argv[1:] if a != "compare3"]
        compare(steps=int(extra[0]) if extra else 100, others=("tf32h", "bf16"), lr=float(extra[1]) if len(extra) > 1 else 1e-4)
    elif "compare" in sys.argv[1:]:
        compare()
    else:
        main()
