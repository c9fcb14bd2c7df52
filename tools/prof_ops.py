"""torch.profiler view of one benchmark-shaped step: which aten ops (with shapes) launch the small element-wise kernels."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gd_amd  # noqa: F401
from gd_amd.finetune import FinetuneGD
from gd_testutil import synthetic_batch


def main():
    dev = torch.device("cuda", 0)
    P, img, N, patch = 32, 518, 300, 14
    hw = (img // patch) ** 2
    eng = FinetuneGD(r=4, backbone="vit_base", patch_size=patch, img_size=img, variant="mast3r", geometry="shared",
                     dtype="bf16", teacher_patch=patch, lora_b_std=1e-3, vit_kwargs=dict(init_values=1.0)).to(dev)
    eng.configure_optimizers()
    batches = [synthetic_batch(P, img, img, N, hw, dev, seed=1234 + i, teacher_patch=patch) for i in range(2)]

    def step(i):
        loss, _ = eng.training_step(batches[i % 2])
        eng.backward(loss)
        eng.optimizer_step()
    for s in range(2):
        step(s)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(2)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.name.startswith("aten::") and e.device_time > 0:
            k = (e.name, str(e.input_shapes)[:120])
            agg[k][0] += 1; agg[k][1] += e.device_time
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{v[1]:9.1f} us x{v[0]:4d} {k[0]:22s} {k[1]}")


main()
