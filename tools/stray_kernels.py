"""Which Python lines launch the torch-side (non-libgd_hip) kernels of one bench-shaped step?  torch.profiler with stacks;
prints kernel name, count, total us and the innermost frame inside this repository."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gd_amd  # noqa: F401
from gd_amd.finetune import FinetuneGD
from gd_testutil import synthetic_batch
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
img, N, patch, P = 518, 300, 14, 32
hw = (img // patch) ** 2
eng = FinetuneGD(r=4, backbone="vit_base", patch_size=patch, img_size=img, variant="mast3r", geometry="shared",
                 dtype=os.environ.get("DT", "tf32h"), teacher_patch=patch, lora_b_std=1e-3, vit_kwargs=dict(init_values=1.0)).to(dev)
eng.configure_optimizers(lr=1e-4)
from gd_amd.teacher_cache import cache_cost_targets
batch = cache_cost_targets(synthetic_batch(P, img, img, N, hw, dev, seed=7, teacher_patch=patch))      # as bench.py's Job holds its batches


def step():          # what bench.py times: fit_step (weight gradients accumulate straight into the flat buffer)
    eng.fit_step(batch)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
by = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    chain, q = [], e
    while q is not None:
        chain.append(q.name)
        q = q.cpu_parent
    for k in e.kernels:
        if not ("at::native" in k.name or "Memcpy" in k.name or "Memset" in k.name or "rocclr" in k.name):
            continue
        st = [f for f in (e.stack or []) if "site-packages" not in f and "dist-packages" not in f and "<built-in" not in f]
        where = " <- ".join(x.strip().split("/")[-1] for x in st[:3]) if st else " < ".join(chain[:5])
        key = (where + "   [" + chain[0] + "]", k.name[:50])
        by[key][0] += 1
        by[key][1] += k.duration
rows = sorted(by.items(), key=lambda kv: -kv[1][1])
print(f"torch-side kernels in one step: {sum(v[0] for _, v in rows)} launches, {sum(v[1] for _, v in rows):.0f} us")
for (chain, kn), (n, us) in rows[:70]:
    print(f"{us:7.1f} us {n:3d}x {kn:50s} {chain}")
