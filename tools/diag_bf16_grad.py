"""Where does the bf16 engine's gradient error at ViT-L / 518^2 come from?  (GPU box; diagnostics, not a test.)
For each case: per-tensor error of the depth head's gradient, the per-group table, a token-uniformity figure of the normed tap
grids (|common component| / |token-specific component|), with the normed taps kept in the engine dtype and in fp32."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p_)
import gd_amd  # noqa: E402,F401
from gd_amd.finetune import FinetuneGD  # noqa: E402
from gd_amd.synthetic import synthetic_batch  # noqa: E402
from test_gpu_step import OracleTrainer  # noqa: E402
from test_gpu_fullsize import _group_table  # noqa: E402

out = {}
for name, backbone, img, N in (("vit_large_518", "vit_large", 518, 300), ("vit_base_518", "vit_base", 518, 300), ("vit_large_336", "vit_large", 336, 200)):
    ref = None
    for tap_dt in (None, torch.float32):
        torch.manual_seed(0)
        eng = FinetuneGD(r=4, backbone=backbone, patch_size=14, img_size=img, variant="vggt", geometry="shared", dtype="bf16",
                         teacher_patch=14, lora_b_std=1e-3, vit_kwargs=dict(init_values=1.0)).cuda()
        eng.model.tap_norm_dtype = tap_dt
        batch = synthetic_batch(1, img, img, N, (img // 14) ** 2, "cuda", seed=1234, teacher_patch=14)
        if ref is None:
            orc = OracleTrainer(eng)
            ref = orc.step(batch, 1)
            names = orc.names
        _, _, ref_grads, _, _ = ref
        eng.configure_optimizers()
        loss, terms = eng.training_step(batch)
        eng.backward(loss)
        ps = eng.trainable_parameters()
        g_hip = [q.grad.detach().double().cpu() for q in ps]
        tot = torch.cat([g.reshape(-1) for g in ref_grads]).norm().item()
        rec = {"groups": {k: round(v["rel_fro"], 4) for k, v in _group_table(names, g_hip, ref_grads, tot).items() if not k.startswith("lora")},
               "head": {n: round(float((a - b).norm() / b.norm()), 4) for n, a, b in zip(names, g_hip, ref_grads) if n.startswith("depth_head")},
               "grad_rel_fro": float((torch.cat([g.reshape(-1) for g in g_hip]) - torch.cat([g.reshape(-1) for g in ref_grads])).norm() / tot)}
        with torch.no_grad():
            rgbs = torch.cat([batch["rgb_1"], batch["rgb_2"]], 0)
            taps, x, normed = eng.model.forward_all(rgbs, (4, 5, 6, 7), norm_taps=True)
            uni = []
            for t in normed:
                t = t.float()[:, 1:]
                common = t.mean(1, keepdim=True)
                uni.append(float(common.norm() * (t.shape[1] ** 0.5) / (t - common).norm()))
            rec["token_uniformity_normed_taps"] = [round(u, 2) for u in uni]
        out[f"{name}/taps_{'f32' if tap_dt else 'bf16'}"] = rec
        print(name, "taps", "f32" if tap_dt else "bf16", json.dumps(rec), flush=True)
        del eng
        torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag_bf16_grad.json"), "w"), indent=1)
