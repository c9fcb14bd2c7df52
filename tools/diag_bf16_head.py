"""Does the depth head amplify feature noise?  Keypoint features of the bf16 engine vs the fp64 oracle, and the depth-head
gradients computed IN FP64 TORCH from either feature set (the head math is then identical: any difference is feature noise)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p_)
import gd_amd  # noqa: E402,F401
import gd_oracle as O  # noqa: E402
from gd_amd.finetune import FinetuneGD  # noqa: E402
from gd_amd.synthetic import export_params, synthetic_batch  # noqa: E402

for name, backbone, img, N in (("vit_large_518", "vit_large", 518, 300), ("vit_large_336", "vit_large", 336, 200), ("vit_base_518", "vit_base", 518, 300)):
    torch.manual_seed(0)
    eng = FinetuneGD(r=4, backbone=backbone, patch_size=14, img_size=img, variant="vggt", geometry="shared", dtype="bf16",
                     teacher_patch=14, lora_b_std=1e-3, vit_kwargs=dict(init_values=1.0)).cuda()
    batch = synthetic_batch(1, img, img, N, (img // 14) ** 2, "cuda", seed=1234, teacher_patch=14)
    p, tr, refine, head, cfg = export_params(eng)
    p = {k: v.double() for k, v in p.items()}
    tr = {a: {i: {k: v.double() for k, v in d.items()} for i, d in dd.items()} for a, dd in tr.items()}
    refine = {k: v.double() for k, v in refine.items()}
    with torch.no_grad():
        rgbs = torch.cat([batch["rgb_1"], batch["rgb_2"]], 0)
        kp = torch.cat([batch["kp_1"], batch["kp_2"]], 0)
        eng.clear_cache()
        eng.model.prepare_trainables(None)
        f_hip = eng.get_intermediate_feature(rgbs, pts=kp).double().cpu()          # [2, N, D]
        eng.model.release_trainables()
        f_ref = torch.stack([O.student_features(batch[f"rgb_{v}"].cpu().double(), batch[f"kp_{v}"].cpu(), p, cfg, tr, refine)[0][0] for v in (1, 2)])
    d1 = O.extract_kp_depth(batch["depth_1"][0].cpu().double(), batch["kp_1"].cpu())
    d2 = O.extract_kp_depth(batch["depth_2"][0].cpu().double(), batch["kp_2"].cpu())
    res = {"feat_rel_fro": float((f_hip - f_ref).norm() / f_ref.norm())}
    diff = f_ref[0][None] - f_ref[0][:, None]
    res["rms_pair_diff_over_rms_feat"] = float(diff.pow(2).mean().sqrt() / f_ref.pow(2).mean().sqrt())
    grads = {}
    for tag, f in (("ref", f_ref), ("hip", f_hip)):
        hp = {k: v.double().clone().requires_grad_(True) for k, v in head.items()}
        ff = f.clone().requires_grad_(True)
        l1, r = O.depth_losses(hp, ff[0:1], ff[1:2], d1, d2)
        (eng.depth_loss_weight * l1 + eng.intra_depth_loss_weight * r).backward()
        grads[tag] = {k: v.grad.clone() for k, v in hp.items()}
        grads[tag]["feat"] = ff.grad.clone()
        res[f"loss_{tag}"] = (float(l1), float(r))
    res["head_grad_err_from_feature_noise"] = {k: round(float((grads["hip"][k] - grads["ref"][k]).norm() / grads["ref"][k].norm()), 4) for k in grads["ref"]}
    # pre-LN activations of the head on pair differences: how small is W1 diff + b1 relative to the LN epsilon?
    with torch.no_grad():
        h = torch.nn.functional.linear(diff, head["w1"].double(), head["b1"].double())
        res["head_preLN_std_mean"] = float(h.std(-1).mean())
        res["head_preLN_std_min"] = float(h.std(-1).min())
    print(name, json.dumps(res), flush=True)
    del eng
    torch.cuda.empty_cache()
