"""Full-size parity of the benched workload (BASELINE config 2: ViT-B/14 + LoRA, 518^2, 1370 tokens, hw = 1369,
300 keypoints) and of the other BASELINE widths / modes at their real token counts, against the fp64 CPU oracle on the
same weights: total loss and every loss term within the north_star's 1e-3 rel, gradient direction, updated weights.

The numbers of every case are also written to gpurun_out/fullsize_parity.json (scratch; copied under profiles/ per round).
"""
import json
import os

import pytest
import torch

from gd_testutil import synthetic_batch
from test_gpu_step import _oracle_step

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-3          # north_star: "loss parity to CPU reference within 1e-3 rel"
TERMS = (("ap_loss", "ap"), ("depth_loss", "depth"), ("intra_depth_loss", "intra"), ("kl_loss", "kl"))


def _record(name, rec):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "fullsize_parity.json")
    data = {}
    if os.path.exists(path):
        with open(path) as fh:
            data = json.load(fh)
    data[name] = rec
    with open(path, "w") as fh:
        json.dump(data, fh, indent=1, sort_keys=True)


def _run_case(name, backbone, variant, dtype, img=518, P=2, N=300, vit_kwargs=None, counts=None, eng_kwargs=None):
    from gd_amd.finetune import FinetuneGD
    torch.manual_seed(0)
    vk = dict(init_values=1.0) if vit_kwargs is None else vit_kwargs
    eng = FinetuneGD(r=4, backbone=backbone, patch_size=14, img_size=img, variant=variant, geometry="shared", dtype=dtype,
                     teacher_patch=14, lora_b_std=1e-3, vit_kwargs=vk, **(eng_kwargs or {})).cuda()
    hw = (img // 14) ** 2
    batch = synthetic_batch(P, img, img, N, hw, "cuda", seed=1234, teacher_patch=14, counts=counts)
    ref_loss, ref_terms, ref_grads, ref_params, ref_norm = _oracle_step(eng, batch, P)
    flat = eng.configure_optimizers()
    before = [q.detach().clone() for q in eng.trainable_parameters()]
    loss, terms = eng.training_step(batch)
    eng.backward(loss)
    rec = {"loss": loss.item(), "ref_loss": ref_loss, "rel_err": abs(loss.item() - ref_loss) / abs(ref_loss), "terms": {}}
    for a, b in TERMS:
        for q in range(P):
            r = ref_terms[q][b]
            rec["terms"][f"{b}[{q}]"] = {"hip": terms[a][q].item(), "ref": r,
                                         "rel_err": abs(terms[a][q].item() - r) / max(1e-3, abs(r))}
    ps = eng.trainable_parameters()
    # gradient of the whole trainable vector: relative Frobenius error and cosine against the oracle
    g_hip = torch.cat([q.grad.detach().double().cpu().reshape(-1) for q in ps])
    g_ref = torch.cat([g.reshape(-1) for g in ref_grads])
    rec["grad_rel_fro"] = ((g_hip - g_ref).norm() / g_ref.norm()).item()
    rec["grad_cos"] = (torch.dot(g_hip, g_ref) / (g_hip.norm() * g_ref.norm())).item()
    norm = eng.optimizer_step()
    rec["grad_norm"], rec["ref_grad_norm"] = norm.item(), ref_norm.item()
    # updated weights: the step moved every element by <= lr; compare the UPDATE vectors (post - pre), not the weights
    upd_hip = torch.cat([(q.detach() - b0).double().cpu().reshape(-1) for q, b0 in zip(ps, before)])
    upd_ref = torch.cat([(r - b0.double().cpu()).reshape(-1) for r, b0 in zip(ref_params, before)])
    w_ref = torch.cat([r.reshape(-1) for r in ref_params])
    w_hip = torch.cat([q.detach().double().cpu().reshape(-1) for q in ps])
    rec["update_cos"] = (torch.dot(upd_hip, upd_ref) / (upd_hip.norm() * upd_ref.norm())).item()
    rec["weights_rel_fro"] = ((w_hip - w_ref).norm() / w_ref.norm()).item()
    rec["max_weight_diff_over_lr"] = ((w_hip - w_ref).abs().max() / flat["lr"]).item()
    _record(name, rec)
    return rec


def _check(rec, tol=TOL, cos=0.99):
    assert rec["rel_err"] < tol, rec
    for k, t in rec["terms"].items():
        assert t["rel_err"] < tol, (k, t)
    assert abs(rec["grad_norm"] - rec["ref_grad_norm"]) < 2e-2 * rec["ref_grad_norm"], rec
    assert rec["grad_cos"] > cos, rec
    assert rec["max_weight_diff_over_lr"] <= 2.1, rec       # AdamW step 1 moves an element by at most lr (sign flips: 2 lr)
    assert rec["weights_rel_fro"] < 1e-3, rec


# BASELINE config 2 (the benched workload) in both engine dtypes and both loss variants
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("variant", ["mast3r", "vggt"])
def test_vit_base_518_step_matches_oracle(variant, dtype):
    rec = _run_case(f"vit_base_518_{variant}_{dtype}", "vit_base", variant, dtype, counts=[300, 211])
    _check(rec, cos=0.999 if dtype == "f32" else 0.99)


# BASELINE config 3: ViT-L/14 + VGGT losses (dense cost volume at C = 1024, hw = 1369)
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_vit_large_518_vggt_step_matches_oracle(dtype):
    rec = _run_case(f"vit_large_518_vggt_{dtype}", "vit_large", "vggt", dtype, P=1)
    _check(rec, cos=0.999 if dtype == "f32" else 0.99)


# BASELINE config 5: CLIP-style pre-norm ViT-L/14 (no LayerScale, LN eps 1e-5, timm pos-embed resample), bf16, all four losses
def test_prenorm_vit_large_bf16_step_matches_oracle():
    rec = _run_case("prenorm_vit_large_336_vggt_bf16", "vit_large", "vggt", "bf16", img=336, P=1, N=200,
                    vit_kwargs=dict(pre_norm=True, ln_eps=1e-5, pos_interp="timm"))
    _check(rec)


# BASELINE config 5 as worded: CLIP-style ViT-L/14 student + MASt3R teacher, "mixed corr + depth + cost losses", bf16 — the MASt3R
# trainer's loss kernels (masked rows -> softmax KL, keypoint patch masks) with the depth L1 term switched on as well
def test_prenorm_vit_large_bf16_mast3r_mixed_losses_step_matches_oracle():
    rec = _run_case("prenorm_vit_large_336_mast3r_all_losses_bf16", "vit_large", "mast3r", "bf16", img=336, P=1, N=200,
                    vit_kwargs=dict(pre_norm=True, ln_eps=1e-5, pos_interp="timm"), eng_kwargs=dict(depth_loss_weight=1.0))
    _check(rec)
