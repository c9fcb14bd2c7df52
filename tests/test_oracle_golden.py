"""CPU: the oracle/ restatement against the fixtures generated from the REFERENCE's own code
(tools/make_golden.py).  These pin the checker; they run without a GPU."""
import torch
import torch.nn.functional as F

import gd_oracle as O
from conftest import load_golden, rel_err


def test_sigmoid():
    g = load_golden("g01_sigmoid")
    assert rel_err(O.sigmoid_t(g["x"], g["temp"]), g["y"]) < 1e-7


def test_interpolate_features():
    for P in (14, 16):
        g = load_golden(f"g02_interp_p{P}")
        desc = g["desc"].clone().requires_grad_(True)
        out = O.interpolate_features(desc, g["pts"], g["h"], g["w"], False, P, P)
        assert rel_err(out, g["out"]) < 1e-6
        (out * g["gout"]).sum().backward()
        assert rel_err(desc.grad, g["gdesc"]) < 1e-6
        assert rel_err(O.interpolate_features(desc.detach(), g["pts"], g["h"], g["w"], True, P, P), g["out_norm"]) < 1e-6


def test_patch_mask():
    g = load_golden("g03_patch_mask")
    assert torch.equal(O.patch_mask_from_kp(g["kp"], g["H"], g["W"], g["patch"]), g["mask"])


def test_masked_cost_and_kl():
    g = load_golden("g04_masked_cost")
    assert rel_err(O.masked_patch_cost(g["cost"], g["row_mask"]), g["renorm"]) < 1e-6
    assert rel_err(O.masked_patch_cost(g["cost"] * 4 - 2, g["row_mask"], use_softmax=True, temperature=0.7),
                   g["softmax_t07"]) < 1e-6
    assert rel_err(O.kl_divergence_map(g["kl_t"], g["kl_p"]), torch.as_tensor(g["kl"])) < 1e-6


def test_cost_loss_variants():
    for v in ("vggt", "mast3r"):
        g = load_golden(f"g06_cost_{v}")
        f1 = g["f1"][None].clone().requires_grad_(True)
        f2 = g["f2"][None].clone().requires_grad_(True)
        loss = O.cost_volume_kl(f1, f2, g["t1"][None], g["t2"][None], g["m1"], g["m2"], v)
        assert abs(loss.item() - g["loss"]) < 1e-5 * abs(g["loss"])
        loss.backward()
        assert rel_err(f1.grad[0], g["g1"]) < 1e-4 and rel_err(f2.grad[0], g["g2"]) < 1e-4


def _hp(g):
    return {k: g["hp_" + k] for k in ("w1", "b1", "ln_w", "ln_b", "w2", "b2")}


def test_ranking_loss():
    g = load_golden("g07_ranking")
    hp = {k: v.clone().requires_grad_(True) for k, v in _hp(g).items()}
    feats = g["feats"].clone().requires_grad_(True)
    loss = O.pairwise_ranking_loss(hp, feats, g["depths"], g["thr"])
    assert abs(loss.item() - g["loss"]) < 1e-5
    loss.backward()
    assert rel_err(feats.grad, g["gfeats"]) < 1e-4
    for k in hp:
        assert rel_err(hp[k].grad, g["g_" + k]) < 1e-4, k


def test_matching_loss():
    for v in ("vggt", "mast3r"):
        g = load_golden(f"g08_match_{v}")
        d1 = g["desc1"].clone().requires_grad_(True)
        d2 = g["desc2"].clone().requires_grad_(True)
        loss = O.smooth_ap_loss(d1, d2, g["pts3d_1"], g["pts3d_2"], v)
        assert abs(loss.item() - g["loss"]) < 1e-6
        loss.backward()
        assert rel_err(d1.grad, g["gdesc1"]) < 1e-5 and rel_err(d2.grad, g["gdesc2"]) < 1e-5


def test_kp_depth():
    g = load_golden("g09_kp_depth")
    assert rel_err(O.extract_kp_depth(g["depth"], g["kp"]), g["out"]) < 1e-6


def test_lora_adapter():
    g = load_golden("g10_lora_adapter")
    x = g["x"].clone().requires_grad_(True)
    lo = {k: g[k].clone().requires_grad_(True) for k in ("a_q", "b_q", "a_v", "b_v")}
    y = O.lora_qkv(x, g["qkv_w"], g["qkv_b"], lo)
    assert rel_err(y, g["y"]) < 1e-6
    (y * g["gy"]).sum().backward()
    assert rel_err(x.grad, g["gx"]) < 1e-5
    for k in lo:
        assert rel_err(lo[k].grad, g["g_" + k]) < 1e-5
    ad = {"down": g["ad_down"].clone().requires_grad_(True), "up": g["ad_up"].clone().requires_grad_(True)}
    x2 = g["ad_x"].clone().requires_grad_(True)
    y2 = O.adapter(x2, ad)
    assert rel_err(y2, g["ad_y"]) < 1e-6
    (y2 * g["ad_gy"]).sum().backward()
    assert rel_err(x2.grad, g["ad_gx"]) < 1e-5
    assert rel_err(ad["down"].grad, g["ad_g_down"]) < 1e-5 and rel_err(ad["up"].grad, g["ad_g_up"]) < 1e-5


def test_depth_loss():
    g = load_golden("g11_depth_loss")
    hp = _hp(g)
    d1 = O.extract_kp_depth(g["depth_1"], g["kp_1"])
    d2 = O.extract_kp_depth(g["depth_2"], g["kp_2"])
    l1, intra = O.depth_losses(hp, g["kf1"], g["kf2"], d1, d2)
    assert abs(l1.item() - g["depth_loss"]) < 1e-6 and abs(intra.item() - g["intra_loss"]) < 1e-6


def _vit_inputs(size):
    g = load_golden(f"g12_vit_{size}")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    tr = {"lora": {}, "adapter": {}}
    for bi in (4, 5):
        tr["lora"][bi] = {k: g[f"lora_{bi}_{k}"] for k in ("a_q", "b_q", "a_v", "b_v")}
        tr["adapter"][bi] = {k: g[f"adapter_{bi}_{k}"] for k in ("down", "up")}
    cfg = dict(patch=14, dim=64, depth=6, heads=4, ln_eps=1e-6, pos_interp="dinov2")
    return g, sd, tr, cfg


def test_vit_forward_backward():
    for size in (56, 70):
        g, sd, tr, cfg = _vit_inputs(size)
        for d in tr.values():
            for blk in d.values():
                for k in blk:
                    blk[k] = blk[k].clone().requires_grad_(True)
        taps, x = O.vit_forward(g["img"], sd, cfg, tr, taps=(4, 5))
        assert rel_err(taps[0], g["tap4"]) < 2e-5 and rel_err(taps[1], g["tap5"]) < 2e-5
        assert rel_err(O.final_norm(x, sd, cfg), g["xnorm"]) < 2e-5
        ((taps[0] * g["wt4"]).sum() + (taps[1] * g["wt5"]).sum()).backward()
        for bi in (4, 5):
            for k in ("a_q", "b_q", "a_v", "b_v"):
                assert rel_err(tr["lora"][bi][k].grad, g[f"g_{k}_{bi}"]) < 1e-4, (bi, k)
            for k in ("down", "up"):
                assert rel_err(tr["adapter"][bi][k].grad, g[f"g_{k}_{bi}"]) < 1e-4, (bi, k)


def test_stride_override_tokens_g21():
    """src/evaluate_timm.py:262-272 on the reference's in-tree ViT (stride 7 on patch 14, `_fix_pos_enc` bound): fixture G21."""
    g = load_golden("g21_stride_override")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    cfg = dict(patch=14, dim=64, depth=2, heads=1, ln_eps=1e-6, pos_interp="dinov2", patch_stride=(7, 7))
    for tag in ("sq", "rect"):
        img = g[f"{tag}.img"]
        h, w = img.shape[-2:]
        gh, gw = 1 + (h - 14) // 7, 1 + (w - 14) // 7
        assert rel_err(O.fix_pos_enc(sd["pos_embed"], 14, (7, 7), gh * gw, h, w), g[f"{tag}.pos"]) < 1e-6
        x = O.vit_tokens(O.normalize_image(img, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)), sd, cfg)
        for i in range(2):
            x = O.vit_block(x, sd, i, cfg, None, None)
        assert rel_err(O.final_norm(x, sd, cfg), g[f"{tag}.xnorm"]) < 2e-5


def test_rope2d():
    g = load_golden("g13_rope2d")
    tok = g["tokens_bhnd"].transpose(1, 2).contiguous()
    out = O.rope_2d(tok, g["positions"], g["base"], 1.0)
    assert rel_err(out.transpose(1, 2), g["out_bhnd"]) < 1e-5
    assert rel_err(O.rope_2d(out, g["positions"], g["base"], -1.0), tok) < 1e-5


def test_teacher_glue_vggt_side():
    g = load_golden("g15_teacher_glue")
    pm = O.unproject_depth(g["depth"], torch.stack([g["E1"], g["E2"]]), torch.stack([g["K1"], g["K2"]]))
    assert rel_err(pm, g["point_maps"]) < 1e-5
    m1, m2 = O.coview_masks(g["point_maps"][0], g["point_maps"][1], g["K1"], g["E1"], g["K2"], g["E2"], tuple(g["depth"].shape[1:]))
    assert torch.equal(m1, g["mask_1"]) and torch.equal(m2, g["mask_2"])
    assert torch.equal(O.nms_keypoints(g["mask_1"], g["conf"], 400, g["nms_min_distance"]), g["nms_kps"])
    assert rel_err(O.point_cloud_to_depth(g["pc_points"], g["K1"], g["depth"].shape[2], g["depth"].shape[1]), g["pc_depth"]) < 1e-6
    assert torch.equal(O.filter_kp_by_conf(g["fk_kp"], g["fk_mask"])[1], g["fk_idx"])


def test_reciprocal_nns():
    g = load_golden("g16_reciprocal_nns")
    xy1, xy2 = O.reciprocal_nns(g["desc1"], g["desc2"], g["subsample"])
    assert torch.equal(xy1, g["xy1"].long()) and torch.equal(xy2, g["xy2"].long())
    k1, k2 = O.mast3r_keypoint_filter(xy1, xy2, g["conf1"], g["conf2"])
    assert torch.equal(k1[0], g["kp1_filtered"]) and torch.equal(k2[0], g["kp2_filtered"])


def test_g14_cross_view_attention_maps():
    """VGGT teacher cross-view maps (vggt/layers/attention.py:51-85 + head mean): oracle vs the reference's output."""
    g = load_golden("g14_cross_view_attn")
    got = O.cross_view_attention_maps(g["q"], g["k"], float(g["scale"]), float(g["temperature"]), int(g["prefix"]))
    assert got.shape == g["maps"].shape
    assert float((got - g["maps"]).abs().max()) < 1e-6
    assert float((got.sum(-1) - 1).abs().max()) < 1e-5      # every row is a mean of softmax rows


def test_g08_matching_loss_me_variant():
    """ME variant (src/finetune_timm_me.py:191-220, dynamic positives): oracle vs the reference's loss and gradients."""
    g = load_golden("g08_match_me")
    d1 = g["desc1"].clone().requires_grad_(True)
    d2 = g["desc2"].clone().requires_grad_(True)
    loss = O.smooth_ap_loss_me(d1, d2, g["pts3d_1"], g["pts3d_2"])
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    assert float((d1.grad - g["gdesc1"]).abs().max()) < 1e-6 and float((d2.grad - g["gdesc2"]).abs().max()) < 1e-6


def test_g17_mast3r_tgt_attn_map():
    """MASt3R teacher target map (dust3r/dust3r/model.py:346-366): oracle vs the reference's forward."""
    g = load_golden("g17_mast3r_tgt_attn_map")
    got = O.mast3r_tgt_attn_map(list(g["tgt"]), list(g["src"]), float(g["temperature"]))
    assert float((got - g["out"]).abs().max()) < 1e-6


def test_g16b_filter_and_match_oracle():
    """The oracle's MASt3R keypoint pipeline equals the reference's filter_and_match_keypoints output (G16b)."""
    g = load_golden("g16b_filter_and_match")
    d1, d2, c1, c2 = (g[k].float() for k in ("desc1", "desc2", "conf1", "conf2"))
    o1, o2 = O.reciprocal_nns(d1, d2, subsample=16)
    k1, k2 = O.mast3r_keypoint_filter(o1, o2, c1, c2)
    assert torch.equal(k1[0], g["kp1"]) and torch.equal(k2[0], g["kp2"])


def test_full_optimisation_step_g18():
    """G18: the reference's own training_step (both trainers) + clip_grad_norm_ + its configure_optimizers' AdamW on two
    pairs; the oracle reproduces every loss term, the gradients, the clip norm and the updated weights."""
    import pytest
    from gd_testutil import g18_oracle_inputs, g18_pair_batch, g18_unpack
    for variant in ("vggt", "mast3r"):
        g = load_golden(f"g18_full_step_{variant}")
        sd, before, after, grads, pairs = g18_unpack(g)
        tr, refine, hp, leaves, cfg = g18_oracle_inputs(before, variant, g)
        w = {"ap": 1.0, "depth": 1.0 if variant == "vggt" else 0.0, "intra": 1.0, "kl": 1.0}
        total = 0
        for t in pairs:
            terms = O.pair_losses(g18_pair_batch(t), sd, cfg, tr, refine, hp)
            for k in ("ap", "depth", "intra", "kl"):
                assert terms[k].item() == pytest.approx(float(t[f"term_{k}"]), rel=2e-5), (variant, k)
            total = total + O.total_loss(terms, w) / len(pairs)
        assert total.item() == pytest.approx(float(g["loss"]), rel=2e-5)
        total.backward()
        og = [x.grad if x.grad is not None else torch.zeros_like(x) for x in leaves]
        for i, (a, b) in enumerate(zip(og, grads)):
            if b.abs().max() > 0:
                assert rel_err(a, b) < 5e-4, (variant, i)
        live = [int(i) for i in g["n_live"]]
        params = [x.detach().clone() for x in leaves]
        state = [(torch.zeros_like(x), torch.zeros_like(x)) for x in params]
        norm = O.clip_and_adamw([params[i] for i in live], [og[i] for i in live], [state[i] for i in live], 1)
        assert norm.item() == pytest.approx(float(g["clip_norm"]), rel=2e-5)
        for i, (p1, a1, b0) in enumerate(zip(params, after, before)):
            da, db = (p1 - b0).double(), (a1 - b0).double()
            assert (da - db).norm() <= 2e-3 * db.norm() + 1e-12, (variant, i)


def test_load_fn_g20():
    """G20: the reference's load_and_preprocess_images (PIL bicubic resize, ToTensor, crop / white padding) — the oracle's
    integer restatement of Pillow's resampler is bit-exact."""
    g = load_golden("g20_load_fn")
    for mode in ("crop", "pad"):
        ins = [g[f"{mode}.in{i}"].numpy() for i in range(2)]
        out = O.preprocess_images(ins, mode=mode)
        assert torch.equal(out, g[f"{mode}.out_u8"].float() / 255)
