"""GPU parity: teacher -> target glue kernels against fixtures produced by the reference's own functions
(tools/make_golden.py G15 / G16) and the oracle."""
import pytest
import torch

import gd_oracle as O
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def test_vggt_side_golden():
    from gd_amd import teacher_glue as tg
    g = load_golden("g15_teacher_glue")
    cu = lambda k: g[k].cuda()
    pm = tg.unproject_depth_map_to_point_map(cu("depth"), torch.stack([cu("E1"), cu("E2")]), torch.stack([cu("K1"), cu("K2")]))
    assert rel_err(pm, g["point_maps"]) < 1e-5
    m1, m2 = tg.get_coview_masks(cu("point_maps")[0], cu("point_maps")[1], cu("K1"), cu("E1"), cu("K2"), cu("E2"),
                                 tuple(g["depth"].shape[1:]))
    # a projection that lands within rounding of an image border may flip: allow a couple of pixels
    assert int((m1.cpu() != g["mask_1"]).sum()) <= 2 and int((m2.cpu() != g["mask_2"]).sum()) <= 2
    kps = tg.sample_keypoints_nms(cu("mask_1"), cu("conf"), 400, g["nms_min_distance"])
    assert torch.equal(kps.cpu(), g["nms_kps"])
    d = tg.point_cloud_to_depth(cu("pc_points"), cu("K1"), g["depth"].shape[2], g["depth"].shape[1])
    assert rel_err(d, g["pc_depth"]) < 1e-6
    _, idx = tg.filter_kp_by_conf(cu("fk_kp"), cu("fk_mask"))
    assert torch.equal(idx.cpu(), g["fk_idx"])


def test_reciprocal_nns_golden():
    from gd_amd import teacher_glue as tg
    g = load_golden("g16_reciprocal_nns")
    xy1, xy2 = tg.fast_reciprocal_NNs(g["desc1"].cuda(), g["desc2"].cuda(), g["subsample"])
    assert torch.equal(xy1.cpu(), g["xy1"].long()) and torch.equal(xy2.cpu(), g["xy2"].long())
    k1, k2 = tg.filter_and_match_keypoints(g["desc1"].cuda(), g["desc2"].cuda(), g["conf1"].cuda(), g["conf2"].cuda(),
                                           subsample=g["subsample"])
    assert torch.equal(k1[0].cpu(), g["kp1_filtered"]) and torch.equal(k2[0].cpu(), g["kp2_filtered"])


def test_nn_argmax_and_nms_at_teacher_scale():
    from gd_amd import teacher_glue as tg
    gen = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn(768, 24, generator=gen, device="cuda")
    db = torch.randn(384 * 512, 24, generator=gen, device="cuda")       # MASt3R 384x512 dense descriptors
    act = torch.rand(768, generator=gen, device="cuda") > 0.2
    idx = tg.nn_argmax(q, db, act)
    ref = (q @ db.t()).argmax(1)
    assert torch.equal(idx[act].long(), ref[act]) and bool((idx[~act] == -1).all())
    H = W = 518                                                            # VGGT frame
    mask = torch.rand(H, W, generator=gen, device="cuda") > 0.3
    conf = 1 + torch.rand(H, W, generator=gen, device="cuda")
    kps = tg.sample_keypoints_nms(mask, conf, 10 ** 9, 5)
    want = O.nms_keypoints(mask.cpu(), conf.cpu(), 10 ** 9, 5)
    assert torch.equal(kps.cpu(), want)
    sub = tg.sample_keypoints_nms(mask, conf, 300, 5)
    assert sub.shape == (300, 2)
    lin = set((want[:, 0] * W + want[:, 1]).tolist())
    assert all(int(r) * W + int(c) in lin for r, c in sub.cpu())


def test_cross_view_attention_maps_golden_and_full_size():
    """gd_cross_view_attn against the reference fixture (f32), and at the teacher's real size against the oracle (bf16 / f32),
    including the block mean folded in through weight / accumulate."""
    from gd_amd import teacher_glue as TG
    import gd_oracle as O
    g = load_golden("g14_cross_view_attn")
    q, k = g["q"].cuda(), g["k"].cuda()
    out = TG.cross_view_attention_maps(q, k, float(g["scale"]), float(g["temperature"]), int(g["prefix"]))
    assert out.shape == g["maps"].shape
    assert float((out.cpu() - g["maps"]).abs().max()) < 2e-6
    # real size: 2 views x (5 + 37*37) tokens, 16 heads; two "blocks" averaged
    B, H, n, prefix = 1, 16, 1369, 5
    N = 2 * (n + prefix)
    gen = torch.Generator(device="cuda").manual_seed(7)
    qs = [torch.randn(B, H, N, 64, device="cuda", generator=gen) for _ in range(2)]
    ks = [torch.randn(B, H, N, 64, device="cuda", generator=gen) for _ in range(2)]
    for dt, tol in ((torch.float32, 2e-6), (torch.bfloat16, 2e-3)):
        acc = None
        for i in range(2):
            acc = TG.cross_view_attention_maps(qs[i].to(dt), ks[i].to(dt), 0.125, 0.7, prefix, out=acc, weight=1.0 / (2 * H),
                                               accumulate=i > 0)
        ref = sum(O.cross_view_attention_maps(qs[i].to(dt).double(), ks[i].to(dt).double(), 0.125, 0.7, prefix) for i in range(2)) / 2
        assert float((acc.double() - ref).abs().max()) < tol * float(ref.abs().max()) + 1e-9, dt
        assert float((acc.sum(-1) - 1).abs().max()) < 1e-3


def test_mast3r_tgt_attn_map_golden_and_from_qk():
    from gd_amd import teacher_glue as TG
    g = load_golden("g17_mast3r_tgt_attn_map")
    out = TG.mast3r_tgt_attn_map([t.cuda() for t in g["tgt"]], [t.cuda() for t in g["src"]], float(g["temperature"]))
    assert float((out.cpu() - g["out"]).abs().max()) < 2e-6
    # from q / k (12 heads x 64, 768 tokens = the 24 x 32 MASt3R grid), 3 layers: never forms a per-head map
    B, H, N, L, scale = 2, 12, 768, 3, 0.125
    gen = torch.Generator(device="cuda").manual_seed(11)
    mk = lambda: [torch.randn(B, H, N, 64, device="cuda", generator=gen) for _ in range(L)]
    q1, k2, q2, k1 = mk(), mk(), mk(), mk()
    got = TG.mast3r_tgt_attn_map_from_qk(q1, k2, q2, k1, scale, 3.0)
    tgt = [(a.double() @ b.double().transpose(-1, -2)) * scale for a, b in zip(q1, k2)]
    src = [(a.double() @ b.double().transpose(-1, -2)) * scale for a, b in zip(q2, k1)]
    ref = O.mast3r_tgt_attn_map(tgt, src, 3.0)
    assert float((got.double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    assert float((got[:, :, 1:].sum(-1) - ref[:, :, 1:].sum(-1).float()).abs().max()) < 1e-4
