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
    assert torch.equal(m1.cpu(), g["mask_1"].bool()) and torch.equal(m2.cpu(), g["mask_2"].bool())     # bool work: exact
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
    # (the filtered keypoints of this fixture were written by the oracle; the reference-generated ones are G16b below)


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


def test_filter_and_match_keypoints_reference_fixture():
    """G16b: keypoints produced by the reference's own `FinetuneMASt3RTIMM.filter_and_match_keypoints` called unbound
    (tools/make_golden_g16b.py): reciprocal NNs at subsample 16, border filter and union-of-confidence filter all fire."""
    from gd_amd import teacher_glue as tg
    g = load_golden("g16b_filter_and_match")
    k1, k2 = tg.filter_and_match_keypoints(g["desc1"].float().cuda(), g["desc2"].float().cuda(), g["conf1"].float().cuda(),
                                           g["conf2"].float().cuda(), subsample=16, min_conf_thr=10)
    assert torch.equal(k1[0].cpu(), g["kp1"]) and torch.equal(k2[0].cpu(), g["kp2"])


@pytest.mark.parametrize("k,H,W,holes", [(3, 42, 60, 0.6), (5, 37, 49, 0.85), (3, 336, 512, 0.5), (3, 24, 24, 0.0)])
def test_post_process_depth_vs_oracle(k, H, W, holes):
    """post_process_depth (utils/functions.py:262-345) against the oracle's restatement (kornia's filters restated: parity
    unpinned): sparse rasterised depth with holes, no holes at all, and the MASt3R frame size, batched and single."""
    from gd_amd import teacher_glue as tg
    g = torch.Generator().manual_seed(H + W)
    d = 0.5 + 3 * torch.rand(2, H, W, generator=g)
    d[torch.rand(2, H, W, generator=g) < holes] = 0.0
    d[1, H // 3:H // 2, W // 4:W // 2] = 0.0                        # a hole wider than both fill kernels
    out = tg.post_process_depth(d.cuda(), kernel_size=k)
    assert out.shape == (2, H, W)
    for p in range(2):
        ref = O.post_process_depth(d[p].double(), kernel_size=k)
        # the 3-sigma outlier switch is discontinuous: a pixel whose |q - mean| sits within fp32 rounding of 3 std may take
        # the other branch; everything else agrees to fp32 accuracy
        diff = (out[p].cpu().double() - ref).abs()
        assert float((diff > 1e-4 * (1 + ref.abs())).float().mean()) < 2e-3
        assert float(diff.median()) < 1e-5
    one = tg.post_process_depth(d[0].cuda(), kernel_size=k)
    assert one.shape == (H, W) and torch.equal(one, out[0])


def ops_stats(e):
    from gd_amd import ops
    return ops.cost_volume_teacher_stats(e["cost_1"][None], e["cost_2"][None])[0]


def test_target_composites_and_cache_feed_the_step():
    """extract_vggt_targets / extract_mast3r_targets -> TeacherTargetCache -> collate -> FinetuneGD.training_step: the
    composites chain the glue kernels exactly as the reference's extract_* / sample_keypoints / filter_and_match_keypoints do
    (checked against the oracle's pieces), the cache runs the producer once per pair, and a cached batch steps."""
    from gd_amd import teacher_glue as tg
    from gd_amd.finetune import FinetuneGD
    from gd_amd.teacher_cache import TeacherTargetCache
    gen = torch.Generator().manual_seed(77)
    Himg = Wimg = 56                       # 4 x 4 patches of 14
    n, prefix, Hh = 16, 5, 2
    N = 2 * (n + prefix)
    g15 = load_golden("g15_teacher_glue")
    E = torch.stack([g15["E1"], g15["E2"]]).cuda()
    K = torch.stack([g15["K1"], g15["K2"]]).cuda()
    K[:, :2, 2] = 28.0
    depth = (1.0 + 2 * torch.rand(2, Himg, Wimg, generator=gen)).cuda()
    conf = (1 + torch.rand(2, Himg, Wimg, generator=gen)).cuda()
    qk = [(torch.randn(1, Hh, N, 64, generator=gen).cuda(), torch.randn(1, Hh, N, 64, generator=gen).cuda()) for _ in range(2)]
    calls = []

    def vggt_producer():
        calls.append(1)
        return tg.extract_vggt_targets(qk, depth, conf, E, K, track_fn=lambda kp: (kp + 1).clamp(0, Wimg - 1), num_keypoints=4000,
                                       min_distance=3)
    cache = TeacherTargetCache()
    e = cache.get("pair0", vggt_producer, temperature=1.0)
    cache.get("pair0", vggt_producer, temperature=1.0)
    assert len(calls) == 1 and cache.hits == 1 and cache.misses == 1
    # pieces against the oracle
    ref_cost = sum(O.cross_view_attention_maps(q.cpu().double(), k.cpu().double(), 0.125, 1.0, prefix) for q, k in qk) / 2
    hw = n
    assert rel_err(e["cost_1"][:, :hw], ref_cost[0]) < 1e-4 and rel_err(e["cost_2"][:, :hw], ref_cost[1]) < 1e-4
    assert e["cost_1"].shape[1] % 4 == 0
    pm = O.unproject_depth(depth.cpu(), E.cpu(), K.cpu())
    om1, _ = O.coview_masks(pm[0], pm[1], K[0].cpu(), E[0].cpu(), K[1].cpu(), E[1].cpu(), (Himg, Wimg))
    want = O.nms_keypoints(om1, conf[0].cpu(), 4000, 3)
    if want is not None:
        kp1 = want[:, [1, 0]]
        ok = (kp1[:, 0] >= 3) & (kp1[:, 0] < Wimg - 3) & (kp1[:, 1] >= 3) & (kp1[:, 1] < Himg - 3) & (kp1[:, 0] + 1 < Wimg - 3) & (kp1[:, 1] + 1 < Himg - 3)
        assert torch.equal(e["kp_1"].cpu().long(), kp1[ok])
    assert torch.equal(e["kp_2"], e["kp_1"] + 1)
    assert rel_err(e["pts3d_1"], pm[0][e["kp_1"][:, 1].long().cpu(), e["kp_1"][:, 0].long().cpu()]) < 1e-5
    # a second pair with fewer keypoints, then a step on the collated batch
    cache.put("pair1", {k: (v[:max(1, v.shape[0] // 2)] if k in ("kp_1", "kp_2", "pts3d_1", "pts3d_2") else v) for k, v in e.items()
                        if not k.startswith("_")})
    rgb = torch.rand(4, 3, Himg, Wimg, device="cuda")
    batch = cache.collate(["pair0", "pair1"], rgb[:2], rgb[2:])
    assert batch["counts"].tolist() == [e["kp_1"].shape[0], max(1, e["kp_1"].shape[0] // 2)] and batch["cost_tstats"].shape == (2, 2, hw, 4)
    eng = FinetuneGD(r=4, backbone="vit_tiny_test", patch_size=14, img_size=56, variant="vggt", geometry="shared", dtype="f32",
                     lora_b_std=0.05, vit_kwargs=dict(init_values=1.0), teacher_patch=14).cuda()
    eng.configure_optimizers()
    loss, _, norm = eng.fit_step(batch)
    assert torch.isfinite(loss) and torch.isfinite(norm)
    # MASt3R side: the composite = reciprocal NNs + filters + rasterised, post-processed depth
    g = load_golden("g16b_filter_and_match")
    d1, d2, c1, c2 = (g[k].float().cuda() for k in ("desc1", "desc2", "conf1", "conf2"))
    Hm, Wm = c1.shape
    pts = torch.rand(Hm, Wm, 3, generator=gen).cuda() + torch.tensor([0.0, 0.0, 1.5], device="cuda")
    Km = torch.tensor([[60.0, 0, Wm / 2], [0, 60.0, Hm / 2], [0, 0, 1]], device="cuda")
    cost = torch.softmax(torch.randn(24, 24, generator=gen), -1).cuda()
    t = tg.extract_mast3r_targets(d1, d2, c1, c2, pts, pts + 0.01, pts, cost, cost, intrinsic=Km)
    assert torch.equal(t["kp_1"].cpu(), g["kp1"]) and torch.equal(t["kp_2"].cpu(), g["kp2"])
    rd = O.post_process_depth(O.point_cloud_to_depth(pts.reshape(-1, 3).cpu(), Km.cpu(), Wm, Hm)[0, 0].double(), kernel_size=3)
    assert float(((t["depth_1"].cpu().double() - rd).abs() > 1e-4 * (1 + rd.abs())).float().mean()) < 5e-3
    # the MASt3R temperature schedule (src/finetune_timm_mast3r.py:217-227) moves every epoch: without logits or a cost producer
    # the whole entry is rebuilt ...
    calls.clear()
    cache.get("m0", lambda: (calls.append(1), t)[1], temperature=1.0)
    cache.get("m0", lambda: (calls.append(1), t)[1], temperature=0.9)
    assert len(calls) == 2
    # ... with a cost producer only the cost maps are, and the keypoints / depth stay the cached tensors
    kp_before = cache.get("m0", temperature=0.9)["kp_1"]
    cost9 = torch.softmax(torch.randn(24, 24, generator=gen), -1).cuda()
    e9 = cache.get("m0", lambda: (calls.append(1), t)[1], temperature=0.8, cost_producer=lambda T: (cost9, cost9))
    assert len(calls) == 2 and cache.cost_refreshes == 1 and e9["kp_1"] is kp_before
    assert torch.equal(e9["cost_1"][:, :24], cost9) and e9["_temperature"] == 0.8
    # ... and with the pre-softmax score maps kept in the entry the cache re-applies the target kernel itself (G17's formula)
    keep = TeacherTargetCache(keep_logits=True)
    L_, n_ = 3, 24
    tgt = [torch.randn(2, 4, n_, n_, generator=gen).cuda() for _ in range(L_)]
    src = [torch.randn(2, 4, n_, n_, generator=gen).cuda() for _ in range(L_)]
    recip = tg.mast3r_recip_logits(tgt, src)
    full = lambda T: dict(t, cost_1=tg.mast3r_tgt_attn_map(tgt, src, T)[1], cost_2=tg.mast3r_tgt_attn_map(tgt, src, T)[0], cost_recip=recip)
    calls.clear()
    keep.get("m1", lambda: (calls.append(1), full(1.0))[1], temperature=1.0)
    e5 = keep.get("m1", lambda: (calls.append(1), full(0.5))[1], temperature=0.5)
    assert len(calls) == 1 and keep.cost_refreshes == 1
    want = O.mast3r_tgt_attn_map([x.cpu().double() for x in tgt], [x.cpu().double() for x in src], 0.5)
    assert rel_err(e5["cost_1"][:, :n_], want[1]) < 1e-5 and rel_err(e5["cost_2"][:, :n_], want[0]) < 1e-5
    fresh = ops_stats(e5)
    assert rel_err(e5["cost_tstats"], fresh) < 1e-6
    assert "cost_recip" not in cache.put("m2", full(1.0), 1.0)            # dropped unless the cache was built with keep_logits
    # a pair without surviving keypoints: the producers return None (src/finetune_timm_mast3r.py:604-607); remembered, never re-run
    calls.clear()
    assert cache.get("empty", lambda: (calls.append(1), None)[1], temperature=1.0) is None
    assert cache.get("empty", lambda: (calls.append(1), None)[1], temperature=0.7) is None and len(calls) == 1
    # ... and it STAYS in the batch as a zero-loss sample (counts = 0): the reference's step returns a constant zero for it, so the mean's
    # divisor, the image rows and every data-parallel rank's pair count are what the caller passed
    b2 = cache.collate(["pair0", "empty", "pair1"], rgb[:3], rgb[1:4])
    assert b2["counts"].tolist()[1] == 0 and b2["counts"].shape[0] == 3 and torch.equal(b2["rgb_1"], rgb[:3])
    assert torch.equal(b2["rgb_2"], rgb[1:4]) and bool((b2["kp_1"][1] == -1).all())
    loss3, terms3 = eng.training_step(b2)
    both = cache.collate(["pair0", "pair1"], rgb[[0, 2]], rgb[[1, 3]])
    loss2, _ = eng.training_step(both)
    assert bool(torch.isfinite(loss3)) and abs(loss3.item() * 3 - loss2.item() * 2) < 1e-3 * abs(loss2.item() * 2)
    assert cache.collate(["empty"], rgb[:1], rgb[1:2]) is None


def test_vggt_teacher_runner_with_a_fake_teacher():
    """teacher_runner.VGGTTeacherRunner end to end on the GPU around a FAKE teacher laid out like vggt.models.vggt.VGGT
    (aggregator.global_blocks[i].attn with qkv / q_norm / k_norm / scale, attn_indices, temperature; camera / depth / point / track
    heads): q and k are captured by hooks, the blocks' `return_attn` maps are never formed, and the targets equal the composite
    called on the tensors directly.  (The hooks against the REFERENCE's aggregator: tests/test_teacher_runner_ref.py, CPU.)"""
    import torch.nn as nn
    from gd_amd import teacher_glue as tg
    from gd_amd.teacher_runner import VGGTTeacherRunner
    torch.manual_seed(3)
    Himg = Wimg = 56
    n, prefix, D, Hh = 16, 5, 128, 2
    g15 = load_golden("g15_teacher_glue")
    E = torch.stack([g15["E1"], g15["E2"]]).cuda()
    K = torch.stack([g15["K1"], g15["K2"]]).cuda()
    K[:, :2, 2] = 28.0

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.num_heads, self.head_dim, self.scale = Hh, D // Hh, (D // Hh) ** -0.5
            self.qkv, self.proj = nn.Linear(D, 3 * D), nn.Linear(D, D)
            self.q_norm, self.k_norm, self.rope = nn.LayerNorm(D // Hh), nn.LayerNorm(D // Hh), None
            self.formed_maps = 0

        def forward(self, x, pos=None, return_attn=False, temperature=1.0):
            B, N, C = x.shape
            q, k, v = self.qkv(x).reshape(B, N, 3, Hh, C // Hh).permute(2, 0, 3, 1, 4).unbind(0)
            q, k = self.q_norm(q), self.k_norm(k)
            out = self.proj(torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, C))
            if return_attn:
                self.formed_maps += 1
                return out, torch.zeros(2 * B, Hh, n, n, device=x.device)
            return out

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.attn = Attn()

        def forward(self, x, pos=None, return_attn=False, temperature=1.0):
            if return_attn:
                y, a = self.attn(x, pos=pos, return_attn=True, temperature=temperature)
                return x + y, a
            return x + self.attn(x, pos=pos)

    class Agg(nn.Module):
        def __init__(self):
            super().__init__()
            self.global_blocks = nn.ModuleList([Blk() for _ in range(3)])
            self.attn_indices, self.aa_block_size, self.temperature = [0, 2], 1, 0.9
            self.embed = nn.Linear(3, D)

        def forward(self, images):
            B, S = images.shape[:2]
            tok = self.embed(torch.nn.functional.adaptive_avg_pool2d(images.flatten(0, 1), (4, 4)).flatten(2).transpose(1, 2))   # [B*S, 16, D]
            tok = torch.cat([tok.new_zeros(B * S, prefix, D), tok], 1).reshape(B, S * (prefix + n), D)
            maps = []
            for i, blk in enumerate(self.global_blocks):
                tok, a = blk(tok, pos=None, return_attn=True, temperature=self.temperature)
                if i in self.attn_indices:
                    maps.append(a)
            return [tok], prefix, torch.mean(torch.stack(maps), 0)

    depth = (1.0 + 2 * torch.rand(1, 2, Himg, Wimg, 1)).cuda()
    conf = (1 + torch.rand(1, 2, Himg, Wimg)).cuda()

    class Fake(nn.Module):
        def __init__(self):
            super().__init__()
            self.aggregator = Agg()
            self.camera_head = lambda toks: [torch.zeros(1, 2, 9, device="cuda")]
            self.depth_head = lambda toks, img, ps: (depth, conf)
            self.point_head = lambda toks, img, ps: (None, conf)
            self.track_head = lambda toks, img, ps, query_points: ([torch.stack([query_points[0], query_points[0] + 1])[None]], None, None)
    teacher = Fake().cuda().eval()
    runner = VGGTTeacherRunner(teacher, dtype=torch.float32, prefix=prefix, pose_decoder=lambda pe, hw: (E[None], K[None]))
    img = torch.rand(1, 2, 3, Himg, Wimg, device="cuda")
    t = runner.targets(img, num_keypoints=4000, min_distance=3)
    assert sum(b.attn.formed_maps for b in teacher.aggregator.global_blocks) == 1       # only the unselected block built its maps
    # the same targets from the tensors directly
    from gd_amd.teacher_runner import QKCapture
    sel = [teacher.aggregator.global_blocks[i].attn for i in (0, 2)]
    with torch.no_grad(), QKCapture(sel) as cap:
        teacher.aggregator(img)
    want = tg.extract_vggt_targets(cap.pairs(), depth[0], conf[0], E, K, track_fn=lambda kp: kp + 1, scale=sel[0].scale, temperature=0.9,
                                   prefix=prefix, num_keypoints=4000, min_distance=3)
    for k in ("cost_1", "cost_2", "kp_1", "kp_2", "pts3d_1", "depth_2", "mask_1"):
        assert torch.equal(t[k], want[k]), k
    ref = sum(O.cross_view_attention_maps(q.cpu().double(), k.cpu().double(), sel[0].scale, 0.9, prefix) for q, k in cap.pairs()) / 2
    assert rel_err(t["cost_1"], ref[0]) < 1e-4
