"""Device-side teacher -> target glue: the reference's functions of the same names, without the
numpy / CPU round-trips of `extract_vggt_features`, `sample_keypoints`, `filter_and_match_keypoints`
(src/finetune_timm_vggt.py:357-449, src/finetune_timm_mast3r.py:345-469).  Heavy parts are HIP kernels over the C ABI;
torch is used only for the final dynamic-size `unique` / `sort` / boolean indexing of a few hundred keypoints."""
import torch

from ._lib import check, lib, ptr, stream, GdHipError


def _f(t):
    return t.contiguous().float()


def unproject_depth_map_to_point_map(depth_map, extrinsics_cam, intrinsics_cam):
    """vggt/utils/geometry.py:12-42 on the device: depth [S,H,W(,1)], extrinsic [S,3,4], intrinsic [S,3,3] -> [S,H,W,3]."""
    d = _f(depth_map.squeeze(-1) if depth_map.dim() == 4 else depth_map)
    S, H, W = d.shape
    e, k = _f(extrinsics_cam), _f(intrinsics_cam)
    out = torch.empty(S, H, W, 3, dtype=torch.float32, device=d.device)
    check(lib().gd_unproject_depth(ptr(d), ptr(e), ptr(k), ptr(out), S, H, W, stream()), "gd_unproject_depth")
    return out


def get_coview_masks(point_map_view1, point_map_view2, intrinsic1, extrinsic1, intrinsic2, extrinsic2, image_shape):
    """utils/functions.py:462-472.  Point maps [H,W,3] (or [P,H,W,3] with [P,3,3]/[P,3,4] cameras) -> bool masks."""
    single = point_map_view1.dim() == 3
    pm1, pm2 = _f(point_map_view1), _f(point_map_view2)
    if single:
        pm1, pm2 = pm1[None], pm2[None]
    P, H, W, _ = pm1.shape
    if tuple(image_shape) != (H, W):
        raise GdHipError("image_shape must equal the point-map resolution")
    K1, E1, K2, E2 = (_f(t).reshape(P, -1) for t in (intrinsic1, extrinsic1, intrinsic2, extrinsic2))
    m1 = torch.empty(P, H, W, dtype=torch.uint8, device=pm1.device)
    m2 = torch.empty_like(m1)
    check(lib().gd_coview_masks(ptr(pm1), ptr(pm2), ptr(K1), ptr(E1), ptr(K2), ptr(E2), ptr(m1), ptr(m2), P, H, W,
                                stream()), "gd_coview_masks")
    m1, m2 = m1.bool(), m2.bool()
    return (m1[0], m2[0]) if single else (m1, m2)


def sample_keypoints_nms(mask, conf, N, min_distance, generator=None):
    """utils/functions.py:475-507: (row, col) of the NMS survivors, row-major; a random subset of N when more survive
    (the reference draws torch.randperm on the device — same distribution, different stream).  mask/conf [H,W]."""
    H, W = mask.shape
    mk = mask.contiguous().to(torch.uint8)[None]
    cf = _f(conf)[None]
    cap = H * W
    keep = torch.empty(1, H, W, dtype=torch.uint8, device=mask.device)
    idx = torch.empty(1, cap, dtype=torch.int32, device=mask.device)
    cnt = torch.empty(1, dtype=torch.int32, device=mask.device)
    check(lib().gd_nms_keypoints(ptr(mk), ptr(cf), int(min_distance), ptr(keep), ptr(idx), ptr(cnt), 1, H, W, cap,
                                 stream()), "gd_nms_keypoints")
    M = int(cnt.item())                      # the only host sync: the keypoint count sizes everything downstream
    if M == 0:
        return None
    lin = idx[0, :M].long()
    if M > N:
        lin = lin[torch.randperm(M, device=mask.device, generator=generator)[:N]]
    return torch.stack([lin // W, lin % W], 1)


def nn_argmax(queries, database, active=None, out=None):
    """argmax_j <q_i, db_j> for the active queries (dist='dot' of mast3r/fast_nn.py:11-62) -> int32 [Nq]."""
    q, db = _f(queries), _f(database)
    Nq, D = q.shape
    if out is None:
        out = torch.full((Nq,), -1, dtype=torch.int32, device=q.device)
    keys = torch.empty(Nq, dtype=torch.int64, device=q.device)
    act = active.contiguous().to(torch.uint8) if active is not None else None
    check(lib().gd_nn_argmax(ptr(q), ptr(db), ptr(act), ptr(out), ptr(keys), Nq, db.shape[0], D, stream()), "gd_nn_argmax")
    return out


def fast_reciprocal_NNs(pts1, pts2, subsample_or_initxy1=8, max_iter=10):
    """mast3r/fast_nn.py:109-188 (integer subsample, pixel_tol=0, dist='dot', ret_xy=True) with the whole iteration
    on the device: no numpy round-trip per step, the convergence masks are tensors, and the loop always runs max_iter
    rounds (converged seeds are skipped inside the kernel), so there is no host sync until the final `unique`.
    pts [H,W,D] descriptors -> (xy1 [M,2], xy2 [M,2]) int64 (x,y), unique, sorted on the view-1 index."""
    H1, W1, D = pts1.shape
    H2, W2, _ = pts2.shape
    dev = pts1.device
    p1, p2 = _f(pts1).reshape(-1, D), _f(pts2).reshape(-1, D)
    S = int(subsample_or_initxy1)
    ys, xs = torch.meshgrid(torch.arange(S // 2, H1, S, device=dev), torch.arange(S // 2, W1, S, device=dev), indexing="ij")
    xy1 = (xs.reshape(-1) + W1 * ys.reshape(-1)).to(torch.int32)     # strictly increasing: already unique
    xy2 = torch.full_like(xy1, -1)
    old1, old2 = xy1.clone(), xy2.clone()
    notyet = torch.ones_like(xy1, dtype=torch.bool)
    for it in range(max_iter):
        nn_argmax(p1[xy1.long()], p2, notyet, out=xy2)
        notyet = notyet & (old2 != xy2)
        nn_argmax(p2[xy2.clamp_min(0).long()], p1, notyet, out=xy1)
        notyet = notyet & (old1 != xy1)
        if it + 1 < max_iter:
            old2, old1 = xy2.clone(), xy1.clone()
    conv = ~notyet
    key = torch.unique(xy1[conv].long() * (H2 * W2 + 1) + xy2[conv].long())
    i1, i2 = key // (H2 * W2 + 1), key % (H2 * W2 + 1)
    return torch.stack([i1 % W1, i1 // W1], 1), torch.stack([i2 % W2, i2 // W2], 1)


def filter_kp_by_conf(kp, conf_mask):
    """utils/functions.py:199-207."""
    k = kp[0]
    valid = conf_mask[k[:, 1].round().long(), k[:, 0].round().long()]
    idx = valid.nonzero(as_tuple=False).squeeze(1)
    return kp[:, idx], idx


def filter_and_match_keypoints(desc_1, desc_2, conf_1, conf_2, subsample=16, min_conf_thr=10):
    """src/finetune_timm_mast3r.py:414-459 on the device: reciprocal NNs, 3-px border filter, union of the two
    percentile-confidence filters.  -> kp_1, kp_2 float [1,N,2] (x,y)."""
    kp1, kp2 = fast_reciprocal_NNs(desc_1, desc_2, subsample)
    H1, W1 = conf_1.shape
    H2, W2 = conf_2.shape
    ok = ((kp1[:, 0] >= 3) & (kp1[:, 0] < W1 - 3) & (kp1[:, 1] >= 3) & (kp1[:, 1] < H1 - 3)
          & (kp2[:, 0] >= 3) & (kp2[:, 0] < W2 - 3) & (kp2[:, 1] >= 3) & (kp2[:, 1] < H2 - 3))
    a, b = kp1[ok].float()[None], kp2[ok].float()[None]
    th1 = conf_1.reshape(-1).sort()[0][int(conf_1.numel() * float(min_conf_thr) * 0.01)]
    th2 = conf_2.reshape(-1).sort()[0][int(conf_2.numel() * float(min_conf_thr) * 0.01)]
    _, i1 = filter_kp_by_conf(a, conf_1 >= th1)
    _, i2 = filter_kp_by_conf(b, conf_2 >= th2)
    keep = torch.unique(torch.cat([i1, i2], 0))
    return a[:, keep], b[:, keep]


def point_cloud_to_depth(points, K, w, h, device=None):
    """utils/functions.py:218-259: points [N,3] (camera frame), K [3,3] -> depth [1,1,h,w]."""
    pts, k = _f(points)[None], _f(K).reshape(1, 9)
    depth = torch.empty(1, h, w, dtype=torch.float32, device=pts.device)
    cnt = torch.empty_like(depth)
    check(lib().gd_point_cloud_to_depth(ptr(pts), ptr(k), ptr(depth), ptr(cnt), 1, pts.shape[1], w, h, stream()),
          "gd_point_cloud_to_depth")
    return depth.view(1, 1, h, w)


def post_process_depth(depth_img, kernel_size=5, bilateral_d=3, bilateral_sigma_color=0.1, bilateral_sigma_space=1.0, guided_r=8,
                       guided_eps=1e-2):
    """utils/functions.py:262-345 on the device: depth [H,W] (or [1,H,W] / [1,1,H,W], or a batch [P,H,W]) -> filtered
    depth, `.squeeze()`d like the reference's.  kornia's filters are restated (parity unpinned, see include/gd_hip.h)."""
    d = _f(depth_img)
    batched = d.dim() == 3 and d.shape[0] > 1
    d = d.reshape(-1, d.shape[-2], d.shape[-1])
    P, H, W = d.shape
    out = torch.empty_like(d)
    ws = torch.empty(lib().gd_post_process_depth_workspace_bytes(P, H, W), dtype=torch.uint8, device=d.device)
    check(lib().gd_post_process_depth(ptr(d), ptr(out), P, H, W, int(kernel_size), int(bilateral_d), float(bilateral_sigma_color),
                                      float(bilateral_sigma_space), int(guided_r), float(guided_eps), ptr(ws), stream()),
          "gd_post_process_depth")
    return out if batched else out[0]


def cross_view_attention_maps(q, k, scale, temperature=1.0, prefix=5, out=None, weight=None, accumulate=False):
    """Head-averaged cross-view attention maps of one VGGT global block (vggt/layers/attention.py:51-85, `return_attn`,
    followed by `.mean(dim=1)` of src/finetune_timm_vggt.py:390-392) without the [2B, H, n, n] intermediate.
    q, k: [B, H, N, 64] (f32 | bf16) -> [2B, n, n] fp32, n = N/2 - prefix.  Averaging over several blocks
    (vggt/models/aggregator.py:273): call once per block with the same `out`, weight = 1 / (H * n_blocks) and
    accumulate=True from the second block on."""
    if not (q.is_cuda and k.is_cuda and q.dtype == k.dtype and q.shape == k.shape and q.dim() == 4):
        raise GdHipError("cross_view_attention_maps: q, k must be CUDA tensors [B, H, N, 64] of one dtype")
    B, H, N, d = q.shape
    q, k = q.contiguous(), k.contiguous()
    n = N // 2 - prefix
    if out is None:
        out = torch.empty(2 * B, max(n, 0), max(n, 0), dtype=torch.float32, device=q.device)
    ws = torch.empty(max(1, lib().gd_cross_view_attn_workspace_bytes(B, H, N, prefix)), dtype=torch.uint8, device=q.device)
    dt = 1 if q.dtype == torch.bfloat16 else 0
    if q.dtype not in (torch.float32, torch.bfloat16):
        raise GdHipError("cross_view_attention_maps: dtype must be float32 or bfloat16")
    w = (1.0 / H) if weight is None else float(weight)
    check(lib().gd_cross_view_attn(ptr(q), ptr(k), ptr(out), B, H, N, int(prefix), d, float(scale), float(temperature), w,
                                   1 if accumulate else 0, dt, ptr(ws), stream()), "gd_cross_view_attn")
    return out


def _mast3r_target(recip, temperature):
    L, B, N1, N2 = recip.shape
    out = torch.empty(B, N1, N2, dtype=torch.float32, device=recip.device)
    ws = torch.empty(L * B * N1, dtype=torch.float32, device=recip.device)
    check(lib().gd_mast3r_attn_target(ptr(recip), L, B, N1, N2, float(temperature), ptr(out), ptr(ws), stream()),
          "gd_mast3r_attn_target")
    return out


def mast3r_recip_logits(tgt_camaps, src_camaps):
    """The temperature-INDEPENDENT part of `tgt_attn_map` (dust3r/dust3r/model.py:346-350): per decoder layer the head mean of
    the raw cross-attention scores, averaged with the transposed other direction.  -> [L, B, N1, N2] fp32; with the symmetrised
    two-pair batch of the trainer B = 2 and `_mast3r_target(recip, T)[1] / [0]` are cost_1 / cost_2 at temperature T — what
    teacher_cache.TeacherTargetCache keeps as `cost_recip` so that the annealed temperature never re-runs the teacher."""
    return torch.stack([(t.float().mean(dim=1) + s.float().mean(dim=1).transpose(-1, -2)) * 0.5
                        for t, s in zip(tgt_camaps, src_camaps)]).contiguous()


def mast3r_tgt_attn_map(tgt_camaps, src_camaps, temperature=3.0):
    """`tgt_attn_map` of dust3r/dust3r/model.py:346-366 (reciprocity on) from the decoder's per-layer raw cross-attention
    score maps, tgt [B,H,N1,N2] / src [B,H,N2,N1] (CrossAttention's `attn_map`): head mean and reciprocity average in
    torch (two reductions per layer), softmax / min-fill / layer mean in one HIP pass.  -> [B, N1, N2] fp32."""
    return _mast3r_target(mast3r_recip_logits(tgt_camaps, src_camaps), temperature)


def mast3r_tgt_attn_map_from_qk(q1s, k2s, q2s, k1s, scale, temperature=3.0):
    """Same target without ever forming a per-head map: the head MEAN of raw scores is linear, so per layer
    mean_h(q_h k_h^T) * scale = scale / H * Q K^T over the concatenated heads, and the reciprocity average is
    scale / (2H) * (Q1 K2^T + K1 Q2^T) — two accumulating MFMA GEMMs.  q*/k*: per-layer lists of [B, H, N, 64]
    (after RoPE, dust3r/croco/models/blocks.py:150-172).  -> [B, N1, N2] fp32."""
    from . import ops
    L = len(q1s)
    B, H, N1, d = q1s[0].shape
    N2 = k2s[0].shape[2]
    flat = lambda t: t.permute(0, 2, 1, 3).reshape(t.shape[0], t.shape[2], H * d).contiguous()
    recip = torch.empty(L, B, N1, N2, dtype=torch.float32, device=q1s[0].device)
    a = float(scale) / (2.0 * H)
    for l in range(L):
        for b in range(B):
            ops.gemm_nt(flat(q1s[l])[b], flat(k2s[l])[b], out=recip[l, b], alpha=a)
            ops.gemm_nt(flat(k1s[l])[b], flat(q2s[l])[b], out=recip[l, b], alpha=a, accumulate=True)
    return _mast3r_target(recip, temperature)


# ------------------------------------------------------------------------------------------------------------------
# Composites: one call per image pair from the frozen teacher's raw outputs to the pair's distillation targets in the layout
# of teacher_cache.TeacherTargetCache / FinetuneGD.training_step.  The teacher networks themselves (VGGT aggregator + heads,
# MASt3R encoder / decoder / heads, the VGGT track head) stay PyTorch-ROCm inference and are passed in as tensors / callables.
# ------------------------------------------------------------------------------------------------------------------
def _gather_xy(map_hw_c, kp_xy):
    """map [H, W, C], kp [N, 2] (x, y) -> [N, C]   (`point_map[kp[..., 1].long(), kp[..., 0].long()]`, finetune_timm_vggt.py:540)"""
    return map_hw_c[kp_xy[:, 1].long(), kp_xy[:, 0].long()]


def _border_ok(kp, h, w):
    return (kp[:, 0] >= 3) & (kp[:, 0] < int(w) - 3) & (kp[:, 1] >= 3) & (kp[:, 1] < int(h) - 3)


def extract_vggt_targets(global_qk, depth_map, point_conf, extrinsic, intrinsic, track_fn, scale=64 ** -0.5, temperature=1.0,
                         prefix=5, num_keypoints=300, min_distance=5, generator=None):
    """`extract_vggt_features` + `sample_keypoints` (src/finetune_timm_vggt.py:357-449) after the teacher forward.
    global_qk : list over the selected global-attention blocks of (q, k), each [1, H, N, 64] after q/k-norm and RoPE, N = the
                two frames' tokens (the aggregator's `return_attn` blocks, vggt/models/aggregator.py:273)
    depth_map [2, H, W(,1)], point_conf [2, H, W], extrinsic [2, 3, 4], intrinsic [2, 3, 3]: the depth / camera heads' outputs
    track_fn  : kp_1 int [N, 2] (x, y) -> kp_2 [N, 2] — the teacher's track head (finetune_timm_vggt.py:439-440)
    -> dict of targets (None when the NMS leaves no keypoint: the reference then skips the step, :585-588)."""
    out = None
    nb = len(global_qk)
    for i, (q, k) in enumerate(global_qk):
        out = cross_view_attention_maps(q, k, scale, temperature, prefix, out=out, weight=1.0 / (q.shape[1] * nb), accumulate=i > 0)
    d = depth_map.squeeze(-1) if depth_map.dim() == 4 else depth_map
    Himg, Wimg = d.shape[-2:]
    pm = unproject_depth_map_to_point_map(d, extrinsic, intrinsic)                    # [2, H, W, 3]
    m1, m2 = get_coview_masks(pm[0], pm[1], intrinsic[0], extrinsic[0], intrinsic[1], extrinsic[1], (Himg, Wimg))
    rc = sample_keypoints_nms(m1, point_conf[0], num_keypoints, min_distance, generator=generator)
    if rc is None:
        return None
    kp1 = rc[:, [1, 0]].int()                                                            # (row, col) -> (x, y)
    kp2 = track_fn(kp1).int()
    ok = _border_ok(kp1, Himg, Wimg) & _border_ok(kp2, Himg, Wimg)
    kp1, kp2 = kp1[ok].float(), kp2[ok].float()
    return {"cost_1": out[0], "cost_2": out[1], "kp_1": kp1, "kp_2": kp2, "pts3d_1": _gather_xy(pm[0], kp1),
            "pts3d_2": _gather_xy(pm[1], kp2), "depth_1": _f(d[0]), "depth_2": _f(d[1]), "mask_1": m1, "mask_2": m2}


def extract_mast3r_targets(desc_1, desc_2, conf_1, conf_2, pts3d_1, pts3d_2_from_1, pts3d_2, cost_1, cost_2, intrinsic=None,
                           depth_1=None, depth_2=None, subsample=16, min_conf_thr=10, depth_kernel_size=3, cost_recip=None):
    """`extract_mast3r_features` tail + `filter_and_match_keypoints` + the depth branch of `training_step`
    (src/finetune_timm_mast3r.py:345-469, 617-633) after the teacher forward.
    desc_k [H, W, 24], conf_k [H, W], pts3d_* [H, W, 3] (view-1 frame), cost_k [hw, hw] = `tgt_attn_map` rows
    (mast3r_tgt_attn_map / mast3r_tgt_attn_map_from_qk at the trainer's current `teacher_temperature`);
    depth_k given (the batch's depth maps) or rasterised from the point clouds with `intrinsic` [3, 3] and post-processed.
    -> dict of targets (None when no keypoint survives, :604-607)."""
    kp1, kp2 = filter_and_match_keypoints(desc_1, desc_2, conf_1, conf_2, subsample=subsample, min_conf_thr=min_conf_thr)
    if kp1.shape[1] == 0:
        return None
    H, W = conf_1.shape
    if depth_1 is None:
        if intrinsic is None:
            raise GdHipError("extract_mast3r_targets: give depth_1 / depth_2 or the intrinsic matrix to rasterise them")
        depth_1 = post_process_depth(point_cloud_to_depth(pts3d_1.reshape(-1, 3), intrinsic, W, H)[0, 0], kernel_size=depth_kernel_size)
        depth_2 = post_process_depth(point_cloud_to_depth(pts3d_2.reshape(-1, 3), intrinsic, W, H)[0, 0], kernel_size=depth_kernel_size)
    k1, k2 = kp1[0], kp2[0]
    out = {"cost_1": _f(cost_1), "cost_2": _f(cost_2), "kp_1": k1, "kp_2": k2, "pts3d_1": _gather_xy(_f(pts3d_1), k1),
           "pts3d_2": _gather_xy(_f(pts3d_2_from_1), k2), "depth_1": _f(depth_1), "depth_2": _f(depth_2)}
    if cost_recip is not None:     # [L, 2, N, N] pre-softmax score maps: lets the cache follow the annealed temperature on its own
        out["cost_recip"] = _f(cost_recip)
    return out
