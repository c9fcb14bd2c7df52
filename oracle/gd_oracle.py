"""CPU oracle for the geometric-distillation hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain-PyTorch (CPU, fp32 or fp64) restatement of the reference's
arithmetic for SURVEY.md section 8(a).  It is the checker for the HIP kernels: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  The
product package (3d-vlm-gd_amd/) never imports anything from oracle/.

Pinning: every function below is compared against the reference's own code (imported
from /root/reference in the build container by tools/make_golden.py) and against the
fixtures that script commits under tests/golden/.  One boundary is "parity unpinned":
timm==0.9.10's VisionTransformer (requirements.txt:21) is not vendored in the
reference; `vit_forward` restates its published forward and is pinned
instead on the reference's in-tree DINOv2 ViT (vggt/layers/vision_transformer.py),
which shares the block arithmetic.

All citations are relative to /root/reference.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------
# elementwise / sampling helpers
# ----------------------------------------------------------------------------------


def sigmoid_t(x, temp=1.0):
    """utils/functions.py:24-33 — temperature sigmoid, exponent clamped to +-50."""
    e = torch.clamp(-x / temp, min=-50.0, max=50.0)
    return 1.0 / (1.0 + torch.exp(e))


def keypoint_grid_coords(pts, h, w, patch_size, stride):
    """utils/functions.py:56-65 — pixel (x,y) -> grid_sample coords in [-1,1] with
    patch-centre alignment.  pts [B,N,2]; h,w = image size in pixels."""
    last_h = ((h - patch_size) // stride) * stride + (patch_size / 2)
    last_w = ((w - patch_size) // stride) * stride + (patch_size / 2)
    ah = 2 / (last_h - (patch_size / 2))
    aw = 2 / (last_w - (patch_size / 2))
    bh = 1 - last_h * 2 / (last_h - (patch_size / 2))
    bw = 1 - last_w * 2 / (last_w - (patch_size / 2))
    a = torch.tensor([[aw, ah]], dtype=torch.float32).to(pts.device)
    b = torch.tensor([[bw, bh]], dtype=torch.float32).to(pts.device)
    return a * pts.float() + b


def bilinear_sample_border(desc, g):
    """F.grid_sample(align_corners=True, padding_mode='border', bilinear) written out.
    desc [B,C,H,W]; g [B,N,2] in [-1,1] (x,y) -> [B,C,N]."""
    B, C, H, W = desc.shape
    x = (g[..., 0] + 1) * 0.5 * (W - 1)
    y = (g[..., 1] + 1) * 0.5 * (H - 1)
    x = x.clamp(0, W - 1)
    y = y.clamp(0, H - 1)
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    wx1 = x - x0
    wy1 = y - y0
    x0 = x0.long()
    y0 = y0.long()
    x1 = (x0 + 1).clamp(max=W - 1)
    y1 = (y0 + 1).clamp(max=H - 1)
    flat = desc.reshape(B, C, H * W)

    def take(yy, xx):
        idx = (yy * W + xx).unsqueeze(1).expand(B, C, -1)
        return flat.gather(2, idx)

    w00 = ((1 - wx1) * (1 - wy1)).unsqueeze(1)
    w01 = (wx1 * (1 - wy1)).unsqueeze(1)
    w10 = ((1 - wx1) * wy1).unsqueeze(1)
    w11 = (wx1 * wy1).unsqueeze(1)
    return take(y0, x0) * w00 + take(y0, x1) * w01 + take(y1, x0) * w10 + take(y1, x1) * w11


def interpolate_features(desc, pts, h, w, normalize=True, patch_size=14, stride=14):
    """utils/functions.py:55-76.  desc [B,C,ph,pw], pts [B,N,2] px -> [B,C,N]."""
    g = keypoint_grid_coords(pts, h, w, patch_size, stride).to(desc.dtype)
    out = bilinear_sample_border(desc, g)
    return F.normalize(out, dim=1) if normalize else out


def extract_kp_depth(depth_map, kp, window_size=3):
    """utils/functions.py:348-372 — 3x3 replicate-padded mean of depth at integer kps.
    depth_map [H,W]; kp [B,N,2] (x,y) -> [B,N]."""
    H, W = depth_map.shape[-2:]
    half = window_size // 2
    padded = F.pad(depth_map.reshape(1, 1, H, W), (half,) * 4, mode="replicate")
    means = F.avg_pool2d(padded, window_size, stride=1).reshape(1, H * W)
    idx = (kp[..., 1] * W + kp[..., 0]).long()
    return means.expand(kp.shape[0], -1).gather(1, idx)


def patch_mask_from_kp(kp_xy, H, W, patch_size):
    """utils/functions.py:375-399 — bool [ (H//P)*(W//P) ] of patches holding a keypoint."""
    ph, pw = H // patch_size, W // patch_size
    ok = (kp_xy[:, 0] >= 0) & (kp_xy[:, 0] < W) & (kp_xy[:, 1] >= 0) & (kp_xy[:, 1] < H)
    mask = torch.zeros(ph * pw, dtype=torch.bool, device=kp_xy.device)
    k = kp_xy[ok]
    if k.shape[0]:
        mask[(k[:, 1].long() // patch_size) * pw + (k[:, 0].long() // patch_size)] = True
    return mask


# ----------------------------------------------------------------------------------
# dense cost-volume KL (a14-a16)
# ----------------------------------------------------------------------------------


def masked_patch_cost(cost, row_mask, eps=1e-8, use_softmax=False, temperature=1.0):
    """utils/functions.py:402-422 with mask_patch_2=None: zero masked-out rows, then
    softmax(/T) or divide by clamp_min(row_sum, eps)."""
    c = cost * row_mask.to(cost.dtype).reshape(1, -1, 1)
    if use_softmax:
        return torch.softmax(c / temperature, dim=-1)
    return c / c.sum(-1, keepdim=True).clamp_min(eps)


def kl_divergence_map(t, p, eps=1e-8):
    """utils/losses.py:5-15."""
    t = t.clamp_min(eps)
    p = p.clamp_min(eps)
    return (t * torch.log(t / p)).sum(-1).mean()


def cost_volume_kl(f1, f2, t1, t2, m1, m2, variant):
    """calculate_cost_loss: src/finetune_timm_vggt.py:509-533 (variant='vggt') and
    src/finetune_timm_mast3r.py:522-540 (variant='mast3r').
    f1,f2 [1,hw,C] raw student features; t1,t2 [1,hw,hw] teacher maps; m1,m2 bool [hw]."""
    a = F.normalize(f1, p=2, dim=-1)
    b = F.normalize(f2, p=2, dim=-1)
    s12 = a @ b.transpose(-1, -2)
    s21 = b @ a.transpose(-1, -2)
    tt1 = masked_patch_cost(t1, m1)
    tt2 = masked_patch_cost(t2, m2)
    if variant == "vggt":
        p12 = masked_patch_cost(torch.softmax(s12, -1), m1)
        p21 = masked_patch_cost(torch.softmax(s21, -1), m2)
    elif variant == "mast3r":
        p12 = masked_patch_cost(s12, m1, use_softmax=True)
        p21 = masked_patch_cost(s21, m2, use_softmax=True)
    else:
        raise ValueError(variant)
    return 0.5 * (kl_divergence_map(tt1, p12) + kl_divergence_map(tt2, p21))


# ----------------------------------------------------------------------------------
# sparse correspondence smooth-AP (a8)
# ----------------------------------------------------------------------------------


def smooth_ap_loss(desc1, desc2, pts3d_1, pts3d_2, variant, thres3d_neg=0.1, temp=0.01):
    """calculate_matching_loss on L2-normalised descriptors.
    src/finetune_timm_vggt.py:543-572 (variant='vggt': ap1 uses sigmoid(1-pos)),
    src/finetune_timm_mast3r.py:560-589 (variant='mast3r': ap1 uses sigmoid(pos-1)).
    desc [1,N,C]; pts3d [1,N,3]."""
    N = desc1.shape[1]
    eye = torch.eye(N, dtype=torch.bool, device=desc1.device).unsqueeze(0)
    neg = ((torch.cdist(pts3d_1, pts3d_2) > thres3d_neg) & ~eye)[0].to(desc1.dtype)
    sim = (desc1 @ desc2.transpose(-1, -2))[0]
    pos = sim.diagonal()
    if variant == "vggt":
        rpos1 = sigmoid_t(1.0 - pos, temp) + 1
    else:
        rpos1 = sigmoid_t(pos - 1.0, temp) + 1
    ap1 = rpos1 / (rpos1 + (sigmoid_t(sim - 1.0, temp) * neg).sum(-1))
    rpos2 = sigmoid_t(1.0 - pos, temp) + 1
    ap2 = rpos2 / (rpos2 + (sigmoid_t(sim - pos[:, None], temp) * neg).sum(-1))
    return torch.mean(1.0 - (ap1 + ap2) / 2)


def smooth_ap_loss_me(desc1, desc2, pts3d_1, pts3d_2, thres3d_pos=5e-3, thres3d_neg=0.1, temp=0.01):
    """ME variant, src/finetune_timm_me.py:191-220: positives are every (i,j) with
    3-D distance < thres3d_pos (dynamic count), negatives distance > thres3d_neg."""
    d = torch.cdist(pts3d_1, pts3d_2)[0]
    sim = (desc1 @ desc2.transpose(-1, -2))[0]
    negm = (d > thres3d_neg).to(sim.dtype)
    ii, jj = torch.nonzero(d < thres3d_pos, as_tuple=True)
    pos = sim[ii, jj]
    rows = sim[ii]
    nrows = negm[ii]
    rpos = sigmoid_t(pos - 1.0, temp) + 1
    ap1 = rpos / (rpos + (sigmoid_t(rows - 1.0, temp) * nrows).sum(-1))
    rpos2 = sigmoid_t(1.0 - pos, temp) + 1
    ap2 = rpos2 / (rpos2 + (sigmoid_t(rows - pos[:, None], temp) * nrows).sum(-1))
    return torch.mean(1.0 - (ap1 + ap2) / 2)


# ----------------------------------------------------------------------------------
# relative-depth head and ranking loss (a10-a12)
# ----------------------------------------------------------------------------------


def depth_head(x, hp):
    """DepthAwareFeatureFusion.forward with depths=None (utils/model.py:101-127):
    tanh(Linear(128->1)(GELU(LayerNorm(Linear(D->128)(x))))).  hp: dict of tensors
    'w1'[128,D] 'b1'[128] 'ln_w'[128] 'ln_b'[128] 'w2'[1,128] 'b2'[1]."""
    h = F.linear(x, hp["w1"], hp["b1"])
    h = F.layer_norm(h, (h.shape[-1],), hp["ln_w"], hp["ln_b"], 1e-5)
    h = F.gelu(h)
    return torch.tanh(F.linear(h, hp["w2"], hp["b2"])).squeeze(-1)


def pairwise_ranking_loss(hp, feats, depths, depth_threshold=0.05):
    """utils/losses.py:18-41 — head(f_j - f_i) against sign(d_j - d_i) over pairs with
    |d_j - d_i| > threshold; log(1+exp(-alpha*s)) mean.  feats [1,N,D], depths [1,N]."""
    f = feats[0]
    d = depths[0]
    diff = f.unsqueeze(0) - f.unsqueeze(1)  # [i, j, :] = f_j - f_i
    s = depth_head(diff, hp)
    dd = d.unsqueeze(0) - d.unsqueeze(1)  # d_j - d_i
    alpha = torch.sign(dd)
    valid = dd.abs() > depth_threshold
    if not bool(valid.any()):
        return torch.zeros((), dtype=feats.dtype, device=feats.device)
    per = torch.log(1.0 + torch.exp(-alpha * s))
    return per[valid].mean()


def depth_losses(hp, kp_feat_1, kp_feat_2, kp_depth_1, kp_depth_2, depth_threshold=0.05, aux=None):
    """calculate_depth_loss tail (src/finetune_timm_vggt.py:472-485,
    src/finetune_timm_mast3r.py:487-501): L1(head(f1-f2), tanh(d1-d2)) and the mean of
    the two intra-view ranking losses.  aux (dict, optional) receives the L1 residuals pred - target per keypoint: |.| has a kink
    at zero, and a keypoint whose residual is smaller than an implementation's feature noise takes either sign there."""
    pred = depth_head(kp_feat_1 - kp_feat_2, hp)
    if aux is not None:
        aux["l1_residual"] = (pred - torch.tanh(kp_depth_1 - kp_depth_2)).detach().reshape(-1)
    l1 = (pred - torch.tanh(kp_depth_1 - kp_depth_2)).abs().mean()
    r = 0.5 * (pairwise_ranking_loss(hp, kp_feat_1, kp_depth_1, depth_threshold)
               + pairwise_ranking_loss(hp, kp_feat_2, kp_depth_2, depth_threshold))
    return l1, r


# ----------------------------------------------------------------------------------
# student ViT (a1-a3): timm/DINOv2-style pre-LN transformer with LoRA(q,v) + adapters
# ----------------------------------------------------------------------------------


def resample_pos_embed(pos_embed, gh, gw, num_prefix, mode="dinov2", interpolate_offset=0.1):
    """Learned abs-pos-embed resampled to a gh x gw token grid.
    mode='dinov2': vggt/layers/vision_transformer.py:181-213 (bicubic, scale_factor kludge
    with +0.1 offset, no antialias).  mode='timm': timm 0.9.10 resample_abs_pos_embed
    (bicubic, antialias=True, explicit size) — unpinned (timm not vendored)."""
    pe = pos_embed.float()
    prefix, grid = pe[:, :num_prefix], pe[:, num_prefix:]
    n = grid.shape[1]
    m = int(math.sqrt(n))
    assert m * m == n
    dim = pe.shape[-1]
    if m == gh and m == gw:
        return pe
    g = grid.reshape(1, m, m, dim).permute(0, 3, 1, 2)
    if mode == "dinov2":
        if interpolate_offset:
            g = F.interpolate(g, mode="bicubic", antialias=False,
                              scale_factor=(float(gh + interpolate_offset) / m, float(gw + interpolate_offset) / m))
        else:
            g = F.interpolate(g, mode="bicubic", antialias=False, size=(gh, gw))
    else:
        g = F.interpolate(g, mode="bicubic", antialias=True, size=(gh, gw), align_corners=False)
    assert g.shape[-2:] == (gh, gw)
    g = g.permute(0, 2, 3, 1).reshape(1, gh * gw, dim)
    return torch.cat([prefix, g], dim=1)


def lora_qkv(x, w, b, lora):
    """utils/model.py:57-71 — qkv = Wx+b; q-slice += B_q A_q x; v-slice += B_v A_v x."""
    qkv = F.linear(x, w, b)
    if lora is not None:
        D = x.shape[-1]
        dq = F.linear(F.linear(x, lora["a_q"]), lora["b_q"])
        dv = F.linear(F.linear(x, lora["a_v"]), lora["b_v"])
        qkv = torch.cat([qkv[..., :D] + dq, qkv[..., D:2 * D], qkv[..., 2 * D:] + dv], dim=-1)
    return qkv


def adapter(x, ad):
    """utils/model.py:7-25 — BlockWithAdapter: out + up(relu(down(out)))."""
    return x + F.linear(F.relu(F.linear(x, ad["down"])), ad["up"])


def vit_block(x, p, i, cfg, lora=None, ad=None):
    """One pre-LN block (vggt/layers/block.py:81-134 eval path):
    x += ls1(proj(sdpa(qkv(norm1 x)))) ; x += ls2(fc2(gelu(fc1(norm2 x)))) ; [adapter]."""
    pre = f"blocks.{i}."
    B, N, D = x.shape
    h = cfg["heads"]
    d = D // h
    eps = cfg["ln_eps"]
    y = F.layer_norm(x, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps)
    qkv = lora_qkv(y, p[pre + "attn.qkv.weight"], p.get(pre + "attn.qkv.bias"), lora)
    q, k, v = qkv.reshape(B, N, 3, h, d).permute(2, 0, 3, 1, 4).unbind(0)
    if cfg.get("attention_sdpa"):
        # the same softmax(q k^T / sqrt d) v through torch's fused CPU kernel — the call the student's timm blocks make themselves (timm
        # Attention.forward with fused_attn: F.scaled_dot_product_attention; the vendored vggt/layers/attention.py:33,98 carries the same switch).
        # No [h, N, N] matrix is materialised, forward or backward: at the reference geometry's 4 801 / 6 401 tokens it is 8 x faster on the host
        # than the checkpointed row blocks below (0.74 s against 6.25 s per layer and image on 8 cores) and equal to them to 5e-7
        # (tests/test_oracle_attention_modes.py).  fp32 oracles only.
        a = F.scaled_dot_product_attention(q, k, v)
    elif cfg.get("attention_chunk"):
        # the same softmax(q k^T / sqrt d) v, evaluated for `attention_chunk` query rows at a time under gradient checkpointing: at the reference
        # geometry's 6 401 tokens the [h, N, N] score and probability matrices of every trainable block (3.9 GB each in fp64, four big forwards
        # per pair) do not have to stay alive for the backward.  Row blocks of a softmax are independent: identical values and gradients.
        from torch.utils.checkpoint import checkpoint

        def rows(qc, k_, v_):
            return torch.softmax((qc * d ** -0.5) @ k_.transpose(-1, -2), dim=-1) @ v_
        c = int(cfg["attention_chunk"])
        a = torch.cat([checkpoint(rows, q[:, :, i:i + c], k, v, use_reentrant=False) for i in range(0, N, c)], dim=2)
    else:
        s = (q * d ** -0.5) @ k.transpose(-1, -2)
        a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, N, D)
    a = F.linear(a, p[pre + "attn.proj.weight"], p.get(pre + "attn.proj.bias"))
    if pre + "ls1.gamma" in p:
        a = a * p[pre + "ls1.gamma"]
    x = x + a
    y = F.layer_norm(x, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)
    y = F.gelu(F.linear(y, p[pre + "mlp.fc1.weight"], p.get(pre + "mlp.fc1.bias")))
    y = F.linear(y, p[pre + "mlp.fc2.weight"], p.get(pre + "mlp.fc2.bias"))
    if pre + "ls2.gamma" in p:
        y = y * p[pre + "ls2.gamma"]
    x = x + y
    if ad is not None:
        x = adapter(x, ad)
    return x


def fix_pos_enc(pos_embed, patch_size, stride_hw, npatch, w, h):
    """The resampler src/evaluate_timm.py:268-269 binds onto the model when it overrides the patch-conv stride
    (utils/functions.py:169-196, `_fix_pos_enc(patch_size, stride_hw)(self, x, w, h)`): token counts from the STRIDE, bicubic
    with the +0.1 scale-factor offset.  w / h follow the DINO call convention (`B, nc, w, h = x.shape`: w is the image's
    first spatial extent).  pos_embed [1, 1+N, D]; npatch = x.shape[1] - 1."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    class_pos, patch_pos = pos_embed[:, 0], pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    w0 = 1 + (w - patch_size) // stride_hw[1]
    h0 = 1 + (h - patch_size) // stride_hw[0]
    assert w0 * h0 == npatch, (w0, h0, npatch)
    m = int(math.sqrt(N))
    w0, h0 = w0 + 0.1, h0 + 0.1
    grid = F.interpolate(patch_pos.reshape(1, m, m, dim).permute(0, 3, 1, 2),
                         scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic", align_corners=False,
                         recompute_scale_factor=False)
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    return torch.cat((class_pos.unsqueeze(0), grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)


def vit_tokens(img, p, cfg):
    """patch-embed conv (PxP stride P) + cls + resampled pos-embed (+ norm_pre when
    cfg['pre_norm']).  img [B,3,H,W] already normalised.  cfg['patch_stride'] = (sy, sx): the evaluation-time stride
    override of src/evaluate_timm.py:262-269 (overlapping patches, position table from `fix_pos_enc`)."""
    P = cfg["patch"]
    st = cfg.get("patch_stride")
    if st is not None and tuple(st) != (P, P):
        x = F.conv2d(img, p["patch_embed.proj.weight"], p.get("patch_embed.proj.bias"), stride=tuple(st))
        B, D, gh, gw = x.shape
        x = x.flatten(2).transpose(1, 2)
        x = torch.cat([p["cls_token"].expand(B, -1, -1), x], dim=1)
        x = x + fix_pos_enc(p["pos_embed"], P, tuple(st), gh * gw, img.shape[-2], img.shape[-1]).to(x.dtype)
        if cfg.get("pre_norm", False):
            x = F.layer_norm(x, (D,), p["norm_pre.weight"], p["norm_pre.bias"], cfg["ln_eps"])
        return x
    x = F.conv2d(img, p["patch_embed.proj.weight"], p.get("patch_embed.proj.bias"), stride=P)
    B, D, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([p["cls_token"].expand(B, -1, -1), x], dim=1)
    x = x + resample_pos_embed(p["pos_embed"], gh, gw, 1, cfg.get("pos_interp", "dinov2")).to(x.dtype)
    if cfg.get("pre_norm", False):
        x = F.layer_norm(x, (D,), p["norm_pre.weight"], p["norm_pre.bias"], cfg["ln_eps"])
    return x


def vit_forward(img, p, cfg, trainable=None, taps=()):
    """Returns (list of tap outputs [B,Nt,D] in `taps` order, x after the last block).
    trainable: {'lora': {blk: {...}}, 'adapter': {blk: {...}}} or None."""
    x = vit_tokens(img, p, cfg)
    outs = {}
    for i in range(cfg["depth"]):
        lo = trainable["lora"].get(i) if trainable else None
        ad = trainable["adapter"].get(i) if trainable else None
        x = vit_block(x, p, i, cfg, lo, ad)
        if i in taps:
            outs[i] = x
    return [outs[i] for i in taps], x


def final_norm(x, p, cfg):
    return F.layer_norm(x, (x.shape[-1],), p["norm.weight"], p["norm.bias"], cfg["ln_eps"])


def normalize_image(img, mean, std):
    """a0: timm Normalize (src/finetune_timm_mast3r.py:103-104,153)."""
    m = torch.tensor(mean, dtype=img.dtype, device=img.device).reshape(1, 3, 1, 1)
    s = torch.tensor(std, dtype=img.dtype, device=img.device).reshape(1, 3, 1, 1)
    return (img - m) / s


def resize_bilinear(img, size):
    """a0: torchvision 0.16.2 tensor resize = bilinear, align_corners=False, no antialias."""
    if tuple(img.shape[-2:]) == tuple(size):
        return img
    return F.interpolate(img, size=tuple(size), mode="bilinear", align_corners=False, antialias=False)


# ----------------------------------------------------------------------------------
# feature extractors (a4-a6) and the full loss / optimiser step (a21)
# ----------------------------------------------------------------------------------


def keypoint_geometry(h, w, cfg):
    """Token grid used for keypoint features.  'reference': target_res/downsample_factor
    (src/finetune_timm_vggt.py:265-270); 'shared': the cost grid itself."""
    if cfg.get("geometry", "reference") == "shared":
        return h // cfg["teacher_patch"], w // cfg["teacher_patch"]
    tr, ds = cfg.get("target_res", 640), cfg.get("downsample_factor", 8)
    if h > w:
        tgt = (tr, int(w * tr / h))
    else:
        tgt = (int(h * tr / w), tr)
    return tgt[0] // ds, tgt[1] // ds


def student_features(img, kp, p, cfg, trainable, refine):
    """One image through the three extractors of the reference step.
    img [1,3,h,w] in [0,1]; kp [1,N,2] px (x,y) in the (h,w) frame.
    Returns kp_feat [1,N,D] (get_intermediate_feature), desc [1,N,D] unit (get_feature),
    cost_feat [1,hw,D] (get_feature_cost).
    src/finetune_timm_vggt.py:256-355, src/finetune_timm_mast3r.py:242-342."""
    P = cfg["patch"]
    h, w = img.shape[-2:]
    gh, gw = keypoint_geometry(h, w, cfg)
    big = normalize_image(resize_bilinear(img, (gh * P, gw * P)), cfg["mean"], cfg["std"])
    sc = torch.tensor([(gw * P) / w, (gh * P) / h], dtype=torch.float32)
    pts = kp * sc
    taps, x = vit_forward(big, p, cfg, trainable, taps=(4, 5, 6, 7))

    def grid(t):
        return t[:, 1:].reshape(1, gh, gw, -1).permute(0, 3, 1, 2)

    samp = [interpolate_features(grid(final_norm(t, p, cfg)), pts, gh * P, gw * P, False, P, P) for t in taps]
    kp_feat = torch.stack(samp, 0).mean(0).permute(0, 2, 1)
    fmap = F.conv2d(grid(final_norm(x, p, cfg)), refine["weight"], refine["bias"], padding=1)
    # the ME trainer calls interpolate_features(feature, pts, h=patch_h * 14, w=patch_w * 14) with the default
    # patch_size = stride = 14 (src/finetune_timm_me.py:155); the other two pass the model's patch size
    Pi = 14 if cfg["variant"] == "me" else P
    desc = F.normalize(interpolate_features(fmap, pts, gh * Pi, gw * Pi, False, Pi, Pi).permute(0, 2, 1), dim=-1)

    tp = cfg["teacher_patch"]
    ch, cw = h // tp, w // tp
    small = normalize_image(resize_bilinear(img, (ch * P, cw * P)), cfg["mean"], cfg["std"])
    if cfg["variant"] == "vggt":
        ctaps = (7,)
    else:
        ctaps = (4, 5, 6, 7)
    if (ch, cw) == (gh, gw):
        ct = [taps[(4, 5, 6, 7).index(i)] for i in ctaps]
    else:
        ct, _ = vit_forward(small, p, cfg, trainable, taps=ctaps)
    cost_feat = torch.stack([t[:, 1:] for t in ct], 0).mean(0)
    return kp_feat, desc, cost_feat


def pair_losses(batch, p, cfg, trainable, refine, hp, aux=None):
    """Loss terms of one image pair = training_step body after the teacher
    (src/finetune_timm_vggt.py:599-616, src/finetune_timm_mast3r.py:635-653)."""
    f1 = student_features(batch["rgb_1"], batch["kp_1"], p, cfg, trainable, refine)
    f2 = student_features(batch["rgb_2"], batch["kp_2"], p, cfg, trainable, refine)
    d1 = extract_kp_depth(batch["depth_1"], batch["kp_1"])
    d2 = extract_kp_depth(batch["depth_2"], batch["kp_2"])
    depth_l1, intra = depth_losses(hp, f1[0], f2[0], d1, d2, aux=aux)
    h, w = batch["rgb_1"].shape[-2:]
    if cfg["variant"] == "vggt":
        m1, m2 = batch["mask_patch_1"], batch["mask_patch_2"]
    else:
        m1 = patch_mask_from_kp(batch["kp_1"][0], h, w, cfg["patch"])
        m2 = patch_mask_from_kp(batch["kp_2"][0], h, w, cfg["patch"])
    kl = cost_volume_kl(f1[2], f2[2], batch["cost_1"], batch["cost_2"], m1, m2, cfg["variant"])
    ap = smooth_ap_loss(f1[1], f2[1], batch["pts3d_1"], batch["pts3d_2"], cfg["variant"])
    return {"ap": ap, "depth": depth_l1, "intra": intra, "kl": kl}


def me_pair_loss(batch, p, cfg, trainable, refine, thres3d_pos=5e-3, thres3d_neg=0.1):
    """FinetuneTIMM.training_step (src/finetune_timm_me.py:191-220): get_feature on both views, then the smooth-AP loss
    with dynamic positives (every keypoint pair closer than thres3d_pos in 3-D)."""
    d1 = student_features(batch["rgb_1"], batch["kp_1"], p, cfg, trainable, refine)[1]
    d2 = student_features(batch["rgb_2"], batch["kp_2"], p, cfg, trainable, refine)[1]
    return smooth_ap_loss_me(d1, d2, batch["pts3d_1"], batch["pts3d_2"], thres3d_pos, thres3d_neg)


def total_loss(terms, weights):
    return (weights["ap"] * terms["ap"] + weights["depth"] * terms["depth"]
            + weights["intra"] * terms["intra"] + weights["kl"] * terms["kl"])


def clip_and_adamw(params, grads, state, step, lr=1e-5, wd=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0):
    """Lightning gradient_clip_val=1.0 (global L2 norm, src/main.py:153) followed by
    torch.optim.AdamW(lr=1e-5, weight_decay=1e-4) (src/finetune_timm_vggt.py:642-648).
    In place on `params`; `state` = list of (exp_avg, exp_avg_sq); `step` 1-based."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for q, g, (m, v) in zip(params, grads, state):
        g = g * coef
        q.mul_(1 - lr * wd)
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1 = 1 - betas[0] ** step
        bc2 = 1 - betas[1] ** step
        q.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)
    return total


# ----------------------------------------------------------------------------------
# teacher-side native op: 2-D RoPE
# ----------------------------------------------------------------------------------


def rope_2d(tokens, positions, base=100.0, fwd=1.0):
    """dust3r/croco/models/curope/curope.cpp:11-47 / kernels.cu:17-82.
    tokens [B,N,H,D] (returns a new tensor); positions int64 [B,N,2] (y,x);
    D = 4 quarters [u_Y, v_Y, u_X, v_X]; theta = pos*fwd / base^(i/Q)."""
    B, N, H, D = tokens.shape
    Q = D // 4
    inv = fwd / (base ** (torch.arange(Q, dtype=torch.float32) / Q))
    out = tokens.clone()
    for ax in range(2):
        th = positions[..., ax].float().unsqueeze(-1) * inv  # [B,N,Q]
        c, s = torch.cos(th).unsqueeze(2), torch.sin(th).unsqueeze(2)
        u = tokens[..., (2 * ax) * Q:(2 * ax + 1) * Q]
        v = tokens[..., (2 * ax + 1) * Q:(2 * ax + 2) * Q]
        out[..., (2 * ax) * Q:(2 * ax + 1) * Q] = u * c - v * s
        out[..., (2 * ax + 1) * Q:(2 * ax + 2) * Q] = v * c + u * s
    return out


# ----------------------------------------------------------------------------------
# teacher -> target glue (a18, a19)
# ----------------------------------------------------------------------------------


def unproject_depth(depth, extrinsic, intrinsic):
    """vggt/utils/geometry.py:12-110 — depth [S,H,W], extrinsic [S,3,4] (cam from world), intrinsic [S,3,3]
    -> world points [S,H,W,3]: x_cam = (u-cu) d / fu, then R^T (x_cam - t)."""
    S, H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, dtype=depth.dtype), torch.arange(W, dtype=depth.dtype), indexing="ij")
    out = []
    for s in range(S):
        fu, fv, cu, cv = intrinsic[s, 0, 0], intrinsic[s, 1, 1], intrinsic[s, 0, 2], intrinsic[s, 1, 2]
        cam = torch.stack([(u - cu) * depth[s] / fu, (v - cv) * depth[s] / fv, depth[s]], -1)
        R, t = extrinsic[s, :, :3], extrinsic[s, :, 3]
        Rinv = R.t()
        out.append(cam @ Rinv.t() + (-(Rinv @ t)))
    return torch.stack(out, 0)


def coview_masks(pm1, pm2, K1, E1, K2, E2, image_shape):
    """utils/functions.py:425-472 get_coview_masks.  NOTE the reference converts BOTH point maps with extrinsic1
    (`convert_camera_to_world(point_map_view2, extrinsic1)`, :464) — reproduced as is."""
    H, W = image_shape

    def to_world(pm, E):
        return (pm - E[:, 3].unsqueeze(0)) @ E[:, :3].t()       # torch.matmul(pm - t, R_inv) with R_inv = R.t(), as written (:452-456)

    def project(P, pts):
        flat = pts.reshape(-1, 3)
        ph = (P @ torch.cat([flat, torch.ones(flat.shape[0], 1, dtype=flat.dtype)], 1).t()).t()
        uv = ph[:, :2] / (ph[:, 2:3] + 1e-8)
        return uv.reshape(*pts.shape[:-1], 2)

    def inside(uv):
        return (uv[..., 0] >= 0) & (uv[..., 0] < W) & (uv[..., 1] >= 0) & (uv[..., 1] < H)

    w1, w2 = to_world(pm1, E1), to_world(pm2, E1)
    return inside(project(K2 @ E2, w1)), inside(project(K1 @ E1, w2))


def nms_keypoints(mask, conf, N, min_distance, perm=None):
    """utils/functions.py:475-507 sample_keypoints_nms: local maxima of the masked confidence under a
    (2*min_distance+1)^2 max-pool, as (row, col) in row-major order; when more than N survive the reference draws
    torch.randperm — here the permutation is an argument."""
    score = torch.where(mask, conf, torch.zeros_like(conf)).float()
    k = int(min_distance) * 2 + 1
    pooled = F.max_pool2d(score[None, None], kernel_size=k, stride=1, padding=k // 2)[0, 0]
    keep = ((score - pooled).abs() < 1e-6) & mask
    kps = torch.nonzero(keep, as_tuple=False)
    if kps.shape[0] == 0:
        return None
    if kps.shape[0] > N:
        kps = kps[perm[:N]]
    return kps


def point_cloud_to_depth(points, K, w, h):
    """utils/functions.py:218-259 — mean z of the points that round to each pixel."""
    pts = points[points[:, 2] > 0]
    out = torch.zeros(h * w, dtype=torch.float32)
    if pts.shape[0] == 0:
        return out.view(1, 1, h, w)
    u = torch.round(pts[:, 0] / pts[:, 2] * K[0, 0] + K[0, 2]).long()
    v = torch.round(pts[:, 1] / pts[:, 2] * K[1, 1] + K[1, 2]).long()
    ok = (u >= 0) & (u < w) & (v >= 0) & (v < h)
    idx, z = (v * w + u)[ok], pts[:, 2][ok]
    acc = torch.zeros(h * w, dtype=torch.float32).index_add_(0, idx, z.float())
    cnt = torch.zeros(h * w, dtype=torch.float32).index_add_(0, idx, torch.ones_like(z, dtype=torch.float32))
    hit = cnt > 0
    out[hit] = acc[hit] / cnt[hit]
    return out.view(1, 1, h, w)


def _reflect_pad(x, front, rear):
    return F.pad(x, [front, rear, front, rear], mode="reflect")


def _box_mean_reflect(x, k):
    """kornia.filters.box_blur(x, (k, k), border_type='reflect'): mean over a k x k window; an even k pads (k-1)//2 in
    front and k - 1 - (k-1)//2 behind (window offsets -3..+4 for k = 8)."""
    fr = (k - 1) // 2
    return F.avg_pool2d(_reflect_pad(x, fr, k - 1 - fr), k, stride=1)


def _gauss1d(k, sigma):
    x = torch.arange(k, dtype=torch.float64) - k // 2
    g = torch.exp(-x * x / (2.0 * sigma * sigma))
    return g / g.sum()


def _joint_bilateral(x, guide, k, sigma_color, sigma_space):
    """kornia.filters.joint_bilateral_blur(x, guide, (k, k), sigma_color, (s, s)), border 'reflect', colour distance 'l1'
    (single channel: |difference|): weights gaussian_space * exp(-0.5 (guide_nb - guide_c)^2 / sigma_color^2), normalised."""
    pad = k // 2
    xp, gp = _reflect_pad(x, pad, pad), _reflect_pad(guide, pad, pad)
    H, W = x.shape[-2:]
    g1 = _gauss1d(k, sigma_space).to(x.dtype)
    num = torch.zeros_like(x)
    den = torch.zeros_like(x)
    for dy in range(k):
        for dx in range(k):
            nb, gn = xp[..., dy:dy + H, dx:dx + W], gp[..., dy:dy + H, dx:dx + W]
            wgt = g1[dy] * g1[dx] * torch.exp(-0.5 / (sigma_color * sigma_color) * (gn - guide) ** 2)
            num = num + wgt * nb
            den = den + wgt
    return num / den


def post_process_depth(depth_img, kernel_size=5, bilateral_d=3, bilateral_sigma_color=0.1, bilateral_sigma_space=1.0,
                       guided_r=8, guided_eps=1e-2):
    """utils/functions.py:262-345.  SOURCE ABSENT — PARITY UNPINNED: the function calls kornia (median_blur, bilateral_blur,
    guided_blur, joint_bilateral_blur), which is not installed here and not vendored in the reference (requirements.txt pins
    kornia); those four filters are restated from kornia's published definitions (zero-padded lower median; bilateral /
    joint bilateral with reflect border, gaussian space kernel, l1 colour distance; He et al.'s guided filter on reflect-padded
    box means — called as guided_blur(guidance=depth_bilateral, input=depth_median), the argument order of the call site
    :321), the torch parts (max-pool closing, two hole-filling passes, 3-sigma outlier replacement) follow the source line
    by line.  depth_img [H, W] -> [H, W]."""
    x = depth_img.reshape(1, 1, *depth_img.shape[-2:])
    k = kernel_size
    pad = k // 2
    dil = F.max_pool2d(x, k, stride=1, padding=pad)
    ero = -F.max_pool2d(-dil, k, stride=1, padding=pad)
    for ks in (5, 7):                                     # :283-310 (the `if empty_mask.sum() > 0` guard changes nothing:
        valid = (ero >= 1e-5).to(x.dtype) if ks == 5 else (ero > 0).to(x.dtype)    # with no holes fill_mask is zero)
        ones = torch.ones(1, 1, ks, ks, dtype=x.dtype)
        cnt = F.conv2d(valid, ones, padding=ks // 2)
        val = F.conv2d(ero * valid, ones, padding=ks // 2)
        fill = ((cnt > 0).to(x.dtype) - valid).clamp(0, 1)
        ero = ero * valid + val / (cnt + 1e-8) * fill
    # median_blur: zero padding, the (k*k+1)/2-th smallest of the window
    win = F.unfold(ero, k, padding=pad).reshape(1, k * k, *x.shape[-2:])
    med = win.sort(dim=1)[0][:, (k * k - 1) // 2][:, None]
    bil = _joint_bilateral(med, med, bilateral_d, bilateral_sigma_color, bilateral_sigma_space)
    # guided filter: guidance I = bil, input p = med
    I, p_ = bil, med
    mI, mp = _box_mean_reflect(I, guided_r), _box_mean_reflect(p_, guided_r)
    var = _box_mean_reflect(I * I, guided_r) - mI * mI
    cov = _box_mean_reflect(I * p_, guided_r) - mI * mp
    a = cov / (var + guided_eps)
    b = mp - a * mI
    q = _box_mean_reflect(a, guided_r) * I + _box_mean_reflect(b, guided_r)
    kern = torch.ones(1, 1, k, k, dtype=x.dtype) / (k * k)
    lm = F.conv2d(q, kern, padding=pad)
    lv = F.conv2d(q * q, kern, padding=pad) - lm * lm
    out_mask = ((q - lm).abs() > 3.0 * torch.sqrt(lv.clamp(min=1e-6))).to(x.dtype)
    filt = q * (1 - out_mask) + med * out_mask
    fin = _joint_bilateral(filt, med, bilateral_d, bilateral_sigma_color / 2, bilateral_sigma_space)
    return fin[0, 0]


def filter_kp_by_conf(kp, conf_mask):
    """utils/functions.py:199-207."""
    k = kp[0]
    valid = conf_mask[k[:, 1].round().long(), k[:, 0].round().long()]
    idx = valid.nonzero(as_tuple=False).squeeze(1)
    return kp[:, idx], idx


def reciprocal_nns(desc1, desc2, subsample=16, max_iter=10):
    """mast3r/fast_nn.py:109-188 fast_reciprocal_NNs(dist='dot', pixel_tol=0, ret_xy=True): iterate
    seed -> NN in view 2 -> NN back in view 1 until the cycle closes, keep converged, unique, sorted on view-1 index.
    desc [H,W,D] -> (xy1 [M,2], xy2 [M,2]) int (x,y)."""
    H1, W1, D = desc1.shape
    H2, W2, _ = desc2.shape
    p1, p2 = desc1.reshape(-1, D), desc2.reshape(-1, D)
    S = subsample
    ys, xs = torch.meshgrid(torch.arange(S // 2, H1, S), torch.arange(S // 2, W1, S), indexing="ij")
    xy1 = torch.unique(xs.reshape(-1) + W1 * ys.reshape(-1))
    xy2 = torch.full_like(xy1, -1)
    old1, old2 = xy1.clone(), xy2.clone()
    notyet = torch.ones_like(xy1, dtype=torch.bool)
    niter = 0
    while bool(notyet.any()):
        xy2[notyet] = (p1[xy1[notyet]] @ p2.t()).argmax(1)
        notyet &= old2 != xy2
        if bool(notyet.any()):
            xy1[notyet] = (p2[xy2[notyet]] @ p1.t()).argmax(1)
        notyet &= old1 != xy1
        niter += 1
        if niter >= max_iter:
            break
        old2, old1 = xy2.clone(), xy1.clone()
    conv = ~notyet
    key = torch.unique(xy1[conv] * (H2 * W2 + 1) + xy2[conv])      # unique pairs, sorted on xy1 then xy2
    i1, i2 = key // (H2 * W2 + 1), key % (H2 * W2 + 1)
    return torch.stack([i1 % W1, i1 // W1], 1), torch.stack([i2 % W2, i2 // W2], 1)


def mast3r_keypoint_filter(kp1, kp2, conf1, conf2, min_conf_thr=10):
    """src/finetune_timm_mast3r.py:420-459: drop matches within 3 px of a border, then keep those whose keypoint is
    above the min_conf_thr-th percentile confidence in EITHER view (union, :456).  kp [M,2] (x,y); conf [H,W]."""
    H, W = conf1.shape
    ok = ((kp1[:, 0] >= 3) & (kp1[:, 0] < W - 3) & (kp1[:, 1] >= 3) & (kp1[:, 1] < H - 3)
          & (kp2[:, 0] >= 3) & (kp2[:, 0] < conf2.shape[1] - 3) & (kp2[:, 1] >= 3) & (kp2[:, 1] < conf2.shape[0] - 3))
    a, b = kp1[ok].float()[None], kp2[ok].float()[None]
    th1 = conf1.reshape(-1).sort()[0][int(conf1.numel() * float(min_conf_thr) * 0.01)]
    th2 = conf2.reshape(-1).sort()[0][int(conf2.numel() * float(min_conf_thr) * 0.01)]
    _, i1 = filter_kp_by_conf(a, conf1 >= th1)
    _, i2 = filter_kp_by_conf(b, conf2 >= th2)
    keep = torch.unique(torch.cat([i1, i2], 0))
    return a[:, keep], b[:, keep]


def cross_view_attention_maps(q, k, scale, temperature=1.0, prefix=5):
    """VGGT teacher -> distillation target: the cross-view attention maps of one global block,
    vggt/layers/attention.py:51-85 (`return_attn` branch: q scaled by head_dim^-0.5, the two cross blocks
    q[prefix:N/2].k[N/2+prefix:]^T and q[N/2+prefix:].k[prefix:N/2]^T, softmax(scores / temperature) over keys,
    torch.cat on dim 0), followed by the head mean of src/finetune_timm_vggt.py:390-392.
    q, k [B, H, N, d] -> [2B, n, n] with n = N/2 - prefix; rows 0..B-1 are view-1 queries, B..2B-1 view-2 queries."""
    N = q.shape[2]
    qs = q * scale
    a1 = torch.softmax(qs[..., prefix:N // 2, :] @ k[..., N // 2 + prefix:, :].transpose(-2, -1) / temperature, dim=-1)
    a2 = torch.softmax(qs[..., N // 2 + prefix:, :] @ k[..., prefix:N // 2, :].transpose(-2, -1) / temperature, dim=-1)
    return torch.cat([a1, a2], dim=0).mean(dim=1)


def mast3r_tgt_attn_map(tgt_camaps, src_camaps, temperature=3.0, reciprocity=True):
    """MASt3R teacher -> distillation target, dust3r/dust3r/model.py:346-366.  tgt_camaps / src_camaps: per decoder layer
    the CrossAttention `attn_map` = raw scaled scores [B, H, N1, N2] / [B, H, N2, N1]
    (dust3r/croco/models/blocks.py:150-172).  Head mean -> (tgt + src^T) / 2 -> softmax(. / temperature) over keys ->
    column 0 := min of the whole layer map -> mean over layers.  -> [B, N1, N2]."""
    t = [a.mean(dim=1) for a in tgt_camaps]
    if reciprocity:
        s = [a.mean(dim=1) for a in src_camaps]
        t = [((ct + cs.transpose(-1, -2)) / 2 / temperature).softmax(dim=-1) for ct, cs in zip(t, s)]
    out = []
    for m in t:
        m = m.clone()
        m[:, :, 0] = m.min()
        out.append(m)
    return torch.stack(out, dim=1).mean(dim=1)


# ----------------------------------------------------------------------------------
# input pipeline: load_and_preprocess_images after the file decode
# ----------------------------------------------------------------------------------
# vggt/utils/load_fn.py:12-146 resizes with PIL's Image.resize(..., BICUBIC) on uint8 RGB.  Pillow is a third-party
# dependency (requirements.txt pins pillow; 12.2.0 in the build image); its resampler (src/libImaging/Resample.c) is
# restated here from the published algorithm: separable convolution, horizontal pass then vertical pass with a uint8
# intermediate, support = 2 * max(scale, 1) (antialiasing when shrinking), Keys cubic a = -0.5, per-output coefficients
# normalised in double precision, quantised to 22-bit fixed point, integer accumulation from 1 << 21, >> 22, clip to [0,255].
# Pinned by fixture G20: the reference's load_and_preprocess_images itself, run on synthetic image files.
_PIL_PRECISION_BITS = 32 - 8 - 2


def _pil_bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_resample_coeffs(in_size, out_size):
    """-> (xmin int32 [out], count int32 [out], coeffs int32 [out, ksize]) of Resample.c precompute_coeffs + normalize_coeffs_8bpc."""
    import numpy as np
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin_a = np.zeros(out_size, np.int32)
    cnt_a = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_pil_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << _PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PIL_PRECISION_BITS))
        xmin_a[xx], cnt_a[xx] = xmin, xmax
    return xmin_a, cnt_a, kk


def pil_resize_bicubic_u8(img, new_w, new_h):
    """img uint8 [H, W, C] (numpy) -> uint8 [new_h, new_w, C]: PIL Image.resize((new_w, new_h), BICUBIC)."""
    import numpy as np
    img = np.asarray(img)
    H, W, C = img.shape

    def one_pass(src, in_size, out_size):          # along axis 1 of src [R, in_size, C]
        xmin, cnt, kk = pil_resample_coeffs(in_size, out_size)
        out = np.empty((src.shape[0], out_size, src.shape[2]), np.uint8)
        s64 = src.astype(np.int64)
        for xx in range(out_size):
            acc = np.full((src.shape[0], src.shape[2]), 1 << (_PIL_PRECISION_BITS - 1), np.int64)
            for x in range(cnt[xx]):
                acc += s64[:, xmin[xx] + x, :] * int(kk[xx, x])
            out[:, xx, :] = np.clip(acc >> _PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
        return out
    if new_w != W:
        img = one_pass(img, W, new_w)
    if new_h != H:
        img = one_pass(img.transpose(1, 0, 2), H, new_h).transpose(1, 0, 2)
    return np.ascontiguousarray(img)


def load_fn_geometry(width, height, mode="crop", target_size=518):
    """vggt/utils/load_fn.py:70-84: (new_width, new_height) of the resize."""
    if mode == "pad":
        if width >= height:
            new_width = target_size
            new_height = round(height * (new_width / width) / 14) * 14
        else:
            new_height = target_size
            new_width = round(width * (new_height / height) / 14) * 14
    else:
        new_width = target_size
        new_height = round(height * (new_width / width) / 14) * 14
    return new_width, new_height


def preprocess_images(images_u8, mode="crop", target_size=518):
    """vggt/utils/load_fn.py:12-146 after the decode: list of uint8 RGB arrays [H, W, 3] -> float32 [N, 3, H', W'] in [0, 1]
    (bicubic resize to width 518 / longest side 518 with sides divisible by 14, ToTensor, centre crop of the height (crop mode)
    or white padding to a square (pad mode), white padding to a common shape)."""
    import numpy as np
    if len(images_u8) == 0:
        raise ValueError("At least 1 image is required")
    if mode not in ["crop", "pad"]:
        raise ValueError("Mode must be either 'crop' or 'pad'")
    out = []
    for im in images_u8:
        h, w = im.shape[:2]
        nw, nh = load_fn_geometry(w, h, mode, target_size)
        t = torch.from_numpy(pil_resize_bicubic_u8(im, nw, nh)).permute(2, 0, 1).float() / 255.0
        if mode == "crop" and nh > target_size:
            sy = (nh - target_size) // 2
            t = t[:, sy:sy + target_size, :]
        if mode == "pad":
            hp, wp = target_size - t.shape[1], target_size - t.shape[2]
            if hp > 0 or wp > 0:
                t = torch.nn.functional.pad(t, (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2), value=1.0)
        out.append(t)
    mh, mw = max(t.shape[1] for t in out), max(t.shape[2] for t in out)
    res = []
    for t in out:
        hp, wp = mh - t.shape[1], mw - t.shape[2]
        if hp > 0 or wp > 0:
            t = torch.nn.functional.pad(t, (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2), value=1.0)
        res.append(t)
    return torch.stack(res)


# colour augmentation (data_utils/dataset_mast3r_scannetpp.py:185-207).  albumentations (ColorJitter, GaussianBlur) and the
# OpenCV routines under it are third-party and absent here: restated from their published definitions — PARITY UNPINNED.
def _cv_gray(r, g, b):
    return (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14


def color_jitter_u8(img, factors, order):
    """img uint8 [H, W, 3] (numpy); factors (brightness, contrast, saturation, hue); order: permutation of 0..3, -1 = skip.
    float32 arithmetic in the kernel's operation order (csrc/image_prep.hip jitter_op)."""
    import numpy as np
    f32 = np.float32
    r, g, b = (img[..., c].astype(np.int32) for c in range(3))
    clip_t = lambda v: np.clip(v, f32(0), f32(255)).astype(np.int32)
    clip_r = lambda v: np.clip(np.rint(v), f32(0), f32(255)).astype(np.int32)
    for op in order:
        if op < 0:
            continue
        f = f32(factors[op])
        if op == 0:
            r, g, b = (clip_t(c.astype(f32) * f) for c in (r, g, b))
        elif op == 1:
            mean = f32(np.float64(_cv_gray(r, g, b).astype(np.uint64).sum()) / np.float64(r.size))
            o = f32(mean * f32(f32(1) - f))
            r, g, b = (clip_t(f32(c.astype(f32) * f) + o) for c in (r, g, b))
        elif op == 2:
            gy = (_cv_gray(r, g, b).astype(f32) * f32(f32(1) - f)).astype(f32)
            r, g, b = (clip_r((c.astype(f32) * f).astype(f32) + gy) for c in (r, g, b))
        else:
            fr, fg, fb = r.astype(f32), g.astype(f32), b.astype(f32)
            v = np.maximum(fr, np.maximum(fg, fb))
            mn = np.minimum(fr, np.minimum(fg, fb))
            d = (v - mn).astype(f32)
            ds = np.where(d > 0, d, f32(1))
            h = np.where(v == fr, (fg - fb) / ds, np.where(v == fg, f32(2) + (fb - fr) / ds, f32(4) + (fr - fg) / ds)).astype(f32)
            h = (h * f32(30)).astype(f32)
            h = np.where(h < 0, h + f32(180), h).astype(f32)
            h = np.where(d > 0, h, f32(0))
            s = np.where(v > 0, d / np.where(v > 0, v, f32(1)), f32(0)).astype(f32)
            hi8 = np.rint(h).astype(np.int32)
            hi8 = np.where(hi8 >= 180, hi8 - 180, hi8)
            hq = np.fmod(hi8.astype(f32) + f32(f32(180) * f), f32(180)).astype(np.int32)
            hq = np.where(hq < 0, hq + 180, hq)
            s8 = (np.rint(s * f32(255)) / f32(255)).astype(f32)
            hh = (hq.astype(f32) / f32(30)).astype(f32)
            sector = hh.astype(np.int32) % 6
            fq = (hh - np.floor(hh)).astype(f32)
            one = f32(1)
            p = (v * (one - s8)).astype(f32)
            q = (v * (one - (s8 * fq).astype(f32))).astype(f32)
            t = (v * (one - (s8 * (one - fq).astype(f32)).astype(f32))).astype(f32)
            R = np.choose(sector, [v, q, p, p, t, v])
            G = np.choose(sector, [t, v, v, q, p, p])
            B = np.choose(sector, [p, p, t, v, v, q])
            r, g, b = clip_r(R), clip_r(G), clip_r(B)
    return np.stack([r, g, b], -1).astype(np.uint8)


def gaussian_blur_u8(img, k):
    """cv2.GaussianBlur(img, (k, k), 0) restated: sigma = 0.3 ((k-1)/2 - 1) + 0.8, normalised, BORDER_REFLECT_101, separable."""
    import numpy as np
    if k <= 1:
        return img.copy()
    f32 = np.float32
    sigma = f32(0.3) * (f32(k - 1) * f32(0.5) - f32(1)) + f32(0.8)
    half = k // 2
    w = np.exp(-(np.arange(-half, half + 1).astype(f32) ** 2) / (f32(2) * sigma * sigma)).astype(f32)

    def one(a, axis):
        a = np.moveaxis(a, axis, 0)
        n = a.shape[0]
        acc, ws = np.zeros_like(a, dtype=f32), f32(0)
        for j in range(-half, half + 1):
            idx = np.arange(n) + j
            idx = np.where(idx < 0, -idx, idx)
            idx = np.where(idx >= n, 2 * n - 2 - idx, idx)
            acc = (acc + w[j + half] * a[idx]).astype(f32)
            ws = f32(ws + w[j + half])
        return np.moveaxis((acc / ws).astype(f32), 0, axis)
    t = one(img.astype(f32), 1)
    t = one(t, 0)
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)
