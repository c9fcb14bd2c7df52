"""The reference's loss / feature helpers under their own names and signatures, on HIP kernels.

A caller of the reference that does `from utils.losses import kl_divergence_map, pairwise_logistic_ranking_loss`
or `from utils.functions import get_masked_patch_cost, interpolate_features, sigmoid` binds these instead; argument
order, defaults, shapes and return values are the reference's.  They are autograd-differentiable like the originals.
The fused training step (`finetune.FinetuneGD`) does NOT go through this module: it uses the fused kernels, in which
the hw x hw maps and the N x N x D tensor these signatures imply never exist.  There is no CPU fallback.
"""
import torch

from . import ops


def sigmoid(tensor, temp=1.0):
    """utils/functions.py:24-33 — temperature sigmoid with the exponent clamped to +-50."""
    return ops.sigmoid_temp(tensor, temp)


def kl_divergence_map(mast3r_cost, feat_cost_sim, eps=1e-8):
    """utils/losses.py:5-15 — mean over rows of sum_j t log(t / p) on the eps-clamped maps [B, hw, hw] -> scalar."""
    return ops.kl_divergence_map(mast3r_cost, feat_cost_sim, eps)


def get_masked_patch_cost(cost, mask_patch_1, mask_patch_2=None, eps=1e-8, use_softmax=False, temperature=1.0):
    """utils/functions.py:402-422 — zero the rows (and, with mask_patch_2, the columns) outside the patch masks, then
    softmax(row / temperature) in fp32 or row / max(rowsum, eps).  cost [B, hw, hw2], masks bool [hw] / [hw2]."""
    return ops.masked_patch_cost(cost, mask_patch_1, mask_patch_2, eps, use_softmax, temperature)


def pairwise_logistic_ranking_loss(model, pred_scores, gt_depths, depth_threshold=0.0):
    """utils/losses.py:18-41 — `model` is the DepthAwareFeatureFusion head (anything with `head_params()` or a
    `fusion_layer` laid out like utils/model.py:96-99); pred_scores [B, N, D] keypoint features, gt_depths [B, N].
    The reference's [B, N, N, D] difference tensor is never formed (W1 (f_j - f_i) = u_j - u_i)."""
    return ops.pair_rank_loss(pred_scores, gt_depths, head_params(model), depth_threshold)


def interpolate_features(descriptors, pts, h, w, normalize=True, patch_size=14, stride=14):
    """utils/functions.py:55-76 — descriptors [B, C, gh, gw], pts [B, N, 2] (x, y pixels of an h x w image) ->
    [B, C, N]: bilinear samples at the patch-centre-aligned coordinates (grid_sample, align_corners=True, border
    padding), L2-normalised over C when `normalize`."""
    from .vit import kp_gather
    B, C, gh, gw = descriptors.shape
    grid = descriptors.permute(0, 2, 3, 1).reshape(B, gh * gw, C)             # token-major view of the NCHW map
    out = kp_gather([grid], pts, gh, gw, 1.0, 1.0, h, w, patch_size, stride=stride)        # [B, N, C] fp32
    if normalize:
        out = ops.l2_normalize(out)
    return out.permute(0, 2, 1)


def head_params(model):
    """DepthAwareFeatureFusion-like module -> the dict of tensors the HIP head kernels take (duck-typed: the reference's
    own utils/model.py class works as well as gd_amd.model's)."""
    if hasattr(model, "head_params"):
        return model.head_params()
    fl = model.fusion_layer
    return {"w1": fl[0].weight, "b1": fl[0].bias, "ln_w": fl[1].weight, "ln_b": fl[1].bias, "w2": fl[3].weight,
            "b2": fl[3].bias}


def extract_kp_depth(depth_map, kp, window_size=3):
    """utils/functions.py:348-372 — depth_map [H, W], kp [1, N, 2] (x, y) -> [1, N] mean depth of the 3 x 3 window."""
    if window_size != 3:
        raise ops._lib.GdHipError("extract_kp_depth: the HIP kernel implements the reference's window_size = 3")
    return ops.kp_depth(depth_map[None] if depth_map.dim() == 2 else depth_map, kp)


def get_patch_mask_from_kp_tensor(kp_xy, H, W, patch_size):
    """utils/functions.py:375-399 — kp_xy [N, 2] -> bool [(H // P) * (W // P)]."""
    return ops.patch_mask(kp_xy[None].float(), H, W, patch_size)[0].bool()
