"""The frozen teacher, run ONCE per image pair, feeding `teacher_cache.TeacherTargetCache` (north_star: "the frozen teacher
(mast3r/model.py, vggt/models) runs once per pair on-device via cached PyTorch-ROCm inference").

The teacher networks are the USER'S modules (the reference vendors them; they are not part of this repository and stay plain
PyTorch-ROCm inference).  What lives here is the glue around them:

* `QKCapture` — forward hooks that pick the post-norm, post-RoPE q and k of selected attention modules off a running teacher
  (duck-typed on the attribute names of `vggt/layers/attention.py:24-50`: `qkv`, `q_norm`, `k_norm`, `rope`, `num_heads`), so the
  distillation target maps can be built by `gd_cross_view_attn` WITHOUT the [2B, H, n, n] per-head softmax maps the reference's
  `return_attn` branch materialises (`vggt/layers/attention.py:73-84`: 480 MB per pair and block at n = 1369);
* `VGGTTeacherRunner` — `extract_vggt_features` + `sample_keypoints` (src/finetune_timm_vggt.py:357-449) as one call that returns the
  pair's targets in the cache layout; the reference aggregator's hard-wired `return_attn=True` is neutralised for the duration of the
  call (the selected blocks return a 1-element placeholder instead of the maps);
* `MASt3RTeacherRunner` — `extract_mast3r_features` + `filter_and_match_keypoints` + the depth branch
  (src/finetune_timm_mast3r.py:345-469, 617-633) around the user's `dust3r.inference.inference`.
"""
import contextlib

import torch

from . import teacher_glue as tg
from ._lib import GdHipError


class QKCapture:
    """with QKCapture(attn_modules) as cap: teacher(...);  cap.qk -> [(q, k)] per module call, each [B, H, N, d] as they enter
    the attention product (after q_norm / k_norm and RoPE).  Works on any module shaped like vggt's `Attention`."""

    def __init__(self, attn_modules):
        self.mods = list(attn_modules)
        self.qk, self._h, self._pend = [], [], {}

    def __enter__(self):
        self._active = None
        ropes = {}
        for i, m in enumerate(self.mods):
            # which selected module is executing: the RoPE module is typically ONE object shared by every block of the teacher
            # (vggt/models/aggregator.py:77-96 hands `self.rope` to all frame and global blocks), so its hook must know whose
            # call it is seeing
            self._h.append(m.register_forward_pre_hook(lambda mod, a, i=i: setattr(self, "_active", i)))
            self._h.append(m.register_forward_hook(lambda mod, a, out: setattr(self, "_active", None)))
            rope = getattr(m, "rope", None)
            if rope is not None and isinstance(rope, torch.nn.Module):
                ropes[id(rope)] = rope        # forward() calls self.rope(q, pos) then self.rope(k, pos): outputs arrive in that order
            else:
                for nm in ("q_norm", "k_norm"):
                    if not isinstance(getattr(m, nm, None), torch.nn.Module):
                        raise GdHipError(f"QKCapture: module {type(m).__name__} has neither a rope module nor {nm}")
                self._h.append(m.q_norm.register_forward_hook(lambda mod, a, out: self._push(out)))
                self._h.append(m.k_norm.register_forward_hook(lambda mod, a, out: self._push(out)))
        for rope in ropes.values():
            self._h.append(rope.register_forward_hook(lambda mod, a, out: self._push(out)))
        return self

    def _push(self, out):
        i = self._active
        if i is None:
            return
        lst = self._pend.setdefault(i, [])
        lst.append(out.detach())
        if len(lst) == 2:
            self.qk.append((i, lst[0], lst[1]))
            self._pend[i] = []

    def __exit__(self, *exc):
        for h in self._h:
            h.remove()
        self._h = []
        return False

    def pairs(self):
        """[(q, k)] ordered by module index (one call per module assumed)."""
        return [(q, k) for _, q, k in sorted(self.qk, key=lambda t: t[0])]


@contextlib.contextmanager
def _no_attention_maps(attn_modules):
    """The reference's aggregator calls its global blocks with return_attn=True unconditionally (vggt/models/aggregator.py:257,
    314) and each then forms two [B, H, n, n] softmax maps.  For the duration of the block, the selected attention modules answer
    `return_attn=True` with (output, 1-element placeholder): the aggregator's `torch.stack(attn_list).mean(0)` stays valid, the
    maps are never built."""
    saved = []
    for m in attn_modules:
        orig = m.forward

        def fwd(x, pos=None, return_attn=False, temperature=1.0, _orig=orig):
            out = _orig(x, pos=pos)
            return (out, out.new_zeros(1)) if return_attn else out
        saved.append((m, orig))
        m.forward = fwd
    try:
        yield
    finally:
        for m, orig in saved:
            del m.forward            # drop the instance attribute: the class's forward is visible again


class VGGTTeacherRunner:
    """vggt: a model shaped like vggt.models.vggt.VGGT (aggregator with `global_blocks`, `attn_indices`, `aa_block_size`,
    `temperature`; camera_head, depth_head, point_head, track_head).  `pose_decoder(pose_enc, image_hw) -> (extrinsic, intrinsic)`
    = vggt.utils.pose_enc.pose_encoding_to_extri_intri of the user's vggt package (imported lazily when not given)."""

    def __init__(self, vggt, dtype=torch.bfloat16, prefix=5, pose_decoder=None):
        self.m, self.dtype, self.prefix, self.pose_decoder = vggt, dtype, prefix, pose_decoder
        agg = vggt.aggregator
        per = getattr(agg, "aa_block_size", 1)
        # aggregator.forward appends the map of the LAST global block of every selected aa iteration (aggregator.py:255-260)
        self.sel = [agg.global_blocks[i * per + per - 1].attn for i in agg.attn_indices]

    @torch.no_grad()
    def targets(self, rgb_vggt, num_keypoints=300, min_distance=5, generator=None):
        """rgb_vggt [1, 2, 3, H, W] in [0, 1] -> the pair's target dict (teacher_glue.extract_vggt_targets), or None."""
        m, agg = self.m, self.m.aggregator
        with QKCapture(self.sel) as cap, _no_attention_maps(self.sel):
            with torch.autocast("cuda", dtype=self.dtype, enabled=rgb_vggt.is_cuda):
                tokens_list, ps_idx, _ = agg(rgb_vggt)
            pose_enc = m.camera_head(tokens_list)[-1]
            dec = self.pose_decoder
            if dec is None:
                from vggt.utils.pose_enc import pose_encoding_to_extri_intri as dec       # the user's teacher package
            extrinsic, intrinsic = dec(pose_enc, rgb_vggt.shape[-2:])
            depth_map, _ = m.depth_head(tokens_list, rgb_vggt, ps_idx)
            _, point_conf = m.point_head(tokens_list, rgb_vggt, ps_idx)

        def track(kp1):
            trk, _, _ = m.track_head(tokens_list, rgb_vggt, ps_idx, query_points=kp1[None])
            return trk[-1][0][1]
        qk = [(q.to(self.dtype) if q.dtype not in (torch.float32, torch.bfloat16) else q,
               k.to(self.dtype) if k.dtype not in (torch.float32, torch.bfloat16) else k) for q, k in cap.pairs()]
        scale = float(self.sel[0].scale) if hasattr(self.sel[0], "scale") else qk[0][0].shape[-1] ** -0.5
        return tg.extract_vggt_targets(qk, depth_map[0], point_conf[0], extrinsic[0], intrinsic[0], track, scale=scale,
                                       temperature=float(getattr(agg, "temperature", 1.0)), prefix=self.prefix,
                                       num_keypoints=num_keypoints, min_distance=min_distance, generator=generator)


class MASt3RTeacherRunner:
    """matcher: the user's AsymmetricMASt3R (the reference's fork returns `tgt_attn_map`, dust3r/dust3r/model.py:346-366);
    `inference` / `make_pairs`: dust3r.inference.inference and dust3r.image_pairs.make_pairs of the user's package (lazy import)."""

    def __init__(self, matcher, inference=None, make_pairs=None, min_conf_thr=10, subsample=16, keep_logits=False):
        self.matcher, self.inference, self.make_pairs = matcher, inference, make_pairs
        self.min_conf_thr, self.subsample, self.keep_logits = min_conf_thr, subsample, keep_logits

    @torch.no_grad()
    def targets(self, rgb_mast3r_1, rgb_mast3r_2, temperature=1.0, intrinsic=None, depth_1=None, depth_2=None, device="cuda"):
        inf, mk = self.inference, self.make_pairs
        if inf is None:
            from dust3r.inference import inference as inf
        if mk is None:
            from dust3r.image_pairs import make_pairs as mk
        self.matcher.temperature = temperature          # src/finetune_timm_mast3r.py:215, 224: the annealed target temperature
        # the decoder's raw cross-attention score maps (dust3r/dust3r/model.py:337, `_decoder` -> dec_feats, tgt_camap, src_camap) are
        # picked up on the way: their head mean / reciprocity average is the temperature-independent part of `tgt_attn_map`, which
        # lets TeacherTargetCache(keep_logits=True) follow the annealed temperature without another teacher forward
        seen, dec = [], getattr(self.matcher, "_decoder", None)
        if self.keep_logits and dec is not None:
            def wrapped(*a, **k):
                r = dec(*a, **k)
                if isinstance(r, tuple) and len(r) == 3:
                    seen.append((r[1], r[2]))
                return r
            self.matcher._decoder = wrapped
        try:
            out = inf(mk([rgb_mast3r_1, rgb_mast3r_2], scene_graph="complete", prefilter=None, symmetrize=True), self.matcher, device,
                      verbose=False)
        finally:
            if self.keep_logits and dec is not None:
                del self.matcher._decoder           # the instance attribute shadowing the method
        p1, p2 = out["pred1"], out["pred2"]
        dev = torch.device(device)
        recip = tg.mast3r_recip_logits([t.to(dev) for t in seen[-1][0]], [t.to(dev) for t in seen[-1][1]]) if seen else None
        return tg.extract_mast3r_targets(
            p1["desc"][1].to(dev), p2["desc"][1].to(dev), p1["conf"][1].to(dev), p1["conf"][0].to(dev),
            p1["pts3d"][1].to(dev), p2["pts3d_in_other_view"][1].to(dev), p1["pts3d"][0].to(dev),
            p2["tgt_attn_map"][1].to(dev), p2["tgt_attn_map"][0].to(dev), intrinsic=intrinsic, depth_1=depth_1, depth_2=depth_2,
            subsample=self.subsample, min_conf_thr=self.min_conf_thr, cost_recip=recip)
