"""Geometric-distillation fine-tuning step on HIP kernels — the host-side mirror of the reference's
LightningModules `FinetuneVGGTTIMM` (src/finetune_timm_vggt.py:81-648) and `FinetuneMASt3RTIMM`
(src/finetune_timm_mast3r.py:72-689), with the same method names and argument meaning, batched over P
image pairs (the reference is hard-wired to one pair per rank; P pairs here = P data-parallel ranks of it,
the loss is the mean over pairs).

The frozen teacher (MASt3R / VGGT) is OUTSIDE this module: `training_step` takes its outputs (cost maps,
keypoints, 3-D points, depth maps, co-view masks) as the batch's teacher targets (a18-a20 run as
cached PyTorch-ROCm inference; for benchmarks they are synthetic).

geometry = "reference": keypoint features at target_res/downsample_factor tokens (640/8 = 80 on the long
side), cost features at the teacher grid — the reference's three resolutions (two distinct forwards).
geometry = "shared": one forward per image at the teacher grid feeds all three extractors (BASELINE 518^2).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .options import option
from .model import Adapter, BlockWithAdapter, DepthAwareFeatureFusion, _LoRA_qkv
from .vit import conv3x3_at_keypoints, conv3x3_tokens, create_vit, kp_gather


def _pair_batch(a, b):
    """cat([a, b], 0) — free when the loader already collated the two views as the halves of one buffer."""
    base = a._base
    if (base is not None and base is b._base and base.is_contiguous() and base.shape[0] == a.shape[0] + b.shape[0]
            and base.shape[1:] == a.shape[1:] and a.is_contiguous() and b.is_contiguous()
            and a.data_ptr() == base.data_ptr() and b.data_ptr() == base.data_ptr() + a.numel() * a.element_size()):
        return base
    return torch.cat([a, b], 0).contiguous()


class FinetuneGD(nn.Module):
    def __init__(self, r=4, backbone="vit_base", patch_size=14, img_size=518, variant="vggt", geometry="shared",
                 dtype="bf16", ap_loss_weight=1.0, depth_loss_weight=None, intra_depth_loss_weight=None,
                 kl_loss_weight=None, teacher_patch=None, adapter_start_idx=4, bottleneck_dim=64, vit_kwargs=None,
                 lora_b_std=0.0, seed=0, init_temperature=1.0, final_temperature=None, max_epochs=500):
        super().__init__()
        assert r > 0 and variant in ("vggt", "mast3r", "me") and geometry in ("shared", "reference")
        self.variant, self.geometry = variant, geometry
        me = variant == "me"     # FinetuneTIMM (src/finetune_timm_me.py:30-89): LoRA on the last 4 blocks, AP loss only
        self.ap_loss_weight = ap_loss_weight
        # reference defaults: MASt3R depth_loss_weight=0.0 (finetune_timm_mast3r.py:79-82), VGGT 1.0 (:86-89)
        self.depth_loss_weight = (1.0 if variant == "vggt" else 0.0) if depth_loss_weight is None else depth_loss_weight
        self.intra_depth_loss_weight = (0.0 if me else 1.0) if intra_depth_loss_weight is None else intra_depth_loss_weight
        self.kl_loss_weight = (0.0 if me else 1.0) if kl_loss_weight is None else kl_loss_weight
        # teacher softmax temperature schedule: MASt3R 1.0 -> 0.5 over max_epochs (finetune_timm_mast3r.py:83-84,217-227),
        # VGGT constant 1.0 (finetune_timm_vggt.py:92-93,187-188).  The teacher runner reads `teacher_temperature`.
        self.init_temperature = init_temperature
        self.final_temperature = (0.5 if variant == "mast3r" else init_temperature) if final_temperature is None else final_temperature
        self.teacher_temperature = self.init_temperature
        self.max_epochs, self.current_epoch = max_epochs, 0
        model = create_vit(backbone, patch_size=patch_size, img_size=img_size, dtype=dtype, **(vit_kwargs or {}))
        self.embedding_dim = model.embed_dim
        for p in model.parameters():
            p.requires_grad = False
        # --- LoRA(q,v) + adapters on blocks[adapter_start_idx:]  (src/finetune_timm_vggt.py:134-162) ---
        self.w_As, self.w_Bs = [], []
        self.adapters = nn.ModuleList()
        g = torch.Generator().manual_seed(seed + 1)
        if me:
            adapter_start_idx = len(model.blocks) - 4         # `model.blocks[-4:]`, no adapters (finetune_timm_me.py:51-70)
        for blk_idx in range(adapter_start_idx, len(model.blocks)):
            blk = model.blocks[blk_idx]
            w_qkv_linear = blk.attn.qkv
            self.dim = w_qkv_linear.in_features
            w_a_q, w_b_q = nn.Linear(self.dim, r, bias=False), nn.Linear(r, self.dim, bias=False)
            w_a_v, w_b_v = nn.Linear(self.dim, r, bias=False), nn.Linear(r, self.dim, bias=False)
            self.w_As += [w_a_q, w_a_v]
            self.w_Bs += [w_b_q, w_b_v]
            blk.attn.qkv = _LoRA_qkv(w_qkv_linear, w_a_q, w_b_q, w_a_v, w_b_v)
            if me:
                continue
            adapter = Adapter(dim=self.embedding_dim, bottleneck_dim=bottleneck_dim)
            model.blocks[blk_idx] = BlockWithAdapter(blk, adapter)
            self.adapters.append(adapter)
        self.reset_parameters(lora_b_std, g)
        self.model = model
        self.downsample_factor = 8
        self.refine_conv = nn.Conv2d(self.embedding_dim, self.embedding_dim, kernel_size=3, stride=1, padding=1)
        self.thres3d_neg = 0.1
        self.thresh3d_pos = 5e-3                               # ME positives (finetune_timm_me.py:76)
        self.patch_size = model.patch_embed.patch_size[0]
        self.target_res = 640
        self.depth_diff_head = None if me else DepthAwareFeatureFusion(input_dim=self.embedding_dim, use_tanh=True)
        self.resize_patch_size = teacher_patch or (14 if variant == "vggt" else self.patch_size)
        self._fwd_cache, self._norm_cache, self._fuse_taps = {}, {}, False
        self._flat = None
        self._pending_opt_state = None

    def reset_parameters(self, lora_b_std=0.0, generator=None):
        """A: kaiming-uniform(a=sqrt 5), B: zeros (src/finetune_timm_vggt.py:166-170).  lora_b_std > 0 gives the
        small non-zero B that gradient-parity runs use (zero-init hides bugs)."""
        for w_A in self.w_As:
            nn.init.kaiming_uniform_(w_A.weight, a=math.sqrt(5), generator=generator)
        for w_B in self.w_Bs:
            if lora_b_std > 0:
                nn.init.normal_(w_B.weight, std=lora_b_std, generator=generator)
            else:
                nn.init.zeros_(w_B.weight)

    # ------------------------------------------------------------------ optimiser state (flat buffers)
    def trainable_parameters(self):
        """Order of configure_optimizers (src/finetune_timm_vggt.py:642-648)."""
        head = list(self.depth_diff_head.parameters()) if self.depth_diff_head is not None else []
        return ([l.weight for l in self.w_As] + [l.weight for l in self.w_Bs] + list(self.refine_conv.parameters())
                + head + list(self.adapters.parameters()))

    def configure_optimizers(self, lr=1e-5, weight_decay=1e-4, max_norm=1.0):
        """Re-seat every trainable tensor as a view of ONE flat fp32 buffer (params / grads / Adam moments):
        one fused clip+AdamW launch and one all-reduce message per step."""
        ps = self.trainable_parameters()
        al = lambda k: (k + 3) // 4 * 4          # every tensor starts 16-byte aligned (GEMM operand requirement);
        n = sum(al(p.numel()) for p in ps)       # the pad elements stay zero in params, grads and moments
        dev = ps[0].device
        flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        # parameters that never receive a gradient (depth_attention: utils/model.py:92-97 builds it, no loss evaluates it):
        # torch.optim.AdamW skips a parameter whose .grad is None — no moment update AND no weight decay.  They sit in ONE
        # contiguous span of the flat buffer (depth_diff_head.parameters() lists depth_attention first); the optimiser
        # step runs on the two ranges around it.
        unused = {id(q) for q in self.depth_diff_head.depth_attention.parameters()} if self.depth_diff_head is not None else set()
        off = 0
        views, skip = [], []
        for p in ps:
            k = p.numel()
            flat_p[off:off + k] = p.detach().reshape(-1)
            p.data = flat_p[off:off + k].view(p.shape)
            p.grad = flat_g[off:off + k].view(p.shape)
            views.append(p.grad)
            if id(p) in unused:
                skip.append((off, off + al(k)))
            off += al(k)
        # where the LoRA-A / LoRA-B / adapter tensors start: they sit back to back (all sizes are multiples of 4), in the order the
        # engine stacks them, so the per-step weight packs and the blocks' weight gradients can use views (vit.prepare_trainables)
        spans = None
        nA, nB, nAd = len(self.w_As), len(self.w_Bs), 2 * len(self.adapters)
        packed = ps[:nA + nB] + (ps[len(ps) - nAd:] if nAd else [])       # these must sit gap-free: sizes that are multiples of 4
        if nA and nAd == nA and all(p.numel() % 4 == 0 for p in packed):
            offs, o = [], 0
            for p in ps:
                offs.append(o)
                o += al(p.numel())
            spans = {"A": offs[0], "B": offs[nA], "ad": offs[len(ps) - nAd], "L": nA // 2}
        live, pos = [], 0
        for a, b in skip:                     # complement of the (adjacent) skipped spans
            if a > pos:
                live.append((pos, a))
            pos = max(pos, b)
        if pos < n:
            live.append((pos, n))
        self._flat = {"p": flat_p, "g": flat_g, "views": views, "m": torch.zeros_like(flat_p), "v": torch.zeros_like(flat_p),
                      "step": 0, "lr": lr, "wd": weight_decay, "max_norm": max_norm, "live": live, "spans": spans}
        st, self._pending_opt_state = getattr(self, "_pending_opt_state", None), None
        if st is not None:       # a checkpoint was loaded before the optimiser existed: resume its moments and step count
            self._apply_optimizer_state(st)
        return self._flat

    # ------------------------------------------------------------------ tf32h: the range contract, observed
    def _range_counters(self):
        """Two device counters every scaled fp16 gradient cast of the tf32h engine adds to (ops.set_range_counters): operands that saturated at
        +-65504, and non-zero operands that fell below fp16's normal range (2^-14 after the block's power-of-two scale: fewer than 11 bits, or
        flushed).  Device-side, one atomic per wave and only when a count is non-zero: nothing here synchronises."""
        if getattr(self, "_range", None) is None:
            dev = next(self.parameters()).device
            self._range = torch.zeros(128, dtype=torch.int32, device=dev)      # 2 x 64 partial sums (csrc: GD_RANGE_SLOTS)
        return self._range

    def range_report(self, reset=True):
        """{'saturated': n, 'below_normal': n} since the last reset (ONE host read: call it every N steps, not every step).  A healthy run
        reports saturated == 0; below_normal counts gradient entries more than 2^17 under their block's largest one, which a TF32 tensor core
        would still have carried at full precision (the range contract).  What is counted: every scaled fp16 cast (cast kernels, the
        LayerNorm backward's fp16 copy) and the two dX products per block that leave their GEMM epilogue as saturating fp16 and are read by a
        LayerNorm backward (counted there: entries at the clamp).  The attention output gradient and the GELU-gated MLP gradient are saturating fp16
        stores too, consumed by MFMA kernels only: those are not counted."""
        if getattr(self, "_range", None) is None:
            return {"saturated": 0, "below_normal": 0}
        v = self._range.view(2, 64).sum(1).tolist()
        if reset:
            self._range.zero_()
        return {"saturated": int(v[0]), "below_normal": int(v[1])}

    def zero_grad_flat(self):
        self._flat["g"].zero_()

    def backward(self, loss, pre_gather=None):
        """loss.backward() with the gradients gathered into the flat buffer by ONE multi-tensor copy.  With `p.grad`
        pre-set to views of the flat buffer autograd accumulates into each of the ~60 trainable tensors separately
        (an add and a copy kernel per tensor, ~6 us each on an otherwise busy GPU); with `p.grad = None` it just hands the
        gradient tensors over."""
        ps = self.trainable_parameters()
        views = self._flat["views"]
        for p in ps:
            p.grad = None
        self._flat["g"].zero_()          # BEFORE the backward: blocks prepared with the flat record accumulate straight into it
        loss.backward()
        ops.amax_clear()                 # (tf32h: a gradient scale nobody consumed must not outlive its backward pass)
        self.model.finish_trainable_grads()
        if pre_gather is not None:       # e.g. OverlappedGradReducer.wait_early: hook-launched all-reduces of some p.grad
            pre_gather()
        dst = [v for p, v in zip(ps, views) if p.grad is not None]
        src = [p.grad for p in ps if p.grad is not None]
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(ps, views):
            p.grad = v

    def optimizer_step(self, grad_scale=1.0):
        f = self._flat
        f["step"] += 1
        return ops.clip_adamw_step(f["p"], f["g"], f["m"], f["v"], f["step"], lr=f["lr"], weight_decay=f["wd"],
                                   max_norm=f["max_norm"], grad_scale=grad_scale, ranges=f["live"])

    # ------------------------------------------------------------------ teacher temperature schedule
    def update_temperature(self, current_epoch=None, total_epochs=None):
        """src/finetune_timm_mast3r.py:217-224: linear init -> final over the run's epochs; the result is what the
        teacher's target softmax uses (`self.matcher.temperature` there, `teacher_temperature` here)."""
        if current_epoch is not None:
            self.current_epoch = current_epoch
        total = total_epochs if total_epochs is not None else (self.max_epochs if self.max_epochs and self.max_epochs > 0 else 100)
        ratio = min(self.current_epoch / total, 1.0)
        self.teacher_temperature = self.init_temperature * (1 - ratio) + self.final_temperature * ratio
        return self.teacher_temperature

    def on_train_batch_end(self, *args, **kwargs):
        """src/finetune_timm_mast3r.py:226-227"""
        self.update_temperature()

    def fit_step(self, batch, reducer=None):
        """One optimisation step of the minimal loop that stands in for Lightning's (src/main.py:153-161): forward +
        losses, backward, gradient exchange (dp.OverlappedGradReducer; None = single rank), clip + AdamW.
        -> (loss, terms, pre-clip gradient norm)."""
        # weight packs as views of the flat parameter buffer + LoRA / adapter weight gradients accumulated straight into the flat gradient
        # buffer (GDViT.prepare_trainables(flat)): ~45 fewer torch-side launches per step (no per-block zero-filled scratch, no
        # AccumulateGrad copies of the strided LoRA-B views, no 48-tensor gather).  Step time equal within run-to-run noise on one box
        # (round 2: 539.3 vs 542.9 image-pairs/s); the default since round 3, GD_DIRECT_GRADS=0 restores the autograd gather.
        loss, terms = self.training_step(batch, direct_grads=bool(option("direct_grads")))
        if reducer is None:
            self.backward(loss)
            scale = 1.0
        else:
            self.backward(loss, pre_gather=reducer.wait_early)
            reducer.start()
            scale = reducer.finish()
        return loss, terms, self.optimizer_step(grad_scale=scale)

    # ------------------------------------------------------------------ checkpoint layout
    def on_save_checkpoint(self, checkpoint):
        checkpoint["state_dict"] = {"refine_conv": self.refine_conv.state_dict()}
        for i, l in enumerate(self.w_As):
            checkpoint[f"w_a_{i:03d}"] = l.weight
        for i, l in enumerate(self.w_Bs):
            checkpoint[f"w_b_{i:03d}"] = l.weight
        if self.depth_diff_head is not None:
            checkpoint["depth_diff_head"] = self.depth_diff_head.state_dict()
        for i, a in enumerate(self.adapters):
            checkpoint[f"adapter_{i:03d}"] = a.state_dict()
        # Lightning stores `optimizer_states` beside these keys; here the AdamW state is the flat moment buffers
        if self._flat is not None:
            f = self._flat
            checkpoint["gd_optimizer_state"] = {"exp_avg": f["m"].detach().clone(), "exp_avg_sq": f["v"].detach().clone(),
                                                "step": f["step"], "numel": f["p"].numel()}
        # every tensor — the nested state_dicts included — is a view of the flat parameter buffer: save copies, so that torch.save
        # writes the tensors and not the whole storage, and a later optimiser step cannot change a checkpoint that is still in memory
        def own(v):
            if isinstance(v, torch.Tensor):
                return v.detach().clone()
            if isinstance(v, dict):
                return type(v)((k, own(x)) for k, x in v.items())
            return v
        for k in list(checkpoint.keys()):
            checkpoint[k] = own(checkpoint[k])
        checkpoint["epoch"] = self.current_epoch
        return checkpoint

    def on_load_checkpoint(self, checkpoint):
        with torch.no_grad():
            self.refine_conv.load_state_dict(checkpoint["state_dict"]["refine_conv"])
            for i, l in enumerate(self.w_As):
                l.weight.copy_(checkpoint[f"w_a_{i:03d}"])
            for i, l in enumerate(self.w_Bs):
                l.weight.copy_(checkpoint[f"w_b_{i:03d}"])
            if self.depth_diff_head is not None:
                self.depth_diff_head.load_state_dict(checkpoint["depth_diff_head"])
            for i, a in enumerate(self.adapters):
                a.load_state_dict(checkpoint[f"adapter_{i:03d}"])
            st = checkpoint.get("gd_optimizer_state")
            if st is not None:
                if self._flat is None:       # the natural resume order: load first, configure_optimizers applies it
                    self._pending_opt_state = st
                else:
                    self._apply_optimizer_state(st)
            self.current_epoch = int(checkpoint.get("epoch", self.current_epoch))
            self.update_temperature()

    def _apply_optimizer_state(self, st):
        """AdamW moments + step count of a checkpoint into the flat buffers; a layout mismatch is an error, not a silent restart."""
        n = self._flat["p"].numel()
        if int(st["numel"]) != n:
            raise ValueError(f"checkpoint optimizer state has {int(st['numel'])} elements, this engine's trainable buffer has {n} "
                             "(different backbone / rank / adapter configuration?)")
        with torch.no_grad():
            self._flat["m"].copy_(st["exp_avg"])
            self._flat["v"].copy_(st["exp_avg_sq"])
        self._flat["step"] = int(st["step"])

    # ------------------------------------------------------------------ student forwards
    def _kp_grid(self, h, w):
        if self.geometry == "shared":
            return h // self.resize_patch_size, w // self.resize_patch_size
        tr, ds = self.target_res, self.downsample_factor
        tgt = (tr, int(w * tr / h)) if h > w else (int(h * tr / w), tr)
        return tgt[0] // ds, tgt[1] // ds

    def _forward(self, rgbs, gh, gw):
        """taps [4..7] + last block output at a gh x gw token grid; cached per (image tensor, grid) so that the three
        extractors of one step share a forward where their resolutions coincide."""
        key = (rgbs.data_ptr(), rgbs._version, tuple(rgbs.shape), gh, gw)
        if key not in self._fwd_cache:
            P = self.patch_size
            if getattr(self, "_fuse_taps", False):   # training_step: every tap is also wanted final-normed
                taps, x, normed = self.model.forward_all(rgbs, (4, 5, 6, 7), size=(gh * P, gw * P), norm_taps="deferred")
                self._norm_cache[key] = normed
            else:
                taps, x = self.model.forward_all(rgbs, (4, 5, 6, 7), size=(gh * P, gw * P))
            self._fwd_cache[key] = (taps, x)
        return self._fwd_cache[key]

    def clear_cache(self):
        self._fwd_cache, self._norm_cache, self._kp_cache = {}, {}, {}
        ops.ln_stats_clear()

    def _kp_pair(self, kp_1, kp_2):
        """cat([kp_1, kp_2], 0), once per step (the depth and the matching loss sample the same keypoints)."""
        key = (kp_1.data_ptr(), kp_2.data_ptr(), kp_1._version, kp_2._version, tuple(kp_1.shape))
        c = getattr(self, "_kp_cache", None)
        if c is None:
            c = self._kp_cache = {}
        if key not in c:
            c[key] = torch.cat([kp_1, kp_2], 0)
        return c[key]

    def get_intermediate_feature(self, rgbs, pts=None, n=(4, 5, 6, 7), normalize=True):
        """src/finetune_timm_vggt.py:256-302 (reshape=True path): mean over taps of bilinear samples of the
        (final-normed) tap grids at the keypoints -> [B, N, D] fp32."""
        h, w = rgbs.shape[-2:]
        gh, gw = self._kp_grid(h, w)
        P = self.patch_size
        taps, _ = self._forward(rgbs, gh, gw)
        idx = [(4, 5, 6, 7).index(i) for i in n]
        normed = self._norm_cache.get((rgbs.data_ptr(), rgbs._version, tuple(rgbs.shape), gh, gw))
        if not normalize:
            grids = [taps[j] for j in idx]
        elif normed is not None:
            grids = [normed[j] for j in idx]
        else:
            grids = [self.model.norm(taps[j]) for j in idx]
        return kp_gather(grids, pts, gh, gw, (gw * P) / w, (gh * P) / h, gh * P, gw * P, P)

    def get_feature(self, rgbs, pts, normalize=True):
        """src/finetune_timm_vggt.py:304-332: forward_features -> refine_conv -> bilinear sample -> L2 normalise."""
        h, w = rgbs.shape[-2:]
        gh, gw = self._kp_grid(h, w)
        P = self.patch_size
        _, x = self._forward(rgbs, gh, gw)
        # the ME trainer passes h = patch_h * 14 and the default patch_size = stride = 14 whatever the model's patch is
        # (src/finetune_timm_me.py:155); the two teacher-driven trainers pass the model's patch (finetune_timm_vggt.py:325-327)
        Pi = 14 if self.variant == "me" else P
        xn = self.model.norm(x)
        # conv and bilinear sample are both linear: mix the input patches first, convolve B*Nk rows instead of the whole grid
        feat = conv3x3_at_keypoints(xn, self.refine_conv.weight, self.refine_conv.bias, pts, gh, gw, (gw * P) / w, (gh * P) / h,
                                    gh * Pi, gw * Pi, Pi, x3=getattr(self.model, "opfmt", ""))
        if feat is None:
            fmap, pitch = conv3x3_tokens(xn, self.refine_conv.weight, self.refine_conv.bias, gh, gw)
            feat = kp_gather([fmap], pts, gh, gw, (gw * P) / w, (gh * P) / h, gh * Pi, gw * Pi, Pi, pitch=pitch)
        return ops.l2_normalize(feat) if normalize else feat

    def get_feature_cost(self, rgbs, with_norm=False):
        """src/finetune_timm_vggt.py:335-355 (tap 7) / src/finetune_timm_mast3r.py:321-342 (mean of taps 4-7), no
        norm, prefix dropped -> [B, hw, D] in the engine dtype.  with_norm: -> (features, inverse row norms [B, hw]) — the cost
        loss normalises these rows; their norms are taken while the rows are written."""
        h, w = rgbs.shape[-2:]
        ch, cw = h // self.resize_patch_size, w // self.resize_patch_size
        taps, _ = self._forward(rgbs, ch, cw)
        sel = [taps[3]] if self.variant == "vggt" else list(taps)
        if with_norm and getattr(self.model, "opfmt", "") == "h" and sel[0].dtype == torch.float32:
            with_norm = 2       # tf32h: the fp16 copy of the rows comes out of the same pass (-> (features, inverse norms, fp16 features))
        return ops.tap_mean(sel, prefix=self.model.num_prefix_tokens, with_norm=with_norm)

    # ------------------------------------------------------------------ losses (per-pair vectors [P])
    def calculate_depth_loss(self, depth_1, depth_2, rgbs, kp_1, kp_2, counts=None, indices=(4, 5, 6, 7)):
        """src/finetune_timm_vggt.py:465-485.  rgbs = cat(rgb_1, rgb_2) [2P,3,h,w]; depth_k [P,H,W]."""
        P = kp_1.shape[0]
        kp = self._kp_pair(kp_1, kp_2)
        feat = self.get_intermediate_feature(rgbs, pts=kp, n=indices, normalize=True)       # [2P,N,D]
        d1, d2 = ops.kp_depth(depth_1, kp_1), ops.kp_depth(depth_2, kp_2)
        return ops.depth_losses(feat, d1, d2, self.depth_diff_head.head_params(), counts=counts, depth_threshold=0.05)   # view-major [2P,N,D]

    def calculate_cost_loss(self, rgbs, cost_1, cost_2, kp_1=None, kp_2=None, mask_1=None, mask_2=None, cost_tstats=None):
        """src/finetune_timm_vggt.py:488-533 / src/finetune_timm_mast3r.py:504-540.  cost_k [P, hw, hw] — or [P, hw, ldt]
        padded to 16-byte rows with `cost_tstats` = the cached teacher-row statistics (teacher_cache.TeacherTargetCache)."""
        h, w = rgbs.shape[-2:]
        P = rgbs.shape[0] // 2
        fc = self.get_feature_cost(rgbs, with_norm=True)
        f, inv, f16 = fc if len(fc) == 3 else (fc[0], fc[1], None)
        ph, pw = h // self.resize_patch_size, w // self.resize_patch_size
        kmax = None
        if self.variant == "mast3r" or mask_1 is None:
            m1 = ops.patch_mask(kp_1, h, w, self.patch_size)
            m2 = ops.patch_mask(kp_2, h, w, self.patch_size)
            kmax = max(kp_1.shape[1], kp_2.shape[1])        # a keypoint marks one patch: at most this many rows are kept per view
        else:
            m1 = F.interpolate(mask_1[:, None].float(), size=(ph, pw), mode="nearest").reshape(P, -1) > 0
            m2 = F.interpolate(mask_2[:, None].float(), size=(ph, pw), mode="nearest").reshape(P, -1) > 0
        f1, f2 = ops.split_pairs(f, P)
        return ops.cost_volume_kl(f1, f2, cost_1, cost_2, m1, m2, self.variant, tstats=cost_tstats, inv_norms=(inv[:P], inv[P:]),
                                  x3=getattr(self.model, "opfmt", ""), h16=None if f16 is None else (f16[:P], f16[P:]), kept_rows_max=kmax)

    def calculate_matching_loss(self, rgbs, kp_1, kp_2, pts3d_1, pts3d_2, counts=None):
        """src/finetune_timm_vggt.py:536-574 / src/finetune_timm_mast3r.py:543-589.  pts3d_k [P,N,3] are the
        teacher's 3-D points already gathered at the keypoints."""
        P = kp_1.shape[0]
        desc = self.get_feature(rgbs, self._kp_pair(kp_1, kp_2), normalize=True)
        return ops.smooth_ap_pairs(desc, P, pts3d_1, pts3d_2, counts, self.variant, self.thres3d_neg, 0.01, thres3d_pos=self.thresh3d_pos)

    def training_step(self, batch, direct_grads=False):
        """Loss of P pairs = mean over pairs of the reference's per-pair loss (src/finetune_timm_vggt.py:599-616).
        batch: rgb_1, rgb_2 [P,3,h,w] in [0,1]; kp_1, kp_2 [P,N,2] px; counts int32 [P] (optional);
        pts3d_1, pts3d_2 [P,N,3]; depth_1, depth_2 [P,h,w]; cost_1, cost_2 [P,hw,hw];
        mask_1, mask_2 [P,h,w] bool (vggt)."""
        self.clear_cache()
        ops.set_range_counters(self._range_counters() if getattr(self.model, "opfmt", "") == "h" else None)
        # per-step pack of the LoRA / adapter weights (dropped again below: never stale).  direct_grads (fit_step): built from
        # views of the flat parameter buffer, and the blocks write their weight gradients into the flat gradient buffer — the
        # caller must then use `self.backward(loss)` (it zeroes that buffer first and finishes the LoRA-B transposes)
        self.model.prepare_trainables(self._flat if direct_grads and self._flat is not None else None)
        self._fuse_taps = self.geometry == "shared"   # one forward feeds keypoint AND cost features: norm the taps on the way
        rgbs = _pair_batch(batch["rgb_1"], batch["rgb_2"])
        counts = batch.get("counts")
        if self.variant == "me":          # src/finetune_timm_me.py:191-220: the correspondence AP loss alone
            ap = self.calculate_matching_loss(rgbs, batch["kp_1"], batch["kp_2"], batch["pts3d_1"], batch["pts3d_2"], counts)
            self.clear_cache()
            self._fuse_taps = False
            self.model.release_trainables()
            zero = torch.zeros_like(ap.detach())
            loss, tm = ops.loss_combine(ap, zero, zero, zero, (self.ap_loss_weight, 0.0, 0.0, 0.0), None)
            return loss, {"ap_loss": tm[0], "depth_loss": zero, "intra_depth_loss": zero, "kl_loss": zero}
        depth_loss, intra = self.calculate_depth_loss(batch["depth_1"], batch["depth_2"], rgbs, batch["kp_1"],
                                                      batch["kp_2"], counts)
        kl = self.calculate_cost_loss(rgbs, batch["cost_1"], batch["cost_2"], batch["kp_1"], batch["kp_2"],
                                      batch.get("mask_1"), batch.get("mask_2"), batch.get("cost_tstats"))
        ap = self.calculate_matching_loss(rgbs, batch["kp_1"], batch["kp_2"], batch["pts3d_1"], batch["pts3d_2"], counts)
        # loss = mean over pairs of the weighted sum; a pair whose keypoint filter left nothing contributes the reference's constant zero
        # (src/finetune_timm_mast3r.py:604-607, src/finetune_timm_vggt.py:585-597: zero loss, zero gradient, zero reported terms) — one kernel each way
        loss, tm = ops.loss_combine(ap, depth_loss, intra, kl, (self.ap_loss_weight, self.depth_loss_weight, self.intra_depth_loss_weight,
                                                                self.kl_loss_weight), counts)
        self.clear_cache()
        self._fuse_taps = False
        self.model.release_trainables()
        return loss, {"ap_loss": tm[0], "depth_loss": tm[1], "intra_depth_loss": tm[2], "kl_loss": tm[3]}
