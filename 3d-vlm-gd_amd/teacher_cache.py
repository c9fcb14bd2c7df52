"""Per-pair cache of the frozen teacher's distillation targets (north_star: "the frozen teacher runs once per pair
on-device via cached PyTorch-ROCm inference"; reference call sites src/finetune_timm_vggt.py:357-411,599-603 and
src/finetune_timm_mast3r.py:345-389,597-633, which re-run the teacher on every step).

A cached entry holds device tensors in the layout the fused student step reads them:
  cost_1, cost_2   [hw, ldt] fp32 — teacher cost maps, row stride padded to 16 bytes (zero pad)
  cost_tstats      [2, hw, 4] fp32 — per teacher row {max(rowsum, 1e-8), sum t, sum t log t, 0} (ops.cost_volume_teacher_stats):
                   everything the KL needs from the teacher side alone, so the step reads each map exactly once
  kp_1, kp_2 [N, 2], count, pts3d_1, pts3d_2 [N, 3], depth_1, depth_2 [h, w], mask_1, mask_2 [h, w] (VGGT co-view masks)
  cost_recip       optional [L, 2, N, N] fp32 — MASt3R's pre-softmax reciprocal score maps (temperature-independent; see the class)
`collate` stacks entries into the batch dict of FinetuneGD.training_step (ragged keypoint counts -> `counts` + -1 padding).
"""
import torch

from . import ops

_PER_PAIR = ("cost_1", "cost_2", "cost_tstats", "kp_1", "kp_2", "pts3d_1", "pts3d_2", "depth_1", "depth_2", "mask_1", "mask_2")


def cache_cost_targets(batch):
    """In place on a batch dict: pad cost_1 / cost_2 [P, hw, hw] to 16-byte rows and attach `cost_tstats` (idempotent)."""
    if "cost_tstats" not in batch:
        batch["cost_1"] = ops.pad_teacher_maps(batch["cost_1"])
        batch["cost_2"] = ops.pad_teacher_maps(batch["cost_2"])
        batch["cost_tstats"] = ops.cost_volume_teacher_stats(batch["cost_1"], batch["cost_2"])
    return batch


class TeacherTargetCache:
    """key (any hashable pair id) -> the pair's targets on `device`.  `get(key, producer)` runs `producer()` — the
    teacher forward + target extraction (teacher_glue.extract_vggt_targets / extract_mast3r_targets) — only on a miss.

    Temperature.  The MASt3R cost maps depend on the teacher softmax temperature, which src/finetune_timm_mast3r.py:217-227
    anneals every epoch (1.0 -> 0.5 over 500 epochs in the default yaml: ~1e-3 per epoch), so a cache keyed on the temperature
    alone would miss on every pair of every epoch.  Everything else in an entry (keypoints, 3-D points, depth maps) does not
    depend on it, so an entry whose temperature has moved by more than `temp_tol` refreshes ONLY its cost maps, from the cheapest
    source available:
      1. `cost_recip` kept in the entry — the pre-softmax reciprocal score maps [L, 2, N, N] (view-2 target first, as
         `tgt_attn_map` indexes them; teacher_glue.mast3r_recip_logits): one gd_mast3r_attn_target launch, no teacher call.
         Costs L * 2 * N^2 * 4 bytes per pair (180 MB at N = 1369, L = 12), so it is opt-in (`keep_logits=True` keeps what
         the producer hands over, otherwise the field is dropped);
      2. `cost_producer(temperature) -> (cost_1, cost_2)` given to `get` — e.g. the decoder alone;
      3. the full `producer()`.
    `hits` / `misses` / `cost_refreshes` count the three outcomes.

    Empty pairs.  The producers return None when no keypoint survives (the reference then skips the step:
    src/finetune_timm_vggt.py:585-588, src/finetune_timm_mast3r.py:604-607).  Such a pair is remembered as empty: `get` returns
    None for it without calling the producer again, and `collate` leaves it out of the batch."""

    def __init__(self, device="cuda", max_pairs=None, temp_tol=1e-6, keep_logits=False):
        self.device, self.max_pairs, self.temp_tol, self.keep_logits = device, max_pairs, temp_tol, keep_logits
        self._d = {}
        self.hits = self.misses = self.cost_refreshes = 0

    def __len__(self):
        return len(self._d)

    def nbytes(self):
        return sum(v.numel() * v.element_size() for e in self._d.values() for v in e.values() if isinstance(v, torch.Tensor))

    def _set_costs(self, e, c1, c2):
        c1 = c1[None] if c1.dim() == 2 else c1
        c2 = c2[None] if c2.dim() == 2 else c2
        p1, p2 = ops.pad_teacher_maps(c1.float().to(self.device)), ops.pad_teacher_maps(c2.float().to(self.device))
        e["cost_1"], e["cost_2"], e["cost_tstats"] = p1[0], p2[0], ops.cost_volume_teacher_stats(p1, p2)[0]

    def put(self, key, targets, temperature=None):
        if targets is None:
            e = {"_empty": True, "_temperature": temperature}
        else:
            e = {}
            for k, v in targets.items():
                if k == "cost_recip" and not self.keep_logits:
                    continue
                e[k] = v.to(self.device) if isinstance(v, torch.Tensor) else v
            if "cost_tstats" not in e:
                self._set_costs(e, e["cost_1"], e["cost_2"])
            e["_temperature"] = temperature
        if self.max_pairs is not None and len(self._d) >= self.max_pairs and key not in self._d:
            self._d.pop(next(iter(self._d)))           # FIFO eviction: epochs walk the dataset in a fixed order
        self._d[key] = e
        return None if targets is None else e

    def get(self, key, producer=None, temperature=None, cost_producer=None):
        e = self._d.get(key)
        if e is not None and e.get("_empty"):       # keypoints do not depend on the temperature: still empty
            self.hits += 1
            return None
        stale = e is not None and temperature is not None and e["_temperature"] is not None and \
            abs(e["_temperature"] - temperature) > self.temp_tol
        if e is not None and stale and ("cost_recip" in e or cost_producer is not None):
            if "cost_recip" in e:
                from .teacher_glue import _mast3r_target
                tgt = _mast3r_target(e["cost_recip"], temperature)                 # [2, N1, N2]
                c1, c2 = tgt[1], tgt[0]                                           # p2['tgt_attn_map'][1] / [0]
            else:
                c1, c2 = cost_producer(temperature)
            self._set_costs(e, c1, c2)
            e["_temperature"] = temperature
            self.cost_refreshes += 1
            return e
        if e is None or stale:
            if producer is None:
                raise KeyError(key)
            self.misses += 1
            return self.put(key, producer(), temperature)
        self.hits += 1
        return e

    def collate(self, keys, rgb_1, rgb_2):
        """-> batch dict for FinetuneGD.training_step: cached targets of `keys` stacked, keypoints padded with -1 to the
        longest set, `counts` int32 [P].  A pair cached as empty (no keypoint survived) STAYS in the batch with counts = 0 — the
        reference counts such a step as a zero-loss sample (src/finetune_timm_mast3r.py:604-607) and so does training_step's
        `torch.where(counts > 0, ...)` before the mean: the divisor, the rows of rgb_1 / rgb_2 and, under data parallelism, every
        rank's P stay what the caller passed.  Its dense targets are borrowed from a non-empty pair of the batch (they are never
        evaluated into the loss).  None when every pair is empty (the caller skips the step, as the reference does)."""
        live = [self._d[k] for k in keys if not self._d[k].get("_empty")]
        if not live:
            return None
        empty = dict(live[0])
        empty["kp_1"], empty["kp_2"] = live[0]["kp_1"][:0], live[0]["kp_2"][:0]
        if "pts3d_1" in empty:
            empty["pts3d_1"], empty["pts3d_2"] = live[0]["pts3d_1"][:0], live[0]["pts3d_2"][:0]
        es = [empty if self._d[k].get("_empty") else self._d[k] for k in keys]
        n = [int(e["kp_1"].shape[0]) for e in es]
        N = max(max(n), 1)

        def padkp(t, fill):
            out = t.new_full((N,) + tuple(t.shape[1:]), fill)
            out[:t.shape[0]] = t
            return out
        b = {"rgb_1": rgb_1, "rgb_2": rgb_2, "counts": torch.tensor(n, dtype=torch.int32, device=self.device)}
        for k in _PER_PAIR:
            if k not in es[0]:
                continue
            if k in ("kp_1", "kp_2"):
                b[k] = torch.stack([padkp(e[k].float(), -1.0) for e in es])
            elif k in ("pts3d_1", "pts3d_2"):
                b[k] = torch.stack([padkp(e[k].float(), 0.0) for e in es])
            else:
                b[k] = torch.stack([e[k] for e in es])
        return b
