"""CPU, build container only (skipped where /root/reference is absent, i.e. on the GPU box): the teacher runner's hooks against
the REFERENCE's own VGGT aggregator (vggt/models/aggregator.py, random-init tiny configuration).
  * QKCapture picks up exactly the q / k the selected global blocks multiply (shared RoPE module, q/k-norm on);
  * rebuilding the cross-view maps from the captured q, k (oracle formula = what gd_cross_view_attn computes on the GPU,
    tests/test_gpu_teacher_glue.py) reproduces the aggregator's own `attn_mean`;
  * under `_no_attention_maps` the aggregator's token outputs are unchanged while no [B, H, n, n] map is formed."""
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "vggt")), reason="reference checkout not present")


def test_qk_capture_against_reference_aggregator():
    sys.path.insert(0, REF)
    try:
        from vggt.models.aggregator import Aggregator
    finally:
        sys.path.remove(REF)
    import gd_amd  # noqa: F401
    import gd_oracle as O
    from gd_amd.teacher_runner import QKCapture, _no_attention_maps
    torch.manual_seed(0)
    agg = Aggregator(img_size=56, patch_size=14, embed_dim=128, depth=3, num_heads=2, patch_embed="conv", attn_indices=[1, 2],
                     temperature=0.8).eval()
    img = torch.rand(1, 2, 3, 56, 56)
    with torch.no_grad():
        ref_tokens, ps_idx, ref_attn = agg(img)                       # the reference path: per-head maps materialised
    sel = [agg.global_blocks[i].attn for i in agg.attn_indices]
    with torch.no_grad(), QKCapture(sel) as cap, _no_attention_maps(sel):
        tokens, ps2, placeholder = agg(img)
    assert ps2 == ps_idx == 5 and placeholder.numel() == 1
    for a, b in zip(tokens, ref_tokens):
        assert torch.equal(a, b)                                      # the teacher's outputs do not change
    qk = cap.pairs()
    assert len(qk) == 2 and qk[0][0].shape == (1, 2, 2 * (5 + 16), 64)
    maps = sum(O.cross_view_attention_maps(q.double(), k.double(), sel[0].scale, 0.8, prefix=5) for q, k in qk) / len(qk)
    want = ref_attn.mean(dim=1)                                       # src/finetune_timm_vggt.py:390-392: head mean of attn.chunk(2)
    assert float((maps - want.double()).abs().max()) < 1e-6
