"""CPU: checkpoint compatibility with the reference.  G19 is a checkpoint written by the reference's own
`FinetuneMASt3RTIMM.on_save_checkpoint` (tools/make_golden_g19.py: unbound call on the reference's Adapter /
DepthAwareFeatureFusion / LoRA Linears; the reverse direction — the reference's on_load_checkpoint reading a FinetuneGD checkpoint —
is asserted inside that script, where the reference is importable)."""
import numpy as np
import os
import torch

from conftest import GOLDEN


def _unflatten():
    d = np.load(os.path.join(GOLDEN, "g19_reference_checkpoint.npz"))
    ck = {}
    for k in d.files:
        parts, cur = k.split("/"), ck
        for p_ in parts[:-1]:
            cur = cur.setdefault(p_, {})
        cur[parts[-1]] = torch.from_numpy(d[k])
    return ck


def test_reference_checkpoint_loads_and_key_layout_matches():
    import gd_amd  # noqa: F401
    from gd_amd.finetune import FinetuneGD
    ck = _unflatten()
    eng = FinetuneGD(r=4, backbone="vit_tiny_test", patch_size=14, img_size=56, variant="mast3r", dtype="f32")
    eng.on_load_checkpoint(ck)
    for i, l in enumerate(eng.w_As):
        assert torch.equal(l.weight, ck[f"w_a_{i:03d}"])
    for i, l in enumerate(eng.w_Bs):
        assert torch.equal(l.weight, ck[f"w_b_{i:03d}"])
    for i, a in enumerate(eng.adapters):
        assert torch.equal(a.down.weight, ck[f"adapter_{i:03d}"]["down.weight"]) and torch.equal(a.up.weight, ck[f"adapter_{i:03d}"]["up.weight"])
    for k, v in eng.depth_diff_head.state_dict().items():
        assert torch.equal(v, ck["depth_diff_head"][k]), k
    assert torch.equal(eng.refine_conv.weight, ck["state_dict"]["refine_conv"]["weight"])
    # what FinetuneGD writes has exactly the reference's keys (+ its own optimizer / epoch entries), same shapes, same nesting
    mine = eng.on_save_checkpoint({})

    def layout(d, pre=""):
        out = {}
        for k, v in d.items():
            if isinstance(v, dict):
                out.update(layout(v, pre + k + "/"))
            elif isinstance(v, torch.Tensor):
                out[pre + k] = tuple(v.shape)
        return out
    ref_l, my_l = layout(ck), layout(mine)
    extra = {k for k in my_l if k not in ref_l}
    assert all(k.startswith("gd_optimizer_state") for k in extra), extra
    assert {k: v for k, v in my_l.items() if k in ref_l} == ref_l


def test_saved_checkpoint_owns_its_tensors_and_optimizer_state_survives_the_resume_order():
    """(a) every tensor of a saved checkpoint — nested state_dicts included — is a copy, not a view of the flat parameter buffer
    (a later optimiser step must not reach into a checkpoint that is still in memory); (b) loading BEFORE configure_optimizers,
    the natural resume order, keeps the AdamW moments and the step count; (c) a mismatching optimizer state raises."""
    import pytest
    import gd_amd  # noqa: F401
    from gd_amd.finetune import FinetuneGD
    kw = dict(r=4, backbone="vit_tiny_test", patch_size=14, img_size=56, variant="mast3r", dtype="f32")
    eng = FinetuneGD(**kw)
    flat = eng.configure_optimizers()
    g = torch.Generator().manual_seed(5)
    flat["m"].copy_(torch.randn(flat["m"].shape, generator=g))
    flat["v"].copy_(torch.rand(flat["v"].shape, generator=g))
    flat["step"] = 17
    ck = eng.on_save_checkpoint({})
    lo, hi = flat["p"].data_ptr(), flat["p"].data_ptr() + flat["p"].numel() * 4

    def walk(d):
        for v in d.values():
            if isinstance(v, dict):
                yield from walk(v)
            elif isinstance(v, torch.Tensor):
                yield v
    tensors = list(walk(ck))
    assert len(tensors) > 20
    for t in tensors:
        assert not (lo <= t.data_ptr() < hi), "a checkpoint tensor aliases the live flat parameter buffer"
        assert t.untyped_storage().nbytes() <= max(t.numel() * t.element_size(), 8) + 64
    before = ck["adapter_000"]["down.weight"].clone()
    with torch.no_grad():
        flat["p"].add_(1.0)
    assert torch.equal(ck["adapter_000"]["down.weight"], before)
    other = FinetuneGD(**kw)
    other.on_load_checkpoint(ck)                       # no optimiser yet
    f2 = other.configure_optimizers()
    assert f2["step"] == 17 and torch.equal(f2["m"], ck["gd_optimizer_state"]["exp_avg"]) and torch.equal(f2["v"], ck["gd_optimizer_state"]["exp_avg_sq"])
    bad = dict(ck, gd_optimizer_state=dict(ck["gd_optimizer_state"], numel=ck["gd_optimizer_state"]["numel"] + 4))
    third = FinetuneGD(**kw)
    third.configure_optimizers()
    with pytest.raises(ValueError):
        third.on_load_checkpoint(bad)
