"""GPU parity: fused cost-volume KL (C ABI) against the oracle and the reference-generated fixtures."""
import pytest
import torch

import gd_oracle as O
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _oracle(f1, f2, t1, t2, m1, m2, variant):
    """per-pair loss + grads from the CPU oracle (fp64)."""
    losses, g1, g2 = [], [], []
    for p in range(f1.shape[0]):
        a = f1[p:p + 1].double().cpu().requires_grad_(True)
        b = f2[p:p + 1].double().cpu().requires_grad_(True)
        l = O.cost_volume_kl(a, b, t1[p:p + 1].double().cpu(), t2[p:p + 1].double().cpu(), m1[p].cpu(), m2[p].cpu(), variant)
        l.backward()
        losses.append(l.detach())
        g1.append(a.grad[0])
        g2.append(b.grad[0])
    return torch.stack(losses), torch.stack(g1), torch.stack(g2)


@pytest.mark.parametrize("variant", ["vggt", "mast3r"])
def test_golden_fixture(variant):
    from gd_amd import ops
    g = load_golden(f"g06_cost_{variant}")
    f1 = g["f1"][None].cuda().requires_grad_(True)
    f2 = g["f2"][None].cuda().requires_grad_(True)
    loss = ops.cost_volume_kl(f1, f2, g["t1"][None].cuda(), g["t2"][None].cuda(), g["m1"][None].cuda(),
                              g["m2"][None].cuda(), variant)
    assert abs(loss.item() - g["loss"]) < 2e-5 * abs(g["loss"])
    loss.sum().backward()
    assert rel_err(f1.grad[0], g["g1"]) < 2e-4 and rel_err(f2.grad[0], g["g2"]) < 2e-4


def _teacher(layout, t1, t2):
    """'plain': [P,hw,hw] maps, teacher statistics recomputed inside the op; 'cached': rows padded to 16 bytes + the
    statistics computed once (what teacher_cache.TeacherTargetCache holds per pair) — the single-pass persistent kernel."""
    from gd_amd import ops
    if layout == "plain":
        return t1, t2, None
    p1, p2 = ops.pad_teacher_maps(t1), ops.pad_teacher_maps(t2)
    if p1.shape[-1] != t1.shape[-1]:          # poison the pad entries' neighbours? no: pads must be finite (zero) by contract
        assert float(p1[..., t1.shape[-1]:].abs().max()) == 0.0
    return p1, p2, ops.cost_volume_teacher_stats(p1, p2)


@pytest.mark.parametrize("layout", ["plain", "cached"])
@pytest.mark.parametrize("variant", ["vggt", "mast3r"])
@pytest.mark.parametrize("P,hw,C", [(2, 100, 64), (1, 333, 96), (2, 672, 128), (3, 257, 160)])
def test_vs_oracle_f32(variant, P, hw, C, layout):
    from gd_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(hw)
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t1[0, 3] = 0.0                                  # a teacher row whose sum hits the 1e-8 clamp
    m1 = torch.rand(P, hw, generator=gen, device="cuda") > 0.3
    m2 = torch.rand(P, hw, generator=gen, device="cuda") > 0.3
    m2[0] = False                                   # a fully masked direction
    c1, c2, ts = _teacher(layout, t1, t2)
    loss = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts)
    w = torch.tensor([1.0, 0.5, 2.0][:P], device="cuda")
    (loss * w).sum().backward()
    ol, og1, og2 = _oracle(f1.detach(), f2.detach(), t1, t2, m1, m2, variant)
    assert rel_err(loss, ol) < 1e-5
    assert rel_err(f1.grad, og1 * w.cpu()[:, None, None]) < 1e-4
    assert rel_err(f2.grad, og2 * w.cpu()[:, None, None]) < 1e-4


def test_bf16_full_size():
    """hw = 1369 (37x37), C = 768, bf16 features: loss within 1e-3 rel of the fp64 oracle on the
    same (bf16-rounded) inputs; gradients to bf16 accuracy."""
    from gd_amd import ops
    P, hw, C = 1, 1369, 768
    gen = torch.Generator(device="cuda").manual_seed(7)
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").bfloat16().requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").bfloat16().requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    m1 = torch.rand(P, hw, generator=gen, device="cuda") > 0.3
    m2 = torch.rand(P, hw, generator=gen, device="cuda") > 0.3
    loss = ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "vggt")
    loss.sum().backward()
    ol, og1, og2 = _oracle(f1.detach().float(), f2.detach().float(), t1, t2, m1, m2, "vggt")
    assert rel_err(loss, ol) < 1e-3
    assert rel_err(f1.grad.float(), og1) < 3e-2 and rel_err(f2.grad.float(), og2) < 3e-2
    # the cached-target layout (padded rows + teacher statistics): the persistent single-pass kernel, same numbers
    c1, c2, ts = _teacher("cached", t1, t2)
    loss2 = ops.cost_volume_kl(f1.detach(), f2.detach(), c1, c2, m1, m2, "vggt", tstats=ts)
    assert rel_err(loss2, ol) < 1e-3 and rel_err(loss2, loss) < 1e-5


@pytest.mark.parametrize("variant,dtype,P,hw,C", [("mast3r", torch.float32, 9, 672, 96), ("vggt", torch.bfloat16, 3, 1369, 1024),
                                                  ("mast3r", torch.bfloat16, 5, 768, 384)])
def test_persistent_kernel_many_tiles_per_block(variant, dtype, P, hw, C):
    """The persistent forward with FEW blocks (gd_debug_set("cv_grid", 8)), so every block walks dozens of tiles: ring slots, the tile-parity
    statistics / partial-sum buffers and the one-tile-ahead teacher prefetch all wrap around many times.  f32 operands
    against the fp64 oracle; bf16 (ViT-L width C = 1024 at hw = 1369; ViT-S width at the MASt3R grid hw = 768) against it on the
    bf16-rounded inputs.  Running twice gives bit-identical losses (fixed summation order)."""
    from gd_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(hw + C)
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").to(dtype)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").to(dtype)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    m1 = torch.rand(P, hw, generator=gen, device="cuda") > 0.3
    m2 = torch.rand(P, hw, generator=gen, device="cuda") > 0.3
    c1, c2, ts = _teacher("cached", t1, t2)
    full = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts)
    from gd_amd._lib import check, lib
    check(lib().gd_debug_set(b"cv_grid", 8), "gd_debug_set")
    try:
        few = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts)
        few2 = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts)
    finally:
        lib().gd_debug_set(b"cv_grid", 0)
    assert torch.equal(few, few2)
    assert rel_err(few, full) < 1e-6
    ol, _, _ = _oracle(f1.float(), f2.float(), t1, t2, m1, m2, variant)
    assert rel_err(few, ol) < (1e-5 if dtype == torch.float32 else 1e-3)


def _debug(name, value):
    from gd_amd._lib import check, lib
    check(lib().gd_debug_set(name.encode(), value), "gd_debug_set")


@pytest.mark.parametrize("mode,P,hw,C,masked", [("f32", 3, 672, 96, True), ("bf16", 3, 1369, 768, True), ("bf16", 5, 700, 384, False),
                                                ("h", 3, 1369, 768, True), ("h", 2, 1369, 1024, False)])
def test_persistent_backward_matches_the_tile_kernel_and_the_oracle(mode, P, hw, C, masked):
    """The dense backward's G pass on the persistent kernel (cv_fwd_persist_kernel<.., BWD>: G1 / G2 stored from the accumulator layout; the two
    contractions as ONE batch of 2P) against (a) the one-tile-per-block kernel it replaces (GD_CV_PERSIST=0) and (b) the fp64 oracle — with few
    blocks (cv_grid 8: every block walks many tiles, the keep bits and the teacher prefetch wrap), row masks on both views, a ragged last tile
    (hw = 1369, 700, 672 = 5.25 tiles), per-pair loss gradients four decades apart, and bit-identical results run to run."""
    from gd_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(hw + C + P)
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").to(dt).requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").to(dt).requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    if masked:
        m1 = torch.rand(P, hw, generator=gen, device="cuda") > 0.4
        m2 = torch.rand(P, hw, generator=gen, device="cuda") > 0.4
    else:
        m1 = m2 = torch.ones(P, hw, dtype=torch.bool, device="cuda")
    c1, c2, ts = _teacher("cached", t1, t2)
    kw = dict(tstats=ts)
    if mode == "h":
        kw.update(inv_norms=(1.0 / f1.detach().norm(dim=-1).clamp_min(1e-12), 1.0 / f2.detach().norm(dim=-1).clamp_min(1e-12)), x3="h")
    wts = torch.logspace(0, -4, P, device="cuda")

    def grads():
        f1.grad = f2.grad = None
        (ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, "vggt", **kw) * wts).sum().backward()
        return f1.grad.clone(), f2.grad.clone()

    _debug("cv_grid", 8)
    try:
        g1, g2 = grads()
        h1, h2 = grads()
    finally:
        _debug("cv_grid", 0)
    assert torch.equal(g1, h1) and torch.equal(g2, h2)
    _debug("cv_persist", 0)
    try:
        o1, o2 = grads()
    finally:
        _debug("cv_persist", 1)
    tol = {"f32": 2e-5, "bf16": 1e-2, "h": 2e-3}[mode]       # (the two kernels round G at different points: exp(S) / Z against exp(S - log Z))
    for p in range(P):
        assert rel_err(g1[p], o1[p]) < tol and rel_err(g2[p], o2[p]) < tol, p
    _, og1, og2 = _oracle(f1.detach().float(), f2.detach().float(), t1, t2, m1, m2, "vggt")
    otol = {"f32": 1e-4, "bf16": 2e-2, "h": 4e-3}[mode]
    w = wts.double().view(P, 1, 1).cpu()
    for p in range(P):
        assert rel_err(g1[p], (og1.cpu() * w)[p]) < otol and rel_err(g2[p], (og2.cpu() * w)[p]) < otol, p


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_producer_side_row_norms(dtype):
    """ops.tap_mean(with_norm=True) hands out the inverse L2 norms of the rows it writes, and cost_volume_kl(inv_norms=...) uses them
    instead of its own pass over the features: same features, same norms, bit-identical loss and gradients as the self-contained
    op (gd_tap_mean_norm_fwd / gd_cost_volume_kl_fwd_prenorm)."""
    from gd_amd import ops
    P, hw, C, pre = 3, 672, 256, 1
    gen = torch.Generator(device="cuda").manual_seed(11)
    taps = [torch.randn(2 * P, hw + pre, C, generator=gen, device="cuda").to(dtype).requires_grad_(True) for _ in range(4)]
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    m1 = torch.rand(P, hw, generator=gen, device="cuda") > 0.5
    m2 = torch.rand(P, hw, generator=gen, device="cuda") > 0.5
    c1, c2, ts = _teacher("cached", t1, t2)
    f_plain = ops.tap_mean(taps, prefix=pre)
    f, inv = ops.tap_mean(taps, prefix=pre, with_norm=True)
    assert torch.equal(f, f_plain) and inv.shape == (2 * P, hw) and not inv.requires_grad
    want = 1.0 / f.detach().float().norm(dim=-1).clamp_min(1e-12)
    assert rel_err(inv, want) < 1e-6
    la = ops.cost_volume_kl(f_plain[:P], f_plain[P:], c1, c2, m1, m2, "mast3r", tstats=ts)
    la.sum().backward()
    ga = [t.grad.clone() for t in taps]
    for t in taps:
        t.grad = None
    lb = ops.cost_volume_kl(f[:P], f[P:], c1, c2, m1, m2, "mast3r", tstats=ts, inv_norms=(inv[:P], inv[P:]))
    lb.sum().backward()
    assert torch.equal(la, lb)
    for a, t in zip(ga, taps):
        assert torch.equal(a, t.grad)
    ol, _, _ = _oracle(f[:P].detach().float(), f[P:].detach().float(), t1, t2, m1, m2, "mast3r")
    assert rel_err(lb, ol) < (1e-5 if dtype == torch.float32 else 1e-3)


@pytest.mark.parametrize("variant", ["mast3r", "vggt"])
def test_split_precision_forward(variant):
    """cost_volume_kl(x3=True) on fp32 features (tf32x engine): the similarity matrix of the FORWARD as a split-precision bf16 product
    (gd_split3 + the bf16 tile kernel at K = 3C) — loss within 1e-5 of the fp64 oracle (the bf16 kernel on bf16 features: 1e-3), gradients
    (f32 backward reading the forward's saved logZ) within 1e-4."""
    from gd_amd import ops
    P, hw, C = 2, 1369, 768
    gen = torch.Generator(device="cuda").manual_seed(5)
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    m1 = torch.rand(P, hw, generator=gen, device="cuda") > 0.5
    m2 = torch.rand(P, hw, generator=gen, device="cuda") > 0.5
    c1, c2, ts = _teacher("cached", t1, t2)
    inv1 = 1.0 / f1.detach().norm(dim=-1).clamp_min(1e-12)
    inv2 = 1.0 / f2.detach().norm(dim=-1).clamp_min(1e-12)
    loss = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts, inv_norms=(inv1, inv2), x3=True)
    loss.sum().backward()
    ol, og1, og2 = _oracle(f1.detach(), f2.detach(), t1, t2, m1, m2, variant)
    assert rel_err(loss, ol) < 1e-5
    assert rel_err(f1.grad, og1) < 1e-4 and rel_err(f2.grad, og2) < 1e-4



@pytest.mark.parametrize("variant", ["mast3r", "vggt"])
def test_fp16_operand_forward_and_backward(variant):
    """cost_volume_kl(x3="h") on fp32 features (tf32h engine): S from fp16 copies of the features (forward and the backward's recompute), G = dloss/dS
    as fp16 under a device-side power-of-two scale (|G| ~ 1e-9 unscaled), both contractions on the fp16 MFMA kernels, the gradient through the
    normalisation in fp32: loss within 1e-4, gradients within 2e-3 of the fp64 oracle (bf16 features: 1e-3 / 1e-2), finite for a zero loss gradient."""
    from gd_amd import ops
    P, hw, C = 2, 1369, 768
    gen = torch.Generator(device="cuda").manual_seed(5)
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    m1 = torch.rand(P, hw, generator=gen, device="cuda") > 0.5
    m2 = torch.rand(P, hw, generator=gen, device="cuda") > 0.5
    c1, c2, ts = _teacher("cached", t1, t2)
    inv1 = 1.0 / f1.detach().norm(dim=-1).clamp_min(1e-12)
    inv2 = 1.0 / f2.detach().norm(dim=-1).clamp_min(1e-12)
    loss = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts, inv_norms=(inv1, inv2), x3="h")
    (loss * torch.tensor([1.0, 1e-3], device="cuda")).sum().backward()          # pairs with very different loss gradients share one scale
    ol, og1, og2 = _oracle(f1.detach(), f2.detach(), t1, t2, m1, m2, variant)
    assert rel_err(loss, ol) < 1e-4
    w = torch.tensor([1.0, 1e-3], dtype=torch.float64).view(P, 1, 1)
    assert f1.grad.dtype == torch.float32 and bool(torch.isfinite(f1.grad).all())
    for p in range(P):
        assert rel_err(f1.grad[p], (og1 * w)[p]) < (2e-3 if p == 0 else 2e-2) and rel_err(f2.grad[p], (og2 * w)[p]) < (2e-3 if p == 0 else 2e-2), p
    f1.grad = f2.grad = None
    loss = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts, inv_norms=(inv1, inv2), x3="h")
    (loss * 0.0).sum().backward()
    assert float(f1.grad.abs().max()) == 0.0 and bool(torch.isfinite(f2.grad).all())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("variant", ["mast3r", "vggt"])
def test_kept_row_forward_matches_the_dense_sweep(dtype, variant):
    """cost_volume_kl(kept_rows_max=...) with sparse row masks (keypoint-patch masks: <= N_kp rows kept per view): both directions as compacted row
    problems (gd_cost_volume_kl_fwd_rows) — same loss as the hw x hw sweep to fp32 summation order, same gradients (the backward reads the saved
    logZ / W of the kept rows), the fp64 oracle within the dtype's tolerance; ragged kept counts per pair and view, one view of one pair with NO kept row."""
    from gd_amd import ops
    P, hw, C = 3, 1369, 256
    gen = torch.Generator(device="cuda").manual_seed(17)
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").to(dtype).requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").to(dtype).requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    m1 = torch.zeros(P, hw, dtype=torch.bool, device="cuda")
    m2 = torch.zeros(P, hw, dtype=torch.bool, device="cuda")
    for p, (k1, k2) in enumerate([(300, 211), (129, 0), (7, 256)]):
        m1[p, torch.randperm(hw, generator=gen, device="cuda")[:k1]] = True
        m2[p, torch.randperm(hw, generator=gen, device="cuda")[:k2]] = True
    c1, c2, ts = _teacher("cached", t1, t2)
    inv1 = 1.0 / f1.detach().float().norm(dim=-1).clamp_min(1e-12)
    inv2 = 1.0 / f2.detach().float().norm(dim=-1).clamp_min(1e-12)
    dense = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts, inv_norms=(inv1, inv2))
    dense.sum().backward()
    g1, g2 = f1.grad.clone(), f2.grad.clone()
    f1.grad = f2.grad = None
    rows = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts, inv_norms=(inv1, inv2), kept_rows_max=300)
    rows.sum().backward()
    assert rel_err(rows, dense) < 1e-6
    assert rel_err(f1.grad, g1) < (1e-5 if dtype == torch.float32 else 2e-2) and rel_err(f2.grad, g2) < (1e-5 if dtype == torch.float32 else 2e-2)
    ol, og1, og2 = _oracle(f1.detach().float(), f2.detach().float(), t1, t2, m1, m2, variant)
    assert rel_err(rows, ol) < (1e-5 if dtype == torch.float32 else 1e-3)
    # the kept-row backward (gd_cost_volume_kl_bwd_rows: G for the kept rows of each direction only) against the fp64 oracle, and the dense backward
    # behind the kept-row forward (GD_CV_BWD_ROWS=0) against both
    gtol = 1e-4 if dtype == torch.float32 else 1e-2
    assert rel_err(f1.grad, og1) < gtol and rel_err(f2.grad, og2) < gtol
    assert float(f2.grad[1].abs().max()) > 0 and float(f1.grad[1].float().abs().max()) > 0      # pair 1: view 2 keeps nothing, view 1's direction still reaches both
    from gd_amd.options import set_option
    r1, r2 = f1.grad.clone(), f2.grad.clone()
    f1.grad = f2.grad = None
    old = set_option("cv_bwd_rows", 0)
    try:
        ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts, inv_norms=(inv1, inv2), kept_rows_max=300).sum().backward()
    finally:
        set_option("cv_bwd_rows", old)
    dtol = 1e-5 if dtype == torch.float32 else 5e-3      # (logZ of the two forwards differs in fp32 summation order: bf16 G rounds differently)
    assert rel_err(f1.grad, g1) < dtol and rel_err(f2.grad, g2) < dtol and rel_err(r1, f1.grad) < (1e-5 if dtype == torch.float32 else 2e-2)
    # few persistent blocks: every block walks many tiles
    from gd_amd._lib import lib
    lib().gd_debug_set(b"cv_grid", 8)
    try:
        again = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts, inv_norms=(inv1, inv2), kept_rows_max=300)
    finally:
        lib().gd_debug_set(b"cv_grid", 0)
    assert torch.equal(again, rows)


@pytest.mark.parametrize("variant", ["mast3r", "vggt"])
def test_kept_row_backward_fp16_operands(variant):
    """The tf32h engine's cost volume with sparse row masks at the step's shape (hw 1369, C 768, <= 384 kept rows): kept-row forward and kept-row
    backward on the fp16 copies, G under the device-side scale — gradients within 2e-3 of the fp64 oracle and of the dense fp16 backward, pairs with
    loss gradients three orders of magnitude apart under one scale, exact zeros for a zero loss gradient."""
    from gd_amd import ops
    P, hw, C = 2, 1369, 768
    gen = torch.Generator(device="cuda").manual_seed(23)
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    m1 = torch.zeros(P, hw, dtype=torch.bool, device="cuda")
    m2 = torch.zeros(P, hw, dtype=torch.bool, device="cuda")
    for p, (k1, k2) in enumerate([(384, 257), (301, 128)]):
        m1[p, torch.randperm(hw, generator=gen, device="cuda")[:k1]] = True
        m2[p, torch.randperm(hw, generator=gen, device="cuda")[:k2]] = True
    c1, c2, ts = _teacher("cached", t1, t2)
    inv1 = 1.0 / f1.detach().norm(dim=-1).clamp_min(1e-12)
    inv2 = 1.0 / f2.detach().norm(dim=-1).clamp_min(1e-12)
    wv = torch.tensor([1.0, 1e-3], device="cuda")
    dense = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts, inv_norms=(inv1, inv2), x3="h")
    (dense * wv).sum().backward()
    d1, d2 = f1.grad.clone(), f2.grad.clone()
    f1.grad = f2.grad = None
    loss = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts, inv_norms=(inv1, inv2), x3="h", kept_rows_max=384)
    (loss * wv).sum().backward()
    ol, og1, og2 = _oracle(f1.detach(), f2.detach(), t1, t2, m1, m2, variant)
    assert rel_err(loss, ol) < 1e-4 and rel_err(loss, dense) < 1e-6
    w = wv.double().cpu().view(P, 1, 1)
    assert f1.grad.dtype == torch.float32 and bool(torch.isfinite(f1.grad).all())
    for p in range(P):
        tol = 2e-3 if p == 0 else 2e-2
        assert rel_err(f1.grad[p], (og1 * w)[p]) < tol and rel_err(f2.grad[p], (og2 * w)[p]) < tol, p
        assert rel_err(f1.grad[p], d1[p]) < tol and rel_err(f2.grad[p], d2[p]) < tol, p
    f1.grad = f2.grad = None
    loss = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, variant, tstats=ts, inv_norms=(inv1, inv2), x3="h", kept_rows_max=384)
    (loss * 0.0).sum().backward()
    assert float(f1.grad.abs().max()) == 0.0 and float(f2.grad.abs().max()) == 0.0


def test_kept_row_bound_exceeded_is_loud():
    """kept_rows_max is the caller's promise; a pair whose mask keeps MORE rows than the bound gets a NaN loss and NaN gradients (never a silently
    truncated sum), the other pairs are unaffected."""
    from gd_amd import ops
    P, hw, C = 2, 1369, 256
    gen = torch.Generator(device="cuda").manual_seed(29)
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    m1 = torch.zeros(P, hw, dtype=torch.bool, device="cuda")
    m2 = torch.zeros(P, hw, dtype=torch.bool, device="cuda")
    for p, (k1, k2) in enumerate([(100, 128), (129, 90)]):      # bound 100 -> kcap 128: pair 1 / view 1 keeps one row too many
        m1[p, torch.randperm(hw, generator=gen, device="cuda")[:k1]] = True
        m2[p, torch.randperm(hw, generator=gen, device="cuda")[:k2]] = True
    c1, c2, ts = _teacher("cached", t1, t2)
    inv1 = 1.0 / f1.detach().norm(dim=-1).clamp_min(1e-12)
    inv2 = 1.0 / f2.detach().norm(dim=-1).clamp_min(1e-12)
    dense = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, "mast3r", tstats=ts, inv_norms=(inv1, inv2))
    loss = ops.cost_volume_kl(f1, f2, c1, c2, m1, m2, "mast3r", tstats=ts, inv_norms=(inv1, inv2), kept_rows_max=100)
    assert rel_err(loss[0], dense[0]) < 1e-6 and bool(torch.isnan(loss[1]))
    loss.sum().backward()
    assert bool(torch.isfinite(f1.grad[0]).all()) and bool(torch.isnan(f1.grad[1]).any())
