"""GPU parity: MFMA GEMMs (through the C ABI) against plain torch fp32/fp64 matmul."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _mk(shape, dtype, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, generator=g, device="cuda", dtype=torch.float32).to(dtype)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-6), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 96), (1370, 2304, 768), (77, 8, 768), (513, 768, 3072),
                                   (1000, 64, 768), (640, 768, 64), (200, 136, 8)])
def test_gemm_nt_plain(dtype, tol, M, N, K):
    from gd_amd import ops
    a, w = _mk((M, K), dtype, 1), _mk((N, K), dtype, 2)
    out = ops.gemm_nt(a, w)
    ref = a.double() @ w.double().t()
    assert out.dtype == dtype and rel_err(out, ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-6), (torch.bfloat16, 2e-2)])
def test_gemm_nt_epilogues(dtype, tol):
    from gd_amd import ops
    M, N, K = 333, 200, 160
    a, w = _mk((M, K), dtype, 3), _mk((N, K), dtype, 4)
    bias = _mk((N,), torch.float32, 5)
    lt, lb = _mk((M, 8), torch.float32, 6), _mk((8, N), torch.float32, 7)
    res = _mk((M, N), dtype, 8)
    base = a.double() @ w.double().t() * 0.5 + bias.double() + lt.double() @ lb.double()
    pre = torch.empty(M, N, dtype=dtype, device="cuda")
    out = ops.gemm_nt(a, w, alpha=0.5, bias=bias, lora_t=lt, lora_b=lb, preact=pre, act=1, residual=res)
    assert rel_err(pre, base) < tol
    assert rel_err(out, torch.nn.functional.gelu(base) + res.double()) < tol
    out2 = ops.gemm_nt(a, w, alpha=0.5, bias=bias, act=2)
    assert rel_err(out2, torch.relu(a.double() @ w.double().t() * 0.5 + bias.double())) < tol
    # dact: v * gelu'(src) and v * (src > 0); accumulate into an fp32 output from bf16 operands
    src = _mk((M, N), dtype, 9)
    x = src.double().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    plain = a.double() @ w.double().t()
    assert rel_err(ops.gemm_nt(a, w, dact_src=src, dact=1), plain * x.grad) < tol
    assert rel_err(ops.gemm_nt(a, w, dact_src=src, dact=2), plain * (src.double() > 0)) < tol
    assert rel_err(ops.gemm_nt(a, w, dact_src=src, dact=3), plain * src.double()) < tol
    yy = (plain + bias.double()).requires_grad_(True)
    torch.nn.functional.gelu(yy).sum().backward()
    dgs = torch.empty(M, N, dtype=dtype, device="cuda")
    o3 = ops.gemm_nt(a, w, bias=bias, preact=dgs, act=3)
    assert rel_err(o3, torch.nn.functional.gelu(yy.detach())) < tol
    assert rel_err(dgs, yy.grad) < max(tol, 2e-5)   # GELU' goes through __expf in the f32 path
    acc = _mk((M, N), torch.float32, 10)
    want = acc.double() + plain
    ops.gemm_nt(a, w, out=acc, accumulate=True)
    assert acc.dtype == torch.float32 and rel_err(acc, want) < tol


@pytest.mark.parametrize("M,N,K", [(1500, 512, 256), (20000, 1024, 192), (1100, 768, 64)])
def test_gemm_nt_persistent_epilogues(M, N, K):
    """the 256x256 persistent kernel (bf16 operands, M >= 1024, N >= 256): every compile-time epilogue it is instantiated
    for, ragged last row tile, several tiles per block (20000 x 1024 -> 316 tiles on 256 CUs)."""
    from gd_amd import ops
    dt, tol = torch.bfloat16, 2e-2
    a, w = _mk((M, K), dt, 21), _mk((N, K), dt, 22)
    bias = _mk((N,), torch.float32, 23)
    lt, lb = _mk((M, 8), torch.float32, 24), _mk((8, N), torch.float32, 25)
    res, src = _mk((M, N), dt, 26), _mk((M, N), dt, 27)
    plain = a.double() @ w.double().t()
    gelu = torch.nn.functional.gelu
    assert rel_err(ops.gemm_nt(a, w), plain) < tol
    assert rel_err(ops.gemm_nt(a, w, alpha=0.5, bias=bias), 0.5 * plain + bias.double()) < tol
    assert rel_err(ops.gemm_nt(a, w, out_dtype=torch.float32, bias=bias), plain + bias.double()) < 1e-5
    lora = lt.double() @ lb.double()
    assert rel_err(ops.gemm_nt(a, w, bias=bias, lora_t=lt, lora_b=lb), plain + bias.double() + lora) < tol
    assert rel_err(ops.gemm_nt(a, w, alpha=0.25, lora_t=lt, lora_b=lb), 0.25 * plain + lora) < tol
    assert rel_err(ops.gemm_nt(a, w, bias=bias, act=1), gelu(plain + bias.double())) < tol
    pre = torch.empty(M, N, dtype=dt, device="cuda")
    out = ops.gemm_nt(a, w, bias=bias, preact=pre, act=1)
    assert rel_err(pre, plain + bias.double()) < tol and rel_err(out, gelu(plain + bias.double())) < tol
    x = src.double().requires_grad_(True)
    gelu(x).sum().backward()
    assert rel_err(ops.gemm_nt(a, w, dact_src=src, dact=1), plain * x.grad) < tol
    assert rel_err(ops.gemm_nt(a, w, bias=bias, residual=res), plain + bias.double() + res.double()) < tol
    assert rel_err(ops.gemm_nt(a, w, residual=res), plain + res.double()) < tol
    # act = 3: GELU forward that stores GELU'(v); dact = 3: gate by the stored derivative
    y = (plain + bias.double()).requires_grad_(True)
    gelu(y).sum().backward()
    dg = torch.empty(M, N, dtype=dt, device="cuda")
    out = ops.gemm_nt(a, w, bias=bias, preact=dg, act=3)
    assert rel_err(out, gelu(y.detach())) < tol and rel_err(dg, y.grad) < tol
    assert rel_err(ops.gemm_nt(a, w, dact_src=src, dact=3), plain * src.double()) < tol


@pytest.mark.parametrize("M,N,K", [(1500, 512, 256), (20000, 768, 192)])
def test_gemm_nt_persistent_f32_epilogue_tensors(M, N, K):
    """bf16 operands with fp32 C / preact / dact_src / residual on the persistent kernel (the instantiations the tf32x engine runs its
    3K-wide split operands on): the epilogue arithmetic is fp32 end to end — compared at 1e-5 against the fp64 result of the same
    bf16-rounded operands."""
    from gd_amd import ops
    dt = torch.bfloat16
    a, w = _mk((M, K), dt, 31), _mk((N, K), dt, 32)
    bias = _mk((N,), torch.float32, 33)
    res, src = _mk((M, N), torch.float32, 36), _mk((M, N), torch.float32, 37)
    plain = a.double() @ w.double().t()
    gelu = torch.nn.functional.gelu
    f32 = torch.float32
    assert rel_err(ops.gemm_nt(a, w, out_dtype=f32, bias=bias, residual=res), plain + bias.double() + res.double()) < 1e-5
    assert rel_err(ops.gemm_nt(a, w, out_dtype=f32, dact_src=src, dact=3), plain * src.double()) < 1e-5
    assert rel_err(ops.gemm_nt(a, w, out_dtype=f32, bias=bias, act=1), gelu(plain + bias.double())) < 1e-5
    y = (plain + bias.double()).requires_grad_(True)
    gelu(y).sum().backward()
    dg = torch.empty(M, N, dtype=f32, device="cuda")
    out = ops.gemm_nt(a, w, out_dtype=f32, bias=bias, preact=dg, act=3)
    assert out.dtype == f32 and rel_err(out, gelu(y.detach())) < 1e-5 and rel_err(dg, y.grad) < 1e-5


def _split_matches(got, f):
    """got [M, 3N] bf16 is the [hi | lo | hi] split of f [M, N] f32: the two hi planes are identical, hi is bf16(f) up to the last-ulp
    differences two instantiations of one fp32 epilogue may have (packed vs scalar FMA contraction: <= 1 f32 ulp, which moves a bf16
    rounding on < 1 % of the values), and hi + lo reconstructs f to 2^-16 |f| (a split that lost its lo plane would be at 2^-9)."""
    N = f.shape[1]
    hi, lo, hi2 = got[:, :N], got[:, N:2 * N], got[:, 2 * N:]
    assert torch.equal(hi, hi2)
    assert float((hi != f.bfloat16()).float().mean()) < 0.01
    err = (hi.double() + lo.double() - f.double()).abs()
    assert bool((err <= f.double().abs() * 2.0 ** -16 + 1e-30).all()), float((err / f.double().abs().clamp_min(1e-30)).max())


@pytest.mark.parametrize("M,N,K", [(1300, 320, 192), (2048, 512, 384)])
def test_gemm_nt_split_output_equals_split_of_f32_output(M, N, K):
    """out_split (c_dtype GD_F32X3): the epilogue writes the [hi | lo | hi] operand split of its f32 result — what gd_split3 of the
    f32-output call with the same epilogue gives (GELU, GELU + stored derivative, dact 3), ragged last tiles included; shapes the
    persistent kernel does not serve are refused, never silently rerouted."""
    from gd_amd import ops
    from gd_amd._lib import GdHipError
    dt, f32 = torch.bfloat16, torch.float32
    a, w = _mk((M, K), dt, 51), _mk((N, K), dt, 52)
    bias, src = _mk((N,), f32, 53), _mk((M, N), f32, 54)
    assert ops.split_out_ok(M, N, K)
    got = ops.gemm_nt(a, w, bias=bias, act=1, out_split=True)
    assert got.shape == (M, 3 * N) and got.dtype == dt
    _split_matches(got, ops.gemm_nt(a, w, out_dtype=f32, bias=bias, act=1))
    d0, d1 = torch.empty(M, N, dtype=f32, device="cuda"), torch.empty(M, N, dtype=f32, device="cuda")
    f = ops.gemm_nt(a, w, out_dtype=f32, bias=bias, act=3, preact=d0)
    _split_matches(ops.gemm_nt(a, w, bias=bias, act=3, preact=d1, out_split=True), f)
    assert rel_err(d1, d0) < 1e-6
    _split_matches(ops.gemm_nt(a, w, dact_src=src, dact=3, out_split=True), ops.gemm_nt(a, w, out_dtype=f32, dact_src=src, dact=3))
    # and the product it feeds: split-out . split(w2)^T against the fp64 product of the f32 tensors
    w2 = _mk((256, N), f32, 55)
    h3 = ops.gemm_nt(a, w, bias=bias, act=1, out_split=True)
    hf = ops.gemm_nt(a, w, out_dtype=f32, bias=bias, act=1)
    assert rel_err(ops.gemm_nt(h3, ops.split3(w2, "w"), out_dtype=f32), hf.double() @ w2.double().t()) < 2e-5
    with pytest.raises(GdHipError):
        ops.gemm_nt(a[:512], w, bias=bias, act=1, out_split=True)       # M < 1024: not a persistent-kernel shape
    with pytest.raises(GdHipError):
        ops.gemm_nt(a, w, bias=bias, out_split=True)                    # no plain-epilogue split instantiation


def test_gemm_nt_persistent_ragged_n_and_batched():
    """persistent kernel with a half-empty last column tile (N = 320), and batched with fp32 output (the cost-volume
    backward's G.b products: grid.y = batch)."""
    from gd_amd import ops
    dt = torch.bfloat16
    M, N, K = 1300, 320, 128
    a, w = _mk((M, K), dt, 41), _mk((N, K), dt, 42)
    bias, res = _mk((N,), torch.float32, 43), _mk((M, N), dt, 44)
    lt, lb = _mk((M, 8), torch.float32, 45), _mk((8, N), torch.float32, 46)
    plain = a.double() @ w.double().t()
    assert rel_err(ops.gemm_nt(a, w, bias=bias, residual=res), plain + bias.double() + res.double()) < 2e-2
    assert rel_err(ops.gemm_nt(a, w, bias=bias, lora_t=lt, lora_b=lb), plain + bias.double() + lt.double() @ lb.double()) < 2e-2
    assert rel_err(ops.gemm_nt(a, w, bias=bias, act=1), torch.nn.functional.gelu(plain + bias.double())) < 2e-2
    ab, wb = _mk((3, 1100, 192), dt, 47), _mk((3, 320, 192), dt, 48)
    out = ops.gemm_nt(ab, wb, out_dtype=torch.float32)
    assert out.dtype == torch.float32 and rel_err(out, ab.double() @ wb.double().transpose(1, 2)) < 1e-5


def test_gemm_nt_persistent_schedule_knobs_do_not_change_results():
    """the persistent kernel's schedule options — grouped tile order (gemm_group_m), compute units left to
    collectives (reserve_cus) — change WHEN and WHERE a tile is computed, never its value: outputs are bit-identical to the default schedule, with a
    side tensor and a ragged last tile row in play."""
    from gd_amd import ops
    from gd_amd._lib import lib
    M, N, K = 256 * 37 + 19, 768, 768                      # 114 tiles: fewer than one round at 256 CUs but several per block at reserve_cus = 200
    a, w = _mk((M, K), torch.float16, 71), _mk((N, K), torch.float16, 72) * 0.05
    bias, res = _mk((N,), torch.float32, 73), _mk((M, N), torch.float32, 74)
    run = lambda: ops.gemm_nt(a, w, bias=bias, residual=res, out_dtype=torch.float32)
    ref = run()
    try:
        for name, val in [("gemm_group_m", 4), ("reserve_cus", 200)]:
            assert lib().gd_debug_set(name.encode(), val) == 0
            out = run()
            assert lib().gd_debug_set(name.encode(), 1 if name == "gemm_group_m" else 0) == 0
            assert torch.equal(out, ref), name
    finally:
        for name, val in [("gemm_group_m", 1), ("reserve_cus", 0)]:
            lib().gd_debug_set(name.encode(), val)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1e-2)])
def test_gemm_nt_batched_strided(dtype, tol):
    from gd_amd import ops
    a, w = _mk((3, 150, 72), dtype, 11), _mk((3, 90, 72), dtype, 12)
    assert rel_err(ops.gemm_nt(a, w), a.double() @ w.double().transpose(1, 2)) < tol
    big = _mk((257, 3 * 64), dtype, 13)          # strided view: the "k" slice of a packed qkv
    wv = _mk((40, 64), dtype, 14)
    assert rel_err(ops.gemm_nt(big[:, 64:128], wv), big[:, 64:128].double() @ wv.double().t()) < tol


@pytest.mark.parametrize("ydt,xdt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                     (torch.float32, torch.bfloat16)])
@pytest.mark.parametrize("M,N,K", [(1000, 64, 768), (5000, 8, 2304), (777, 768, 64), (4096, 128, 200)])
def test_gemm_tn(ydt, xdt, M, N, K):
    from gd_amd import ops
    y, x = _mk((M, N), ydt, 15), _mk((M, K), xdt, 16)
    out = ops.gemm_tn(y, x, alpha=0.25)
    assert rel_err(out, 0.25 * y.double().t() @ x.double()) < 5e-6
    ops.gemm_tn(y, x, out=out, alpha=0.25)   # accumulates
    assert rel_err(out, 0.5 * y.double().t() @ x.double()) < 5e-6


def test_gemm_rejects_bad_arguments():
    from gd_amd import ops, _lib
    a = torch.randn(16, 10, device="cuda")   # K*4 = 40 bytes: not a multiple of 16
    with pytest.raises(_lib.GdHipError):
        ops.gemm_nt(a, torch.randn(8, 10, device="cuda"))


@pytest.mark.parametrize("M,D", [(1000, 768), (87, 256), (4100, 1024), (32, 512), (33007, 768), (32800, 1024)])
def test_adapter_fused(M, D):
    """fused bottleneck adapter (utils/model.py:7-25) against fp64 torch: forward (ReLU gate) and the backward-to-input
    form (gate from the saved hidden tile); ragged last 32-row tile; M >= 32768 takes the persistent kernel (weights in
    registers, one block per CU walking tiles)."""
    from gd_amd import ops
    dt = torch.bfloat16
    x = _mk((M, D), dt, 51)
    down, up = _mk((64, D), dt, 52) * 0.05, _mk((D, 64), dt, 53) * 0.05
    assert ops.adapter_fused_supported(x, 64)
    out, hid = ops.adapter_fused(x, down, up)
    h_ref = torch.relu(x.double() @ down.double().t())
    assert rel_err(hid, h_ref) < 1e-2
    assert rel_err(out, x.double() + hid.double() @ up.double().t()) < 1e-2     # the kernel contracts the bf16 hidden tile
    assert rel_err(out - x, h_ref @ up.double().t()) < 5e-2                        # the adapter branch itself
    out2, none = ops.adapter_fused(x, down, up, save_hidden=False)
    assert none is None and torch.equal(out2, out)
    # backward-to-input: dX = dOut + ((dOut @ up) * [h > 0]) @ down
    dout = _mk((M, D), dt, 54)
    up_t, down_t = up.t().contiguous(), down.t().contiguous()
    dx, dh = ops.adapter_fused(dout, up_t, down_t, gate_src=hid)
    dh_ref = (dout.double() @ up.double()) * (hid.double() > 0)
    assert rel_err(dh, dh_ref) < 1e-2
    assert rel_err(dx, dout.double() + dh.double() @ down.double()) < 1e-2
    # same numbers as the two-GEMM formulation it replaces
    hd2 = ops.gemm_nt(x, down, act=2)
    o2 = ops.gemm_nt(hd2, up, residual=x)
    assert rel_err(hid, hd2.double()) < 1e-2 and rel_err(out, o2.double()) < 1e-2


def test_adapter_fused_rejects_unsupported():
    from gd_amd import ops, _lib
    x = torch.randn(64, 384, device="cuda").bfloat16()
    assert not ops.adapter_fused_supported(x, 64)
    with pytest.raises(_lib.GdHipError):
        ops.adapter_fused(x, torch.randn(64, 384, device="cuda").bfloat16(), torch.randn(384, 64, device="cuda").bfloat16())


@pytest.mark.parametrize("M,N,K", [(5000, 8, 2304), (4100, 4, 768), (8192, 8, 128), (4097, 7, 1536)])
def test_gemm_nt_skinny(M, N, K):
    """streaming kernel for N <= 8, M >= 4096 (the LoRA rank projections): fp32 and bf16 outputs, alpha, a strided A view
    (q / v thirds of a packed qkv gradient) and a strided output slice."""
    from gd_amd import ops
    dt = torch.bfloat16
    a, w = _mk((M, K), dt, 61), _mk((N, K), dt, 62)
    ref = a.double() @ w.double().t()
    assert rel_err(ops.gemm_nt(a, w, out_dtype=torch.float32), ref) < 1e-5
    assert rel_err(ops.gemm_nt(a, w, alpha=0.5), 0.5 * ref) < 1e-2
    big = _mk((M, 3 * K), dt, 63)
    sl = big[:, 2 * K:]
    out = torch.zeros(M, 2 * N, dtype=torch.float32, device="cuda")
    ops.gemm_nt(sl, w, out=out[:, N:])
    assert rel_err(out[:, N:], sl.double() @ w.double().t()) < 1e-5 and float(out[:, :N].abs().max()) == 0.0


@pytest.mark.parametrize("M,K,ldx", [(87680, 1536, 2304), (1000, 768, 768), (130, 2048, 3072), (64, 256, 256)])
def test_lora_bwd_fused(M, K, ldx):
    """gd_lora_bwd_fused: dt = dqv . bt^T and gbt += t^T . dqv in one pass over the (dq, dv) gradient block, against fp64 and
    against the two streaming kernels it replaces (ragged last chunk, strided rows, accumulation into a non-zero gbt)."""
    from gd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    buf = torch.randn(M, ldx, generator=g, device="cuda").bfloat16()
    dqv = buf[:, :K]
    t = torch.randn(M, 8, generator=g, device="cuda") * 0.3
    bt = (torch.randn(8, K, generator=g, device="cuda") * 0.05).bfloat16()
    g0 = torch.randn(8, K, generator=g, device="cuda")
    gbt = g0.clone()
    assert ops.lora_bwd_fused_supported(dqv, t, bt, gbt)
    dt = ops.lora_bwd_fused(dqv, t, bt, gbt)
    rdt = dqv.double() @ bt.double().t()
    rg = g0.double() + t.double().t() @ dqv.double()
    assert rel_err(dt, rdt) < 1e-5
    assert rel_err(gbt, rg) < 2e-5
    dt2 = ops.gemm_nt(dqv, bt, out_dtype=torch.float32)
    assert rel_err(dt, dt2) < 1e-5
    # second product alone (bt = NULL): the LoRA-A gradient form
    g2 = g0.clone()
    ops.skinny_tn_mfma(t, dqv, g2)
    assert rel_err(g2, rg) < 2e-5


@pytest.mark.parametrize("M,K,ldx", [(87680, 1536, 2304), (1000, 768, 768), (130, 2048, 3072)])
def test_lora_bwd_fused_fp16_operands_with_device_scales(M, K, ldx):
    """gd_lora_bwd_fused_scaled on fp16 operands (tf32h engine).  (1) the (dq, dv) block carries the step's scale s = 2^12 (its true values ~1e-4:
    below fp16's normal range unscaled): dt = dqv . bt^T and gbt += t^T . dqv come back times 1 / s, within 1e-3 of fp64 on the fp16-rounded
    operands' exact values (t goes in as a high + low fp16 pair: ~2^-22).  (2) the LoRA-A gradient form: a GRADIENT in the t role (values ~1e-5)
    goes in under s and comes out times 1 / s — unscaled it would lose most of its bits to fp16 subnormals."""
    from gd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + K + 1)
    s = torch.tensor([4096.0, 1.0 / 4096.0], device="cuda")
    true = torch.randn(M, ldx, generator=g, device="cuda") * 1e-4
    buf = (true * s[0]).half()                                   # what the attention backward leaves: fp16(dqkv * s)
    dqv = buf[:, :K]
    t = torch.randn(M, 8, generator=g, device="cuda") * 0.3
    bt = (torch.randn(8, K, generator=g, device="cuda") * 0.05).half()
    g0 = torch.randn(8, K, generator=g, device="cuda") * 1e-3
    gbt = g0.clone()
    assert ops.lora_bwd_fused_h_supported(dqv, t, bt, gbt)
    dt = ops.lora_bwd_fused_h(dqv, t, bt, gbt, out_mul=s[1:2])
    xd = dqv.double() / 4096.0
    assert rel_err(dt, xd @ bt.double().t()) < 1e-5
    assert rel_err(gbt - g0, t.double().t() @ xd) < 1e-4 and rel_err(gbt, g0.double() + t.double().t() @ xd) < 1e-4
    # (2) gat = dt^T . y with dt a gradient
    y = torch.randn(M, ldx, generator=g, device="cuda").half()[:, :K]
    dtg = torch.randn(M, 8, generator=g, device="cuda") * 1e-5
    gat = torch.zeros(8, K, device="cuda")
    ops.lora_bwd_fused_h(y, dtg, None, gat, t_mul=s[0:1], out_mul=s[1:2])
    assert rel_err(gat, dtg.double().t() @ y.double()) < 1e-4
    unscaled = torch.zeros(8, K, device="cuda")
    ops.lora_bwd_fused_h(y, dtg, None, unscaled)
    assert rel_err(unscaled, dtg.double().t() @ y.double()) > 10 * rel_err(gat, dtg.double().t() @ y.double())      # why the scale is there


@pytest.mark.parametrize("M,N,K", [(1370, 2304, 768), (5480, 768, 3072), (300, 128, 64)])
def test_split3_product_accuracy(M, N, K):
    """ops.split3 + gemm_nt on the 3K-wide bf16 operands (gd_split3: TF32-class products on the bf16 matrix cores): the planes are
    hi = bf16(x), lo = bf16(x - hi) in the documented order, and the product is within 2e-5 of the fp64 one — well inside TF32's
    ~3e-4 (operands rounded to 10 mantissa bits) and two orders of magnitude inside plain bf16's ~2e-3; epilogue tensors stay fp32."""
    from gd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    a = torch.randn(M, K, generator=g, device="cuda")
    w = torch.randn(N, K, generator=g, device="cuda") * 0.05
    a3, w3 = ops.split3(a, "a"), ops.split3(w, "w")
    hi = a.bfloat16()
    lo = (a - hi.float()).bfloat16()
    assert torch.equal(a3[:, :K], hi) and torch.equal(a3[:, K:2 * K], lo) and torch.equal(a3[:, 2 * K:], hi)
    wh = w.bfloat16()
    assert torch.equal(w3[:, :K], wh) and torch.equal(w3[:, K:2 * K], wh) and torch.equal(w3[:, 2 * K:], (w - wh.float()).bfloat16())
    ref = a.double() @ w.double().t()
    fro = lambda x: float((x.double() - ref).norm() / ref.norm())
    got = ops.gemm_nt(a3, w3, out_dtype=torch.float32)
    assert got.dtype == torch.float32 and fro(got) < 2e-5
    tf32 = lambda x: ((x.view(torch.int32) + 0x1000) & ~0x1FFF).view(torch.float32)
    assert fro(got) < 0.2 * fro(tf32(a.clone()).double() @ tf32(w.clone()).double().t())
    assert fro(got) < 0.02 * fro(ops.gemm_nt(hi, wh, out_dtype=torch.float32))
    bias = torch.randn(N, generator=g, device="cuda")
    res = torch.randn(M, N, generator=g, device="cuda")
    got2 = ops.gemm_nt_x3(a, w3, bias=bias, residual=res)
    assert float((got2.double() - (ref + bias.double() + res.double())).norm() / ref.norm()) < 2e-5


@pytest.mark.parametrize("M,N,K", [(1300, 320, 192), (2048, 768, 768), (300, 64, 136)])
def test_fp16_operand_products_are_tf32_class(M, N, K):
    """The tf32h engine's products: operands rounded to fp16 (11-bit significand = TF32's), fp32 accumulation and epilogue.  The result is the
    fp64 product of the fp16-rounded operands to fp32 round-off, and its error against the product of the ORIGINAL operands is no larger
    than that of TF32 emulated on the same operands (10 explicit mantissa bits, round to nearest).  Persistent (first two shapes) and tile
    kernels (third), with the fp32 epilogue tensors."""
    from gd_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    a = torch.randn(M, K, generator=g, device="cuda")
    w = torch.randn(N, K, generator=g, device="cuda") * 0.05
    ah, wh = ops.cast16(a), ops.cast16(w)
    assert ah.dtype == torch.float16 and torch.equal(ah, a.half()) and torch.equal(wh, w.half())
    ref = a.double() @ w.double().t()
    fro = lambda x, r=ref: float((x.double() - r).norm() / r.norm())
    got = ops.gemm_nt(ah, wh, out_dtype=torch.float32)
    assert got.dtype == torch.float32 and fro(got, ah.double() @ wh.double().t()) < 2e-6
    tf32 = lambda x: ((x.view(torch.int32) + 0x1000) & ~0x1FFF).view(torch.float32)
    assert fro(got) < 1.05 * fro(tf32(a.clone()).double() @ tf32(w.clone()).double().t())
    bias = torch.randn(N, generator=g, device="cuda")
    res = torch.randn(M, N, generator=g, device="cuda")
    exact = ah.double() @ wh.double().t()
    assert fro(ops.gemm_nt(ah, wh, out_dtype=torch.float32, bias=bias, residual=res), exact + bias.double() + res.double()) < 2e-6
    assert fro(ops.gemm_nt(ah, wh, out_dtype=torch.float32, bias=bias, act=1), torch.nn.functional.gelu(exact + bias.double())) < 1e-5


def test_fp16_gradient_operands_carry_a_device_side_scale():
    """A gradient-sized tensor (values ~1e-7: below fp16's normal range) goes through the fp16 product with the power-of-two scale
    of gd_amax_scale — taken and undone on the device (cast16(scale_dev=), gemm_nt(alpha_dev=)) — and comes back TF32-class; the same
    tensor cast without a scale loses everything.  fp16 C output (the next product's operand) saturates instead of overflowing."""
    from gd_amd import ops
    M, N, K = 2048, 768, 3072
    g = torch.Generator(device="cuda").manual_seed(3)
    d = torch.randn(M, K, generator=g, device="cuda") * 1e-7
    d[5, 7] = 3e-4                                                          # one outlier sets the scale; the bulk sits 2^-12 below it
    w = torch.randn(N, K, generator=g, device="cuda") * 0.05
    sc = ops.amax_scale(d, target=64.0)
    s, inv = float(sc[0]), float(sc[1])
    assert s * inv == 1.0 and 32.0 < 3e-4 * s <= 64.0 and abs(round(torch.log2(sc[0]).item()) - torch.log2(sc[0]).item()) < 1e-6
    dh, wh = ops.cast16(d, scale_dev=sc[0:1]), ops.cast16(w)
    ref = d.double() @ w.double().t()
    fro = lambda x: float((x.double() - ref).norm() / ref.norm())
    got = ops.gemm_nt(dh, wh, out_dtype=torch.float32, alpha_dev=sc[1:2])
    tf32 = lambda x: ((x.view(torch.int32) + 0x1000) & ~0x1FFF).view(torch.float32)
    assert fro(got) < 1.05 * fro(tf32(d.clone()).double() @ tf32(w.clone()).double().t())
    assert fro(ops.gemm_nt(ops.cast16(d), wh, out_dtype=torch.float32)) > 0.05          # unscaled: the operand is mostly subnormal / zero
    # fp16 C with fp16 epilogue tensors; saturation
    src = torch.rand(M, N, generator=g, device="cuda").half()
    c16 = ops.gemm_nt(dh, wh, out_dtype=torch.float16, dact_src=src, dact=3)
    want = (dh.double() @ wh.double().t()) * src.double()
    assert c16.dtype == torch.float16 and float((c16.double() - want).norm() / want.norm()) < 5e-4
    big = ops.gemm_nt(ops.cast16(torch.full((1024, 64), 60000.0, device="cuda")), ops.cast16(torch.ones(256, 64, device="cuda")),
                      out_dtype=torch.float16)
    assert bool(torch.isfinite(big).all()) and float(big.max()) == 65504.0


def test_gemm_nt_copy16_writes_the_f32_result_and_its_scaled_fp16_copy():
    """gd_gemm_nt_copy16 (tf32h engine): the residual-stream result in f32 and, from the same epilogue, fp16(sat(result * s)) with s a device
    scalar — bit-identical to the plain call followed by gd_cast_f16; other shapes are refused."""
    from gd_amd import ops
    from gd_amd._lib import GdHipError
    M, N, K = 1300, 768, 192
    a, w = _mk((M, K), torch.float32, 61).half(), _mk((N, K), torch.float32, 62).half()
    res, bias = _mk((M, N), torch.float32, 63), _mk((N,), torch.float32, 64)
    sc = ops.amax_scale(res * 1e-5, 8.0)
    out, c16 = ops.gemm_nt_copy16(a, w, res, bias=bias, alpha_dev=sc[1:2], copy_scale=sc[0:1])
    ref = ops.gemm_nt(a, w, out_dtype=torch.float32, bias=bias, residual=res, alpha_dev=sc[1:2])
    assert torch.equal(out, ref) and c16.dtype == torch.float16
    want = ops.cast16(ref, scale_dev=sc[0:1])
    assert float((c16 != want).float().mean()) < 0.01 and rel_err(c16.float(), want.float()) < 1e-3       # (packed vs scalar multiply: last-ulp differences)
    with pytest.raises(GdHipError):
        ops.gemm_nt_copy16(a[:512], w, res[:512])


@pytest.mark.parametrize("M", [8192, 8192 + 50])
def test_adapter_fused_fp16_operands(M):
    """gd_adapter_fused_h (tf32h engine): fp32 x / out, both products on fp16 operands, in one pass.  Forward (ReLU) and backward-to-input
    (gated by the forward's hidden, dOut under a device-side scale, the scaled fp16 copy of dX beside the fp32 one) against the fp64
    arithmetic of the same fp16-rounded operands; shapes the kernel does not serve are refused."""
    from gd_amd import ops
    from gd_amd._lib import GdHipError
    D, bott = 768, 64
    x = _mk((M, D), torch.float32, 71)
    down, up = (_mk((bott, D), torch.float32, 72) * 0.05).half(), (_mk((D, bott), torch.float32, 73) * 0.05).half()
    out, hid, cp = ops.adapter_fused_h(x, down, up)
    assert out.dtype == torch.float32 and hid.dtype == torch.float16 and cp is None
    h_ref = torch.relu(x.half().double() @ down.double().t())
    assert rel_err(hid, h_ref) < 1e-3
    assert rel_err(out, x.double() + hid.double() @ up.double().t()) < 1e-6          # (the kernel's own fp16 hidden tile feeds its second product)
    # backward-to-input: dX = dOut + ((dOut . up) * [h > 0]) . down
    dout = _mk((M, D), torch.float32, 74) * 1e-6
    sc = ops.amax_scale(dout, 8.0)
    s = float(sc[0])
    dx, dh, dx16 = ops.adapter_fused_h(dout, up.t().contiguous(), down.t().contiguous(), gate_src=hid, in_scale=sc[0:1], alpha_dev=sc[1:2],
                                       copy_scale=sc[0:1], want_copy=True)
    d16 = (dout * s).half().double()
    dh_ref = (d16 @ up.double()) * (hid.double() > 0)
    assert rel_err(dh, dh_ref) < 1e-3
    dx_ref = dout.double() + (dh.double() @ down.double()) / s
    assert rel_err(dx, dx_ref) < 1e-6 and rel_err(dx16.double() / s, dx_ref) < 1e-3 and bool(torch.isfinite(dx16.float()).all())
    with pytest.raises(GdHipError):
        ops.adapter_fused_h(x[:1000], down, up)


@pytest.mark.parametrize("M,D", [(32800, 768), (33007, 768), (8192 + 31, 256)])
def test_adapter_fused_with_the_next_blocks_layernorm(M, D):
    """gd_adapter_fused_h_ln (round 5): the adapter pass that also leaves fp16(LayerNorm(out)) and the row statistics — what the next block's norm1 would
    compute from `out` in a pass of its own.  `out` / hidden bit-identical to gd_adapter_fused_h; the statistics and the normed rows against
    gd_layernorm_fwd on that `out` (same two-reduction arithmetic: the statistics to fp32 rounding, the fp16 rows to one rounding), and against fp64.
    Ragged last tile (33007 = 1031 x 32 + 15), rows with a large mean, a massive channel."""
    from gd_amd import ops
    bott = 64
    x = _mk((M, D), torch.float32, 91)
    x[:, 5] += 30.0
    x[: M // 4] += 4.0
    down, up = (_mk((bott, D), torch.float32, 92) * 0.05).half(), (_mk((D, bott), torch.float32, 93) * 0.05).half()
    g, b = 1.0 + 0.2 * _mk((D,), torch.float32, 94), 0.3 * _mk((D,), torch.float32, 95)
    eps = 1e-6
    out0, hid0, _ = ops.adapter_fused_h(x, down, up)
    out, hid, y16, mean, rstd = ops.adapter_fused_h_ln(x, down, up, g, b, eps)
    assert torch.equal(out, out0) and torch.equal(hid, hid0)
    y_ref, m_ref, r_ref = ops.layernorm_fwd(out, g, b, eps, out_dtype=torch.float16)
    assert rel_err(mean, m_ref) < 1e-6 and rel_err(rstd, r_ref) < 1e-6
    assert float((y16.float() - y_ref.float()).abs().max()) <= 2e-3 * float(y_ref.float().abs().max())       # (one fp16 ulp of the largest entries)
    assert rel_err(y16, y_ref) < 1e-3                            # (max-norm: entries whose fp32 value sits on an fp16 rounding boundary differ by one ulp)
    assert float((y16.float() - y_ref.float()).pow(2).mean().sqrt()) < 5e-5 * float(y_ref.float().pow(2).mean().sqrt()) + 1e-6 or \
        float((y16 != y_ref).float().mean()) < 0.01           # ... and they are rare
    want = torch.nn.functional.layer_norm(out.double(), (D,), g.double(), b.double(), eps)
    assert rel_err(y16, want) < 5e-4


def test_gemm_tn_rounds_an_fp32_operand_to_fp16_in_the_kernel():
    """gd_gemm_tn with one fp16 and one fp32 operand (tf32h weight gradients): the fp32 one is rounded to fp16 on its way into LDS — an fp32 Y
    (a gradient) under the scale 1 / alpha_dev, which alpha undoes — and the result equals the cast-then-contract path."""
    from gd_amd import ops
    M, N, K = 8200, 768, 64
    g = torch.Generator(device="cuda").manual_seed(9)
    y = torch.randn(M, N, generator=g, device="cuda") * 1e-6
    x = torch.randn(M, K, generator=g, device="cuda").half()
    sc = ops.amax_scale(y, 8.0)
    got = ops.gemm_tn(y, x, alpha_dev=sc[1:2])
    ref = ops.gemm_tn(ops.cast16(y, scale_dev=sc[0:1]), x, alpha_dev=sc[1:2])
    exact = y.double().t() @ x.double()
    assert rel_err(got, ref) < 1e-5 and rel_err(got, exact) < 1e-3
    x32 = torch.randn(M, N, generator=g, device="cuda")
    y16 = (torch.randn(M, K, generator=g, device="cuda")).half()
    got = ops.gemm_tn(y16, x32)
    assert rel_err(got, y16.double().t() @ x32.half().double()) < 1e-5


