"""Round 6 probes (one process, interleaved rounds; run on the GPU box): python3 tools/probe_r06.py <which> ...
  tn     the adapter weight-gradient contractions of the tf32h step (gd_gemm_tn, one fp32 + one fp16 operand, 87 680 rows) under the TN kernel's
         block-count knob and with its closing atomics switched off (anatomy) — what bounds a 3 TB/s streaming kernel
  step   the benched step (fit_step, 32 pairs) with host/library options given as name=v0,v1 (options.py names, lib.<knob> for csrc knobs)
"""
import os
import sys
import time

import torch

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gd_amd  # noqa: E402,F401
from gd_amd import ops  # noqa: E402
from gd_amd._lib import lib  # noqa: E402


def timeit(fn, warm=5, it=30):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / it


def probe_tn():
    M, D, Bn = 87680, 768, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    dout = torch.randn(M, D, device="cuda", generator=g) * 1e-4
    x2 = torch.randn(M, D, device="cuda", generator=g)
    hd = torch.randn(M, Bn, device="cuda", generator=g).half()
    one = torch.ones(1, device="cuda")
    z_up, z_dn = torch.zeros(D, Bn, device="cuda"), torch.zeros(Bn, D, device="cuda")
    cases = {"g_up   = dOut^T hd   [768, 64]  (fp32 Y, fp16 X)": (lambda: ops.gemm_tn(dout, hd, out=z_up, alpha_dev=one), M * D * 4 + M * Bn * 2),
             "g_down = dhp^T x2    [64, 768]  (fp16 Y, fp32 X)": (lambda: ops.gemm_tn(hd, x2, out=z_dn, alpha_dev=one), M * D * 4 + M * Bn * 2),
             "both fp16            [768, 64]": (lambda: ops.gemm_tn(dout.half(), hd, out=z_up, alpha_dev=one) if False else None, 0)}
    d16 = dout.half()
    cases["both fp16            [768, 64]"] = (lambda: ops.gemm_tn(d16, hd, out=z_up, alpha_dev=one), M * D * 2 + M * Bn * 2)
    for name, (fn, by) in cases.items():
        line = f"{name}:"
        for blocks in (0, 512, 768, 1024, 1536):
            lib().gd_debug_set(b"tn_blocks", blocks)
            t = timeit(fn)
            line += f"  blocks {blocks or 'auto'}: {t * 1e6:6.1f} us {by / t / 1e12:4.2f} TB/s"
        lib().gd_debug_set(b"tn_blocks", 0)
        lib().gd_debug_set(b"gemm_anat", 4)
        t = timeit(fn)
        lib().gd_debug_set(b"gemm_anat", 0)
        line += f"  | no closing atomics: {t * 1e6:6.1f} us"
        print(line, flush=True)


def probe_step(specs, rounds=3):
    import bench
    from gd_amd.options import option, set_option
    dev = torch.device("cuda", 0)
    job = bench.Job("vit_base", "mast3r", os.environ.get("AB_DTYPE", "tf32h"), "shared", 32, 518, 300, dev, 0, 1)

    def timed(steps=6, warm=2):
        for i in range(warm):
            job.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            job.step(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    def setv(name, v):
        if name.startswith("lib."):
            lib().gd_debug_set(name[4:].encode(), int(v))
        else:
            set_option(name, int(v))

    def getv(name):
        return lib().gd_debug_get(name[4:].encode()) if name.startswith("lib.") else option(name)
    timed(2, 2)
    for spec in specs:
        name, vals = spec.split("=")
        vals = [int(v) for v in vals.split(",")]
        keep = getv(name)
        res = {v: [] for v in vals}
        for _ in range(rounds):
            for v in vals:
                setv(name, v)
                res[v].append(timed())
        setv(name, keep)
        print(f"step A/B {name}: " + " | ".join(f"{v}: best {min(r):.2f} ms (all {[round(x, 2) for x in r]})" for v, r in res.items()), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["tn"]
    if "tn" in which:
        probe_tn()
    specs = [w for w in which if "=" in w and not w.startswith("attnbwd:")]
    if specs:
        probe_step(specs)



def probe_attn_bwd(knob, vals, rounds=3):
    """attention backward (dQ + dK/dV as the step runs them), 64 x 12 x 1370 and 8 x 12 x 6401 fp16, under a library knob"""
    for B, N, H in ((64, 1370, 12), (8, 6401, 12)):
        qkv = torch.randn(B * N, 3 * H * 64, device="cuda").half()
        dout = (torch.randn(B * N, H * 64, device="cuda") * 0.1).half()
        o, lse = ops.attention_fwd(qkv, B, N, H)
        res, outs = {v: [] for v in vals}, {}
        for _ in range(rounds):
            for v in vals:
                lib().gd_debug_set(knob.encode(), v)
                res[v].append(timeit(lambda: ops.attention_bwd(qkv, o, dout, lse, B, N, H, vfirst=True), warm=3, it=15))
                outs[v] = ops.attention_bwd(qkv, o, dout, lse, B, N, H, vfirst=True)
        lib().gd_debug_set(knob.encode(), vals[-1])
        d = float((outs[vals[0]].float() - outs[vals[-1]].float()).abs().max()) / float(outs[vals[0]].float().abs().max())
        print(f"attn bwd {B} x {H} x {N} fp16, {knob}: " + " | ".join(f"{v}: {min(r) * 1e6:7.1f} us (all {[round(x * 1e6, 1) for x in r]})" for v, r in res.items())
              + f"   max |difference| / max |dqkv| between {vals[0]} and {vals[-1]}: {d:.2e}", flush=True)


if __name__ == "__main__":
    for w in sys.argv[1:]:
        if w.startswith("attnbwd:"):
            k, vs = w[8:].split("=")
            probe_attn_bwd(k, [int(x) for x in vs.split(",")])
