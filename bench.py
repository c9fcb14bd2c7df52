#!/usr/bin/env python
"""Benchmark of the geometric-distillation student step (BASELINE.json metric: image-pairs/s at 518^2,
ViT-B/14 + LoRA) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process only LAUNCHES `python -m torch.distributed.run --nproc-per-node N
bench.py ...` as a child (before any GPU call) and exits with its code — one rank per GPU over RCCL; under torchrun
(WORLD_SIZE set) it is a rank.  The reference's counterpart is Lightning DDP with devices=-1 (src/main.py:147-151).

A "step" = one pass of the hot path over one batch of synthetic pairs per GPU: student ViT forward (one
shared 37x37-token forward per image, taps 4-7 + final), the three distillation losses against synthetic
teacher targets, backward through LoRA/adapters/refine_conv/depth head, flat-gradient all-reduce (N>1), global-norm
clip + AdamW.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

The headline engine is `tf32h`: fp32 tensors, every matrix product (GEMMs, attention, cost volume) on fp16 operands — TF32's 11-bit significand, fp32
accumulation — i.e. the precision class the reference's benched MASt3R path computes in (torch.backends.cuda.matmul.allow_tf32, dust3r/croco/models/
croco.py:12; gfx950 has no TF32 MFMA).  The bf16 engine (16-bit storage, faster, narrower than the reference) is reported beside it as config.bf16_run.

Extra objects on the line:
  roofline              the dominant kernel (persistent gemm_nt, MFMA-bound) from HIP events around every launch in the timed region
  roofline_cost_volume  the HBM-bound fused cost-volume KL, timed on its own after the run (also as roofline.cost_volume_kl_fwd);
                        frac = bytes that must move / time / peak, for the benched row masks and for every row kept (`unmasked`)
  parity                engine loss vs the CPU oracle (oracle/gd_oracle.py, fp32) on the same weights and the same pairs
  bf16                  the same workload on the bf16 engine (16-bit storage: the throughput mode, narrower than the reference's arithmetic): same
                        --steps / --warmup, own roofline and parity (also as config.bf16_run and roofline.bf16_engine)
  f32                   the same workload on the f32 engine (the reference's arithmetic precision): same --steps / --warmup, own
                        roofline and parity (also as config.reference_precision_run and roofline.f32_engine)
  (tf32h                when --dtype is not tf32h: the fp16-operand TF32-class engine as a companion run, config.tf32_class_run)
  tf32x                 the same workload on the split-precision engine (fp32 storage, 3-term bf16 split products on the matrix cores): same
                        --steps / --warmup, own roofline and parity (also as config.tf32x_run and roofline.tf32x_engine)
  other_configs         short runs of BASELINE configs 3 / 5-like and of the reference's own token geometry
  comm                  N > 1: all-reduce time of the two gradient chunks and the exposed fraction of the step
  cpu_baseline          the CPU oracle on a bounded sample of the same workload, rank 0 / N=1 only
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "tf32x": 2500.0 / 3, "tf32h": 2500.0}   # dense MFMA peaks, MI355X_MICROARCH.md; tf32x: three bf16 MFMAs per product
PEAK_HBM_GBS = 8000.0
PATCH = 14


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, help="a reference yaml (config/finetune_timm_*.yaml): variant and loss "
                    "weights are taken from it (gd_amd.config); explicit flags below win")
    ap.add_argument("--pairs-per-gpu", type=int, default=32)
    ap.add_argument("--dtype", default="tf32h", choices=["tf32h", "bf16", "f32", "tf32x"],
                    help="engine dtype.  tf32h (the headline): fp32 storage, every matrix product on fp16 operands = TF32's 11-bit significand — the "
                         "precision class the reference's benched (MASt3R) path computes its matmuls in (allow_tf32); bf16: the 16-bit-storage "
                         "throughput mode (narrower than the reference: reported as config.bf16_run); f32: exact-f32 MFMA; tf32x: 3-term bf16 "
                         "splits (better than TF32)")
    ap.add_argument("--variant", default=None, choices=["mast3r", "vggt"])
    ap.add_argument("--backbone", default=None)
    ap.add_argument("--img", type=int, default=518)
    ap.add_argument("--keypoints", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle legs (parity + cpu_baseline)")
    ap.add_argument("--no-extras", action="store_true", help="skip the f32 / other_configs companion measurements")
    ap.add_argument("--geometry", default="shared", choices=["shared", "reference"],
                    help="shared: one 37x37-token forward per image feeds all extractors (BASELINE headline); reference: the "
                         "reference's geometry (80x80-token forwards for the keypoint features + the teacher-grid forward)")
    ap.add_argument("--gemm-shapes", action="store_true", help="per-shape gemm_nt breakdown on stderr")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--steps-only", action="store_true", help="profiling aid: the timed steps and nothing else (no cost-volume roofline leg, "
                    "no CPU legs, no companion runs): a kernel trace of this run holds the step's kernels only")
    ap.add_argument("--reserve-cus", type=int, default=0, help="N > 1: compute units the persistent kernels leave to RCCL's kernels "
                    "(csrc/gd_knobs.h reserve_cus); the `comm` object A/Bs it against 0 / 8 in the same run")
    ap.add_argument("--exchange", default="torch", choices=["torch", "direct"],
                    help="gradient exchange: torch = torch.distributed all-reduce in two overlapped chunks (default); direct = "
                         "one gd_flat_allreduce (C ABI, RCCL reduce-scatter + all-gather) after the backward")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); 'gloo' lets two "
                    "ranks share one GPU for a functional check of the N>1 path")
    args = ap.parse_args()
    if args.steps_only:
        args.no_cpu_baseline = args.no_extras = args.no_kernel_events = True
    return args


def launch_ranks(args, script=None, argv=None):
    """--gpus N from a plain shell: start the N ranks as a child job and relay its exit code.  Nothing in this process has
    touched the GPU yet (and it never will): a process that initialised HIP must not be re-exec'ed.
    script / argv: what the ranks run (default: this file with this process's arguments; tests/test_bench_preflight.py starts its CPU stand-in
    through the same launcher)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv))
    return subprocess.call(cmd)


def vit_flops_per_image(D, L, Nt, P, La, r=4, bott=64):
    """Algorithmic FLOPs: forward all L blocks + patch embed; backward dX only through the La adapted blocks."""
    fwd = L * (24 * Nt * D * D + 4 * Nt * Nt * D) + 2 * (Nt - 1) * D * 3 * P * P + La * 4 * Nt * D * bott + La * 8 * Nt * D * r
    bwd = La * (24 * Nt * D * D + 8 * Nt * Nt * D) + 3 * (La * 4 * Nt * D * bott + La * 8 * Nt * D * r)
    return fwd, bwd


def flops_per_pair(eng, hw, geometry):
    D, L = eng.embedding_dim, len(eng.model.blocks)
    fwd, bwd = vit_flops_per_image(D, L, hw + 1, PATCH, L - 4)
    fl = 2 * (fwd + bwd)
    if geometry == "reference":   # + the two 80x80-token forwards / backwards per image
        f2, b2 = vit_flops_per_image(D, L, (eng.target_res // eng.downsample_factor) ** 2 + 1, PATCH, L - 4)
        fl += 2 * 2 * (f2 + b2)
    return fl


def device_sync():
    if torch.cuda.is_available():      # (the CPU preflight of the N > 1 control flow, tests/test_bench_preflight.py, has no device to wait for)
        torch.cuda.synchronize()


def barrier(world):
    if world > 1:
        torch.distributed.barrier()
    device_sync()


class Job:
    """One engine + its resident synthetic batches + the gradient exchange: `step(i)` is the timed unit."""

    def __init__(self, backbone, variant, dtype, geometry, P, img, N, dev, rank, world, vit_kwargs=None, weights=None,
                 exchange="torch", build=None):
        """build: () -> FinetuneGD (e.g. gd_amd.config.build_engine of one of the reference's yaml presets) instead of the /14 constructor; img is
        then the teacher-side image size, an int or (h, w), and the cost grid is img // eng.resize_patch_size per axis."""
        from gd_amd import dp
        from gd_amd.finetune import FinetuneGD
        from gd_amd.synthetic import synthetic_batch
        h, w = (img, img) if isinstance(img, int) else img
        self.P, self.world, self.geometry, self.dtype = P, world, geometry, dtype
        if build is not None:
            self.eng = build().to(dev)
        else:
            self.eng = FinetuneGD(r=4, backbone=backbone, patch_size=PATCH, img_size=img, variant=variant, geometry=geometry,
                                  dtype=dtype, teacher_patch=PATCH, lora_b_std=1e-3,
                                  vit_kwargs=dict(init_values=1.0) if vit_kwargs is None else vit_kwargs, **(weights or {})).to(dev)
        tp = self.eng.resize_patch_size if build is not None else PATCH
        self.hw = (h // tp) * (w // tp)
        flat = self.eng.configure_optimizers()
        # gradient exchange in two chunks: refine_conv + depth head from grad hooks (under the ViT backward), the rest after it
        early = list(self.eng.refine_conv.parameters()) + list(self.eng.depth_diff_head.parameters())
        if exchange == "direct" and world > 1:
            self.reducer = dp.DirectGradReducer(flat["g"], dp.RcclComm(rank, world))
        else:
            self.reducer = dp.OverlappedGradReducer(self.eng.trainable_parameters(), flat["views"], flat["g"], early, world)
            self.reducer.model = self.eng.model      # per-block late chunks: the model reports each block's finished weight-gradient slices
        self.reducer.attach()
        self.flat = flat
        # a few distinct synthetic batches per rank (seed 1234 + 1000*rank + i), resident on the device
        # teacher targets as the per-pair cache holds them: cost maps with 16-byte rows + the teacher-row statistics
        from gd_amd.teacher_cache import cache_cost_targets
        self.batches = [cache_cost_targets(synthetic_batch(P, h, w, N, self.hw, dev, seed=1234 + 1000 * rank + i,
                                                           teacher_patch=tp)) for i in range(2)]

    def step(self, i):
        return self.eng.fit_step(self.batches[i % len(self.batches)], self.reducer)[0]

    def timed(self, steps, warmup, dev, prof=None):
        """-> (seconds for `steps` steps = max over ranks, last loss); barrier + synchronize on both sides."""
        from gd_amd import dp, ops
        for i in range(warmup):
            self.step(i)
        barrier(self.world)
        if prof is not None:
            ops.set_gemm_profiler(prof)
        t0 = time.perf_counter()
        for i in range(steps):
            loss = self.step(i)
        barrier(self.world)
        dt = time.perf_counter() - t0
        ops.set_gemm_profiler(None)
        return dp.max_over_ranks(dt, dev), loss


def gemm_roofline(prof, dtype, dt, steps):
    # the dominant kernel: the 256x256 persistent MFMA kernel (bf16: every gd_gemm_nt launch with M >= 1024, N >= 256);
    # the N <= 8 LoRA projections run on an HBM-bound streaming kernel and are not part of this figure
    big = (lambda t: t[0] >= 1024 and t[1] >= 256 and t[6] in ("bfloat16", "float16") and "+" not in t[4]) if dtype in ("bf16", "tf32x", "tf32h") else None
    fl, ms, n = prof.totals(big)
    if dtype == "tf32x":
        fl /= 3.0            # the profiler sees the 3K-wide bf16 GEMMs: algorithmic FLOPs = executed / 3
    ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    kname = {"bf16": "gemm_nt_persist_kernel<bf16> (256x256 persistent tile kernel, all epilogue instantiations)",
             "f32": "gemm_nt_kernel<float> (128x128 tiles, exact-f32 MFMA v_mfma_f32_16x16x4_f32)",
             "tf32x": "bf16 256x256 tile kernels on 3-term split operands (gd_split3): algorithmic FLOPs, peak = bf16 MFMA peak / 3",
             "tf32h": "gemm_nt_persist_kernel<f16> (256x256 persistent tile kernel on fp16 operands: TF32's significand, one MFMA per term)"}[dtype]
    out = {"kernel": kname, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
           "frac": round(ach / PEAK_TFLOPS[dtype], 4), "traffic": None, "launches": n,
           "avg_launch_us": round(ms / max(n, 1) * 1e3, 2), "share_of_step": round(ms / (dt * 1e3), 3)}
    if big is not None and ms > 0:
        # the same launches against their OWN bounds: a launch with fp32 epilogue tensors at K = 768 has more bytes to move than MFMA work to do
        # (tf32h: A + residual + C = 5 B per output element against 2 K / peak) — per launch max(FLOPs / MFMA peak, algorithmic bytes / HBM peak)
        lim, ms2, nb, n2 = prof.roofline_time(big, PEAK_TFLOPS[dtype] * 1e12 * (3.0 if dtype == "tf32x" else 1.0), PEAK_HBM_GBS * 1e9)
        ab, nab = prof.algorithmic_bytes(big)
        out["algorithmic_bytes_per_launch"] = round(ab / max(nab, 1))      # beside `traffic` (PMC): their ratio is the kernel's over-fetch
        out["per_launch_bounds"] = {"sum_of_bounds_ms_per_step": round(lim / steps, 3), "measured_ms_per_step": round(ms2 / steps, 3),
                                    "frac_of_bound": round(lim / ms2, 4), "launches_bound_by_hbm": nb, "launches": n2,
                                    "what": "sum over the persistent-kernel launches of max(FLOPs / MFMA peak, algorithmic bytes / 8 TB/s) divided by their "
                                            "measured time: `frac` above prices every launch against the MFMA peak alone"}
    if dtype in ("bf16", "tf32h"):     # (fp16 and bf16 MFMAs run at one rate; the ceilings were measured on the bf16 instantiation)
        # context, not the yardstick (NOT measured in this run; profiles/r03_gemm_anatomy.txt, profiles/r02_micro_mfma_gap.txt):
        #  * back-to-back 16x16x32 bf16 MFMAs from registers, no memory traffic: 2328 TFLOP/s — the part's power-limited MFMA clock;
        #  * this kernel's own LDS-read + MFMA + barrier loop with the operand DMA switched off: 1668 TFLOP/s-equivalent at 4096^3
        #    (1219-1350 on the step's K = 768 shapes, where a tile is 12 K steps + its epilogue): what the chip sustains once
        #    fragments come from LDS — the four-wave / 512-register form of the same loop reaches the same 1628;
        #  * with the DMA back on, 1423 at 4096^3 (hipBLASLt's hand-written kernel: 1532), and the C stores cost another 13-17 % at K = 768.
        out["context_ceilings_tflops"] = {"mfma_only_registers": 2327.6, "lds_fed_mfma_loop_4096": 1667.9, "lds_fed_mfma_loop_k768": 1265.3,
                                          "frac_of_lds_fed_loop_k768": round(ach / 1265.3, 4),
                                          "source": "profiles/r03_gemm_anatomy.txt, profiles/r02_micro_mfma_gap.txt (NOT measured in this run)"}
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import gd_amd  # noqa: F401
    from gd_amd import config as gd_config
    from gd_amd import dp, ops

    cfg = gd_config.load(args.config) if args.config else gd_config.preset("finetune_timm_mast3r_objaverse")
    variant = args.variant or cfg["variant"]
    backbone = args.backbone or "vit_base"
    weights = {k: cfg[k] for k in ("ap_loss_weight", "depth_loss_weight", "intra_depth_loss_weight", "kl_loss_weight") if k in cfg}
    if args.variant and args.variant != cfg["variant"]:   # a different trainer: its own default weights, not the yaml's
        weights = {}

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":      # functional check only: every rank may share device 0
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    rank, local, world = dp.init_from_env(backend=args.backend)
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started {world} ranks"
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    P, img, N = args.pairs_per_gpu, args.img, args.keypoints

    if world > 1 and args.reserve_cus:
        dp.reserve_cus_for_collectives(args.reserve_cus)
    job = Job(backbone, variant, args.dtype, args.geometry, P, img, N, dev, rank, world, weights=weights, exchange=args.exchange)
    eng, hw = job.eng, job.hw
    prof = None if args.no_kernel_events else ops.GemmProfiler()
    if args.dtype == "tf32h":
        job.eng.range_report()                       # reset: the report below covers the warm-up + timed steps of this run
    dt, loss = job.timed(args.steps, args.warmup, dev, prof)
    pairs_per_s = P * world * args.steps / dt
    # tf32h: how many scaled fp16 gradient operands saturated / fell below fp16's normal range over these steps (device counters, ONE host read here)
    fp16_range = dict(job.eng.range_report(), steps=args.steps + args.warmup,
                      what="scaled fp16 gradient operands of the tf32h engine: elements clamped at +-65504 / elements below 2^-14 after the block's "
                           "power-of-two scale (DESIGN.md 4, range contract).  Counted: the cast kernels, the LayerNorm backward's fp16 copy, and the two "
                           "dX products per block that leave their GEMM as saturating fp16 (counted where the LayerNorm backward reads them); NOT counted: "
                           "the attention output gradient and the GELU-gated gradient, fp16 C stores consumed by MFMA kernels only") if args.dtype == "tf32h" else None

    comm = comm_report(job, args, dev, dt) if world > 1 else None

    out = None
    if rank == 0:
        flop_pair = flops_per_pair(eng, hw, args.geometry)
        out = {"metric": "image-pairs/sec (518^2, ViT-B/14 LoRA) student distillation step", "value": round(pairs_per_s, 3),
               "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"{cfg['name']}: {backbone}/14 + LoRA(r=4,q,v)+adapters "
                                      f"blocks 4-11, {img}^2 pairs, {'shared-518' if args.geometry == 'shared' else 'reference (80x80-token)'} geometry, {variant} losses "
                                      f"(AP+depth+intra+cost-KL), {P} pairs/GPU, {N} keypoints/pair, hw={hw}",
                          "pairs_per_gpu": P, "global_pairs": P * world, "parallelism": f"dp{world}",
                          "world_size_seen": world, "backend": args.backend or "nccl (RCCL)", "exchange": args.exchange,
                          "teacher": "excluded from the timed region: targets are synthetic and resident, as a warm TeacherTargetCache "
                                     "holds them (the frozen teacher runs once per pair, outside the student step)",
                          "engine_dtype": args.dtype,
                          "dtype_note": {"tf32h": "fp32 storage; every matrix product on fp16 operands (TF32's 11-bit significand), fp32 accumulation: the "
                                                  "reference's matmul precision class on this config (allow_tf32); product error <= 1.05 x emulated TF32 "
                                                  "(tests/test_gpu_gemm.py), full step: loss 2e-7, gradient 0.5 %, ten-step dW 1 % of the fp64 / fp32 oracle",
                                         "bf16": "16-bit storage and operands: NARROWER than the reference's fp32 / TF32 arithmetic (throughput mode)",
                                         "f32": "exact-f32 MFMA: the reference's fp32 arithmetic", "tf32x": "fp32 storage, 3-term bf16 split products "
                                         "(error 4e-6: better than TF32)"}[args.dtype]},
               "loss": round(float(loss.detach()), 6),
               **({"fp16_range": fp16_range} if fp16_range else {}),
               "vit_algorithmic_tflops": round(pairs_per_s / world * flop_pair / 1e12, 2),
               "vit_frac_of_mfma_peak": round(pairs_per_s / world * flop_pair / 1e12 / PEAK_TFLOPS[args.dtype], 4)}
        if prof:
            out["roofline"] = gemm_roofline(prof, args.dtype, dt, args.steps)
            # memory-side bytes per launch: PMC counters cannot be read from inside the timed run, so this REPLAYS the committed
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE measurement of this same command (profiles/README.md), default workload only
            pmc = next((q for q in (os.path.join(ROOT, "profiles", f"r{r:02d}_pmc_gemm_traffic_{args.dtype}.json") for r in (6, 5, 4, 3)) if os.path.exists(q)), None)
            if (pmc is not None and backbone == "vit_base" and P == 32
                    and args.geometry == "shared" and variant == "mast3r"):
                with open(pmc) as fh:
                    t = json.load(fh)
                out["roofline"]["traffic"] = round(t["hbm_side_mb_per_launch"] * 1e6)
                out["roofline"]["traffic_replayed_from"] = (f"profiles/{os.path.basename(pmc)} (2 x FETCH_SIZE + WRITE_SIZE per "
                                                            "persistent-kernel launch, separate --pmc passes; NOT measured in this run"
                                                            + (f"; taken on commit {t['commit']}, libgd_hip.so sha256 {t['libgd_hip_sha256_16']}" if "commit" in t else "") + ")")
                if out["roofline"].get("algorithmic_bytes_per_launch"):
                    out["roofline"]["traffic_over_algorithmic"] = round(out["roofline"]["traffic"] / out["roofline"]["algorithmic_bytes_per_launch"], 3)
            if args.gemm_shapes:
                for k, (cnt, sms, tf) in sorted(prof.by_shape().items(), key=lambda kv: -kv[1][1]):
                    print(f"gemm_nt {str(k):58s} x{cnt:4d} {sms / args.steps:8.3f} ms/step {tf:7.1f} TF/s", file=sys.stderr)
        if not args.steps_only:
            cvr = cost_volume_roofline(job, args, dev, variant)
            out["roofline_cost_volume"] = cvr
            if "roofline" in out:       # the driver's parser keeps `roofline` and `config`: the second north-star kernel rides inside
                out["roofline"]["cost_volume_kl_fwd"] = cvr
        if comm:
            out["comm"] = comm
        # the CPU oracle legs (parity of the benched batch, cpu_baseline) and the companion measurements run at N = 1 only: on N > 1
        # ranks they would only make the other ranks wait (tier contract: "on rank 0 at N=1 only")
        if not args.no_cpu_baseline and world == 1:
            par, base = parity_and_cpu_baseline(job, args)
            out["parity"] = par
            out["cpu_baseline"] = base
    del job, eng
    torch.cuda.empty_cache()
    if not args.no_extras and world == 1:
        extras = companion_runs(args, variant, backbone, weights, dev, rank, world)
        if rank == 0:
            out.update(extras)
            if "f32" in extras:       # first-class beside the headline, where the driver's parser keeps it
                f = extras["f32"]
                out["config"]["reference_precision_run"] = {
                    "dtype": "f32", "value": f["value"], "unit": "image-pairs/s", "ms_per_step": f["ms_per_step"], "steps": f["steps"],
                    "warmup": f["warmup"], "vit_frac_of_mfma_peak": f["vit_frac_of_mfma_peak"],
                    "parity_rel_err": f.get("parity", {}).get("rel_err"), "note": "same workload on the f32 engine (exact-f32 MFMA): the "
                    "reference trains in fp32; gfx950 has no TF32"}
                if "roofline" in out and "roofline" in f:
                    out["roofline"]["f32_engine"] = f["roofline"]
            if "bf16" in extras:
                f = extras["bf16"]
                out["config"]["bf16_run"] = {
                    "dtype": "bf16", "value": f["value"], "unit": "image-pairs/s", "ms_per_step": f["ms_per_step"], "steps": f["steps"],
                    "warmup": f["warmup"], "vit_frac_of_mfma_peak": f["vit_frac_of_mfma_peak"], "parity_rel_err": f.get("parity", {}).get("rel_err"),
                    "note": "same workload on the bf16 engine (16-bit storage and operands, fp32 accumulation / losses / optimiser): the throughput mode — "
                            "NARROWER than the reference's arithmetic (fp32, TF32 matmuls), loss parity 1e-3, gradients ~3 % (DESIGN.md section 4)"}
                if "roofline" in out and "roofline" in f:
                    out["roofline"]["bf16_engine"] = f["roofline"]
            if "tf32h" in extras:
                f = extras["tf32h"]
                out["config"]["tf32_class_run"] = {
                    "dtype": "tf32h", "value": f["value"], "unit": "image-pairs/s", "ms_per_step": f["ms_per_step"], "steps": f["steps"],
                    "warmup": f["warmup"], "vit_frac_of_mfma_peak": f["vit_frac_of_mfma_peak"], "parity_rel_err": f.get("parity", {}).get("rel_err"),
                    "note": "same workload, fp32 storage; every big GEMM and the attention products take fp16 operands (gd_cast_f16, GD_F16): "
                            "TF32's 11-bit significand, one MFMA per term, gradient operands under a per-block power-of-two scale taken on the "
                            "device (gd_amax_scale) — the precision class the reference's MASt3R path computes its matmuls in "
                            "(dust3r/croco/models/croco.py:12, allow_tf32); at full size: loss 2e-7, gradient 0.5 % of the fp64 oracle "
                            "(tests/test_gpu_fullsize.py)"}
                if "roofline" in out and "roofline" in f:
                    out["roofline"]["tf32h_engine"] = f["roofline"]
            if "tf32x" in extras:
                f = extras["tf32x"]
                out["config"]["tf32x_run"] = {
                    "dtype": "tf32x", "value": f["value"], "unit": "image-pairs/s", "ms_per_step": f["ms_per_step"], "steps": f["steps"],
                    "warmup": f["warmup"], "parity_rel_err": f.get("parity", {}).get("rel_err"),
                    "note": "same workload, fp32 storage, every big GEMM and the attention products as three bf16 MFMAs of (hi, lo) operand splits "
                            "(gd_split3, GD_F32X3): product error ~4e-6 against ~3e-4 for TF32 — the precision the reference's MASt3R path "
                            "computes its matmuls in (dust3r/croco/models/croco.py:12, allow_tf32); gradient 3e-4 of the fp64 oracle at full size"}
                if "roofline" in out and "roofline" in f:
                    out["roofline"]["tf32x_engine"] = f["roofline"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def cost_volume_roofline(job, args, dev, variant):
    """The cost-volume KL forward on its own (HBM-bound).  `frac` = the bytes the launch MUST move / its time / 8 TB/s — masked
    teacher rows never enter the loss and are not fetched, so they are not credited: with the benched trainer's row masks the
    needed bytes are features + kept teacher rows + masks; with every row kept (`unmasked`) they are SURVEY 8d's algorithmic
    byte count (2 hw D s + 2 hw^2 4 + 2 hw per pair)."""
    from gd_amd import ops
    P, hw, D = job.P, job.hw, job.eng.embedding_dim
    fmt = getattr(job.eng.model, "opfmt", "")          # "h": the kernel contracts the fp16 copies the feature producer writes beside the fp32 rows
    es = 2 if (args.dtype == "bf16" or fmt == "h") else 4      # bytes per feature element the contraction READS
    b = job.batches[0]
    Tt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    f1 = torch.randn(P, hw, D, device=dev).to(Tt).requires_grad_(True)
    f2 = torch.randn(P, hw, D, device=dev).to(Tt).requires_grad_(True)
    h16 = (f1.detach().half(), f2.detach().half()) if fmt == "h" else None
    # the row masks of the benched trainer, from the benched batch: keypoint-patch masks (MASt3R: src/finetune_timm_mast3r.py:515-519)
    # or the co-view masks down-sampled to the patch grid (VGGT: src/finetune_timm_vggt.py:504-509) — as calculate_cost_loss builds them
    h = w = args.img
    if variant == "mast3r":
        m1, m2 = ops.patch_mask(b["kp_1"], h, w, PATCH), ops.patch_mask(b["kp_2"], h, w, PATCH)
    else:
        g = h // PATCH
        m1 = torch.nn.functional.interpolate(b["mask_1"][:, None].float(), size=(g, g), mode="nearest").reshape(P, -1) > 0
        m2 = torch.nn.functional.interpolate(b["mask_2"][:, None].float(), size=(g, g), mode="nearest").reshape(P, -1) > 0
    feat_bytes = P * (2 * hw * D * es + 2 * hw)
    fwd_bytes = feat_bytes + P * 2 * hw * hw * 4               # every teacher row
    bwd_bytes = fwd_bytes + P * 2 * hw * D * es

    # as the step calls it: the inverse row norms come with the features from their producer (ops.tap_mean(with_norm=True))
    inv = (1.0 / f1.detach().float().norm(dim=-1).clamp_min(1e-12), 1.0 / f2.detach().float().norm(dim=-1).clamp_min(1e-12))

    def timed(ma, mb, backward=False, own_norm=False, kmax=None):
        kw = dict(tstats=b["cost_tstats"]) if own_norm else dict(tstats=b["cost_tstats"], inv_norms=inv, x3=fmt, h16=h16, kept_rows_max=kmax)

        def fwd():
            with torch.no_grad():
                return ops.cost_volume_kl(f1, f2, b["cost_1"], b["cost_2"], ma, mb, variant, **kw)

        def fb():
            f1.grad = f2.grad = None
            ops.cost_volume_kl(f1, f2, b["cost_1"], b["cost_2"], ma, mb, variant, **kw).sum().backward()
        return ops.time_on_stream(fb if backward else fwd, 2, 5)

    def replay(name):    # HBM traffic per launch: PMC counters cannot be read inside the run; committed passes of this configuration (the newest round's file)
        name = next((n for n in (name.replace("r05_", f"r{r:02d}_") for r in (6, 5)) if os.path.exists(os.path.join(ROOT, "profiles", n))), name)
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path) and (P, hw, D) == (32, 1369, 768) and es == 2:      # (one kernel template on the two 16-bit types: same traffic)
            with open(path) as fh:
                return json.load(fh)["fwd_hbm_bytes_per_launch"], f"profiles/{name} (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, width-corrected; NOT measured in this run)"
        return None, None

    def leg(ma, mb, pmc_file, kmax=None):
        t = timed(ma, mb, kmax=kmax)
        kept = int(ma.sum()) + int(mb.sum())
        needed = feat_bytes + kept * hw * 4
        traffic, src = replay(pmc_file) if pmc_file else (None, None)
        return {"kept_row_fraction": round(kept / (2 * P * hw), 4), "us_per_pair_fwd": round(t / P * 1e6, 2),
                "needed_bytes_per_launch": needed, "achieved": round(needed / t / 1e9, 1),
                "frac": round(needed / t / 1e9 / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_replayed_from": src,
                "traffic_over_needed": round(traffic / needed, 3) if traffic else None}, t
    ones = torch.ones(P, hw, dtype=torch.bool, device=dev)
    # keypoint-patch masks keep at most N_kp rows per view: the trainer (finetune.calculate_cost_loss) passes that bound and the op runs its kept-row form
    kmax = int(b["kp_1"].shape[1]) if variant == "mast3r" else None
    # (the kernels' own PMC passes of round 5, tools/prof_r05.sh — the kept-row one taken on the fp16 copies; the bf16 instantiation moves the same bytes)
    bench_masks, tf = leg(m1, m2, "r05_pmc_cost_volume_traffic_rows.json" if kmax else None, kmax=kmax)
    unmasked, tfu = leg(ones, ones, "r05_pmc_cost_volume_traffic_full.json")
    tfb = timed(m1, m2, backward=True, kmax=kmax)
    tfbu = timed(ones, ones, backward=True)
    out = {"kernel": "cost_volume_kl fwd (persistent MFMA contraction + softmax-KL; sparse keypoint masks: both directions as compacted kept-row problems, "
                     "gd_cost_volume_kl_fwd_rows; dense masks / `unmasked`: one hw x hw sweep serving both directions; row norms from the producer, teacher-row "
                     "statistics cached per pair)",
           "bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "masks": "keypoint-patch masks of the benched batch" if variant == "mast3r" else "co-view masks of the benched batch",
           "frac_definition": "bytes the launch must move (features + KEPT teacher rows + masks) / time / peak: skipped rows earn nothing"}
    out.update(bench_masks)
    out["unmasked"] = dict(unmasked, algorithmic_bytes_per_launch_survey_8d=fwd_bytes,
                           fwd_bwd_GBps=round((fwd_bytes + bwd_bytes) / tfbu / 1e9, 1), us_per_pair_fwd_bwd=round(tfbu / P * 1e6, 2))
    out["us_per_pair_fwd_bwd"] = round(tfb / P * 1e6, 2)      # (sparse keypoint masks: kept-row forward AND kept-row backward, gd_cost_volume_kl_bwd_rows)
    out["row_norms"] = ("from the feature producer (gd_tap_mean_norm_fwd -> gd_cost_volume_kl_fwd_prenorm), as in the step; with the op's own "
                        f"norm pass over the features: {timed(ones, ones, own_norm=True) / P * 1e6:.2f} us/pair unmasked")
    out["unmasked"]["attainable"] = cost_volume_attainable(job, dev, variant, fwd_bytes)
    return out


def cost_volume_attainable(job, dev, variant, fwd_bytes):
    """A MEASURED ceiling for the dense cost-volume forward to read `unmasked.frac` against: the persistent kernel's phases timed one at a time through its
    anatomy instantiation (csrc/cost_volume.hip `cv_fwd_persist_kernel<bf16, true>`, gd_debug_set("cv_dbg", bits): 1 = no teacher loads, 2 = no epilogue
    arithmetic, 4 = no fragment reads / MFMAs; 7 leaves the LDS-DMA ring that brings the feature tiles in).  With every phase running ALONE the launch cannot
    be faster than the slowest one; the shipped kernel's time is close to their SUM (profiles/r05_pmc_cost_volume_stall.json: 52 % of the wave cycles parked
    on waitcnt / barrier, MFMA and VALU co-executing in 1.4 %) — what a kernel of this tile shape (128 x 128 x K = 768, 393 KB of features staged per tile)
    could reach if the phases overlapped perfectly is bytes / slowest phase.  bf16 features (the anatomy build exists for bf16 only; the fp16 instantiation
    the tf32h engine runs is the same code on the other 16-bit MFMA)."""
    from gd_amd import ops
    from gd_amd._lib import lib
    P, hw, D = job.P, job.hw, job.eng.embedding_dim
    if (P * hw * D) <= 0 or (D * 2) % 128 != 0:
        return None
    b = job.batches[0]
    f1 = torch.randn(P, hw, D, device=dev).bfloat16()
    f2 = torch.randn(P, hw, D, device=dev).bfloat16()
    inv = (1.0 / f1.float().norm(dim=-1).clamp_min(1e-12), 1.0 / f2.float().norm(dim=-1).clamp_min(1e-12))
    ones = torch.ones(P, hw, dtype=torch.bool, device=dev)

    def t(bits):
        lib().gd_debug_set(b"cv_dbg", bits)
        try:
            with torch.no_grad():
                return ops.time_on_stream(lambda: ops.cost_volume_kl(f1, f2, b["cost_1"], b["cost_2"], ones, ones, variant, tstats=b["cost_tstats"], inv_norms=inv), 2, 5) * 1e6
        finally:
            lib().gd_debug_set(b"cv_dbg", 0)
    whole, ring, ring_mfma, ring_teacher, ring_epi, teacher, no_ring = t(32), t(7), t(3), t(6), t(5), t(64 | 6), t(64)
    phases = {"feature_ring_alone (LDS-DMA, L2 -> LDS)": ring, "fragment_reads_and_mfma (marginal)": max(ring_mfma - ring, 0.0),
              "teacher_loads (marginal)": max(ring_teacher - ring, 0.0), "softmax_kl_epilogue (marginal)": max(ring_epi - ring, 0.0)}
    slow, tot = max(phases.values()), sum(phases.values())
    # round 6 (anatomy bit 64 = no feature DMA): the kernel's two memory streams with no arithmetic, alone and together, and the kernel WITHOUT its feature ring
    streams = {"teacher_loads_alone (no feature DMA, no arithmetic)": round(teacher, 1), "feature_ring_alone": round(ring, 1),
               "both_streams_no_arithmetic": round(ring_teacher, 1), "everything_but_the_feature_DMA": round(no_ring, 1),
               "frac_of_teacher_stream_alone": round(fwd_bytes / (teacher * 1e-6) / 1e9 / PEAK_HBM_GBS, 4) if teacher > 0 else None,
               "reading": "the two streams DO overlap (both ~ the slower one); the slower one is the teacher stream, which by itself — 480 MB as one 131 KB "
                          "burst per tile and CU into the 64 registers the epilogue frees — runs at 2.3-2.5 TB/s (the same maps stream at 5.9 TB/s through "
                          "gd_cost_volume_teacher_stats, profiles/r06_probe_cv_streams.txt), and the kernel without any feature DMA takes ~0.9 of the whole: "
                          "the chain burst -> land -> epilogue -> next burst is the critical path, `frac_of_teacher_stream_alone` its ceiling for this kernel "
                          "family.  A form that streams the teacher through two loader waves and LDS was built (tools/experiments/cv_stream.h): -6 %, shelved"}
    return {"what": "dense forward, every row kept, 16-bit features: the persistent kernel's phases timed one at a time (anatomy build; whole op = statistics "
                    "init + tile kernel + finalize + loss, as `unmasked` times it)",
            "us_per_launch": {"whole_op_anatomy_build_nothing_off": round(whole, 1), **{k: round(v, 1) for k, v in phases.items()}},
            "memory_streams_us": streams,
            "sum_of_phases_us": round(tot, 1), "measured_over_sum_of_phases": round(whole / tot, 3) if tot > 0 else None,
            "slowest_phase_us": round(slow, 1),
            "frac_if_phases_overlapped": round(fwd_bytes / (slow * 1e-6) / 1e9 / PEAK_HBM_GBS, 4) if slow > 0 else None,
            "note": "the ceiling of THIS tile shape under perfect overlap, not of the chip: two re-structurings that tried to overlap the phases (256-row tiles on the "
                    "GEMM skeleton, round 3; row-panel-stationary with the teacher by inline-asm loads, round 5) measured slower and are shelved under tools/experiments/"}


def comm_report(job, args, dev, dt):
    """N > 1: stand-alone all-reduce time of the two gradient chunks and the share of the step the exchange exposes
    (step time with the exchange switched off, same ranks, same inputs, measured right after the timed region)."""
    import torch.distributed as dist
    from gd_amd import dp
    flat_g = job.flat["g"]
    late = getattr(job.reducer, "late", [(0, flat_g.numel())])
    n_late = sum(b - a for a, b in late)
    n_early = max(flat_g.numel() - n_late, 1)

    def ar_ms(n):
        buf = torch.zeros(n, dtype=torch.float32, device=dev)
        for _ in range(2):
            dist.all_reduce(buf)
        barrier(job.world)
        t0 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(buf)
        device_sync()
        return dp.max_over_ranks((time.perf_counter() - t0) / 10 * 1e3, dev)
    early_ms, late_ms = ar_ms(n_early), ar_ms(n_late)
    world = job.reducer.world
    job.reducer.world = 1            # no collective is issued; the 1/world scale goes with it (timing only)
    job.reducer.detach()
    dt0, _ = job.timed(args.steps, 1, dev)
    job.reducer.world = world
    job.reducer.attach()
    out = {"exchange": args.exchange, "reserve_cus": args.reserve_cus,
           "allreduce_ms_early_chunk": round(early_ms, 3), "early_chunk_MB": round(n_early * 4 / 1e6, 2),
           "allreduce_ms_late_chunk": round(late_ms, 3), "late_chunk_MB": round(n_late * 4 / 1e6, 2),
           "ms_per_step_without_exchange": round(dt0 / args.steps * 1e3, 3),
           "exposed_comm_frac": round(max(0.0, (dt - dt0) / dt), 4)}
    # the same ranks, inputs and steps with (a) the OTHER exchange mode and (b) the other CU reservation: what the first multi-GPU run should look at.
    # Every rank must take the same branch (the timed steps hold barriers and all-reduces): a variant runs only if EVERY rank could set it up
    # (`all_ok`), a communicator built for a variant is closed again, and a failure inside the timed steps is not caught — ranks that diverged
    # there cannot be brought back in step, the launcher's timeout is the right outcome.
    def all_ok(flag):
        return dp.max_over_ranks(0.0 if flag else 1.0, dev) == 0.0

    variants = {}
    flat, eng, keep = job.flat, job.eng, job.reducer
    keep.detach()
    alt_red, alt_comm, err = None, None, None
    try:
        if args.exchange == "torch":
            alt_comm = dp.RcclComm(dist.get_rank(), world)
            alt_red = dp.DirectGradReducer(flat["g"], alt_comm)
        else:
            early = list(eng.refine_conv.parameters()) + list(eng.depth_diff_head.parameters())
            alt_red = dp.OverlappedGradReducer(eng.trainable_parameters(), flat["views"], flat["g"], early, world)
            alt_red.model = eng.model
    except Exception as e:       # e.g. two ranks sharing one GPU: no RCCL communicator
        err = repr(e)[:200]
    if all_ok(err is None):
        job.reducer = alt_red
        job.reducer.attach()
        dtx, _ = job.timed(args.steps, 1, dev)
        job.reducer.detach()
        other = "direct" if args.exchange == "torch" else "torch"
        variants["exchange_" + other] = {"ms_per_step": round(dtx / args.steps * 1e3, 3), "exposed_comm_frac": round(max(0.0, (dtx - dt0) / dtx), 4)}
    else:
        variants["exchange_error"] = err or "another rank could not set the variant up"
    if alt_comm is not None:
        alt_comm.close()
    job.reducer = keep
    job.reducer.world = world
    job.reducer.attach()
    if isinstance(job.reducer, dp.OverlappedGradReducer) and job.reducer.model is not None:
        # the blocks' weight-gradient slices exchanged as each block's backward returns (the default) against all of them after the backward
        job.reducer.detach()
        job.reducer.per_block = not job.reducer.per_block
        job.reducer.attach()
        dtb, _ = job.timed(args.steps, 1, dev)
        variants["per_block_chunks_" + ("on" if job.reducer.per_block else "off")] = {"ms_per_step": round(dtb / args.steps * 1e3, 3),
                                                                                      "exposed_comm_frac": round(max(0.0, (dtb - dt0) / dtb), 4)}
        job.reducer.detach()
        job.reducer.per_block = not job.reducer.per_block
        job.reducer.attach()
    alt = 0 if args.reserve_cus else 8
    dp.reserve_cus_for_collectives(alt)
    dtr, _ = job.timed(args.steps, 1, dev)
    job.reducer.world = 1
    job.reducer.detach()
    dtr0, _ = job.timed(args.steps, 1, dev)
    job.reducer.world = world
    job.reducer.attach()
    dp.reserve_cus_for_collectives(args.reserve_cus)
    variants[f"reserve_cus_{alt}"] = {"ms_per_step": round(dtr / args.steps * 1e3, 3), "ms_per_step_without_exchange": round(dtr0 / args.steps * 1e3, 3),
                                      "exposed_comm_frac": round(max(0.0, (dtr - dtr0) / dtr), 4)}
    out["variants"] = variants
    return out


def physical_cores():
    try:
        seen = set()
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        return len(seen) or None
    except OSError:
        return None


def parity_and_cpu_baseline(job, args, ns=4):
    """The CPU oracle (a port of the reference arithmetic, fp32, host cores) on the first NS pairs of batch 0 with the
    engine's CURRENT weights: (a) parity — the engine's per-pair loss terms on the same pairs and weights (checker only,
    outside every timed region); (b) cpu_baseline — the oracle's forward + backward time per pair."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gd_oracle as O
    import torch.nn.functional as F
    from gd_amd.synthetic import export_params as oracle_params
    eng, batch = job.eng, job.batches[0]
    with torch.no_grad():
        _, terms = eng.training_step(batch)
    terms = {k: v.float().cpu() for k, v in terms.items()}
    cores = min(os.cpu_count(), 32)       # torch CPU ops stop scaling (and oversubscribe) well before 256 threads
    torch.set_num_threads(cores)
    p, tr, refine, head, cfg = oracle_params(eng)
    # the fp32 oracle's attention through torch's fused CPU kernel — the call timm's blocks make themselves; the fastest form of the CPU path (the baseline
    # below is not handicapped by an explicit [heads, N, N] softmax) and the only affordable one at the reference geometry's 4 801 / 6 401 tokens;
    # equal to the explicit form to 5e-7 (tests/test_oracle_attention_modes.py)
    cfg["attention_sdpa"] = True
    for d in tr.values():
        for blk in d.values():
            for k in blk:
                blk[k] = blk[k].requires_grad_(True)
    refine = {k: v.requires_grad_(True) for k, v in refine.items()}
    head = {k: v.requires_grad_(True) for k, v in head.items()}
    NS = min(ns, batch["rgb_1"].shape[0])                     # bounded sample: NS pairs, ~10-15 s of host work
    weights = {"ap": eng.ap_loss_weight, "depth": eng.depth_loss_weight, "intra": eng.intra_depth_loss_weight,
               "kl": eng.kl_loss_weight}
    names = (("ap_loss", "ap"), ("depth_loss", "depth"), ("intra_depth_loss", "intra"), ("kl_loss", "kl"))
    worst_term, tot_hip, tot_ref = 0.0, 0.0, 0.0
    # ONE untimed warm-up pair first (SURVEY 8d plans warm-up + timed passes): pair NS - 1 runs once before the clock starts — first-touch page faults of
    # the oracle's activations, torch's CPU thread pool and oneDNN primitive caches are then paid — and its gradient is dropped again
    t0 = None
    for it in range(-1, NS):
        q = NS - 1 if it < 0 else it
        if it == 0:
            for d in tr.values():
                for blk in d.values():
                    for k in blk:
                        blk[k].grad = None
            for v in list(refine.values()) + list(head.values()):
                v.grad = None
            t0 = time.perf_counter()
        cb = {k: v[q:q + 1].detach().cpu() for k, v in batch.items()}
        h, w = cb["rgb_1"].shape[-2:]
        tp = cfg["teacher_patch"]
        one = {"rgb_1": cb["rgb_1"], "rgb_2": cb["rgb_2"], "kp_1": cb["kp_1"], "kp_2": cb["kp_2"],
               "depth_1": cb["depth_1"][0], "depth_2": cb["depth_2"][0],
               "cost_1": cb["cost_1"][..., :cb["cost_1"].shape[-2]], "cost_2": cb["cost_2"][..., :cb["cost_2"].shape[-2]],   # drop the row padding
               "pts3d_1": cb["pts3d_1"], "pts3d_2": cb["pts3d_2"],
               "mask_patch_1": F.interpolate(cb["mask_1"][:, None].float(), size=(h // tp, w // tp), mode="nearest").bool().view(-1),
               "mask_patch_2": F.interpolate(cb["mask_2"][:, None].float(), size=(h // tp, w // tp), mode="nearest").bool().view(-1)}
        ot = O.pair_losses(one, p, cfg, tr, refine, head)
        ref = O.total_loss(ot, weights)
        (ref / NS).backward()
        if it < 0:
            continue
        tot_ref += ref.item()
        tot_hip += sum(weights[b] * terms[a][q].item() for a, b in names)
        for a, b in names:
            if weights[b] != 0:
                worst_term = max(worst_term, abs(terms[a][q].item() - ot[b].item()) / max(1e-3, abs(ot[b].item())))
    dt = (time.perf_counter() - t0) / NS
    rel = abs(tot_hip - tot_ref) / abs(tot_ref)
    par = {"rel_err": float(f"{rel:.3e}"), "tol": 1e-3, "ok": bool(rel < 1e-3), "worst_term_rel_err": float(f"{worst_term:.3e}"),
           "loss_hip": round(tot_hip / NS, 6), "loss_oracle": round(tot_ref / NS, 6), "engine_dtype": job.dtype,
           "what": f"mean loss of {NS} pairs of the benched batch, engine (HIP, {job.dtype}) vs oracle/gd_oracle.py (CPU fp32), same weights"}
    base = {"value": round(1.0 / dt, 4), "unit": "image-pairs/s", "cores": cores, "host_logical_cpus": os.cpu_count(),
            "host_physical_cores": physical_cores(), "kind": "port",
            "sample": f"{NS} pairs of the same workload (fwd+bwd, fp32, oracle/gd_oracle.py, {cores} of {os.cpu_count()} logical CPUs: torch's CPU ops stop "
                      f"scaling before that; attention through torch's fused CPU kernel) after ONE untimed warm-up pair: {dt:.1f} s per pair"}
    return par, base


def _preset_engine(name, dtype):
    from gd_amd import config
    return config.build_engine(config.preset(name), geometry="reference", dtype=dtype, lora_b_std=1e-3)[0]


def companion_runs(args, variant, backbone, weights, dev, rank, world):
    """Driver-observed runs beside the headline (same process, after it): the f32 engine — the reference's arithmetic precision —
    on the SAME workload with the SAME --steps / --warmup, its own GEMM roofline and its own parity check against the CPU oracle;
    and, short (2 steps), ViT-L/14 with the VGGT losses (BASELINE config 3), a CLIP-style pre-norm ViT-L/14 (config 5) and the
    reference's own token geometry."""
    from gd_amd import ops
    out = {}
    P, img, N = args.pairs_per_gpu, args.img, args.keypoints

    def run(bb, var, dtype, geometry, pairs, vit_kwargs=None, steps=2, warmup=1, prof=False, wts=None, parity=False, build=None, size=None):
        job = Job(bb, var, dtype, geometry, pairs, size or img, N, dev, rank, world, vit_kwargs=vit_kwargs, weights=wts, build=build)
        pr = ops.GemmProfiler() if prof else None
        dt, loss = job.timed(steps, warmup, dev, pr)
        pps = pairs * world * steps / dt
        rec = {"value": round(pps, 3), "unit": "image-pairs/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
               "warmup": warmup, "dtype": dtype, "pairs_per_gpu": pairs, "backbone": bb, "variant": var, "geometry": geometry,
               "loss": round(float(loss.detach()), 6)}
        if build is None:      # (the FLOP model below is written for the /14 constructor's token counts)
            rec["vit_frac_of_mfma_peak"] = round(pps / world * flops_per_pair(job.eng, job.hw, geometry) / 1e12 / PEAK_TFLOPS[dtype], 4)
        else:
            rec["image_hw"] = list(size) if not isinstance(size, int) else [size, size]
            rec["patch"] = job.eng.patch_size
        if pr is not None:
            rec["roofline"] = gemm_roofline(pr, dtype, dt, steps)
        if parity and not args.no_cpu_baseline:
            rec["parity"] = parity_and_cpu_baseline(job, args, ns=parity if isinstance(parity, int) and not isinstance(parity, bool) else 4)[0]
        del job
        torch.cuda.empty_cache()
        return rec

    if args.dtype != "bf16":
        out["bf16"] = run(backbone, variant, "bf16", args.geometry, P, steps=args.steps, warmup=args.warmup,
                          prof=not args.no_kernel_events, wts=weights, parity=world == 1)
    if args.dtype != "f32":
        out["f32"] = run(backbone, variant, "f32", args.geometry, P, steps=args.steps, warmup=args.warmup,
                         prof=not args.no_kernel_events, wts=weights, parity=world == 1)
    if args.dtype != "tf32x":
        out["tf32x"] = run(backbone, variant, "tf32x", args.geometry, P, steps=args.steps, warmup=args.warmup,
                           prof=not args.no_kernel_events, wts=weights, parity=world == 1)
    if args.dtype != "tf32h":
        out["tf32h"] = run(backbone, variant, "tf32h", args.geometry, P, steps=args.steps, warmup=args.warmup,
                           prof=not args.no_kernel_events, wts=weights, parity=world == 1)
    if world == 1 and args.geometry == "shared" and backbone == "vit_base":
        out["other_configs"] = {
            # (each with a ONE-pair parity object: engine vs the CPU oracle on the first pair of its own batch, its own weights)
            "vit_large_vggt": run("vit_large", "vggt", args.dtype, "shared", 16, parity=1),
            "prenorm_vit_large_all_losses": run("vit_large", "vggt", args.dtype, "shared", 16,
                                                vit_kwargs=dict(pre_norm=True, ln_eps=1e-5, pos_interp="timm"), parity=1),
            "reference_geometry": run(backbone, variant, args.dtype, "reference", 8, wts=weights, parity=1),
            # the reference's OWN backbone, from its yaml preset with no override but the engine dtype (config.preset: ViT-B/16 with a pre-norm, CLIP
            # mean / std, conv without bias), in its own token geometry: 384 x 512 images -> a 24 x 32 cost grid and 4 801-token keypoint forwards
            "reference_backbone": run("ViT-B-16", "mast3r", args.dtype, "reference", 8, parity=1, size=(384, 512),
                                      build=lambda: _preset_engine("finetune_timm_mast3r_objaverse", args.dtype))}
    return out


if __name__ == "__main__":
    main()
