#!/usr/bin/env python
"""Benchmark of the geometric-distillation student step (BASELINE.json metric: image-pairs/s at 518^2,
ViT-B/14 + LoRA) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, 1 rank/GPU)

A "step" = one pass of the hot path over one batch of synthetic pairs per GPU: student ViT forward (one
shared 37x37-token forward per image, taps 4-7 + final), the three distillation losses against synthetic
teacher targets, backward through LoRA/adapters/refine_conv/depth head, flat-gradient all-reduce (N>1), global-norm
clip + AdamW.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

Extra objects on the line: `roofline` for the dominant kernel (gemm_nt, MFMA-bound) from HIP events recorded
around every launch inside the timed region; `roofline_cost_volume` (HBM-bound fused cost-volume KL, timed on
its own after the run); `cpu_baseline` = the CPU oracle (oracle/gd_oracle.py) on a bounded sample, rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs-per-gpu", type=int, default=32)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--variant", default="mast3r", choices=["mast3r", "vggt"])
    ap.add_argument("--backbone", default="vit_base")
    ap.add_argument("--img", type=int, default=518)
    ap.add_argument("--keypoints", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--geometry", default="shared", choices=["shared", "reference"],
                    help="shared: one 37x37-token forward per image feeds all extractors (BASELINE headline); reference: the "
                         "reference's geometry (80x80-token forwards for the keypoint features + the teacher-grid forward)")
    ap.add_argument("--gemm-shapes", action="store_true", help="per-shape gemm_nt breakdown on stderr")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); 'gloo' lets two "
                    "ranks share one GPU for a functional check of the N>1 path")
    return ap.parse_args()


def vit_flops_per_image(D, L, Nt, P, La, r=4, bott=64):
    """Algorithmic FLOPs: forward all L blocks + patch embed; backward dX only through the La adapted blocks."""
    fwd = L * (24 * Nt * D * D + 4 * Nt * Nt * D) + 2 * (Nt - 1) * D * 3 * P * P + La * 4 * Nt * D * bott + La * 8 * Nt * D * r
    bwd = La * (24 * Nt * D * D + 8 * Nt * Nt * D) + 3 * (La * 4 * Nt * D * bott + La * 8 * Nt * D * r)
    return fwd, bwd


def main():
    args = parse()
    import gd_amd  # noqa: F401
    from gd_amd import dp, ops
    from gd_amd.finetune import FinetuneGD
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gd_testutil import synthetic_batch

    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    ndev = torch.cuda.device_count()
    if args.backend == "gloo":      # functional check only: every rank may share device 0
        os.environ["LOCAL_RANK"] = str(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    rank, local, world = dp.init_from_env(backend=args.backend)
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    P, img, N = args.pairs_per_gpu, args.img, args.keypoints
    patch = 14
    hw = (img // patch) ** 2

    eng = FinetuneGD(r=4, backbone=args.backbone, patch_size=patch, img_size=img, variant=args.variant,
                     geometry=args.geometry, dtype=args.dtype, teacher_patch=patch, lora_b_std=1e-3,
                     vit_kwargs=dict(init_values=1.0)).to(dev)
    flat = eng.configure_optimizers()
    # gradient exchange in two chunks: refine_conv + depth head from grad hooks (under the ViT backward), the rest after it
    early = list(eng.refine_conv.parameters()) + list(eng.depth_diff_head.parameters())
    reducer = dp.OverlappedGradReducer(eng.trainable_parameters(), flat["views"], flat["g"], early, world)
    reducer.attach()
    # a few distinct synthetic batches per rank (seed 1234 + 1000*rank + i), resident on the device
    batches = [synthetic_batch(P, img, img, N, hw, dev, seed=1234 + 1000 * rank + i, teacher_patch=patch) for i in range(2)]

    def step(i):
        loss, terms = eng.training_step(batches[i % len(batches)])
        eng.backward(loss, pre_gather=reducer.wait_early)   # gradients land in the flat buffer through one multi-tensor copy
        reducer.start()
        scale = reducer.finish()
        eng.optimizer_step(grad_scale=scale)
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    prof = None if args.no_kernel_events else ops.GemmProfiler()
    barrier()
    if prof:
        ops.set_gemm_profiler(prof)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    barrier()
    dt = time.perf_counter() - t0
    ops.set_gemm_profiler(None)
    dt = dp.max_over_ranks(dt, dev)
    pairs_per_s = P * world * args.steps / dt

    out = None
    if rank == 0:
        D = eng.embedding_dim
        L = len(eng.model.blocks)
        Nt = hw + 1
        fwd, bwd = vit_flops_per_image(D, L, Nt, patch, L - 4)
        flop_pair = 2 * (fwd + bwd)
        if args.geometry == "reference":   # + the two 80x80-token forwards / backwards per image
            f2, b2 = vit_flops_per_image(D, L, (eng.target_res // eng.downsample_factor) ** 2 + 1, patch, L - 4)
            flop_pair += 2 * 2 * (f2 + b2)
        out = {"metric": "image-pairs/sec (518^2, ViT-B/14 LoRA) student distillation step", "value": round(pairs_per_s, 3),
               "unit": "image-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": f"finetune_timm_{args.variant}_objaverse: {args.backbone}/14 + LoRA(r=4,q,v)+adapters "
                                      f"blocks 4-11, {img}^2 pairs, {'shared-518' if args.geometry == 'shared' else 'reference (80x80-token)'} geometry, {args.variant} losses "
                                      f"(AP+depth+intra+cost-KL), {P} pairs/GPU, {N} keypoints/pair, hw={hw}",
                          "pairs_per_gpu": P, "global_pairs": P * world, "parallelism": f"dp{world}"},
               "loss": round(float(loss.detach()), 6),
               "vit_algorithmic_tflops": round(pairs_per_s / world * flop_pair / 1e12, 2),
               "vit_frac_of_mfma_peak": round(pairs_per_s / world * flop_pair / 1e12 / PEAK_TFLOPS[args.dtype], 4)}
        if prof:
            # the dominant kernel: the 256x256 persistent MFMA kernel (bf16: every gd_gemm_nt launch with M >= 1024, N >= 256);
            # the N <= 8 LoRA projections run on an HBM-bound streaming kernel and are not part of this figure
            big = (lambda t: t[0] >= 1024 and t[1] >= 256 and t[6] == "bfloat16" and "+" not in t[4]) if args.dtype == "bf16" else None
            fl, ms, n = prof.totals(big)
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            kname = "gemm_nt_persist_kernel<bf16> (256x256 persistent tile kernel, all epilogue instantiations)" if args.dtype == "bf16" \
                else f"gd_gemm_nt<{args.dtype}> (gemm_nt_kernel)"
            out["roofline"] = {"kernel": kname, "bound": "mfma", "achieved": round(ach, 2),
                               "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_TFLOPS[args.dtype], 4), "traffic": None,
                               "launches": n, "avg_launch_us": round(ms / max(n, 1) * 1e3, 2),
                               "share_of_step": round(ms / (dt * 1e3), 3)}
            # memory-side bytes per launch: PMC counters cannot be read from inside the timed run, so this is the committed
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE measurement of this same command (profiles/README.md), default workload only
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_gemm_traffic.json")
            if (os.path.exists(pmc) and args.dtype == "bf16" and args.backbone == "vit_base" and P == 32
                    and args.geometry == "shared" and args.variant == "mast3r"):
                with open(pmc) as fh:
                    t = json.load(fh)
                out["roofline"]["traffic"] = round(t["hbm_side_mb_per_launch"] * 1e6)
                out["roofline"]["traffic_source"] = "profiles/r01_pmc_gemm_traffic.json (2 x FETCH_SIZE + WRITE_SIZE per persistent-kernel launch, separate --pmc passes)"
            if args.gemm_shapes:
                for k, (cnt, sms, tf) in sorted(prof.by_shape().items(), key=lambda kv: -kv[1][1]):
                    print(f"gemm_nt {str(k):58s} x{cnt:4d} {sms / args.steps:8.3f} ms/step {tf:7.1f} TF/s", file=sys.stderr)
        # ---- cost-volume kernel on its own (HBM-bound; algorithmic bytes per SURVEY 8d) ----
        es = 2 if args.dtype == "bf16" else 4
        b = batches[0]
        Tt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        f1 = torch.randn(P, hw, D, device=dev).to(Tt).requires_grad_(True)
        f2 = torch.randn(P, hw, D, device=dev).to(Tt).requires_grad_(True)
        m1 = torch.rand(P, hw, device=dev) > 0.3
        m2 = torch.rand(P, hw, device=dev) > 0.3

        def cv_fwd():
            with torch.no_grad():
                return ops.cost_volume_kl(f1, f2, b["cost_1"], b["cost_2"], m1, m2, args.variant)

        def cv_fb():
            f1.grad = f2.grad = None
            ops.cost_volume_kl(f1, f2, b["cost_1"], b["cost_2"], m1, m2, args.variant).sum().backward()
        tf = ops.time_on_stream(cv_fwd, 2, 5)
        tfb = ops.time_on_stream(cv_fb, 2, 5)
        fwd_bytes = P * (2 * hw * D * es + 2 * hw * hw * 4 + 2 * hw)
        bwd_bytes = fwd_bytes + P * 2 * hw * D * es
        out["roofline_cost_volume"] = {"kernel": "cost_volume_kl fwd (cv_prep + cv_fwd_tile + cv_finalize)", "bound": "hbm",
                                       "achieved": round(fwd_bytes / tf / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": round(fwd_bytes / tf / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                                       "us_per_pair_fwd": round(tf / P * 1e6, 2),
                                       "fwd_bwd_GBps": round((fwd_bytes + bwd_bytes) / tfb / 1e9, 1),
                                       "us_per_pair_fwd_bwd": round(tfb / P * 1e6, 2)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(eng, batches[0], args)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def cpu_baseline(eng, batch, args):
    """The CPU oracle (a port of the reference arithmetic) on ONE pair of the same workload, fp32, all host cores:
    forward + backward of the three losses through the oracle ViT (checker code, timed beside — never inside — the product path)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gd_oracle as O
    import torch.nn.functional as F
    from gd_testutil import oracle_params
    cores = min(os.cpu_count(), 32)       # torch CPU ops stop scaling (and oversubscribe) well before 256 threads
    torch.set_num_threads(cores)
    p, tr, refine, head, cfg = oracle_params(eng)
    leaves = []
    for d in tr.values():
        for blk in d.values():
            for k in blk:
                blk[k] = blk[k].requires_grad_(True)
                leaves.append(blk[k])
    refine = {k: v.requires_grad_(True) for k, v in refine.items()}
    head = {k: v.requires_grad_(True) for k, v in head.items()}
    NS = min(4, batch["rgb_1"].shape[0])                      # bounded sample: NS pairs, ~10-15 s of host work
    weights = {"ap": eng.ap_loss_weight, "depth": eng.depth_loss_weight, "intra": eng.intra_depth_loss_weight,
               "kl": eng.kl_loss_weight}
    t0 = time.perf_counter()
    for q in range(NS):
        cb = {k: v[q:q + 1].detach().cpu() for k, v in batch.items()}
        h, w = cb["rgb_1"].shape[-2:]
        tp = cfg["teacher_patch"]
        one = {"rgb_1": cb["rgb_1"], "rgb_2": cb["rgb_2"], "kp_1": cb["kp_1"], "kp_2": cb["kp_2"],
               "depth_1": cb["depth_1"][0], "depth_2": cb["depth_2"][0], "cost_1": cb["cost_1"], "cost_2": cb["cost_2"],
               "pts3d_1": cb["pts3d_1"], "pts3d_2": cb["pts3d_2"],
               "mask_patch_1": F.interpolate(cb["mask_1"][:, None].float(), size=(h // tp, w // tp), mode="nearest").bool().view(-1),
               "mask_patch_2": F.interpolate(cb["mask_2"][:, None].float(), size=(h // tp, w // tp), mode="nearest").bool().view(-1)}
        terms = O.pair_losses(one, p, cfg, tr, refine, head)
        (O.total_loss(terms, weights) / NS).backward()
    dt = (time.perf_counter() - t0) / NS
    return {"value": round(1.0 / dt, 4), "unit": "image-pairs/s", "cores": cores, "kind": "port",
            "sample": f"{NS} pairs of the same workload (fwd+bwd, fp32, oracle/gd_oracle.py, {cores} threads): "
                      f"{dt:.1f} s per pair"}


if __name__ == "__main__":
    main()
