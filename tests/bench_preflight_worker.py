"""One rank of tests/test_bench_preflight.py: bench.py's N > 1 control flow (Job.timed's barriers and max-over-ranks clock, comm_report with
every variant it A/Bs) on the CPU over gloo, with a stand-in for the engine — a flat gradient buffer laid out as FinetuneGD.configure_optimizers
lays it out, a "backward" that fires the loss-side gradient hooks and then reports the blocks top-down exactly as vit._BlockFn.backward does, and
the real dp.OverlappedGradReducer in between.  Every step checks its own result (the exchanged buffer = the mean over ranks), so a variant that
loses or doubles a range fails here and not on the first 8-GPU node.  Started by bench.launch_ranks (the launcher bench.py --gpus N uses)."""
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p_ in (ROOT, HERE):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import bench  # noqa: E402


class StandInJob(bench.Job):
    def __init__(self, rank, world):        # (not bench.Job.__init__: no engine, no device)
        from gd_amd import dp
        self.P, self.world, self.rank, self.geometry, self.dtype = 2, world, rank, "shared", "f32"
        L, nA, nB, nAd = 4, 24, 24, 64
        self.L = L
        conv = torch.nn.Conv2d(4, 4, 3, padding=1)
        head = torch.nn.Linear(4, 3)
        early = list(conv.parameters()) + list(head.parameters())
        sizes = [nA] * L + [nB] * L + [p.numel() for p in early] + [nAd] * L
        al = lambda k: (k + 3) // 4 * 4
        n = sum(al(k) for k in sizes)
        flat_p, flat_g = torch.zeros(n), torch.zeros(n)
        ps = [torch.nn.Parameter(torch.zeros(nA)) for _ in range(L)] + [torch.nn.Parameter(torch.zeros(nB)) for _ in range(L)] + early + \
             [torch.nn.Parameter(torch.zeros(nAd)) for _ in range(L)]
        views, offs, off = [], [], 0
        for p in ps:
            k = p.numel()
            p.data = flat_p[off:off + k].view(p.shape)
            views.append(flat_g[off:off + k].view(p.shape))
            offs.append(off)
            off += al(k)
        self.offs, self.ps, self.early = offs, ps, early
        self.spans = [[(offs[i], offs[i] + nA), (offs[len(ps) - L + i], offs[len(ps) - L + i] + nAd)] for i in range(L)]
        self.lateB = (offs[L], offs[2 * L - 1] + nB)
        model = types.SimpleNamespace(block_grad_hook=None)
        self.eng = types.SimpleNamespace(refine_conv=conv, depth_diff_head=head, trainable_parameters=lambda: ps, model=model)
        self.flat = {"g": flat_g, "p": flat_p, "views": views}
        self.reducer = dp.OverlappedGradReducer(ps, views, flat_g, early, world)
        self.reducer.model = model
        self.reducer.attach()
        self.truth = torch.arange(n, dtype=torch.float32)
        self.steps_run = 0

    def step(self, i):
        """FinetuneGD.fit_step in miniature: zero, backward (hooks + per-block reports), wait_early, gather, start, finish; self-checking."""
        g, red, model = self.flat["g"], self.reducer, self.eng.model
        local = self.truth * (self.rank + 1 + i)
        for p in self.ps:
            p.grad = None
        g.zero_()
        loss = sum((p * local[o:o + p.numel()].view(p.shape)).sum() for p, o in zip(self.ps, self.offs) if any(p is e for e in self.early))
        loss.backward()                                               # the loss-side tensors: their post-accumulate hooks start the early chunk
        for b in reversed(range(self.L)):                            # the blocks, top down: slices final -> reported (when the reducer asked for it)
            for a, e in self.spans[b]:
                g[a:e] = local[a:e]
            if model.block_grad_hook is not None:
                model.block_grad_hook(b, self.spans[b])
        g[self.lateB[0]:self.lateB[1]] = local[self.lateB[0]:self.lateB[1]]      # finish_trainable_grads: LoRA-B lands after the backward
        red.wait_early()
        for p, v in zip(self.ps, self.flat["views"]):
            if p.grad is not None:
                v.copy_(p.grad)
        red.start()
        scale = red.finish()
        w = red.world
        mean = self.truth * (sum(r + 1 + i for r in range(self.world)) / self.world if w > 1 else (self.rank + 1 + i))
        pad = torch.ones_like(g, dtype=torch.bool)
        for p, o in zip(self.ps, self.offs):
            pad[o:o + p.numel()] = False
        got = g * scale
        assert torch.allclose(got[~pad], mean[~pad], rtol=1e-6, atol=1e-6), f"rank {self.rank}: exchanged gradient is not the mean (world seen by the reducer: {w})"
        assert float(got[pad].abs().max()) == 0.0 if bool(pad.any()) else True
        self.steps_run += 1
        return loss.detach()


def main():
    from gd_amd import dp
    rank, _, world = dp.init_from_env(backend="gloo")
    dev = torch.device("cpu")
    args = types.SimpleNamespace(steps=2, warmup=1, exchange="torch", reserve_cus=int(os.environ.get("PREFLIGHT_RESERVE", "0")))
    job = StandInJob(rank, world)
    dt, loss = job.timed(args.steps, args.warmup, dev)
    comm = bench.comm_report(job, args, dev, dt)
    from gd_amd._lib import lib
    assert lib().gd_debug_get(b"reserve_cus") == args.reserve_cus            # the A/B put the reservation back
    assert job.reducer.world == world and job.eng.model.block_grad_hook is not None and job.reducer.per_block
    if rank == 0:
        with open(sys.argv[1], "w") as fh:
            json.dump({"world": world, "comm": comm, "steps_run": job.steps_run, "ms_per_step": dt / args.steps * 1e3}, fh)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
