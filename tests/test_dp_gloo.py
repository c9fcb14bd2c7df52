"""CPU, world_size 2, gloo: the data-parallel gradient exchange (flat buffer all-reduce) and pair sharding."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import gd_amd  # noqa: F401
    from gd_amd import dp
    r, l, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = dp.shard_pairs(8, r, w)
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)          # rank-dependent "gradient"
    red = dp.FlatGradReducer(g, w)
    red.start()
    scale = red.finish()
    t = dp.max_over_ranks(float(rank + 1), "cpu")
    out[rank] = (lo, hi, (g * scale).tolist(), t)
    torch.distributed.destroy_process_group()


def test_flat_grad_allreduce_world2():
    world, port = 2, 29611
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    want = (torch.arange(10, dtype=torch.float32) * 1.5).tolist()     # mean of x1 and x2
    assert out[0][:2] == (0, 4) and out[1][:2] == (4, 8)
    assert out[0][2] == want and out[1][2] == want
    assert out[0][3] == 2.0 and out[1][3] == 2.0


def _worker_overlap(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import gd_amd  # noqa: F401
    from gd_amd import dp
    dp.init_from_env(backend="gloo")
    # five "parameters" as views of one flat buffer (16-byte aligned slices with pads, like FinetuneGD.configure_optimizers)
    shapes = [(3, 4), (5,), (2, 6), (7,), (4, 4)]
    al = lambda k: (k + 3) // 4 * 4
    n = sum(al(torch.Size(s).numel()) for s in shapes)
    flat_p, flat_g = torch.zeros(n), torch.zeros(n)
    params, views, off = [], [], 0
    for s in shapes:
        k = torch.Size(s).numel()
        p = torch.nn.Parameter(torch.zeros(s))
        p.data = flat_p[off:off + k].view(s)
        params.append(p)
        views.append(flat_g[off:off + k].view(s))
        off += al(k)
    early = [params[2], params[3]]                                 # reduced from hooks while "the backward" is still running
    red = dp.OverlappedGradReducer(params, views, flat_g, early, world)
    red.attach()
    x = torch.arange(1, 6, dtype=torch.float32) * (rank + 1)
    loss = sum((p * x[i]).sum() for i, p in enumerate(params))     # d loss / d p_i = x[i] everywhere
    for p in params:
        p.grad = None
    loss.backward()
    red.wait_early()
    flat_g.zero_()
    torch._foreach_copy_(views, [p.grad for p in params])
    red.start()
    scale = red.finish()
    out[rank] = [float((v * scale).mean()) for v in views] + [red.late]
    torch.distributed.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_overlapped_grad_reducer_world2(world):
    """early chunk from post-accumulate-grad hooks + late ranges of the flat buffer == one mean all-reduce of everything
    (world 8: the node size the bench is launched with, src/main.py:147-151 `devices=-1`)"""
    port = 29613 + world
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_overlap, args=(world, port, out), nprocs=world, join=True)
    mean_rank = sum(r + 1 for r in range(world)) / world
    want = [mean_rank * i for i in range(1, 6)]                      # mean over ranks of x[i] * (rank + 1)
    for r in range(world):
        assert out[r][:5] == want, out[r]
        assert out[r][5] == [(0, 20), (39, 56)]      # flat layout 12 | 5+3 pad | [12 | 7] +1 pad | 16; the late ranges are the complement


def _worker_blocks(rank, world, port, out):
    """The N-chunk late exchange: a stand-in model lays its flat gradient buffer out as GDViT.prepare_trainables(flat) does — [L LoRA-A tensors |
    L LoRA-B tensors | an early (loss-side) tensor | L adapter tensors] — and its "backward" finishes the blocks from the last to the first,
    calling the hook the reducer installed after each block exactly as vit._BlockFn.backward does; the LoRA-B slices only become final after the
    backward (finish_trainable_grads).  The result must equal ONE mean all-reduce of the whole buffer, and every element must be exchanged once."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import gd_amd  # noqa: F401
    from gd_amd import dp
    import torch.distributed as dist
    dp.init_from_env(backend="gloo")
    L, nA, nB, nE, nAd = 3, 8, 12, 20, 16
    oA, oB, oE, oAd = 0, L * nA, L * (nA + nB), L * (nA + nB) + nE
    n = oAd + L * nAd
    flat_g = torch.zeros(n)
    pE = torch.nn.Parameter(torch.zeros(nE))
    vE = flat_g[oE:oE + nE]
    # (the A / B / adapter tensors only matter as spans here: the engine writes their gradients into the flat buffer itself)
    dummy = [torch.nn.Parameter(torch.zeros(1)) for _ in range(3)]
    views = [flat_g[oA:oA + L * nA], flat_g[oB:oB + L * nB], vE, flat_g[oAd:oAd + L * nAd]]
    params = [dummy[0], dummy[1], pE, dummy[2]]

    class Model:
        block_grad_hook = None
    model = Model()
    red = dp.OverlappedGradReducer(params, views, flat_g, [pE], world)
    assert red.late == [(0, oE), (oAd, n)]
    red.attach(model=model)
    assert model.block_grad_hook is not None
    truth = torch.arange(n, dtype=torch.float32) * (rank + 1)          # this rank's gradient of every element
    seen = torch.zeros(n)
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):                                 # every element of the flat buffer goes through exactly one collective
        if t.untyped_storage().data_ptr() == flat_g.untyped_storage().data_ptr():
            seen[t.storage_offset():t.storage_offset() + t.numel()] += 1
        return real_all_reduce(t, *a, **k)
    dist.all_reduce = counting_all_reduce
    # ---- the "backward": the loss-side tensor first (its hook fires), then the blocks from the top down
    (pE * truth[oE:oE + nE]).sum().backward()
    for i in reversed(range(L)):
        flat_g[oA + i * nA:oA + (i + 1) * nA] = truth[oA + i * nA:oA + (i + 1) * nA]
        flat_g[oAd + i * nAd:oAd + (i + 1) * nAd] = truth[oAd + i * nAd:oAd + (i + 1) * nAd]
        model.block_grad_hook(i, [(oA + i * nA, oA + (i + 1) * nA), (oAd + i * nAd, oAd + (i + 1) * nAd)])
    in_flight = len(red.works)
    flat_g[oB:oB + L * nB] = truth[oB:oB + L * nB]                      # finish_trainable_grads: LoRA-B lands after the backward
    left = red.remaining_late()
    red.wait_early()
    vE.copy_(pE.grad)                                                    # the gather of the hook-reduced tensor
    red.start()
    scale = red.finish()
    seen[oE:oE + nE] += 1                                                # (reduced as p.grad, not through the flat buffer)
    red.detach()
    dist.all_reduce = real_all_reduce
    out[rank] = ((flat_g * scale).tolist(), in_flight, left, seen.tolist(), model.block_grad_hook is None, red.done)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_per_block_late_chunks_world2(world):
    port = 29637 + world
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_blocks, args=(world, port, out), nprocs=world, join=True)
    L, nA, nB, nE, nAd = 3, 8, 12, 20, 16
    n = L * (nA + nB) + nE + L * nAd
    want = (torch.arange(n, dtype=torch.float32) * (sum(r + 1 for r in range(world)) / world)).tolist()
    for r in range(world):
        vals, in_flight, left, seen, unhooked, done = out[r]
        assert vals == want
        assert in_flight == 1 + 2 * L                                    # the hook's all-reduce + two ranges per block, all launched before the backward ended
        assert left == [(L * nA, L * (nA + nB))]                         # only the LoRA-B region is left for start()
        assert seen == [1.0] * n
        assert unhooked and done == []


class _GlooComm:
    """Stand-in for dp.RcclComm on CPU: the same two members (`world`, `all_reduce_(flat, algo)`), with algo 1 spelled the way
    gd_flat_allreduce spells it — reduce-scatter onto the rank's n / world slice, then all-gather in place."""

    def __init__(self, rank, world):
        self.rank, self.world, self.calls = rank, world, []

    def all_reduce_(self, flat, algo=0):
        import torch.distributed as dist
        self.calls.append(int(algo))
        if algo == 0:
            dist.all_reduce(flat)
            return flat
        assert flat.numel() % self.world == 0
        per = flat.numel() // self.world
        mine = torch.empty(per)
        # gloo has no reduce_scatter: the slice arithmetic of comm.hip (mine = buf + rank * per) on top of all_reduce
        tmp = flat.clone()
        dist.all_reduce(tmp)
        mine.copy_(tmp[self.rank * per:(self.rank + 1) * per])
        parts = [torch.empty(per) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        flat.copy_(torch.cat(parts))
        return flat


def _worker_direct(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import gd_amd  # noqa: F401
    from gd_amd import dp
    dp.init_from_env(backend="gloo")
    res = []
    for algo in (None, 0, 1):
        g = torch.arange(12, dtype=torch.float32) * (rank + 1)
        comm = _GlooComm(rank, world)
        red = dp.DirectGradReducer(g, comm, algo=algo)
        red.attach()
        red.wait_early()               # nothing is in flight before start(): the whole buffer goes in one exchange
        assert comm.calls == []
        red.start()
        scale = red.finish()
        red.detach()
        res.append(((g * scale).tolist(), comm.calls))
    out[rank] = res
    torch.distributed.destroy_process_group()


def test_direct_grad_reducer_contract_world2():
    """dp.DirectGradReducer (the `--exchange direct` path) against a communicator stand-in: same wait_early / start / finish
    contract as the overlapped reducer, exactly one exchange per step, the mean of the ranks' buffers for both algorithms."""
    world, port = 2, 29619
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_direct, args=(world, port, out), nprocs=world, join=True)
    want = (torch.arange(12, dtype=torch.float32) * 1.5).tolist()
    for r in range(world):
        for (vals, calls), algo in zip(out[r], (0, 0, 1)):
            assert vals == want and calls == [algo]


def test_one_hop_slice_arithmetic():
    """dp.one_hop_slices (the layout a hand-written one-hop xGMI exchange would use; host arithmetic only): the slices tile the buffer exactly,
    boundaries are page-aligned, every rank sends one message to every peer, and in every slot the targets are a permutation of the ranks —
    simulated on the host, reduce-scatter + ordered sum + all-gather reproduce the all-reduce for ragged sizes."""
    import torch
    from gd_amd.dp import one_hop_slices
    for n, world, align in [(6_397_440, 8, 1024), (6_397_440, 4, 1024), (5000, 8, 1024), (0, 2, 1024), (12_345_678, 8, 256), (1024 * 8, 8, 1024)]:
        bounds, plan = one_hop_slices(n, world, align)
        assert bounds[0][0] == 0 and bounds[-1][1] == n and all(bounds[j][1] == bounds[j + 1][0] for j in range(world - 1))
        assert all(a % align == 0 for a, _ in bounds if a < n)
        for r in range(world):
            assert sorted(p for p, _, _ in plan[r]) == [j for j in range(world) if j != r]
        for slot in range(world - 1):
            assert sorted(plan[r][slot][0] for r in range(world)) == list(range(world))
        if 0 < n <= 100_000:
            bufs = [torch.arange(n, dtype=torch.float64) * (r + 1) for r in range(world)]
            owned = []
            for j in range(world):
                a, b = bounds[j]
                owned.append(sum(bufs[r][a:b] for r in range(world)))          # rank order: deterministic
            full = torch.cat(owned)
            assert torch.equal(full, sum(bufs))
