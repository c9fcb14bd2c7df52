"""Data parallelism for the distillation step: one process per GPU, image pairs sharded by
rank, and ONE exchange per step — a sum all-reduce of the flat fp32 gradient buffer (6.39 M floats = 25.6 MB
for ViT-B) over RCCL/xGMI (backend "nccl" on ROCm; "gloo" for the CPU tests).  The 1/world mean and the
1/pairs factor are folded into the fused clip+AdamW kernel's grad_scale, so no extra pass touches the buffer.
Replaces Lightning's DDP(strategy='ddp_find_unused_parameters_true') (src/main.py:147-151): the set of
parameters without gradient (depth_attention) is static here — their slice of the flat buffer just stays zero."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class FlatGradReducer:
    """Sum-all-reduce of the flat gradient buffer, launched asynchronously so the caller can overlap it with the
    remaining backward work (the refine_conv / depth-head slices are complete first: they sit after the ViT)."""

    def __init__(self, flat_grad, world):
        self.flat, self.world, self.work = flat_grad, world, None

    def start(self):
        if self.world > 1:
            self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return 1.0 / self.world     # the mean factor for the optimiser's grad_scale


class OverlappedGradReducer:
    """Two-chunk exchange: the tensors whose gradients are complete at the START of the backward pass — the
    loss-side parameters: `refine_conv` (21 of the 25.6 MB for ViT-B) and the depth head — are sum-all-reduced from a
    post-accumulate-grad hook while the ViT backward is still running; the LoRA / adapter slices of the flat buffer follow
    after the backward.  Use with FinetuneGD.backward(loss, pre_gather=reducer.wait_early):

        reducer = OverlappedGradReducer(params, views, flat_g, early, world); reducer.attach()
        eng.backward(loss, pre_gather=reducer.wait_early); reducer.start(); scale = reducer.finish()

    params / views: the trainable tensors and their views into flat_g (same order); early: the subset reduced from hooks.

    Per-block late chunks (`attach(model=...)`): with the weight gradients accumulated straight into the flat buffer
    (GDViT.prepare_trainables(flat): FinetuneGD.fit_step's default) a block's LoRA-A and adapter slices are FINAL when that block's backward
    returns, eleven-to-one blocks before the backward ends: the model calls `_block_done(i, spans)` there and the slices are all-reduced
    under the backward of the blocks below (2 ranges per block: 6 144 + 98 304 floats for ViT-B at r = 4, bottleneck 64).  `start()` then
    exchanges only what is left of the late ranges — the LoRA-B slices, which `finish_trainable_grads` transposes into place after the
    backward (3 % of the buffer), and alignment pads."""

    def __init__(self, params, views, flat_grad, early, world):
        self.world, self.flat = world, flat_grad
        self.model, self.done = None, []
        self.per_block = True      # False: the blocks' slices wait for start() with the rest of the late ranges (bench.py's comm.variants measures both)
        early_ids = {id(p) for p in early}
        self.early = [p for p in params if id(p) in early_ids]
        base = flat_grad.data_ptr()
        spans = sorted(((v.data_ptr() - base) // 4, v.numel()) for p, v in zip(params, views) if id(p) in early_ids)
        # the complement of the early spans inside the flat buffer = what is reduced after the backward
        self.late, pos = [], 0
        for off, n in spans:
            if off > pos:
                self.late.append((pos, off))
            pos = max(pos, off + n)
        if pos < flat_grad.numel():
            self.late.append((pos, flat_grad.numel()))
        self.works, self.handles = [], []

    def attach(self, model=None):
        if self.world > 1 and not self.handles:
            for p in self.early:
                self.handles.append(p.register_post_accumulate_grad_hook(self._hook))
        if model is not None:
            self.model = model
        if self.world > 1 and self.model is not None and self.per_block:
            self.model.block_grad_hook = self._block_done

    def detach(self):
        for h in self.handles:
            h.remove()
        self.handles = []
        if self.model is not None:
            self.model.block_grad_hook = None

    def _block_done(self, i, spans):
        """called by the model when block i's weight gradients are complete in the flat buffer; spans: [(a, b), ...] element ranges of it"""
        for a, b in spans:
            self.works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))
        self.done.extend(spans)

    def remaining_late(self):
        """the late ranges minus what `_block_done` already exchanged during this backward"""
        out = []
        done = sorted(self.done)
        for a, b in self.late:
            pos = a
            for c, d in done:
                if d <= pos or c >= b:
                    continue
                if c > pos:
                    out.append((pos, c))
                pos = max(pos, d)
            if pos < b:
                out.append((pos, b))
        return out

    def _hook(self, p):
        if p.grad is not None:
            if not p.grad.is_contiguous():      # collectives need dense buffers (a gradient may arrive as a strided view)
                p.grad = p.grad.contiguous()
            self.works.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True))

    def wait_early(self):
        """before the gradients are gathered into the flat buffer: the early all-reduces must have landed"""
        for w in self.works:
            w.wait()
        self.works = []

    def start(self):
        if self.world > 1:
            self.works += [dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True) for a, b in self.remaining_late()]
        self.done = []

    def finish(self):
        self.wait_early()
        return 1.0 / self.world


class RcclComm:
    """The C ABI's own RCCL communicator (gd_comm_init / gd_flat_allreduce in include/gd_hip.h): the flat-buffer exchange
    without torch.distributed in the data path.  The 128-byte unique id travels over the already-initialised
    torch.distributed group (any backend) once, at construction."""

    def __init__(self, rank, world):
        import ctypes
        from ._lib import check, lib
        self.rank, self.world = rank, world
        buf = (ctypes.c_char * 128)()
        if rank == 0:
            check(lib().gd_comm_unique_id(buf), "gd_comm_unique_id")
        ids = [bytes(buf)]
        if world > 1:
            dist.broadcast_object_list(ids, src=0)
        self._id = ctypes.create_string_buffer(ids[0], 128)
        self._h = ctypes.c_void_p()
        check(lib().gd_comm_init(ctypes.byref(self._h), world, rank, self._id), "gd_comm_init")

    def all_reduce_(self, flat, algo=None):
        """In place sum over ranks of a contiguous fp32 CUDA tensor, on the current stream.  algo 0 / None: one ncclAllReduce;
        algo 1: ncclReduceScatter + ncclAllGather on the rank's slice (length must divide by the world size).  RCCL picks the
        schedule either way; which of the two is faster on an 8-GPU xGMI node is unmeasured (no such node in rounds 1-3)."""
        from ._lib import check, lib, stream
        assert flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()
        if algo is None:
            algo = 0
        check(lib().gd_flat_allreduce(self._h, flat.data_ptr(), flat.numel(), self.world, self.rank, int(algo),
                                      stream()), "gd_flat_allreduce")
        return flat

    def close(self):
        from ._lib import lib
        if self._h:
            lib().gd_comm_destroy(self._h)
            self._h = None


class DirectGradReducer:
    """Same contract as OverlappedGradReducer (wait_early / start / finish), one gd_flat_allreduce of the whole flat
    gradient buffer after the backward, on the compute stream: no bucketing, no hooks, no torch.distributed in the step.
    EXPERIMENTAL: no N > 1 hardware run exists yet (`bench.py --exchange direct`); `comm` is anything with `.world` and
    `.all_reduce_(flat)` (RcclComm on GPUs; tests/test_dp_gloo.py drives the contract with a torch.distributed stand-in)."""

    def __init__(self, flat_grad, comm, algo=None):
        self.flat, self.comm, self.world, self.algo = flat_grad, comm, comm.world, algo

    def attach(self):
        pass

    def detach(self):
        pass

    def wait_early(self):
        pass

    def start(self):
        if self.world > 1:
            self.comm.all_reduce_(self.flat) if self.algo is None else self.comm.all_reduce_(self.flat, self.algo)

    def finish(self):
        return 1.0 / self.world


def reserve_cus_for_collectives(k):
    """Leave k compute units out of every persistent kernel's grid (csrc/gd_knobs.h `reserve_cus`): the backward of this engine is a chain of
    one-block-per-CU launches that hold all 256 CUs, so an RCCL kernel issued from a gradient hook finds no CU until a launch ends.  Whether the
    overlapped exchange needs this — and which k pays — can only be measured on a multi-GPU node: bench.py --reserve-cus k (A/B in its `comm`
    object).  Process-wide; 0 restores the full grid."""
    from ._lib import check, lib
    check(lib().gd_debug_set(b"reserve_cus", int(k)), "gd_debug_set(reserve_cus)")


def one_hop_slices(n, world, align=1024):
    """The slice arithmetic of a hand-written one-hop exchange over the fully connected xGMI mesh (NOT built: it needs peer-mapped
    buffers and a multi-GPU node to validate — `gd_flat_allreduce` algo 0 / 1 are RCCL-scheduled): the flat buffer of n floats is cut into
    `world` contiguous slices whose boundaries are multiples of `align` floats (whole 4-KB pages of fp32: one DMA descriptor per slice and peer);
    rank j OWNS slice j — every rank pushes its copy of slice j to rank j over its own link (reduce-scatter: world - 1 messages in, all at
    once), rank j sums the world copies in rank order (deterministic) and pushes the result back to every peer (all-gather).
    -> (bounds, push_plan): bounds[j] = (start, stop) of slice j (the last one takes the remainder; empty slices when n < world * align);
    push_plan[r] = [(peer, start, stop), ...] the reduce-scatter messages rank r sends, in the order that starts with its right-hand neighbour
    so that no two ranks target the same peer in the same slot."""
    assert n >= 0 and world >= 1 and align >= 1
    per = (n // world) // align * align
    bounds = []
    for j in range(world):
        a = min(j * per, n)
        b = n if j == world - 1 else min((j + 1) * per, n)
        bounds.append((a, b))
    plan = []
    for r in range(world):
        plan.append([((r + k) % world,) + bounds[(r + k) % world] for k in range(1, world)])
    return bounds, plan


def shard_pairs(n_pairs, rank, world):
    """Contiguous split of a global batch of pairs."""
    per = n_pairs // world
    assert per * world == n_pairs, "global batch must divide by the world size"
    return rank * per, (rank + 1) * per


def max_over_ranks(x, device):
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()
