"""Child job of test_gpu_step.py::test_two_ranks_on_one_gpu_equal_one_rank_on_the_whole_batch: one rank of a 2-rank gloo
group, every rank on GPU 0.  Takes its contiguous half of a seeded 4-pair batch (dp.shard_pairs), runs one
FinetuneGD.fit_step with the hook-driven gradient exchange (loss-side tensors from grad hooks, the blocks' slices as each block's
backward returns, the rest after the backward) and (rank 0) saves loss / summed gradient / updated weights."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p_ in (ROOT, HERE):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def make_engine(geometry=None):
    """geometry "reference": two forwards per image through the same blocks (a 16 x 20-token keypoint grid — target_res 160 — and the 4 x 5 cost grid):
    every block has TWO backward nodes per step, and the per-block exchange must wait for the second (vit.BlockGradGate)."""
    import gd_amd  # noqa: F401
    from gd_amd.finetune import FinetuneGD
    geometry = geometry or os.environ.get("DP_GEOMETRY", "shared")
    torch.manual_seed(0)
    eng = FinetuneGD(r=4, backbone="vit_tiny_test", patch_size=14, img_size=56, variant="vggt", geometry=geometry, dtype="f32",
                     lora_b_std=0.05, vit_kwargs=dict(init_values=1.0), teacher_patch=14).cuda()
    if geometry == "reference":
        eng.target_res = 160
    return eng


def make_batch(lo, hi):
    from gd_testutil import synthetic_batch
    full = synthetic_batch(4, 56, 70, 12, 20, "cuda", seed=21)
    return {k: v[lo:hi].contiguous() for k, v in full.items()}


def main():
    from gd_amd import dp
    torch.cuda.set_device(0)
    rank, _, world = dp.init_from_env(backend="gloo")
    lo, hi = dp.shard_pairs(4, rank, world)
    eng = make_engine()
    flat = eng.configure_optimizers()
    early = list(eng.refine_conv.parameters()) + list(eng.depth_diff_head.parameters())
    class Reducer(dp.OverlappedGradReducer):       # (records which blocks reported their weight-gradient slices while the backward was running)
        blocks = []

        def _block_done(self, i, spans):
            self.blocks.append((i, len(self.works)))
            super()._block_done(i, spans)
    red = Reducer(eng.trainable_parameters(), flat["views"], flat["g"], early, world)
    red.attach(model=eng.model)
    loss, _, norm = eng.fit_step(make_batch(lo, hi), red)
    lm = torch.tensor([loss.item()], dtype=torch.float64)
    torch.distributed.all_reduce(lm)
    if rank == 0:
        torch.save({"loss_mean": lm.item() / world, "norm": norm.item(), "grad": flat["g"].cpu(), "params": flat["p"].cpu(),
                    "blocks": [b for b, _ in red.blocks]}, sys.argv[1])
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
