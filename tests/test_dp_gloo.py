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
