"""CPU preflight of `bench.py --gpus N` (src/main.py:147-151 is the reference's DDP launch): the launcher bench.py uses for N > 1, Job.timed's
barrier / max-over-ranks clock and comm_report's whole A/B sequence (exchange off, the other exchange mode — which cannot be set up without RCCL
and must be skipped by EVERY rank —, per-block chunks off, the other CU reservation) at world 2 and world 8 over gloo, with a self-checking
stand-in for the engine (tests/bench_preflight_worker.py).  No scaling number comes out of this: it exists so that the first run on an 8-GPU node
cannot fail on plumbing."""
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_launcher_and_comm_report_control_flow(world, tmp_path):
    import bench
    out = str(tmp_path / "rank0.json")
    worker = os.path.join(ROOT, "tests", "bench_preflight_worker.py")
    rc = bench.launch_ranks(types.SimpleNamespace(gpus=world), script=worker, argv=[out])
    assert rc == 0
    with open(out) as fh:
        rec = json.load(fh)
    comm = rec["comm"]
    assert rec["world"] == world
    for k in ("allreduce_ms_early_chunk", "allreduce_ms_late_chunk", "ms_per_step_without_exchange", "exposed_comm_frac", "variants"):
        assert k in comm, comm
    v = comm["variants"]
    assert "exchange_error" in v                                   # no RCCL communicator on the CPU: every rank skipped the variant together
    assert "per_block_chunks_off" in v and "reserve_cus_8" in v
    # timed steps: the first run (1 warm-up + 2), exchange off (1 + 2), per-block off (1 + 2), the other reservation with and without exchange (2 x (1 + 2))
    assert rec["steps_run"] == 5 * 3
