"""Host-side logic that needs no GPU: the option table and the kept-row capacity rule of the cost-volume op."""
import pytest


def test_kept_row_capacity_rule():
    """ops.kept_row_capacity: the bound rounded up to whole 128-row tiles, 0 (= dense sweep) without a bound or when the two compacted
    problems would cover more rows than the dense one (src/finetune_timm_mast3r.py:515-519 keeps <= N_kp of the hw patch rows)."""
    from gd_amd import ops
    assert ops.kept_row_capacity(1369, 300) == 384          # the benched MASt3R step: 37 x 37 patches, 300 keypoints
    assert ops.kept_row_capacity(1369, 1) == 128
    assert ops.kept_row_capacity(1369, 384) == 384 and ops.kept_row_capacity(1369, 385) == 512
    assert ops.kept_row_capacity(1369, 641) == 0            # 2 x 768 > 1369: the dense sweep is cheaper
    assert ops.kept_row_capacity(1369, None) == 0 and ops.kept_row_capacity(1369, 0) == 0
    assert ops.kept_row_capacity(200, 50) == 0              # small grids: one 128-row tile per direction already exceeds half of hw
    assert ops.kept_row_capacity(256, 5000) == 0            # a bound above hw is clipped to hw first


def test_option_table_defaults_and_override():
    """options.py: every switch has an environment variable and a default, set_option returns the old value and rejects unknown names."""
    from gd_amd import options
    for name, (env, default) in options._DEFS.items():
        assert env.startswith("GD_") and isinstance(default, int), name
    assert options.option("cv_bwd_rows") in (0, 1)
    old = options.set_option("cv_bwd_rows", 0)
    try:
        assert options.option("cv_bwd_rows") == 0
    finally:
        options.set_option("cv_bwd_rows", old)
    assert options.option("cv_bwd_rows") == old
    with pytest.raises(KeyError):
        options.set_option("no_such_option", 1)


def test_block_grad_gate_fires_after_the_last_backward_node_only():
    """vit.BlockGradGate (data parallelism, the per-block gradient exchange): a block that ran in TWO forwards of one step (geometry="reference":
    the keypoint grid and the cost grid through the same blocks) has two backward nodes accumulating into the same slices of the flat gradient
    buffer — the exchange hook may only run when the second one is done, and exactly once; a node that never runs backward leaves the slices to
    the reducer's late ranges (the hook stays silent)."""
    from gd_amd.vit import BlockGradGate
    calls = []
    gate = BlockGradGate(lambda i, spans: calls.append((i, spans)), 5, [(0, 8), (64, 96)])
    gate.armed()
    gate.armed()                       # second forward through the block
    assert gate.node_done() is False and calls == []
    assert gate.node_done() is True and calls == [(5, [(0, 8), (64, 96)])]
    import pytest
    with pytest.raises(RuntimeError):
        gate.armed()                   # a third forward after the slices were handed over would accumulate into memory under exchange
    one = BlockGradGate(lambda i, spans: calls.append(i), 1, [])
    one.armed()
    assert one.node_done() is True and calls[-1] == 1
    idle = BlockGradGate(lambda i, spans: calls.append("never"), 2, [])
    idle.armed()
    idle.armed()
    assert idle.node_done() is False and "never" not in calls      # the other node's output did not reach the loss: nothing is exchanged early
