"""GPU parity: fused cost-volume KL (C ABI) against the oracle and the reference-generated fixtures."""
import pytest
import torch

import gd_oracle as O
from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


def _oracle(f1, f2, t1, t2, m1, m2, variant):
    """per-pair loss + grads from the CPU oracle (fp64)."""
    losses, g1, g2 = [], [], []
    for p in range(f1.shape[0]):
        a = f1[p:p + 1].double().cpu().requires_grad_(True)
        b = f2[p:p + 1].double().cpu().requires_grad_(True)
        l = O.cost_volume_kl(a, b, t1[p:p + 1].double().cpu(), t2[p:p + 1].double().cpu(), m1[p].cpu(), m2[p].cpu(), variant)
        l.backward()
        losses.append(l.detach())
        g1.append(a.grad[0])
        g2.append(b.grad[0])
    return torch.stack(losses), torch.stack(g1), torch.stack(g2)


@pytest.mark.parametrize("variant", ["vggt", "mast3r"])
def test_golden_fixture(variant):
    from gd_amd import ops
    g = load_golden(f"g06_cost_{variant}")
    f1 = g["f1"][None].cuda().requires_grad_(True)
    f2 = g["f2"][None].cuda().requires_grad_(True)
    loss = ops.cost_volume_kl(f1, f2, g["t1"][None].cuda(), g["t2"][None].cuda(), g["m1"][None].cuda(),
                              g["m2"][None].cuda(), variant)
    assert abs(loss.item() - g["loss"]) < 2e-5 * abs(g["loss"])
    loss.sum().backward()
    assert rel_err(f1.grad[0], g["g1"]) < 2e-4 and rel_err(f2.grad[0], g["g2"]) < 2e-4


@pytest.mark.parametrize("variant", ["vggt", "mast3r"])
@pytest.mark.parametrize("P,hw,C", [(2, 100, 64), (1, 333, 96), (2, 672, 128)])
def test_vs_oracle_f32(variant, P, hw, C):
    from gd_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(hw)
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    m1 = torch.rand(P, hw, generator=gen, device="cuda") > 0.3
    m2 = torch.rand(P, hw, generator=gen, device="cuda") > 0.3
    m2[0] = False                                   # a fully masked direction
    loss = ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, variant)
    w = torch.tensor([1.0, 0.5][:P], device="cuda")
    (loss * w).sum().backward()
    ol, og1, og2 = _oracle(f1.detach(), f2.detach(), t1, t2, m1, m2, variant)
    assert rel_err(loss, ol) < 1e-5
    assert rel_err(f1.grad, og1 * w.cpu()[:, None, None]) < 1e-4
    assert rel_err(f2.grad, og2 * w.cpu()[:, None, None]) < 1e-4


def test_bf16_full_size():
    """hw = 1369 (37x37), C = 768, bf16 features: loss within 1e-3 rel of the fp64 oracle on the
    same (bf16-rounded) inputs; gradients to bf16 accuracy."""
    from gd_amd import ops
    P, hw, C = 1, 1369, 768
    gen = torch.Generator(device="cuda").manual_seed(7)
    f1 = torch.randn(P, hw, C, generator=gen, device="cuda").bfloat16().requires_grad_(True)
    f2 = torch.randn(P, hw, C, generator=gen, device="cuda").bfloat16().requires_grad_(True)
    t1 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    t2 = torch.softmax(3 * torch.randn(P, hw, hw, generator=gen, device="cuda"), -1)
    m1 = torch.rand(P, hw, generator=gen, device="cuda") > 0.3
    m2 = torch.rand(P, hw, generator=gen, device="cuda") > 0.3
    loss = ops.cost_volume_kl(f1, f2, t1, t2, m1, m2, "vggt")
    loss.sum().backward()
    ol, og1, og2 = _oracle(f1.detach().float(), f2.detach().float(), t1, t2, m1, m2, "vggt")
    assert rel_err(loss, ol) < 1e-3
    assert rel_err(f1.grad.float(), og1) < 3e-2 and rel_err(f2.grad.float(), og2) < 3e-2
