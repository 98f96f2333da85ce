"""Fused optimiser step on the flat parameter vector (ngpde_adam_step / ngpde_rprop_step) against the numpy restatement of
the Optimisers.jl rules, and the flat-view plumbing (gradients of a layer land in the flat buffer; one kernel updates
every parameter).  Reference call sites: docs/src/tutorials/graph_node.md:90,122-129, VMH.md:97."""
import numpy as np
import pytest
import torch

import ngpde_amd as ng
from ngpde_amd import optim
from oracle import ngpde_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def flat_of(x):
    data = torch.as_tensor(x, device=DEV).clone()
    return optim.FlatParameters(data, torch.zeros_like(data), [("x", 0, tuple(data.shape))])


@pytest.mark.parametrize("n", [1, 8320, 100003])
def test_adam_matches_rule(n):
    rng = np.random.default_rng(n)
    x = rng.normal(size=n).astype(np.float32)
    flat = flat_of(x)
    rule = optim.Adam(0.01)
    st = optim.setup(rule, flat)
    ost = O.adam_init(x)
    for it in range(6):
        g = (rng.normal(size=n) * 10.0 ** rng.integers(-4, 2)).astype(np.float32)
        flat.grad.copy_(torch.as_tensor(g))
        st = optim.update(st, flat)
        x, ost = O.adam_step(x, g, ost, 0.01)
        np.testing.assert_allclose(flat.data.cpu().numpy(), x, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(st["state"]["m"].cpu().numpy(), ost["m"], rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(st["state"]["v"].cpu().numpy(), ost["v"], rtol=2e-6, atol=1e-20)


def test_adam_first_step_closed_form():
    # t = 1: m/(1-b1) = g, v/(1-b2) = g^2  =>  x1 = x0 - eta * g / (|g| + eps)
    x = np.zeros(5, np.float32)
    g = np.array([3.0, -0.5, 1e-3, -7.0, 0.0], np.float32)
    flat = flat_of(x)
    st = optim.setup(optim.Adam(0.1), flat)
    flat.grad.copy_(torch.as_tensor(g))
    optim.update(st, flat)
    np.testing.assert_allclose(flat.data.cpu().numpy(), -0.1 * g / (np.abs(g) + 1e-8), rtol=1e-5, atol=1e-9)


def test_rprop_matches_rule_through_sign_flips():
    rng = np.random.default_rng(4)
    n = 5000
    x = rng.normal(size=n).astype(np.float32)
    flat = flat_of(x)
    rule = optim.Rprop(1e-6, (0.5, 1.2), (1e-8, 10.0))             # the tutorial's setting (VMH.md:97)
    st = optim.setup(rule, flat)
    ost = O.rprop_init(x, 1e-6)
    for it in range(12):
        g = rng.normal(size=n).astype(np.float32)
        g[rng.random(n) < 0.1] = 0.0
        flat.grad.copy_(torch.as_tensor(g))
        st = optim.update(st, flat)
        x, ost = O.rprop_step(x, g, ost, (0.5, 1.2), (1e-8, 10.0))
        assert np.array_equal(flat.data.cpu().numpy(), x)           # only comparisons, one multiply, one subtract: bit-exact
        assert np.array_equal(st["state"]["step"].cpu().numpy(), ost["step"])


def test_flat_views_collect_layer_gradients_and_one_step_updates_all():
    g = ng.rand_graph(60, 300, seed=0)
    model = ng.Chain(ng.GCNConv((8, 8), "relu", initialgraph=g), ng.GCNConv((8, 4), "identity", initialgraph=g))
    ps, st = ng.setup(0, model)
    flat, psv = optim.flatten_parameters(ng.to_device(ps, DEV))
    assert flat.numel() == 8 * 8 + 8 + 8 * 4 + 4 and [t[0] for t in flat.table] == ["layer_1.weight", "layer_1.bias", "layer_2.weight", "layer_2.bias"]
    x = torch.randn(8, 60, device=DEV)
    y, _ = model(x, psv, st)
    y.sum().backward()
    # every leaf's gradient is a window of the flat gradient
    ref = torch.cat([psv["layer_1"]["weight"].grad.reshape(-1), psv["layer_1"]["bias"].grad.reshape(-1),
                     psv["layer_2"]["weight"].grad.reshape(-1), psv["layer_2"]["bias"].grad.reshape(-1)])
    assert torch.equal(ref, flat.grad) and float(flat.grad.abs().sum()) > 0
    before = flat.data.clone()
    opt = optim.setup(optim.Adam(0.01), flat)
    optim.update(opt, flat)
    moved = (flat.data - before).abs()
    assert float(moved[flat.grad != 0].min()) > 0                    # one launch moved every parameter with a gradient
    assert torch.equal(psv["layer_2"]["bias"].detach().reshape(-1), flat.data[-4:])   # leaves alias the flat vector
    flat.zero_grad()
    y2, _ = model(x, psv, st)
    assert not torch.equal(y, y2)


def test_optimiser_needs_device_parameters():
    with pytest.raises(ng.NgpdeError):
        optim.setup(optim.Adam(), optim.FlatParameters(torch.zeros(3), torch.zeros(3), []))


def test_native_communicator_world_of_one_and_fused_adam():
    # ngpde_comm_* / ngpde_grad_allreduce_adam (comm.hip): the data-parallel step behind the C ABI, RCCL on the caller's stream.  A
    # 1-GPU box allows a world of one (RCCL refuses two ranks on one device; N > 1 is the driver's measurement): the all-reduce is
    # then the identity and the fused step must equal ngpde_adam_step bit for bit.
    import ngpde_amd as ng
    from ngpde_amd import _lib
    comm = ng.dist.NativeComm(ng.dist.NativeComm.unique_id(), 0, 1)
    assert (comm.rank, comm.world) == (0, 1)
    assert comm.rccl_count_and_rank() == (1, 0)          # what RCCL itself reports (ncclCommCount, ncclCommUserRank)
    rng = np.random.default_rng(5)
    n = 8320
    x0 = torch.as_tensor(rng.normal(size=n).astype(np.float32), device="cuda:0")
    g0 = torch.as_tensor(rng.normal(size=n).astype(np.float32), device="cuda:0")
    g = g0.clone()
    comm.all_reduce(g)
    torch.cuda.synchronize()
    assert torch.equal(g, g0)
    lib, p = _lib.load(), _lib.ptr
    xa, ma, va = x0.clone(), torch.zeros_like(x0), torch.zeros_like(x0)
    xb, mb, vb = x0.clone(), torch.zeros_like(x0), torch.zeros_like(x0)
    for step in (1, 2, 3):
        comm.all_reduce_adam(xa, g0.clone(), ma, va, 1e-3, 0.9, 0.999, 1e-8, step)
        _lib.check(lib.ngpde_adam_step(n, p(xb), p(g0), p(mb), p(vb), 1e-3, 0.9, 0.999, 1e-8, step, 1.0, _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(xa, xb) and torch.equal(ma, mb) and torch.equal(va, vb) and not torch.equal(xa, x0)
    comm.close()
