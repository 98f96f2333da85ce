"""Optimisers on the flat parameter vector.

The reference's training loops hold the parameters as ONE ComponentArray and update it with Optimisers.jl
(/root/reference/docs/src/tutorials/graph_node.md:90,122-129: `ComponentArray(ps)`, `Optimisers.Adam(0.01f0)`,
`Optimisers.setup`, `Optimisers.update`; VMH.md:97: `Rprop(1f-6, (0.5, 1.2), (1f-8, 10))`).  Here:

    flat, ps = flatten_parameters(ps)        # one fp32 buffer in HBM; every leaf of `ps` becomes a view into it,
                                             # and every leaf's .grad a view into one flat gradient buffer
    opt = Adam(0.01); st_opt = setup(opt, flat)
    loss.backward()                          # gradients land in the flat buffer
    st_opt = update(st_opt, flat)            # [all-reduce of the flat gradient over the DP group] + ONE fused launch

`update` enqueues the all-reduce (when torch.distributed is initialised) and the optimiser kernel on the current
stream; the 1/world_size of a data-parallel mean is folded into the kernel.  No CPU path: the step is a HIP kernel.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import _lib


class FlatParameters:
    """the flat parameter vector, its gradient vector and the (name, offset, shape) table -- ComponentArray's role"""

    def __init__(self, data, grad, table):
        self.data, self.grad, self.table = data, grad, table

    def zero_grad(self):
        self.grad.zero_()

    def numel(self):
        return self.data.numel()


def flatten_parameters(ps, device=None):
    """(FlatParameters, tree of views): leaves in insertion order (ComponentArray order); views require grad and own a
    .grad that aliases the flat gradient buffer, so autograd accumulates straight into it."""
    leaves = []

    def walk(t, prefix):
        for k, v in t.items():
            if isinstance(v, dict):
                walk(v, prefix + k + ".")
            else:
                leaves.append((prefix + k, v))
    walk(ps, "")
    if device is None:
        device = next((v.device for _, v in leaves if isinstance(v, torch.Tensor)), torch.device("cpu"))
    total = sum(int(v.numel()) for _, v in leaves)
    data = torch.empty(total, dtype=torch.float32, device=device)
    grad = torch.zeros(total, dtype=torch.float32, device=device)
    table, views, off = [], {}, 0
    for name, v in leaves:
        n = int(v.numel())
        data[off:off + n].copy_(torch.as_tensor(v).detach().to(device, torch.float32).reshape(-1))
        view = data[off:off + n].view(tuple(v.shape)).requires_grad_(True)
        view.grad = grad[off:off + n].view(tuple(v.shape))
        views[name] = view
        table.append((name, off, tuple(v.shape)))
        off += n

    def rebuild(t, prefix):
        return {k: (rebuild(v, prefix + k + ".") if isinstance(v, dict) else views[prefix + k]) for k, v in t.items()}
    return FlatParameters(data, grad, table), rebuild(ps, "")


class Adam:
    """Optimisers.Adam(eta = 0.001, beta = (0.9, 0.999), epsilon = 1e-8)  [UPSTREAM Optimisers.jl]"""

    def __init__(self, eta=0.001, beta=(0.9, 0.999), epsilon=1e-8):
        self.eta, self.beta, self.epsilon = float(eta), (float(beta[0]), float(beta[1])), float(epsilon)

    def init(self, flat):
        return {"m": torch.zeros_like(flat.data), "v": torch.zeros_like(flat.data), "t": 0}

    def apply(self, state, flat, grad_scale):
        state["t"] += 1
        _lib.check(_lib.load().ngpde_adam_step(flat.numel(), _lib.ptr(flat.data), _lib.ptr(flat.grad), _lib.ptr(state["m"]),
                                               _lib.ptr(state["v"]), self.eta, self.beta[0], self.beta[1], self.epsilon,
                                               state["t"], grad_scale, _lib.current_stream()))
        return state


class Rprop:
    """Optimisers.Rprop(eta = 1e-3, l = (0.5, 1.2), gamma = (1e-6, 50))  [UPSTREAM Optimisers.jl]; VMH.md:97"""

    def __init__(self, eta=1e-3, ell=(0.5, 1.2), gamma=(1e-6, 50.0)):
        self.eta, self.ell, self.gamma = float(eta), (float(ell[0]), float(ell[1])), (float(gamma[0]), float(gamma[1]))

    def init(self, flat):
        return {"g": torch.zeros_like(flat.data), "step": torch.full_like(flat.data, self.eta)}

    def apply(self, state, flat, grad_scale):
        _lib.check(_lib.load().ngpde_rprop_step(flat.numel(), _lib.ptr(flat.data), _lib.ptr(flat.grad), _lib.ptr(state["g"]),
                                                _lib.ptr(state["step"]), self.ell[0], self.ell[1], self.gamma[0], self.gamma[1],
                                                grad_scale, _lib.current_stream()))
        return state


def setup(rule, flat):
    """Optimisers.setup(rule, ps)"""
    if not flat.data.is_cuda:
        raise _lib.NgpdeError(_lib.ERR_INVALID_ARGUMENT, "the optimiser step is a HIP kernel: parameters must live on the GPU")
    return {"rule": rule, "state": rule.init(flat)}


def update(st_opt, flat, group=None, average=True, reduced=False):
    """Optimisers.update(st_opt, ps, gs) on the flat vector, in place: one all-reduce(sum) of the flat gradient over the
    data-parallel group (if initialised), then ONE kernel that scales, updates the moments and the parameters.
    With more than one rank `flat.grad` is overwritten IN PLACE by the cross-rank SUM (the 1/world factor of `average` is
    applied inside the kernel only): anything that reads flat.grad afterwards -- logging, clipping, a second optimiser --
    sees the sum, world times the mean.  The kernel itself does not modify the buffer; call flat.zero_grad() before the
    next backward."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world > 1 and not reduced:      # reduced=True: dist.OverlappedGradReduce has already summed flat.grad over the ranks
        dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM, group=group)
    with torch.no_grad():
        st_opt["state"] = st_opt["rule"].apply(st_opt["state"], flat, (1.0 / world) if average else 1.0)
    return st_opt
