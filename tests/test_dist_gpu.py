"""Two-rank rehearsal of the data-parallel path on ONE GPU (the ranks share the device; gloo carries the collective): BASELINE
config 4's sharding -- whole trajectories per rank, replicated parameters, bucketed all-reduce of the flat gradient
overlapped with the pullback -- must give every rank the gradient of the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N_MESH, TRAJ_PER_RANK, H = 512, 2, 64


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard(first, count):
    """MPPDEConv on trajectories first .. first + count - 1 of one fixed 4-trajectory data set (C4's structure at test size)"""
    import ngpde_amd as ng
    from ngpde_amd import synth as S
    traj_all = 2 * TRAJ_PER_RANK
    N_all = N_MESH * traj_all
    u = S.uniform01(7, N_all).reshape(1, N_all).astype(np.float32)
    xs = np.tile(np.arange(N_MESH) / N_MESH, traj_all)[None, :].astype(np.float32)
    th = S.uniform01(8, 2 * traj_all).reshape(2, traj_all).astype(np.float32)
    x_all = S.normal(9, H * N_all).reshape(N_all, H).astype(np.float32)
    R_all = S.normal(10, H * N_all).reshape(N_all, H).astype(np.float32)
    sl = slice(first * N_MESH, (first + count) * N_MESH)
    s, t = S.periodic_mesh_batch(N_MESH, count)
    g = ng.GNNGraph(s, t, num_nodes=N_MESH * count, index_base=0, num_graphs=count, ndata={"u": u[:, sl], "x": xs[:, sl]},
                    gdata={"θ": th[:, first:first + count]})
    phi = ng.Chain(ng.Dense(132, 64, "swish"), ng.Dense(64, 64, "swish"))
    psi = ng.Chain(ng.Dense(130, 64, "swish"), ng.Dense(64, 64))
    layer = ng.MPPDEConv(phi, psi, initialgraph=g)
    ps, st = ng.setup(4, layer)                       # same seed on every rank: replicated parameters
    flat, psv = ng.optim.flatten_parameters(ng.to_device(ps, "cuda:0"))
    x = torch.as_tensor(x_all[sl], device="cuda:0").T
    R = torch.as_tensor(R_all[sl], device="cuda:0").T
    return ng, layer, flat, psv, st, x, R


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ng, layer, flat, psv, st, x, R = _shard(rank * TRAJ_PER_RANK, TRAJ_PER_RANK)
    red = ng.dist.OverlappedGradReduce(flat, psv, [("ψ.",), ("ϕ.",)])
    layer(x, psv, st)[0].backward(R)
    red.finish()
    torch.cuda.synchronize()
    out[rank] = flat.grad.cpu().numpy()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_c4_gradient_equals_the_single_rank_gradient_of_the_whole_batch():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ng, layer, flat, psv, st, x, R = _shard(0, 2 * TRAJ_PER_RANK)
    layer(x, psv, st)[0].backward(R)
    ref = flat.grad.cpu().numpy()
    assert np.array_equal(out[0], out[1])                           # both ranks hold the same reduced vector
    err = np.abs(out[0] - ref).max()
    assert err <= 2e-5 * np.abs(ref).max() + 1e-6, err              # sum of two half-batch gradients == whole-batch gradient
