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


def test_bench_n2_path_end_to_end_with_torchrun_and_gloo():
    # the driver's N > 1 launch line (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    # --master-port P bench.py --gpus N ...) rehearsed with two ranks sharing this box's one GPU: NGPDE_BENCH_BACKEND=gloo carries
    # the collectives through the host (the measured configuration is nccl = RCCL, one rank per GPU).  Rendezvous, per-rank plans,
    # gradient all-reduce, barrier + max over ranks, the C4 / C5 data-parallel legs and ONE JSON line from rank 0.
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NGPDE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "ODE-steps/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert d["value"] == pytest.approx(2 * 50 * 2 / (d["ms_per_step"] * 1e-3 * 2), rel=1e-3)   # whole-job aggregate over both ranks
    sec = d["secondary"]
    assert sec["C4_mppde_data_parallel_step"]["ranks"] == 2 and set(sec) == {"C4_mppde_data_parallel_step"}   # N > 1: the C2 step + this leg only


def _run_bench(extra_env, *argv, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (bench.py asserts nothing about the plan, these tests do: the child runs without the suite's library switches)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT") and not k.startswith("NGPDE_")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], cwd=root, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


def test_bench_plain_launch_spawns_its_ranks():
    # `python3 bench.py --gpus 2 ...` typed as is -- the command form the driver uses at N = 1 -- starts its own two ranks (fresh
    # child processes; the launcher never touches the GPU), relays rank 0's ONE JSON line and exits 0
    r, recs = _run_bench({"NGPDE_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-secondary")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(recs) == 1, r.stdout[-2000:]
    d = recs[0]
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and not d.get("skipped")
    assert d["value"] == pytest.approx(2 * 50 * 2 / (d["ms_per_step"] * 1e-3 * 2), rel=1e-3)
    assert "persistent_fwd" not in d["plan"]          # ranks sharing a device without taking turns: the replayed plan


def test_bench_plain_launch_with_the_persistent_plan_serialised_per_rank(tmp_path):
    # the same with the ranks taking turns on the one device through a lock file: persistent solve + adjoint, the gradient
    # collective and the fused Adam step of every rank on its one stream -- the sequence of the measured nccl configuration
    env = {"NGPDE_BENCH_BACKEND": "gloo", "NGPDE_BENCH_SERIALISE": str(tmp_path / "device.lock")}
    r, recs = _run_bench(env, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-secondary")
    if r.returncode != 0 and "gave up waiting for its neighbours" in r.stderr:
        # Two PROCESSES on one device is the one configuration the persistent plan does not support (include/ngpde.h, ngpde_node_flags):
        # the lock keeps their persistent launches apart, but not the other rank's allocations, copies and optimiser kernels, and the
        # driver may time-slice the two processes' queues -- seen once in ~10 full-suite runs (round 5): a launch's bounded wait ran
        # out and the plan latched its fault, as designed.  One more attempt; the measured configuration is one rank per GPU.
        r, recs = _run_bench(env, "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-secondary")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(recs) == 1, r.stdout[-2000:]
    d = recs[0]
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["fault"] is False
    assert "persistent_fwd" in d["plan"] and "persistent_bwd" in d["plan"]


def test_bench_more_gpus_than_devices_prints_a_skip_record():
    n = torch.cuda.device_count() + 7
    r, recs = _run_bench({}, "--gpus", str(n))
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(recs) == 1 and recs[0]["skipped"] is True and recs[0]["n_gpus"] == n and recs[0]["value"] is None


def test_two_persistent_plans_on_two_streams_take_turns():
    # two persistent solves in flight on one device could starve each other of residency (every workgroup of a launch must be
    # resident); inside a process the launches take turns (node_persistent.hip, turnstile): both finish, no fault, same bits as
    # alone
    import ngpde_amd as ng
    from ngpde_amd import _lib, synth as S
    from ngpde_amd.node import _Plan
    if any(os.environ.get(v) == "1" for v in ("NGPDE_NO_HALO", "NGPDE_NO_PERSISTENT", "NGPDE_NO_PRESCALE", "NGPDE_NO_MASK")):
        pytest.skip("no persistent plan under this switch")
    lib, p = _lib.load(), _lib.ptr
    N, D, STEPS = 16384, 64, 10
    _, s, t = S.closest_pairs_graph(N, 4 * N, seed=2)
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
    h = g.handle((True, None, False))
    dv = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device="cuda:0")
    w1, w2 = dv(S.glorot_uniform(11, D, D).T), dv(S.glorot_uniform(12, D, D).T)
    b1, b2 = torch.zeros(D, device="cuda:0"), torch.zeros(D, device="cuda:0")
    plans = [_Plan(h, D, _lib.ACT["relu"], "tsit5", STEPS, 1.0 / 50, True) for _ in range(2)]
    assert all("persistent_fwd" in pl.flags() for pl in plans)
    u0 = [dv(S.normal(1000 + k, D * N).reshape(N, D)) for k in range(2)]
    seed = torch.ones_like(u0[0])

    def solve(k, stream):
        outs = [torch.empty_like(u0[k]), torch.empty_like(u0[k]), torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)]
        _lib.check(lib.ngpde_node_gcn2_forward(plans[k].ptr, p(u0[k]), p(w1), p(b1), p(w2), p(b2), p(outs[0]), stream))
        _lib.check(lib.ngpde_node_gcn2_backward(plans[k].ptr, p(seed), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(outs[5]), stream))
        return outs

    cur = torch.cuda.current_stream().cuda_stream
    alone = [solve(k, cur) for k in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    for _ in range(5):
        both = [solve(k, streams[k].cuda_stream) for k in range(2)]
    torch.cuda.synchronize()
    for k in range(2):
        assert not plans[k].fault()
        assert all(torch.equal(a, b) for a, b in zip(alone[k], both[k]))
