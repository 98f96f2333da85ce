"""Multi-process path on CPU: gloo, world size 2 -- the gradient all-reduce used by bench.py --gpus N."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ngpde_amd as ng


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(rank)
    grads = {"layer_1": {"weight": torch.full((4, 4), float(rank + 1)), "bias": torch.full((4, 1), 10.0 * (rank + 1))},
             "layer_2": {"weight": torch.arange(16.0).reshape(4, 4) * (rank + 1)}}
    ng.dist.allreduce_gradients(grads)
    ok = (torch.all(grads["layer_1"]["weight"] == 3.0) and torch.all(grads["layer_1"]["bias"] == 30.0)
          and torch.equal(grads["layer_2"]["weight"], torch.arange(16.0).reshape(4, 4) * 3))
    lo, hi = ng.dist.shard_range(7, rank, world)
    out[rank] = (bool(ok), lo, hi)
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_gradients_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0][0] and out[1][0]
    assert (out[0][1], out[0][2], out[1][1], out[1][2]) == (0, 4, 4, 7)   # 7 trajectories -> 4 + 3


def test_single_process_is_a_no_op():
    g = {"w": torch.ones(3)}
    assert ng.dist.allreduce_gradients(g)["w"].tolist() == [1, 1, 1]
    assert ng.dist.shard_range(512, 3, 8) == (192, 256)                   # C4: 512 trajectories, 64 per GPU


class _SgdRule:
    """host stand-in for the fused kernel so the collective of optim.update can run under gloo on CPU"""

    def init(self, flat):
        return {}

    def apply(self, state, flat, grad_scale):
        flat.data -= 0.5 * grad_scale * flat.grad
        return state


def _optim_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps = {"layer_1": {"weight": torch.ones(2, 3), "bias": torch.zeros(2, 1)}}
    flat, psv = ng.optim.flatten_parameters(ps)
    (psv["layer_1"]["weight"] * float(rank + 1)).sum().backward()        # rank-dependent gradient: 1 resp. 2 per weight entry
    (psv["layer_1"]["bias"] * 4.0).sum().backward()
    st = {"rule": _SgdRule(), "state": {}}
    ng.optim.update(st, flat)                                             # all-reduce(sum) of the flat gradient, then mean
    out[rank] = flat.data.tolist()
    dist.barrier()
    dist.destroy_process_group()


def test_optim_update_allreduces_flat_gradient_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_optim_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    expect = [1 - 0.5 * 1.5] * 6 + [0 - 0.5 * 4.0] * 2                    # mean gradient 1.5 on the weights, 4 on the bias
    assert out[0] == expect and out[1] == expect


def test_flatten_parameters_views_alias_the_flat_vector():
    ps = {"a": {"weight": torch.arange(6.0).reshape(2, 3)}, "b": torch.tensor([7.0, 8.0])}
    flat, psv = ng.optim.flatten_parameters(ps)
    assert flat.data.tolist() == [0, 1, 2, 3, 4, 5, 7, 8] and [t[0] for t in flat.table] == ["a.weight", "b"]
    (psv["a"]["weight"].sum() * 2 + (psv["b"] * torch.tensor([1.0, 3.0])).sum()).backward()
    assert flat.grad.tolist() == [2] * 6 + [1, 3]
    flat.data[0] = 42.0
    assert float(psv["a"]["weight"].detach()[0, 0]) == 42.0


def _overlap_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ps = {"ϕ": {"layer_1": {"weight": torch.ones(2, 3), "bias": torch.zeros(2, 1)}}, "ψ": {"layer_1": {"weight": torch.ones(4, 2)}},
          "extra": torch.ones(2)}
    flat, psv = ng.optim.flatten_parameters(ps)
    red = ng.dist.OverlappedGradReduce(flat, psv, [("ψ.",), ("ϕ.",)])      # psi is final first in an MPPDE pullback
    launched = []
    orig = red._launch
    red._launch = lambda b: (launched.append(tuple(b["names"])), orig(b))[1]
    for step in range(2):                                                  # counters reset between steps
        flat.zero_grad()
        launched.clear()
        loss = (psv["ψ"]["layer_1"]["weight"] * float(rank + 1)).sum()
        loss = loss + (psv["ϕ"]["layer_1"]["weight"] * 10.0 * (rank + 1)).sum() + (psv["ϕ"]["layer_1"]["bias"] * 3.0).sum()
        loss.backward()                                                    # "extra" gets no gradient: reduced by finish()
        red.finish()
        st = {"rule": _SgdRule(), "state": {}}
        before = flat.data.clone()
        ng.optim.update(st, flat, reduced=True)                            # no second all-reduce
        out[(rank, step)] = (flat.grad.tolist(), sorted(launched), (before - flat.data).tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucketed_reduce_gloo_world2():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    # flat order: phi.weight (6), phi.bias (2), psi.weight (8), extra (2); sums over ranks 1 + 2
    expect = [30.0] * 6 + [6.0] * 2 + [3.0] * 8 + [0.0] * 2
    for key in [(0, 0), (1, 0), (0, 1), (1, 1)]:
        grad, launched, delta = out[key]
        assert grad == expect
        assert launched == sorted([("ψ.layer_1.weight",), ("ϕ.layer_1.weight", "ϕ.layer_1.bias"), ("extra",)])
        assert delta == [0.5 * 0.5 * g for g in expect]                    # the mean over the two ranks, applied once


def test_overlapped_reduce_without_process_group_is_inert():
    flat, psv = ng.optim.flatten_parameters({"a": torch.ones(3)})
    red = ng.dist.OverlappedGradReduce(flat, psv, [("a",)])
    (psv["a"] * 2).sum().backward()
    red.finish()
    assert flat.grad.tolist() == [2.0, 2.0, 2.0]


def test_bench_gpus_n_without_enough_devices_prints_a_skip_record():
    # `python3 bench.py --gpus 8` on a box with fewer devices (here: none): ONE JSON record that says so, exit code 0, and no GPU
    # call on the way ("self-skips (and says so)")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box has the 8 devices")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "NGPDE_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["skipped"] is True and d["n_gpus"] == 8 and d["value"] is None and d["steps"] == 5 and "8 devices" in d["reason"]
    assert d["devices_counted_by"] in ("kfd topology", "torch.cuda.device_count()")


def test_launcher_counts_gpus_from_the_kernel_drivers_topology_without_the_hip_runtime(tmp_path, monkeypatch):
    # bench.py's launcher parent decides "enough devices for N ranks?" from /sys/class/kfd (GPU agents have simd_count > 0, CPU
    # agents 0) narrowed by the runtime's visibility variables -- never from a HIP call
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k, simd in enumerate([0, 0, 1024, 1024, 1024]):                 # two CPU sockets, three GPUs
        (tmp_path / str(k)).mkdir()
        (tmp_path / str(k) / "properties").write_text(f"cpu_cores_count {96 if simd == 0 else 0}\nsimd_count {simd}\ngfx_target_version 90500\n")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.visible_gpu_count(str(tmp_path)) == (3, "kfd topology")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count(str(tmp_path)) == (2, "kfd topology")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count(str(tmp_path)) == (1, "kfd topology")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count(str(tmp_path))[0] == 0
    assert bench.visible_gpu_count(str(tmp_path / "absent")) == (None, "no /sys/class/kfd")
    # a negative index ends a visibility list ("-1": no device at all)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "-1")
    assert bench.visible_gpu_count(str(tmp_path))[0] == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,-1,2")
    assert bench.visible_gpu_count(str(tmp_path))[0] == 2
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # a container that exposes only some render nodes: sysfs still lists every GPU of the host, the inaccessible ones are not counted
    topo, dri = tmp_path / "topo2", tmp_path / "dri"
    topo.mkdir(); dri.mkdir()
    for k, minor in enumerate([0, 128, 129, 130]):                      # one CPU agent, three GPUs
        (topo / str(k)).mkdir()
        (topo / str(k) / "properties").write_text(f"simd_count {0 if minor == 0 else 1024}\ndrm_render_minor {minor}\n")
    (dri / "renderD128").write_text("")
    (dri / "renderD130").write_text("")
    assert bench.visible_gpu_count(str(topo), str(dri)) == (2, "kfd topology")
    assert bench.visible_gpu_count(str(topo), str(tmp_path / "no_dri")) == (3, "kfd topology")


def _world8_worker(rank, world, port, out):
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # shards of whole trajectories: BASELINE config 4's 512 over 8 ranks, and a count that does not divide
    D = ng.dist
    lo, hi = D.shard_range(512, rank, world)
    lo2, hi2 = D.shard_range(13, rank, world)
    grads = {"layer_1": {"weight": torch.full((4, 3), float(rank + 1)), "bias": torch.full((4,), float(hi2 - lo2))}}
    D.allreduce_gradients(grads)
    # the bench's timing rule: barrier, time the region, the MAX over the ranks is the job's time
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    out[rank] = (lo, hi, lo2, hi2, grads["layer_1"]["weight"][0, 0].item(), grads["layer_1"]["bias"][0].item(), float(tt.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_count_specific_paths_with_eight_gloo_ranks():
    # the paths that only exist at N = 8 (512 trajectories -> 64 per GPU): a rendezvous of eight, shard ranges with and
    # without a remainder, one all-reduce of the flat gradient, max-over-ranks timing.  CPU ranks: the GPU box admits at most six
    # processes on its card, so an eight-rank rehearsal cannot touch the GPU there; the driver's 8-GPU run is the measurement.
    world, port = 8, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_world8_worker, args=(world, port, out), nprocs=world, join=True)
    res = [out[r] for r in range(world)]
    assert [(r[0], r[1]) for r in res] == [(64 * k, 64 * (k + 1)) for k in range(8)]
    cover = [i for r in res for i in range(r[2], r[3])]
    assert cover == list(range(13)) and [r[3] - r[2] for r in res] == [2, 2, 2, 2, 2, 1, 1, 1]
    assert all(r[4] == 36.0 and r[5] == 13.0 for r in res)              # sum of (rank + 1); the shard sizes add up to the item count
    assert all(abs(r[6] - res[0][6]) < 1e-12 and r[6] >= 0.08 for r in res)   # every rank holds the slowest rank's time
