"""K independent trajectories of the BASELINE workload on ONE GPU, each with its own solver plan and HIP stream, in flight
together: how much of a chain's per-node launch floor and latency another chain's kernels can fill.
usage: python tools/concurrent_plans.py [K ...]"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan
lib = _lib.load()
N, PAIRS, D, STEPS = 16384, 65536, 64, 50
_, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
h = g.handle((True, None, False))
dev = "cuda"
w1 = torch.as_tensor(np.ascontiguousarray(S.glorot_uniform(11, D, D).T, np.float32), device=dev)
w2 = torch.as_tensor(np.ascontiguousarray(S.glorot_uniform(12, D, D).T, np.float32), device=dev)
b1 = torch.zeros(D, device=dev); b2 = torch.zeros(D, device=dev)
p = _lib.ptr
for K in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    plans = [_Plan(h, D, 1, "tsit5", STEPS, 1.0 / STEPS, True) for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    u0 = [torch.as_tensor(S.normal(1000 + k, D * N).reshape(N, D).astype(np.float32), device=dev) for k in range(K)]
    uT = [torch.empty_like(u0[0]) for _ in range(K)]; du0 = [torch.empty_like(u0[0]) for _ in range(K)]
    seed = torch.ones_like(u0[0])
    gr = [[torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)] for _ in range(K)]
    torch.cuda.synchronize()

    def step():
        for k in range(K):
            st = streams[k].cuda_stream
            _lib.check(lib.ngpde_node_gcn2_forward(plans[k].ptr, p(u0[k]), p(w1), p(b1), p(w2), p(b2), p(uT[k]), st))
            _lib.check(lib.ngpde_node_gcn2_backward(plans[k].ptr, p(seed), p(du0[k]), p(gr[k][0]), p(gr[k][1]), p(gr[k][2]), p(gr[k][3]), st))
    for _ in range(3): step()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(trajectories_in_flight=K, trajectory_ode_steps_per_s=round(K * STEPS * reps / dt, 1),
                          ms_per_round=round(1e3 * dt / reps, 3))), flush=True)
    del plans
