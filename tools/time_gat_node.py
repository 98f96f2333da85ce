"""C3 as right-hand side (NeuralODE(GATConv 64 => 4 x 16), C2's graph, Tsit5 x 50): the forward launch and the adjoint launch timed apart
(events around the solve and around backward()), median of REPS; BATCH=k times a batch of k members instead."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import synth as S
N, STEPS, REPS, K = 16384, 50, int(os.environ.get("REPS", 10)), int(os.environ.get("BATCH", 1))
_, s, t = S.closest_pairs_graph(N, 65536, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
if K > 1:
    g = ng.batch([g] + [g.copy() for _ in range(K - 1)])
l = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
node = ng.NeuralODE(l, solver="tsit5", n_steps=STEPS, dt=0.02)
ps, st = ng.setup(3, node)
ps = ng.to_device(ps, "cuda")
for v in ps.values():
    v.requires_grad_(True)
u = torch.as_tensor(S.normal(33, 64 * N * K).reshape(N * K, 64).astype(np.float32), device="cuda").T.requires_grad_(True)
tf, tb = [], []
for rep in range(REPS + 2):
    for v in [u] + list(ps.values()):
        v.grad = None
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    uT, _ = node(u, ps, st)
    e[1].record()
    uT.sum().backward()
    e[2].record()
    torch.cuda.synchronize()
    if rep >= 2:
        tf.append(e[0].elapsed_time(e[1])); tb.append(e[1].elapsed_time(e[2]))
plans = [p for pool in node._plans.values() for p in pool]
print(f"members {K}: forward {np.median(tf):.3f} ms, adjoint {np.median(tb):.3f} ms, solve + adjoint {np.median(tf) + np.median(tb):.3f} ms = "
      f"{K * STEPS / ((np.median(tf) + np.median(tb)) * 1e-3):.0f} (trajectory) ODE-steps/s; flags {sorted(plans[0].flags())}; fault {any(p.fault() for p in plans)}; "
      f"checksum u(T) {float(uT.double().sum()):.9e} du0 {float(u.grad.double().sum()):.9e}")
