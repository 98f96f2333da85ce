"""Kernel-trace subject for `rocprofv3 --kernel-trace --stats`: BASELINE config 3 as an ODE right-hand side (GATConv 64 => 4 x 16
on the C2 graph, Tsit5, forward + discrete adjoint) through NeuralODE's generic path.  Every kernel in the trace must be one of
the library's (ngpde::...): the Runge-Kutta combinations are ngpde_rk_stage_combine launches, not torch element-wise kernels.
Prints the solve's wall time per ODE step."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

N, steps = 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 10
CAPTURE = len(sys.argv) > 2 and sys.argv[2] == "capture"
_, s, t = S.closest_pairs_graph(N, 65536, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
l = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
node = ng.NeuralODE(l, solver="tsit5", n_steps=steps, dt=1.0 / steps, capture=CAPTURE)
ps, st = ng.setup(3, node)
ps = ng.to_device(ps, "cuda")
for v in ps.values():
    v.requires_grad_(True)
u0 = torch.as_tensor(S.normal(33, 64 * N).reshape(N, 64).astype(np.float32), device="cuda").T.requires_grad_(True)


def solve():
    for v in list(ps.values()) + [u0]:
        v.grad = None
    uT, _ = node(u0, ps, st)
    uT.backward(ONES)                         # the cotangent of sum(u(T))


ONES = torch.ones(N, 64, device="cuda").T
solve()
solve()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    solve()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"config": "C3 as ODE RHS: GATConv 64=>4x16 on the C2 graph, Tsit5 fixed step, fwd + discrete adjoint (generic NeuralODE path)",
                  "capture": CAPTURE, "ode_steps": steps, "ms_per_solve": round(dt * 1e3, 3), "ode_steps_per_s": round(steps / dt, 1)}))
