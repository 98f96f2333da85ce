"""The device-resident GAT solver (ngpde_node_gat_*) against the generic solver over the one-launch layer (NGPDE_NO_PERSISTENT=1),
output by output: u(T) and du0 must be bitwise equal, parameter gradients agree to rounding.  Then timings.
env: N (nodes, default 16384), PAIRS, STEPS, SOLVER, ACT, HEADS"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S, _lib
if os.environ.get('NGPDE_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['NGPDE_LIB'])       # A/B runs of another build

N = int(os.environ.get("N", 16384)); PAIRS = int(os.environ.get("PAIRS", 4 * N)); STEPS = int(os.environ.get("STEPS", 50))
SOLVER = os.environ.get("SOLVER", "tsit5"); ACT = os.environ.get("ACT", "relu"); H = int(os.environ.get("HEADS", 4))
K = int(os.environ.get("MEMBERS", 1))     # > 1: a block-diagonal batch of K copies of the graph (two members per workgroup)
DEV = "cuda"
_, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g1 = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
g = ng.batch([g1] * K) if K > 1 else g1
layer = ng.GATConv((64, 64 // H), ACT, heads=H, initialgraph=g)
N1, N = N, N * K


def run(persistent, u0, ps0, capture=False):
    if persistent:
        os.environ.pop("NGPDE_NO_PERSISTENT", None)
    else:
        os.environ["NGPDE_NO_PERSISTENT"] = "1"
    node = ng.NeuralODE(layer, solver=SOLVER, n_steps=STEPS, dt=1.0 / STEPS, capture=capture)
    _, st = ng.setup(3, node)
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in ps0.items()}
    u = u0.detach().clone().requires_grad_(True)
    uT, _ = node(u.T, ps, st)
    uT = uT.T
    (uT * R).sum().backward()
    torch.cuda.synchronize()
    plans = [p for pool in node._plans.values() for p in pool]
    return uT.detach(), u.grad, {k: v.grad for k, v in ps.items()}, plans, node, ps, st


ps0, _ = ng.setup(3, layer)
ps0 = ng.to_device(ps0, DEV)
with torch.no_grad():
    for k, v in ps0.items():
        if k == "bias":
            v.copy_(torch.randn_like(v) * 0.1)
u0 = torch.randn(N, 64, device=DEV)
R = torch.randn(N, 64, device=DEV)
a = run(True, u0, ps0)
print("plans (persistent run):", [p.flags() for p in a[3]], "fault:", [p.fault() for p in a[3]], flush=True)
b = run(False, u0, ps0)
print("plans (generic run):", [p.flags() for p in b[3]])
def cmp(name, x, y):
    d = (x - y).abs().max().item(); m = y.abs().max().item()
    print(f"{name}: max|generic|={m:.4e} max diff {d:.3e} rel {d / max(m, 1e-30):.2e} nan={bool(torch.isnan(x).any())} bitwise_equal={bool(torch.equal(x, y))}", flush=True)
cmp("u(T)", a[0], b[0]); cmp("du0", a[1], b[1])
for k in a[2]:
    cmp("d" + k, a[2][k], b[2][k])

def timeit(persistent, capture, reps=5):
    if persistent: os.environ.pop("NGPDE_NO_PERSISTENT", None)
    else: os.environ["NGPDE_NO_PERSISTENT"] = "1"
    node = ng.NeuralODE(layer, solver=SOLVER, n_steps=STEPS, dt=1.0 / STEPS, capture=capture)
    _, st = ng.setup(3, node)
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in ps0.items()}
    u = u0.detach().clone().requires_grad_(True)
    def step():
        for v in list(ps.values()) + [u]: v.grad = None
        uT, _ = node(u.T, ps, st)
        uT.sum().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
if os.environ.get("TIME", "1") == "1":
    tp = timeit(True, False); tg = timeit(False, True)
    print(f"ms per solve + adjoint ({K} member(s)): persistent {tp:.3f} ({K * STEPS / tp * 1e3:.0f} trajectory ODE-steps/s), generic captured {tg:.3f} ({K * STEPS / tg * 1e3:.0f})")
    if K > 1:     # member by member on single plans
        g, layer, N = g1, ng.GATConv((64, 64 // H), ACT, heads=H, initialgraph=g1), N1
        u0 = u0[:N1].contiguous()
        t1 = timeit(True, False)
        print(f"one member on its own plan: {t1:.3f} ms -> {K} members one after the other {K * t1:.3f} ms ({STEPS / t1 * 1e3:.0f} trajectory ODE-steps/s)")
