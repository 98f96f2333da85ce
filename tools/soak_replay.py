"""Race screen for the solver plan's kernels (LDS-DMA staging, paired workgroups, the flag hand-off of the persistent launches):
replays the 50-step C2 solve + adjoint many times and requires every output to be bit-identical to the first run.
usage: [N=16384] [MEMBERS=1] [WIDTH=64] [WEIGHTED=0] python tools/soak_replay.py [replays]     (MEMBERS > 1: the interleaved batch
kernels; N > 16384: the tile-pair kernels, N > 32768: tile rounds; WIDTH=16/32: the widened plan; WEIGHTED=1: a graph with edge
weights, the tile-round kernels with the slot weights in LDS; GRAPH=cora: a preferential-attachment graph with hubs of N nodes and
2 N pairs -- the hub geometry for N <= 8192)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan
lib = _lib.load()
N, D, MEMBERS = int(os.environ.get("N", 16384)), int(os.environ.get("WIDTH", 64)), int(os.environ.get("MEMBERS", 1))
WEIGHTED = os.environ.get("WEIGHTED", "0") == "1"
PAIRS = 4 * N
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if os.environ.get("GRAPH", "") == "cora":
    s, t = S.preferential_pairs_graph(N, 2 * N, seed=1)
else:
    _, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
ew = (0.25 + np.random.default_rng(1).random(s.size)).astype(np.float32) if WEIGHTED else None
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, edge_weight=ew)
plan = _Plan(g.handle((True, g.edge_weight, False)), D, 1, "tsit5", 50, 0.02, True, members=MEMBERS)
print("plan flags:", plan.flags())
dev = "cuda"
torch.manual_seed(0)
u0 = torch.randn(N * MEMBERS, D, device=dev); w1 = torch.randn(D, D, device=dev) * 0.1; w2 = torch.randn(D, D, device=dev) * 0.1
b1 = torch.randn(D, device=dev) * 0.1; b2 = torch.randn(D, device=dev) * 0.1
seed = torch.randn(N * MEMBERS, D, device=dev)
st = torch.cuda.current_stream().cuda_stream; p = _lib.ptr
first, bad = None, 0
for r in range(reps):
    outs = [torch.empty_like(u0), torch.empty_like(u0), torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w1), torch.empty_like(b1)]
    _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(outs[0]), st))
    _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(outs[5]), st))
    torch.cuda.synchronize()
    if first is None:
        first = [o.clone() for o in outs]
    elif not all(torch.equal(a, b) for a, b in zip(first, outs)):
        bad += 1
print(f"N={N} members={MEMBERS} width={D} weighted={WEIGHTED}: {reps} replays, {bad} replays differing from the first; fault={plan.fault()}")
sys.exit(1 if bad else 0)
