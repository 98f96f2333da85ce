"""Where the host time of an eager layer call goes: cProfile over REPS forward + backward calls of the C3 GAT layer (64 => 4 x 16 on the
C2 graph), sorted by cumulative time.  env: REPS (300), TOP (35)"""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

dev = "cuda:0"
N, PAIRS = 16384, 65536
_, s, t = S.closest_pairs_graph(N, PAIRS, seed=7)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
layer = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
ps, st = ng.setup(3, layer)
ps = ng.to_device(ps, dev)
for v in ps.values():
    v.requires_grad_(True)
x = torch.randn(64, N, device=dev, requires_grad=True)
reps = int(os.environ.get("REPS", 300))


def step():
    y, _ = layer(x, ps, st)
    y.sum().backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    step()
torch.cuda.synchronize()
print(f"{1e6 * (time.perf_counter() - t0) / reps:.1f} us per forward + backward call (wall, eager)")
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    step()
pr.disable()
torch.cuda.synchronize()
st_ = pstats.Stats(pr)
st_.sort_stats("cumulative").print_stats(int(os.environ.get("TOP", 35)))
