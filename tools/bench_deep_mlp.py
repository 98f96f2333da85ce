"""VMHConv with the tutorial's four-layer message MLP (VMH.md:75-83), forward + backward of ONE layer call at several graph sizes:
the deep fused pullback (edge_mlp_deep_bwd.hip) against the primitives' pullback (NGPDE_NO_FUSED_EDGE_BWD=1)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import synth as S

dev = "cuda:0"


def leaves(t):
    for v in t.values():
        if isinstance(v, dict):
            yield from leaves(v)
        else:
            yield v


for nv in [int(a) for a in sys.argv[1:]] or [3000, 16384, 65536, 262144]:
    pts = torch.as_tensor(S.uniform01(41, 2 * nv).reshape(2, nv).astype(np.float32), device=dev)
    gv = ng.GNNGraph(ng.knn_graph(pts, 6), ndata={"x": pts})
    phi = ng.Chain(ng.Dense(4, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 40))
    gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 1))
    l = ng.VMHConv(phi, gam, initialgraph=gv)
    ps, st = ng.setup(4, l)
    ps = ng.to_device(ps, dev)
    for v in leaves(ps):
        v.requires_grad_(True)
    u = torch.randn(1, nv, device=dev, requires_grad=True)
    R = torch.randn(1, nv, device=dev)
    res = {}
    for mode in ("fused", "primitives"):
        if mode == "primitives":
            os.environ["NGPDE_NO_FUSED_EDGE_BWD"] = "1"
        else:
            os.environ.pop("NGPDE_NO_FUSED_EDGE_BWD", None)

        def fb():
            for v in [u] + list(leaves(ps)):
                v.grad = None
            l(u, ps, st)[0].backward(R)
        for _ in range(3):
            fb()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fb()
        torch.cuda.synchronize()
        res[mode] = 1e2 * (time.perf_counter() - t0)
    print(f"nodes {nv:7d} edges {gv.num_edges:8d}: fused deep pullback {res['fused']:.3f} ms, primitives {res['primitives']:.3f} ms (forward + backward, eager)", flush=True)
