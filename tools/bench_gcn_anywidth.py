"""Diagnostic: the any-width GCNConv path (Dense on the fp32-MFMA kernels + generic aggregation) at Cora size, forward and
pullback, through the C ABI and replayed from a HIP graph (device-side time) -- GCNConv(1433 => 16), the tutorial's input layer
(docs/src/tutorials/graph_node.md:83), its mirror image and two small shapes.  One JSON line per shape with the compulsory
bytes of the layer (x, W, y once) and the fraction of the 8 TB/s HBM roofline that the forward reaches."""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S

N, PAIRS = 2708, 5278
s, t = S.preferential_pairs_graph(N, PAIRS, seed=1)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
h = g.handle((True, None, False))
lib = _lib.load()
p = _lib.ptr


def replay_us(fn, reps=200):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


for din, dout in ((1433, 16), (16, 1433), (48, 16), (16, 64)):
    x = (torch.rand(N, din, device="cuda") < 0.05).float() if din > 1000 else torch.randn(N, din, device="cuda")
    w = torch.randn(din, dout, device="cuda") * 0.1
    b = torch.zeros(dout, device="cuda")
    y, z = torch.empty(N, dout, device="cuda"), torch.empty(N, dout, device="cuda")
    agg = torch.empty(N, din, device="cuda")
    wsf = torch.empty(lib.ngpde_gcn_workspace_bytes(h.ptr, din, dout, 0), dtype=torch.uint8, device="cuda")
    wsb = torch.empty(lib.ngpde_gcn_workspace_bytes(h.ptr, din, dout, 1), dtype=torch.uint8, device="cuda")
    dy, dx = torch.randn(N, dout, device="cuda"), torch.empty(N, din, device="cuda")
    dw, db = torch.empty_like(w), torch.empty_like(b)

    def fwd():
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.ngpde_gcn_forward(h.ptr, din, dout, 1, p(x), p(w), p(b), p(y), p(agg), p(z), p(wsf), wsf.numel(), st))

    def bwd():
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.ngpde_gcn_backward(h.ptr, din, dout, 1, p(x), p(w), p(z), p(agg), p(dy), p(dx), p(dw), p(db), p(wsb),
                                          wsb.numel(), st))
    uf, ub = replay_us(fwd), replay_us(bwd)
    bytes_fwd = 4.0 * (N * din + din * dout + N * dout) + 8.0 * s.size
    print(json.dumps({"shape": f"GCNConv({din} => {dout}), {N} nodes / {s.size} edges", "us_forward": round(uf, 2),
                      "us_backward": round(ub, 2), "compulsory_MB_forward": round(bytes_fwd / 1e6, 3),
                      "forward_GBs": round(bytes_fwd / uf / 1e3, 1), "forward_frac_of_8TBs": round(bytes_fwd / uf / 1e3 / 8000, 4)}),
          flush=True)
