"""Where does a phase of the device-resident GAT solver go?  Needs the diagnostic library (make -C neuralgraphpde.jl_amd/csrc diag).
Shader-clock stamps of thread 0 of every workgroup in the LAST phase of the forward and of the adjoint launch, C3 size."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
_lib.LIB_PATH = os.path.join(ROOT, "neuralgraphpde.jl_amd", "libngpde_diag.so")
lib = _lib.load()
lib.ngpde_debug_set_gat_stamps.argtypes = [C.c_void_p]
_, s, t = S.closest_pairs_graph(16384, 65536, seed=2)
g = ng.GNNGraph(s, t, num_nodes=16384, index_base=0)
l = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
STEPS = int(os.environ.get("STEPS", 10))
node = ng.NeuralODE(l, solver="tsit5", n_steps=STEPS, dt=0.02)
ps, st = ng.setup(3, node)
ps = ng.to_device(ps, "cuda")
for v in ps.values(): v.requires_grad_(True)
u = torch.randn(16384, 64, device="cuda").T.requires_grad_(True)
buf = torch.zeros(512 * 16, dtype=torch.int64, device="cuda")
def mean_diffs(raw, idx):
    return [float((raw[:, b] - raw[:, a]).mean()) for a, b in zip(idx[:-1], idx[1:])]
for rep in range(3):
    uT, _ = node(u, ps, st)
    torch.cuda.synchronize()
    if rep == 2:
        # (the forward launch has run with the stamps of the previous repetition's setting)
        pass
    lib.ngpde_debug_set_gat_stamps(_lib.ptr(buf) if rep >= 1 else None)
    if rep == 2:
        fw = buf.cpu().numpy().reshape(512, 16).astype(np.float64).copy()
    uT.sum().backward()
    torch.cuda.synchronize()
    if rep == 2:
        bw = buf.cpu().numpy().reshape(512, 16).astype(np.float64).copy()
lib.ngpde_debug_set_gat_stamps(None)
# forward: 0 phase start, 13 after the wait, 1..5, 9 inside the layer code (gat_fwd_compute), 14 stores issued, 15 published
order = [0, 13, 1, 2, 3, 4, 5, 9, 14, 15]
names = ["wait for the neighbours", "DMA issue, a_l / a_r loads", "barrier (rows land)", "W x of the staged rows (MFMA)", "barrier",
         "score halves, barrier, softmax + coefficients", "aggregation", "activation, tape, combination, row store", "drain + barrier + flag"]
print("forward, last phase (cycles): " + "; ".join(f"{n}: {d:.0f}" for n, d in zip(names, mean_diffs(fw, order))) + f" | phase {float((fw[:, 15] - fw[:, 0]).mean()):.0f}")
order = [0, 1, 7, 8, 9, 2, 3, 4, 5, 10, 11, 12, 13, 14, 6]
names = ["by-target metadata, DMA issue, K-bar, dz store", "T: W / alpha loads, dz tile, barrier (rows land)", "T: db partial, MFMA, barrier", "T: d alpha loop",
         "T: softmax pullback, dscore / dal stores", "drain + barrier + flag", "by-source metadata + W copy", "wait for the neighbours",
         "S: DMA issue, alpha / dscore / x loads", "S: barrier (rows land)", "S: aggregation, tiles", "S: barrier", "S: dx and dW products", "S: u accumulators, barrier, dx read, barrier"]
print("adjoint, last phase (cycles): " + "; ".join(f"{n}: {d:.0f}" for n, d in zip(names, mean_diffs(bw, order))) + f" | T + S {float((bw[:, 6] - bw[:, 0]).mean()):.0f}")
