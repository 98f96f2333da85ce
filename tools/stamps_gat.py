"""Where does the one-launch GAT layer go?  Needs the diagnostic library (make -C neuralgraphpde.jl_amd/csrc diag).  Shader-clock
stamps of thread 0 of every workgroup of gat_layer_fwd_kernel at the C3 size; prints the mean cycles between consecutive points
and the spread of the workgroups' start and end times."""
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
ps, st = ng.setup(3, l)
ps = ng.to_device(ps, "cuda")
x = torch.randn(16384, 64, device="cuda").T
buf = torch.zeros(512 * 16, dtype=torch.int64, device="cuda")
with torch.no_grad():
    for rep in range(4):
        lib.ngpde_debug_set_gat_stamps(_lib.ptr(buf) if rep == 3 else None)
        l(x, ps, st)
        torch.cuda.synchronize()
lib.ngpde_debug_set_gat_stamps(None)
raw = buf.cpu().numpy().reshape(512, 16).astype(np.float64)
st_ = raw[:, :11]
w0, w1 = raw[:, 11], raw[:, 12]          # 100 MHz wall clock at the first and the last stamp
names = ["metadata + W loads + DMA issue + v vectors", "barrier (DMA data lands)", "score halves (ar of staged rows, al)", "barrier",
         "softmax + coefficient table", "per-head aggregation", "barrier", "MFMA + tile store", "barrier", "epilogue store"]
d = np.diff(st_, axis=1)
print("; ".join(f"{n}: {d[:, k].mean():.0f}" for k, n in enumerate(names)))
print(f"workgroup total: mean {(st_[:, 10] - st_[:, 0]).mean():.0f} cycles = {(w1 - w0).mean() * 0.01:.2f} us; first start -> last end "
      f"{(w1.max() - w0.min()) * 0.01:.2f} us; start spread {(w0.max() - w0.min()) * 0.01:.2f} us; end spread {(w1.max() - w1.min()) * 0.01:.2f} us")
order = np.argsort(w0)
print("start times (us, sorted, every 64th):", np.round((w0[order][::64] - w0.min()) * 0.01, 2).tolist())
