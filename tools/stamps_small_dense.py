"""Where does the one-launch small Dense pullback go?  Needs the diagnostic library (make -C neuralgraphpde.jl_amd/csrc diag).  100 MHz wall-clock
stamps of thread 0 of every workgroup of dense_small_bwd_kernel: mean time between the points, and the spread of the workgroups' starts / ends."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "neuralgraphpde.jl_amd", "libngpde_diag.so")
lib = _lib.load()
from ngpde_amd import functional as F
lib.ngpde_debug_set_small_dense_stamps.argtypes = [C.c_void_p]
names = ["W + tile loads, dz", "barrier, LDS writes, barrier", "dX product + stores", "dW product", "slab"]
for n, widths, dout, act in [(3000, [60], 60, "tanh"), (18000, [60], 60, "tanh"), (3000, [2], 60, "identity")]:
    blocks = [torch.randn(n, w, device="cuda", requires_grad=True) for w in widths]
    wt = torch.randn(sum(widths), dout, device="cuda", requires_grad=True)
    b = torch.randn(dout, device="cuda", requires_grad=True)
    R = torch.randn(n, dout, device="cuda")
    nb = min((n + 63) // 64, 1024, n // (sum(widths) + 1))
    buf = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
    for rep in range(4):
        y = F.dense(blocks, wt, b, _lib.ACT[act])
        torch.cuda.synchronize()
        lib.ngpde_debug_set_small_dense_stamps(_lib.ptr(buf) if rep == 3 else None)
        y.backward(R)
        torch.cuda.synchronize()
    lib.ngpde_debug_set_small_dense_stamps(None)
    st = buf.cpu().numpy().reshape(nb, 8).astype(np.float64)[:, :6] * 0.01       # us
    d = np.diff(st, axis=1)
    print(f"n={n} {widths}=>{dout} {act}, {nb} workgroups: " + "; ".join(f"{nm}: {d[:, k].mean():.2f}" for k, nm in enumerate(names)) +
          f" | in-kernel mean {(st[:, 5] - st[:, 0]).mean():.2f} us; first start -> last end {st[:, 5].max() - st[:, 0].min():.2f} us; "
          f"start spread {st[:, 0].max() - st[:, 0].min():.2f} us")
