"""Diagnostic (libngpde_diag.so): phase timestamps of dense_pair_fwd_kernel (P and Q of C4 from one pass over h), one steady-state
tile per persistent workgroup."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib
sys.path.insert(0, os.path.join(ROOT, "tests"))
import composed as F      # the primitives' autograd wrappers (tests/composed.py)
_lib.LIB_PATH = os.path.join(ROOT, "neuralgraphpde.jl_amd", "libngpde_diag.so")
lib = _lib.load()
lib.ngpde_debug_set_pair_stamps.argtypes = [C.c_void_p]; lib.ngpde_debug_set_pair_stamps.restype = C.c_int32
DEV = "cuda:0"
n = 524288
h = torch.randn(n, 64, device=DEV); d = torch.randn(n, 2, device=DEV); th = torch.randn(64, 2, device=DEV)
wp = torch.randn(68, 64, device=DEV) * 0.1; wq = torch.randn(66, 64, device=DEV) * 0.1; bp = torch.randn(64, device=DEV)
nb = 512
stamps = torch.zeros(nb * 16, dtype=torch.int64, device=DEV)
with torch.no_grad():
    for _ in range(3): F.dense_pair([h, d, th], wp, bp, 0, [h, d], wq, None, 0, row_divs_a=[1, 1, n // 64], n=n)
    torch.cuda.synchronize()
    lib.ngpde_debug_set_pair_stamps(stamps.data_ptr())
    F.dense_pair([h, d, th], wp, bp, 0, [h, d], wq, None, 0, row_divs_a=[1, 1, n // 64], n=n); torch.cuda.synchronize()
a = stamps.cpu().numpy().reshape(nb, 16)
a = a[a[:, 8] > 0]
names = ["top barrier", "narrow loads + DMA issue + products", "collect next image (vmcnt)", "xn write + barrier", "stage a + barrier",
         "epilogue a", "barrier + stage b + barrier", "epilogue b"]
dd = np.diff(a[:, :9], axis=1)
print(f"{a.shape[0]} workgroups; phase cycles (s_memtime) median / p90")
for k, nm in enumerate(names):
    print(f"  {nm:40s} {np.median(dd[:, k]):8.0f} {np.percentile(dd[:, k], 90):8.0f}")
print(f"  {'tile total':40s} {np.median(a[:, 8] - a[:, 0]):8.0f}")
