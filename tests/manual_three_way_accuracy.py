"""One-off accuracy check of the full bench workload (C2, 50 Tsit5 steps, forward + adjoint): the HIP path and the float32 C port,
each against the float64 numpy oracle (the oracle takes minutes at this size, so this is a tool, not a test)."""
import os, sys, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ngpde_amd as ng
from oracle import ngpde_oracle as O
import test_configs_gpu as T
s, t, D, params, u0 = T.c2_inputs()
t0 = time.time()
uTo, du0o, acc = O.gcn2_node_loss_and_grads(params, O.Graph(s, t, num_nodes=16384, index_base=0), u0, O.TABLEAUS["tsit5"], 1.0 / 50, 50, "relu")
print("numpy f64 oracle", time.time() - t0, "s", flush=True)
lib = C.CDLL("oracle/libngpde_oracle_omp.so")
C.CDLL("libgomp.so.1").omp_set_num_threads(16)
vp = C.c_void_p
lib.ngo_node_gcn2.argtypes = [C.c_int64, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int] + [vp] * 11
f32 = lambda a: np.ascontiguousarray(a, np.float32)
u = f32(u0.T); w = [f32(params[k]["weight"].T) for k in range(2)]; b = [f32(params[k]["bias"].reshape(-1)) for k in range(2)]
outs = [np.zeros_like(u), np.zeros_like(u), np.zeros_like(w[0]), np.zeros_like(b[0]), np.zeros_like(w[1]), np.zeros_like(b[1])]
s64, t64 = np.ascontiguousarray(s, np.int64), np.ascontiguousarray(t, np.int64)
P = lambda a: a.ctypes.data
lib.ngo_node_gcn2(16384, s64.size, P(s64), P(t64), D, 1, 1, 50, 1.0 / 50, 1, P(u), P(w[0]), P(b[0]), P(w[1]), P(b[1]), *[P(o) for o in outs])
g = ng.GNNGraph(s, t, num_nodes=16384, index_base=0)
node, ps, st = T.gcn2_node(g, D, "tsit5", 50, 1.0 / 50, params)
ut = torch.as_tensor(u0.astype(np.float32), device="cuda:0").requires_grad_(True)
uT, _ = node(ut, ps, st); uT.sum().backward()
def rel(a, ref): return float(np.abs(np.asarray(a, np.float64) - ref).max() / np.abs(ref).max())
G = lambda x: x.detach().cpu().double().numpy()
rows = [("uT", G(uT), outs[0].T, uTo), ("du0", G(ut.grad), outs[1].T, du0o),
        ("dW1", G(ps["layer_1"]["weight"].grad), outs[2].T, acc[0]["weight"]), ("dW2", G(ps["layer_2"]["weight"].grad), outs[4].T, acc[1]["weight"]),
        ("db1", G(ps["layer_1"]["bias"].grad).reshape(-1), outs[3], acc[0]["bias"].reshape(-1)), ("db2", G(ps["layer_2"]["bias"].grad).reshape(-1), outs[5], acc[1]["bias"].reshape(-1))]
for name, gpu, cport, ref in rows:
    print(f"{name:4s} rel err vs f64 oracle: HIP {rel(gpu, ref):.2e}   C port {rel(cport, ref):.2e}   (HIP vs C port {rel(gpu, np.asarray(cport, np.float64)):.2e})")
