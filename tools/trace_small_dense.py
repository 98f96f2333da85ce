"""Device time of the Dense pullback at the VMH tutorial's shapes (tools/trace_vmh_node.py): run under
`rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/trace_small_dense.py`, then
`python tools/trace_small_dense.py DIR/t_kernel_trace.csv` prints the average duration of each Dense kernel per shape (call order; SHAPES below).
NGPDE_DENSE_NO_SMALL_BWD=1: the composed path (dz + weight pullback + reduction + input pullback)."""
import csv, os, sys
SHAPES = [(18000, [60], 60, "tanh"), (18000, [60], 40, "identity"), (3000, [60], 60, "tanh"), (3000, [1, 40], 60, "tanh"), (3000, [60], 1, "identity"),
          (3000, [2], 60, "identity")]
REPS = 20
if len(sys.argv) > 1:
    rows = [r for r in csv.DictReader(open(sys.argv[1]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = ("dense_small_bwd_kernel", "dense_dz_kernel", "dense_mfma_bwd_input_kernel", "dense_mfma_bwd_weight_kernel", "dense_weight_reduce_kernel",
             "dense_mfma_fwd_kernel", "dense_small_fwd_kernel")
    for nm in names:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if nm in r["Kernel_Name"]]
        if not d: continue
        per = REPS      # (a kernel that serves only some of the shapes shows fewer groups: they are in SHAPES' order)
        print(nm, [round(sum(d[k * per + 3:(k + 1) * per]) / max(per - 3, 1), 2) for k in range(len(d) // per)], "us per group of", REPS, "calls")
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import functional as F, _lib
for n, widths, dout, act in SHAPES:
    blocks = [torch.randn(n, w, device="cuda", requires_grad=True) for w in widths]
    wt = torch.randn(sum(widths), dout, device="cuda", requires_grad=True)
    b = torch.randn(dout, device="cuda", requires_grad=True)
    R = torch.randn(n, dout, device="cuda")
    for _ in range(REPS):
        F.dense(blocks, wt, b, _lib.ACT[act]).backward(R)
    torch.cuda.synchronize()
print("done")
