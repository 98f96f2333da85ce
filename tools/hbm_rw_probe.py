"""What the box's HBM does for plain device-wide fills / copies / reads at the C4 array sizes (torch kernels, events): the ceiling
the streaming Dense kernels are compared with."""
import json, torch
dev = "cuda"
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (134, 268, 805):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev); c = torch.empty(n, device=dev)
    a.normal_()
    tf = t(lambda: b.fill_(1.0)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum()); t2 = t(lambda: torch.add(a, 1.0, out=b)); 
    print(json.dumps({"MB": mb, "fill_TBs": round(mb / 1e6 / tf, 2), "copy_TBs_rw": round(2 * mb / 1e6 / tc, 2), "sum_read_TBs": round(mb / 1e6 / tr, 2),
                      "add_TBs_rw": round(2 * mb / 1e6 / t2, 2)}), flush=True)
