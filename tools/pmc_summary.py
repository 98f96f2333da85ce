"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit+miss, each collected in its own
run) into per-kernel-role HBM bytes per launch.  gfx950 correction (MI355X_MICROARCH.md, HBM):
FETCH_SIZE reports half the bytes of a wide (16 B/lane) coalesced read stream -> doubled here;
WRITE_SIZE is exact for 16 B/lane streaming stores.  Units of both counters: KiB.

usage: python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_l2 out.json
"""
import collections, csv, glob, json, sys


def role_rows(path):
    rows = list(csv.DictReader(open(glob.glob(path + "/*/*_counter_collection.csv")[0])))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = collections.defaultdict(dict)      # kernel -> dispatch id -> ordinal (several counters share a dispatch)
    for r in rows:
        n = r["Kernel_Name"]
        if "node_fwd_persistent_kernel" in n:          # the persistent plan: one launch per direction
            k, names = "pfwd", ("fwd_persistent_solve", "fwd_persistent_solve")
        elif "node_bwd_persistent_kernel" in n:
            k, names = "pbwd", ("bwd_persistent_adjoint", "bwd_persistent_adjoint")
        elif "gcn_fused_fwd_kernel" in n:
            k, names = "fwd", ("fwd_layer1", "fwd_layer2_stage")
        elif "gcn_fused_bwd_kernel" in n and ", true," in n:
            k, names = "bwd", ("bwd_layer1", "bwd_stage_layer2")
        else:
            continue
        d = seen[k]
        if r["Dispatch_Id"] not in d:
            d[r["Dispatch_Id"]] = len(d)
        out[names[d[r["Dispatch_Id"]] % 2]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {role: {c: sum(v) / len(v) for c, v in cs.items()} for role, cs in out.items()}


def main():
    fetch, write, l2, dst = sys.argv[1:5]
    f, w, h = role_rows(fetch), role_rows(write), role_rows(l2)
    summary = {}
    for role in f:
        fb = f[role]["FETCH_SIZE"] * 1024 * 2.0
        wb = w[role]["WRITE_SIZE"] * 1024
        hit, miss = h[role]["TCC_HIT_sum"], h[role]["TCC_MISS_sum"]
        summary[role] = {"hbm_read_bytes": round(fb), "hbm_write_bytes": round(wb), "hbm_bytes": round(fb + wb),
                         "fetch_size_raw_KiB": round(f[role]["FETCH_SIZE"], 1), "l2_hit_rate": round(hit / (hit + miss), 3)}
    json.dump(summary, open(dst, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
