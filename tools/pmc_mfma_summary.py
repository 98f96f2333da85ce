"""Per-kernel averages of the counters collected by tools/pmc_mfma.sh: MFMA-pipe busy fraction, LDS bank-conflict share."""
import collections, csv, glob, json, sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(root + "/*_S*/*/*_counter_collection.csv"):      # <config>_<counter group>/<pid>/..., config = c2 / c4 / c5 / vmh / vmhb
    cfg = path.split("/")[-3].split("_")[0]
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "ngpde" not in name:
            continue
        short = name.replace("ngpde::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        acc[(cfg, short)][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for (cfg, k), cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    row = {"dispatches": max(len(v) for v in cs.values())}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("SQ_BUSY_CYCLES"):
        row["mfma_busy_cycles"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"])
        row["sq_busy_cycles"] = round(m["SQ_BUSY_CYCLES"])
        row["mfma_busy_over_sq_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"], 4)
        # SQ_BUSY_CYCLES sums over the 32 shader engines (= 32 x the kernel's duration in cycles), the MFMA counter over the
        # 1024 SIMDs: share of the SIMD-cycles in which the matrix pipe was busy = mfma / (1024 * sq / 32)
        row["matrix_pipe_busy_fraction"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * m["SQ_BUSY_CYCLES"]), 4)
    if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
        row["lds_bank_conflict_share"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    for c in ("SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_WAVE_CYCLES"):
        if c in m:
            row[c] = round(m[c])
    out[f"{cfg}:{k}"] = row
print(json.dumps(out, indent=1))
