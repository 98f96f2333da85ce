"""Summarise the round-3 counter passes of tools/profile_round3.sh: per persistent launch (one-member kernel, two-member
interleaved kernel, tile-pair kernel) HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, KiB units, the gfx950 correction of
tools/pmc_summary.py), matrix-pipe busy share of the SIMD cycles and LDS bank-conflict share of the LDS-active cycles.

usage: python tools/pmc_batched_summary.py gpurun_out/r03_a > profiles/r03_a_batched_pmc.json"""
import collections, csv, glob, json, re, sys


def kind(name):
    m = re.search(r"node_(fwd|bwd)_persistent(2?)_kernel<(.*)>", name)
    if not m:
        return None
    args = [a.strip() for a in m.group(3).split(",")]
    if m.group(2) == "2":
        form = "tile_pairs" if args[-1] in ("true", "1") else "two_members"
    else:
        form = "one_member"
    return f"{m.group(1)}_{form}"


def per_kernel(d):
    f = glob.glob(d + "/*/*_counter_collection.csv")
    if not f:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k = kind(r["Kernel_Name"])
        if k:
            per_dispatch[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per_dispatch.items():
        for c, v in cs.items():
            acc[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches": len(next(iter(cs.values())))} for k, cs in acc.items()}


def main():
    o = sys.argv[1]
    fetch, write = per_kernel(o + "/batched_pmc_FETCH_SIZE"), per_kernel(o + "/batched_pmc_WRITE_SIZE")
    mfma = per_kernel(o + "/batched_SQ_VALU_MFMA_BUSY_CYCLES_SQ_BUSY_CYCLES")
    lds = per_kernel(o + "/batched_SQ_LDS_BANK_CONFLICT_SQ_LDS_IDX_ACTIVE")
    out = {}
    for k in sorted(set(fetch) | set(mfma) | set(lds)):
        e = {}
        if k in fetch and k in write:
            rd, wr = fetch[k]["FETCH_SIZE"] * 2048.0, write[k]["WRITE_SIZE"] * 1024.0
            e.update(hbm_read_bytes=round(rd), hbm_write_bytes=round(wr), hbm_bytes=round(rd + wr), launches_seen=fetch[k]["launches"])
        if k in mfma and mfma[k].get("SQ_BUSY_CYCLES"):
            # SQ_BUSY_CYCLES counts per shader engine, MFMA_BUSY per SIMD-cycle /4 (tools/pmc_mfma_summary.py has the same scaling)
            e["mfma_busy_cycles"] = round(mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"]); e["sq_busy_cycles"] = round(mfma[k]["SQ_BUSY_CYCLES"])
        if k in lds and lds[k].get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_share"] = round(lds[k]["SQ_LDS_BANK_CONFLICT"] / lds[k]["SQ_LDS_IDX_ACTIVE"], 3)
        out[k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
