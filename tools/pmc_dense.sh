#!/bin/bash
# SQ wave-state counters of the Dense kernels on the 524 288 x 64 => 64 shape (tools/bench_dense.py): where do the waves spend
# their cycles -- parked (s_waitcnt / barrier), stalled at issue, or issuing?  usage (through gpurun): bash tools/pmc_dense.sh TAG
TAG=${1:-dense}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/sq -- python3 $R/tools/bench_dense.py > $O/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/sq2 -- python3 $R/tools/bench_dense.py > $O/sq2.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections, json
out = {}
for d in ("sq", "sq2"):
    fs = glob.glob("$O/" + d + "/*/*_counter_collection.csv")
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if "dense" not in n: continue
        short = n.replace("ngpde::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        out.setdefault(k, {}).update({c: round(sum(v) / len(v)) for c, v in cs.items()})
        out[k]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))
PY
