#!/bin/bash
# The C5 layer (GNOConv 128 => 128 on the 64 x 64 grid) under rocprofv3: kernel stats, then the counter groups in separate runs
# (matrix pipe busy, LDS conflicts, wave cycles / waits).  usage (through gpurun): bash tools/c5_quick.sh TAG [radius] [extra env assignments]
TAG=${1:-c5q}
RAD=${2:-0.1}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bench_layers.py --only c5 --width 128 --radius $RAD --reps 20 > $O/layers.jsonl 2> $O/layers.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python3 $R/tools/bench_layers.py --only c5 --width 128 --radius $RAD --reps 10 > $O/stats.log 2>&1
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-60)
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/c5_$n -- python3 $R/tools/bench_layers.py --only c5 --width 128 --radius $RAD --reps 2 > $O/c5_$n.log 2>&1
done
cd $R
python3 tools/kstats.py $O/stats 14 > $O/kernel_stats_top.txt 2>&1
python3 - $O <<'PY' > $O/pmc_summary.json
import collections, csv, glob, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/c5_S*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "ngpde" not in name:
            continue
        short = name.replace("ngpde::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    row = {c: round(v) for c, v in m.items()}
    if m.get("SQ_BUSY_CYCLES"):
        row["matrix_pipe_busy_fraction"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (32.0 * m["SQ_BUSY_CYCLES"]), 4)
    if m.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_share"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
    if m.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in m:
                row[c + "_share"] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
    out[k] = row
print(json.dumps(out, indent=1))
PY
cat $O/layers.jsonl $O/kernel_stats_top.txt
python3 -c "
import json,sys
d=json.load(open('$O/pmc_summary.json'))
for k,v in d.items():
    if 'gform' in k or 'gemm128_split_kernel<false, false>' in k or 'gno_apply_mfma_fwd' in k: print(k, json.dumps(v))
"
find $O -name "*counter_collection.csv" -size +8M -delete
