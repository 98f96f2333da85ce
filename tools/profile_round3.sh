#!/bin/bash
# Round-3 additions to tools/profile_round.sh (same conventions: everything lands in gpurun_out/<tag>/, the program itself follows
# `--` under rocprofv3, counters in their own runs): the interleaved batch kernels, the tile-pair kernels, their phase stamps.
# usage (through gpurun):  bash tools/profile_round3.sh r03_a
TAG=${1:-round}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
# 8 members per GPU: two members per workgroup (default) against one after the other, same build
python3 bench.py --no-cpu-baseline --no-secondary > $O/bench_batched.json 2> /dev/null
NGPDE_NO_INTERLEAVE=1 python3 bench.py --no-cpu-baseline --no-secondary > $O/bench_batched_member_by_member.json 2> /dev/null
python3 - <<PY
import json
for f in ("bench_batched.json", "bench_batched_member_by_member.json"):
    d = json.load(open("$O/" + f))
    print(f, d["value"], d["batched"]["value"], d["batched"]["ms_forward_solve"], d["batched"]["ms_backward_solve"], d["batched"]["pipeline"], d.get("larger_graph", {}).get("value"))
PY
cd /tmp && export TMPDIR=/tmp
# kernel stats of the run WITH the batched and the larger-graph legs (node_*_persistent2_kernel<.., PAIR>)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/batched_stats -o k -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/batched_stats.log 2>&1
# matrix-pipe / LDS counters of the same run
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $O/batched_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/batched_$n.log 2>&1
done
# HBM traffic of the interleaved launches
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/batched_pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/batched_pmc_$c.log 2>&1
done
cd $R
# interleaved vs member-by-member, output by output (bitwise), and where a slot-phase goes
N=16384 K=8 TAB=tsit5 STEPS=50 python3 tools/debug_interleave.py > $O/interleave_ab.txt 2>&1
[ -f neuralgraphpde.jl_amd/libngpde_diag.so ] && timeout 200 python3 tools/stamps_interleaved.py > $O/stamps_interleaved.txt 2>&1
[ -f neuralgraphpde.jl_amd/libngpde_diag.so ] && timeout 200 python3 tools/stamps_persistent.py > $O/stamps_persistent.txt 2>&1
# tile pairs: 24 576 and 32 768 nodes against the replayed plan
N=32768 PAIRS=131072 STEPS=50 python3 tools/debug_persistent.py > $O/tile_pairs_32k.txt 2>&1
N=24576 PAIRS=98304 STEPS=50 python3 tools/debug_persistent.py > $O/tile_pairs_24k.txt 2>&1
# race screens on this build
timeout 300 python3 tools/soak_replay.py 200 > $O/soak_replay.txt 2>&1
MEMBERS=4 timeout 400 python3 tools/soak_replay.py 60 >> $O/soak_replay.txt 2>&1
N=32768 timeout 400 python3 tools/soak_replay.py 100 >> $O/soak_replay.txt 2>&1
timeout 300 python3 tools/soak_layers.py 100 > $O/soak_layers.jsonl 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
tail -3 $O/interleave_ab.txt $O/tile_pairs_32k.txt $O/soak_replay.txt
ls $O | head -40
