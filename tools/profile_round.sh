#!/bin/bash
# One measurement pass on the GPU box: bench line, rocprofv3 kernel stats, PMC passes (each counter group in its own run, no
# trace domains combined with --pmc), layer benches, graph-build timings.  Everything lands in gpurun_out/<tag>/.
# usage (through gpurun):  bash tools/profile_round.sh r01_d
TAG=${1:-round}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --batched 0 --no-secondary > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batched 0 --no-secondary > $O/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batched 0 --no-secondary > $O/pmc_l2.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_l2 $O/pmc_traffic.json > $O/pmc_summary.log 2>&1
# the replayed plan (NGPDE_NO_PERSISTENT=1) for comparison: bench line + kernel stats
NGPDE_NO_PERSISTENT=1 python3 bench.py --no-cpu-baseline > $O/bench_replayed.json 2> /dev/null
python3 tools/bench_layers.py --only c3 --reps 20 > $O/layers.jsonl 2>/dev/null
python3 tools/bench_layers.py --only c4 --traj 64 --reps 10 >> $O/layers.jsonl 2>/dev/null
python3 tools/bench_layers.py --only c5 --width 128 --reps 10 >> $O/layers.jsonl 2>/dev/null
python3 tools/bench_layers.py --only c5 --width 128 --radius 0.1 --reps 10 >> $O/layers.jsonl 2>/dev/null
# the C3 layer's kernels (one-launch GAT layer + two-launch pullback), the any-width GCN path, the generic NeuralODE path
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -o k -- python3 $R/tools/bench_layers.py --only c3 --reps 50 > $O/c3_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gnode_stats -o k -- python3 $R/tools/trace_generic_node.py 10 > $O/gnode_stats.log 2>&1
cd $R
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -o k -- python3 $R/tools/bench_layers.py --only c4 --traj 64 --reps 10 > $O/c4_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -o k -- python3 $R/tools/bench_layers.py --only c5 --width 128 --radius 0.1 --reps 10 > $O/c5_stats.log 2>&1
cd $R
# NeuralODE(VMHConv) on the device-resident plan: outputs against the generic solver, kernel stats, phase stamps (diagnostic library)
STEPS=20 TAB=tsit5 REPS=5 timeout 200 python3 tools/debug_vmh_node.py > $O/vmh_node.txt 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/vmh_stats -o k -- python3 $R/tools/debug_vmh_node.py > $O/vmh_stats.log 2>&1
cd $R
[ -f neuralgraphpde.jl_amd/libngpde_diag.so ] && timeout 200 python3 tools/stamps_vmh.py > $O/vmh_stamps.txt 2>/dev/null
python3 tools/hbm_rw_probe.py > $O/hbm_rw_probe.jsonl 2>/dev/null
[ -f neuralgraphpde.jl_amd/libngpde_diag.so ] && timeout 200 python3 tools/stamps_pair.py > $O/pair_stamps.txt 2>/dev/null
# does the fp32 matrix instruction run beside the VALU?
hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/ovl 2>/dev/null && timeout 120 /tmp/ovl > $O/mfma_valu_overlap.jsonl 2>/dev/null
[ -f neuralgraphpde.jl_amd/libngpde_diag.so ] && timeout 200 python3 tools/stamps_edge64.py 64 > $O/edge64_stamps.txt 2>/dev/null
python3 tools/bench_gcn_anywidth.py > $O/gcn_anywidth.jsonl 2>/dev/null
python3 tools/trace_generic_node.py 50 > $O/generic_node.jsonl 2>/dev/null
python3 tools/trace_generic_node.py 50 capture >> $O/generic_node.jsonl 2>/dev/null
python3 tools/bench_graph_build.py > $O/graph_build.jsonl 2>/dev/null
NGPDE_HOST_GRAPH_BUILD=1 python3 tools/bench_graph_build.py | sed 's/"graph"/"builder": "host", "graph"/' >> $O/graph_build.jsonl 2>/dev/null
python3 tools/bench_dense.py > $O/dense.jsonl 2>/dev/null
# keep only what is small
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
cat $O/layers.jsonl $O/graph_build.jsonl
ls -la $O | head -30
