#!/bin/bash
# MFMA-pipe and LDS counters for the headline kernels (bench.py), the C4 / C5 layers and the VMH solver, each counter
# group in its own rocprofv3 run (no trace domains next to --pmc).  usage (through gpurun): bash tools/pmc_mfma.sh r01_e
TAG=${1:-round}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES"; do
  n=$(echo $grp | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $O/c2_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batched 0 --no-secondary > $O/c2_$n.log 2>&1
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $O/c4_$n -- python3 $R/tools/bench_layers.py --only c4 --traj 64 --reps 2 > $O/c4_$n.log 2>&1
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $O/c5_$n -- python3 $R/tools/bench_layers.py --only c5 --width 128 --radius 0.1 --reps 2 > $O/c5_$n.log 2>&1
  # NeuralODE(VMHConv) on the device-resident plan: one cloud of 3 000 points (half-tile units), a batch of 8 (tile rounds)
  REPS=1 timeout 600 rocprofv3 --pmc $grp --output-format csv -d $O/vmh_$n -- python3 $R/tools/debug_vmh_node.py > $O/vmh_$n.log 2>&1
  NB=8 REPS=1 timeout 600 rocprofv3 --pmc $grp --output-format csv -d $O/vmhb_$n -- python3 $R/tools/bench_vmh_batch.py > $O/vmhb_$n.log 2>&1
done
cd $R
python3 tools/pmc_mfma_summary.py $O > $O/mfma_summary.json 2> $O/mfma_summary.err
cat $O/mfma_summary.json
find $O -name "*counter_collection.csv" -size +8M -delete
