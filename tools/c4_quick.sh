#!/bin/bash
# BASELINE config 4 only: layer times, rocprofv3 kernel stats, matrix-pipe / LDS counters (each group in its own run).
# usage (through gpurun): bash tools/c4_quick.sh r05_b [pmc]
TAG=${1:-c4}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
python3 tools/bench_layers.py --only c4 --traj 64 --reps 20 > $O/layers.jsonl 2>$O/layers.err
cat $O/layers.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -o k -- python3 $R/tools/bench_layers.py --only c4 --traj 64 --reps 10 > $O/c4_stats.log 2>&1
cd $R
python3 tools/kstats.py $O/c4_stats 12 | tee $O/c4_kstats.txt
if [ "$2" = "pmc" ]; then
  cd /tmp
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
    n=$(echo $grp | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/c4_$n -- python3 $R/tools/bench_layers.py --only c4 --traj 64 --reps 2 > $O/c4_$n.log 2>&1
  done
  cd $R
  python3 tools/pmc_mfma_summary.py $O | tee $O/mfma_summary.json
  find $O -name "*counter_collection.csv" -size +8M -delete
fi
find $O -name "*_kernel_trace.csv" -size +8M -delete
