#!/bin/bash
# BASELINE config 4 under rocprofv3 for a list of NGPDE_DENSE_DEPHASE values (start delay, in shader cycles, of the second workgroup of
# every CU in the streaming Dense launches): per-kernel averages of the five streaming kernels + the layer times.  DESIGN 5.4.
# usage (through gpurun): bash tools/dephase_sweep.sh TAG "0 4000 8000 16000"
TAG=${1:-dephase}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for D in $2; do
  export NGPDE_DENSE_DEPHASE=$D
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d$D -o k -- python3 $R/tools/bench_layers.py --only c4 --traj 64 --reps 10 > $O/d$D.log 2>&1 || exit 1
  echo "== NGPDE_DENSE_DEPHASE=$D" | tee -a $O/summary.txt
  grep ms_forward $O/d$D.log | cut -c1-200 | tee -a $O/summary.txt
  python3 $R/tools/kstats.py $O/d$D 9 | tee -a $O/summary.txt
  find $O -name "*_kernel_trace.csv" -delete
done
