#!/bin/bash
# same-box A/B of two builds (neuralgraphpde.jl_amd/ab_{base,noslp}.so) on BASELINE config 4: layer times and per-kernel averages
cd $GRAFT_REPO_ROOT
L=neuralgraphpde.jl_amd
export TMPDIR=/tmp
for rep in 1 2; do
for v in base noslp; do
  cp $L/ab_$v.so $L/libngpde_hip.so
  echo "== $v pass $rep"
  python3 tools/bench_layers.py --only c4 --traj 64 --reps 30 2>/dev/null | cut -c80-170
  (cd /tmp && rm -rf /tmp/abk && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -o k -- python3 $GRAFT_REPO_ROOT/tools/bench_layers.py --only c4 --traj 64 --reps 10 > /dev/null 2>&1)
  python3 tools/kstats.py /tmp/abk 8 | cut -c1-110
done
done
