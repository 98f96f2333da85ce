#!/bin/bash
# same-box A/B of two builds on the Cora-shaped C1 solve (tools/bench_c1_cora.py): neuralgraphpde.jl_amd/ab_{base,noslp}.so
cd $GRAFT_REPO_ROOT
L=neuralgraphpde.jl_amd
for rep in 1 2 3; do
for v in base noslp; do
  cp $L/ab_$v.so $L/libngpde_hip.so
  echo "== $v pass $rep"; python3 tools/bench_c1_cora.py 2>/dev/null | grep -i "cora\|pairs" | cut -c1-150
done
done
