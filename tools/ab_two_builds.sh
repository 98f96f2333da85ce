#!/bin/bash
# A/B of two builds on ONE box: alternating passes of bench_layers (C4) and the bench headline legs
cd $GRAFT_REPO_ROOT
L=neuralgraphpde.jl_amd
for rep in 1 2; do
for v in base noslp; do
  cp $L/ab_$v.so $L/libngpde_hip.so
  echo "== $v pass $rep"
  python3 tools/bench_layers.py --only c4 --traj 64 --reps 30 2>/dev/null | cut -c80-170
  python3 bench.py --no-rocprof --steps 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=d['secondary']
print('headline', d['value'], 'events', d['roofline'].get('frac_events'), 'batched', d['batched']['value'], '32k', d['larger_graph']['value'], '65k', d['larger_graph']['x4']['value'])
print({k:(v.get('ms_forward'),v.get('ms_forward_backward'),v.get('value')) for k,v in s.items()})
"
done
done
