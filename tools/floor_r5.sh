#!/bin/bash
# round 5: the exchange floor with tagged quads (modes 8 / 9) beside the per-tile flags (mode 1); see tools/persistent_floor.hip
set -e
mkdir -p gpurun_out
hipcc -O2 --offload-arch=gfx950 tools/persistent_floor.hip -o /tmp/pf 2>/dev/null
out=gpurun_out/r05_a_floor_tagged_quads.jsonl
: > $out
for work in 0 100 200; do
  timeout -k 10 60 /tmp/pf 1 1200 $work 1 0 0 >> $out
  for sl in 0 4 16; do timeout -k 10 60 /tmp/pf 8 1200 $work 1 0 $sl >> $out; done
  timeout -k 10 60 /tmp/pf 9 1200 $work 1 0 0 >> $out
done
cat $out
