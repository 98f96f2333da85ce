#!/bin/bash
# secondary (C3 / C4 / C5 layer) numbers of bench.py with the layer-level entries and with the composed layers
for mode in new composed; do
  if [ $mode = composed ]; then export NGPDE_LAYERS_COMPOSED=1; else unset NGPDE_LAYERS_COMPOSED; fi
  python3 bench.py --no-cpu-baseline --batched 0 --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
s=d['secondary']
for k in ('C4_mppde_shard_layer','C5_gno_128_r0.05_layer','C5_gno_128_r0.1_layer'):
    v=s[k]; print('$mode',k,v['ms_forward'],v['ms_forward_backward'],'eager',v['ms_forward_eager_api'],v['ms_forward_backward_eager_api'])
"
done
