#!/bin/bash
# The -m gpu suite once per environment switch of the library (INTEGRATION.md "Environment switches"): every switch selects a form
# the tests must also pass on.  usage (through gpurun, at most four switches per call):  bash tools/switch_matrix.sh TAG VAR=VAL [VAR=VAL ...]
TAG=$1; shift
O=gpurun_out/switches_$TAG
mkdir -p $O
for sw in "$@"; do
  name=$(echo $sw | tr '=' '_')
  env $sw timeout -k 10 420 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/$name.log 2>&1
  echo "$sw: $(tail -1 $O/$name.log)" | tee -a $O/summary.txt
done
