O=gpurun_out/r06i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_layer_abi_gpu.py tests/test_mp_gpu.py -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -4 $O/pytest.txt
SECONDS=0; python bench.py > $O/bench.json 2> $O/bench.err; echo "default bench.py took $SECONDS s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06i/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'frac', d['roofline']['frac'], d['roofline'].get('frac_events'))
PY
