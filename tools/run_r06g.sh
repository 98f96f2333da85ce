O=gpurun_out/r06g; mkdir -p $O
ROUNDS=2 timeout -k 10 200 python tools/exp_own_first.py 2>&1 | grep round | tee $O/exp.txt
timeout -k 10 120 python tools/stamps_persistent.py > $O/stamps.txt 2>&1; grep -E "^bwd" $O/stamps.txt
timeout -k 10 900 python -m pytest tests/test_gcn_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py tests/test_hub_plan_gpu.py -x -q -k "node or solve or persistent or c2 or c1 or soak or plan" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout -k 10 300 python bench.py --no-rocprof --no-secondary --batched 8 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06g/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'frac', d['roofline']['frac'], 'batched', d.get('batched',{}).get('value'))
PY
