O=gpurun_out/r06m; mkdir -p $O
ROUNDS=2 timeout -k 10 200 python tools/exp_own_first.py 2>&1 | grep round | tee $O/exp.txt
for rep in 1 2; do for m in 0 1; do
NGPDE_NO_OWN_FIRST=$m timeout -k 10 300 python bench.py --no-rocprof --no-secondary --batched 8 > $O/bench_$m.json 2> $O/bench_$m.err; python - $m <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r06m/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('NO_OWN_FIRST',sys.argv[1],'value', d['value'], 'frac', d['roofline']['frac'], 'batched', d.get('batched',{}).get('value'))
PY
done; done 2>&1 | tee $O/bench.txt
timeout -k 10 900 python -m pytest tests/test_gcn_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py tests/test_hub_plan_gpu.py tests/test_c_abi_gpu.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
