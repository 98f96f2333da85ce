set -x
mkdir -p gpurun_out/r06b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_layer_abi_gpu.py tests/test_mp_gpu.py -x -q -k "gno or gform" > gpurun_out/r06b/pytest_gno.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r06b/pytest_gno.txt
tail -5 gpurun_out/r06b/pytest_gno.txt
timeout -k 10 300 python -m pytest tests/test_configs_gpu.py -x -q -k "c5" > gpurun_out/r06b/pytest_c5.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r06b/pytest_c5.txt
tail -3 gpurun_out/r06b/pytest_c5.txt
for r in 0.1 0.05; do
  timeout -k 10 200 python tools/bench_layers.py --only c5 --width 128 --radius $r --reps 20 > gpurun_out/r06b/c5_gform_r$r.txt 2>&1
  NGPDE_NO_GNO_GFORM=1 timeout -k 10 200 python tools/bench_layers.py --only c5 --width 128 --radius $r --reps 20 > gpurun_out/r06b/c5_bysource_r$r.txt 2>&1
done
cat gpurun_out/r06b/c5_*.txt | grep -v "^$" | tail -20
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r06b/prof_c5 -o c5 -- python tools/bench_layers.py --only c5 --width 128 --radius 0.1 --reps 10 > gpurun_out/r06b/prof_c5.log 2>&1
timeout -k 10 200 python tools/exp_own_first.py > gpurun_out/r06b/exp_own_first.txt 2>&1
tail -4 gpurun_out/r06b/exp_own_first.txt
