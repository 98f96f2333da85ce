O=gpurun_out/r06d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_layer_abi_gpu.py tests/test_mp_gpu.py tests/test_configs_gpu.py -x -q -k "gno or gform or c5" > $O/pytest_gno.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gno.txt
tail -4 $O/pytest_gno.txt
for c in 16 32; do for r in 0.1 0.05; do
  NGPDE_GNO_GFORM_CHUNK=$c timeout -k 10 200 python tools/bench_layers.py --only c5 --width 128 --radius $r --reps 30 2>/dev/null | tail -1 > $O/c5_chunk${c}_r$r.txt
  echo "chunk $c radius $r: $(cat $O/c5_chunk${c}_r$r.txt)"
done; done
cd /tmp && export TMPDIR=/tmp
for c in 16 32; do
NGPDE_GNO_GFORM_CHUNK=$c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats$c -o k -- python3 $GRAFT_REPO_ROOT/tools/bench_layers.py --only c5 --width 128 --radius 0.1 --reps 10 > $GRAFT_REPO_ROOT/$O/stats$c.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kstats.py $GRAFT_REPO_ROOT/$O/stats$c 12
done
