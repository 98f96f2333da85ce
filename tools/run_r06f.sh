O=gpurun_out/r06f; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
for r in 0.1 0.05; do
  timeout -k 10 200 python tools/bench_layers.py --only c5 --width 128 --radius $r --reps 30 2>/dev/null | tail -1 > $O/c5_r$r.txt
  echo "radius $r: $(cat $O/c5_r$r.txt)"
done
