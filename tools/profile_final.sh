R=$PWD
O=$R/gpurun_out/r05_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --batched 0 --no-secondary > $O/stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batched 0 --no-secondary > $O/pmc_$c.log 2>&1 || exit 1
done
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --batched 0 --no-secondary > $O/pmc_l2.log 2>&1 || exit 1
cd $R
python3 tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_l2 $O/pmc_traffic.json > $O/pmc_summary.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -o k -- python3 $R/tools/bench_layers.py --only c4 --traj 64 --reps 10 > $O/c4_stats.log 2>&1
cd $R
find $O -name "*_kernel_trace.csv" -size +2M -delete
find $O -name "*counter_collection.csv" -size +2M -delete
python3 tools/kstats.py $O/stats 4; python3 tools/kstats.py $O/c4_stats 8; cat $O/pmc_traffic.json | head -c 900
