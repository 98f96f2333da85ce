set -e
mkdir -p gpurun_out/r05_c
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for w in 64 32; do
  for act in swish; do
    NGPDE_EDGE64_WGS_PER_XCD=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_c/w${w}_$act -o k -- python3 $R/tools/bench_layers.py --only c4 --traj 64 --reps 5 --act $act > $R/gpurun_out/r05_c/w${w}_$act.log 2>&1
    echo "== wgs_per_xcd $w act $act"; python3 $R/tools/kstats.py $R/gpurun_out/r05_c/w${w}_$act 4
  done
done
