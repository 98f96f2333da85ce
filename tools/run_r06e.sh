cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06e; mkdir -p $O
for d in 0 1 2 4 3 5 6 7; do
NGPDE_GFORM_DEBUG=$d NGPDE_GNO_GFORM_CHUNK=32 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$d -o k -- python3 $GRAFT_REPO_ROOT/tools/bench_layers.py --only c5 --width 128 --radius 0.1 --reps 5 > $O/s$d.log 2>&1
echo "dbg=$d $(python3 $GRAFT_REPO_ROOT/tools/kstats.py $O/s$d 12 | grep gform_fwd)"
done
