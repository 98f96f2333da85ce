O=gpurun_out/r06o; mkdir -p $O
L=neuralgraphpde.jl_amd
cp $L/libngpde_hip.so $L/ab_keep.so
for rep in 1 2; do for v in $VARIANTS; do
  cp $L/ab_$v.so $L/libngpde_hip.so
  echo "== $v pass $rep"
  timeout -k 10 120 python tools/time_gat_node.py 2>&1 | grep members
done; done | tee $O/ab.txt
cp $L/ab_keep.so $L/libngpde_hip.so
