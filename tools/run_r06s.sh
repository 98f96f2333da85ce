O=gpurun_out/r06s; mkdir -p $O
timeout -k 10 500 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_suite.txt 2>&1; echo "default: $(tail -1 $O/gpu_suite.txt)"
bash tools/switch_matrix.sh r06s NGPDE_NO_OWN_FIRST=1 NGPDE_OWN_FIRST_ADJOINT=1
