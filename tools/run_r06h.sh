O=gpurun_out/r06h; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_c_abi_gpu.py tests/test_node_vmh_gpu.py tests/test_configs_gpu.py tests/test_gcn_gpu.py -x -q > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -15 $O/pytest.txt
