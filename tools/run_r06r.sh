O=gpurun_out/r06r; mkdir -p $O
timeout -k 10 500 python tools/fuzz_hub_node.py 40 6 > $O/fuzz_hub.txt 2>&1; tail -2 $O/fuzz_hub.txt
timeout -k 10 300 python tools/fuzz_gcn_node.py 30 6 > $O/fuzz_gcn.txt 2>&1; tail -1 $O/fuzz_gcn.txt
REPLAYS=200 timeout -k 10 300 python tools/soak_replay.py > $O/soak.txt 2>&1; tail -2 $O/soak.txt
