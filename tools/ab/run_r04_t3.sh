mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_dist_sharded.py -q 2>&1 | grep -E "AssertionError|passed|failed|assert |Error|^tests|^E " | cut -c1-400 > gpurun_out/r04/t_dist1.log; tail -30 gpurun_out/r04/t_dist1.log
