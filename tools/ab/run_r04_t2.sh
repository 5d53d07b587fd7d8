mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_block_cg.py tests/test_dist_sharded.py -q -k "grouping or soak or small or rccl or pair_sum" 2>&1 | grep -E "AssertionError|passed|failed|assert |Error|^tests" | cut -c1-300 > gpurun_out/r04/t_new1.log; tail -20 gpurun_out/r04/t_new1.log
