mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_dist_sharded.py tests/test_gpu_parity.py -q -k "two_ranks or uncached or rccl or error_path or tiled_X or crossbar_with or two_devices or current_solve" 2>&1 | grep -E "AssertionError|passed|failed|assert |Error|^tests|^E " | cut -c1-400 > gpurun_out/r04/t_cache1.log; tail -15 gpurun_out/r04/t_cache1.log
