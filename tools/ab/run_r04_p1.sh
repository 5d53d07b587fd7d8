mkdir -p gpurun_out/r04
timeout -k 10 600 python -u -m pytest tests/test_dist_sharded.py -x -q -s -k "peer_write" > gpurun_out/r04/t_p1.log 2>&1; tail -30 gpurun_out/r04/t_p1.log | cut -c1-400
timeout -k 10 900 python -u -m pytest tests/test_dist_sharded.py tests/test_gpu_block_cg.py -x -q > gpurun_out/r04/t_p1b.log 2>&1; tail -5 gpurun_out/r04/t_p1b.log | cut -c1-300
