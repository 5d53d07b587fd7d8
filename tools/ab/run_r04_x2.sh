mkdir -p gpurun_out/r04
timeout -k 10 600 python -u -m pytest tests/test_gpu_block_cg.py -x -q -k smooth > gpurun_out/r04/t_x2.log 2>&1; tail -6 gpurun_out/r04/t_x2.log | cut -c1-300
