mkdir -p gpurun_out/r04
timeout -k 10 600 python -u -m pytest tests/test_gpu_block_cg.py tests/test_dist_sharded.py -x -q > gpurun_out/r04/t_x3.log 2>&1; tail -4 gpurun_out/r04/t_x3.log | cut -c1-300
bash tools/ab/run_r04_bench.sh > gpurun_out/r04/bench_final.log 2>&1; tail -3 gpurun_out/r04/bench_driver_cmd.time
