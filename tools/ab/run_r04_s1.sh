mkdir -p gpurun_out/r04
( time DKMC_SLOW_TESTS=1 timeout -k 10 900 python -u -m pytest tests/test_gpu_scale.py -x -q -k tile20 ) > gpurun_out/r04/t_s1.log 2>&1; tail -25 gpurun_out/r04/t_s1.log | cut -c1-300
