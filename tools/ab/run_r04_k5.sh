mkdir -p gpurun_out/r04
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r04/t_k5.log 2>&1; tail -4 gpurun_out/r04/t_k5.log | cut -c1-300
bash tools/ab/run_r04_k2.sh
