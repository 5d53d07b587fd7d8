mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -q -m gpu 2>&1 | grep -E "AssertionError|passed|failed|assert |Error|^tests" | cut -c1-300 > gpurun_out/r04/t_full1.log; tail -60 gpurun_out/r04/t_full1.log
