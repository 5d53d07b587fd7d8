mkdir -p gpurun_out/r04
( time timeout -k 10 1100 python -u -m pytest tests -q -m gpu -v ) > gpurun_out/r04/t_full.log 2>&1; tail -8 gpurun_out/r04/t_full.log | cut -c1-300
