mkdir -p gpurun_out/r04
( time timeout -k 10 900 python -u -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q -k "tile10 or potential or CB_edge or superstep_sequence or K_blocked or crossbar_log" ) > gpurun_out/r04/t_s3.log 2>&1; tail -12 gpurun_out/r04/t_s3.log | cut -c1-300
