mkdir -p gpurun_out/r04
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_dist_sharded.py -x -q -k "stop_word or K_blocked or peer_write or emulated or rccl" > gpurun_out/r04/t_v1.log 2>&1; tail -4 gpurun_out/r04/t_v1.log | cut -c1-300
bash tools/ab/run_r04_mfma_pmc.sh > gpurun_out/r04/mfma_pmc.log 2>&1; tail -30 gpurun_out/r04/mfma_pmc.log
