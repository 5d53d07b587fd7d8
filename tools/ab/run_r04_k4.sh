mkdir -p gpurun_out/r04
for e in 0 1; do
DKMC_KCG_PADDED=$e timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -x -q -k "crossbar_log" > gpurun_out/r04/t_k4_$e.log 2>&1; tail -3 gpurun_out/r04/t_k4_$e.log | cut -c1-300; grep -n "Error" gpurun_out/r04/t_k4_$e.log | head -3
done
