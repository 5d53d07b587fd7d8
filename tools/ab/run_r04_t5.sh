mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_dist_sharded.py tests/test_gpu_block_cg.py -q 2>&1 | grep -E "AssertionError|passed|failed|assert |Error|^tests|^E " | cut -c1-400 > gpurun_out/r04/t_dist3.log; tail -12 gpurun_out/r04/t_dist3.log
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
timeout -k 10 300 python bench.py --workload tile:10 --steps 2 --warmup 1 $B > gpurun_out/r04/b5_t10.json 2> gpurun_out/r04/b5_t10.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b5_t10.json'));print('tile:10',d['ms_per_step'],d['per_step']['cg_iters_X'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us']); print(d['strong_scaling_model']['by_n_gpus'])"
timeout -k 10 300 python bench.py --workload 7.5nm --steps 10 --warmup 2 $B > gpurun_out/r04/b5_7p5.json 2> gpurun_out/r04/b5_7p5.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b5_7p5.json'));print('7.5nm',d['ms_per_step'],d['per_step']['cg_iters_X'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us'])"
DKMC_BENCH_BACKEND=gloo DKMC_BENCH_SINGLE_DEVICE=1 timeout -k 10 400 python3 bench.py --gpus 2 --workload tile:5 --steps 3 --warmup 1 --no-replicas > gpurun_out/r04/b5_2rank_t5.json 2> gpurun_out/r04/b5_2rank_t5.err; tail -c 1800 gpurun_out/r04/b5_2rank_t5.json; tail -3 gpurun_out/r04/b5_2rank_t5.err | cut -c1-300
