mkdir -p gpurun_out/r04
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
timeout -k 10 300 python bench.py --workload 7.5nm --steps 10 --warmup 2 $B > gpurun_out/r04/b8_7p5.json 2> gpurun_out/r04/b8_7p5.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b8_7p5.json'));print('7.5nm',d['ms_per_step'],d['per_step']['cg_iters_X'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us'],d['roofline']['tile_runs'])"
timeout -k 10 300 python bench.py --workload 2.5nm --steps 10 --warmup 2 $B > gpurun_out/r04/b8_2p5.json 2> gpurun_out/r04/b8_2p5.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b8_2p5.json'));print('2.5nm',d['ms_per_step'],d['per_step']['cg_iters_X'],d['roofline']['avg_launch_us'])"
timeout -k 10 600 python -m pytest tests/test_gpu_block_cg.py tests/test_dist_sharded.py -q 2>&1 | tail -3
