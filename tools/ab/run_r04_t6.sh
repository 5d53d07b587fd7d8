mkdir -p gpurun_out/r04
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
timeout -k 10 300 python bench.py --workload tile:10 --steps 2 --warmup 1 $B > gpurun_out/r04/b6_t10.json 2> gpurun_out/r04/b6_t10.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b6_t10.json'));print('tile:10',d['ms_per_step'],d['per_step']['cg_iters_X'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us'],d['steady'])"
timeout -k 10 300 python bench.py --workload tile:5 --steps 3 --warmup 1 $B > gpurun_out/r04/b6_t5.json 2> gpurun_out/r04/b6_t5.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b6_t5.json'));print('tile:5',d['ms_per_step'],d['per_step']['cg_iters_X'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us'])"
timeout -k 10 400 python -m pytest tests/test_gpu_scale.py tests/test_gpu_block_cg.py -q 2>&1 | tail -3
