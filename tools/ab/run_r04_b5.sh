mkdir -p gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_gpu_block_cg.py -x -q 2>&1 | grep -E "AssertionError|passed|failed|assert |Error" | cut -c1-300 > gpurun_out/r04/t_block8.log; cat gpurun_out/r04/t_block8.log
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance"
for s in 16; do
timeout -k 10 300 python bench.py --workload tile:5 --steps 3 --warmup 1 --x-block $s $B > gpurun_out/r04/b3_t5_s$s.json 2> gpurun_out/r04/b3_t5_s$s.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b3_t5_s$s.json'));print('tile:5 s$s',d['ms_per_step'],d['per_step']['cg_iters_X'],d['split_ms'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us'])"
done
timeout -k 10 400 python bench.py --workload tile:10 --steps 3 --warmup 0 --x-block 16 $B > gpurun_out/r04/b3_t10_s16.json 2> gpurun_out/r04/b3_t10_s16.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b3_t10_s16.json'));print('tile:10 s16',d['ms_per_step'],d['per_step']['cg_iters_X'],d['split_ms'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us'])"
