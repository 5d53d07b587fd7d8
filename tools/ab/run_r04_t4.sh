mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_dist_sharded.py tests/test_gpu_block_cg.py -q 2>&1 | grep -E "AssertionError|passed|failed|assert |Error|^tests|^E " | cut -c1-400 > gpurun_out/r04/t_dist2.log; tail -12 gpurun_out/r04/t_dist2.log
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
for s in 16 12; do
timeout -k 10 300 python bench.py --workload tile:10 --steps 2 --warmup 1 --x-block $s $B > gpurun_out/r04/b4_t10_s$s.json 2> gpurun_out/r04/b4_t10_s$s.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b4_t10_s$s.json'));print('tile:10 s$s',d['ms_per_step'],d['per_step']['cg_iters_X'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us'], d['cold_step'])"
done
timeout -k 10 300 python bench.py --workload 7.5nm --steps 10 --warmup 2 --x-block 12 $B > gpurun_out/r04/b4_7p5_s12.json 2> gpurun_out/r04/b4_7p5_s12.err; python -c "
import json;d=json.load(open('gpurun_out/r04/b4_7p5_s12.json'));print('7.5nm s12',d['ms_per_step'],d['per_step']['cg_iters_X'],d['roofline']['avg_launch_us'],d['roofline']['row_kernel_us'])"
