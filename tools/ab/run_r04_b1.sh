mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_block_cg.py -x -q -k small 2>&1 | tail -15 > gpurun_out/r04/t_block4.log; cat gpurun_out/r04/t_block4.log
for s in 1 8 16; do
timeout -k 10 200 python bench.py --workload 7.5nm --steps 10 --warmup 2 --x-block $s --no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance > gpurun_out/r04/b_7p5_s$s.json 2> gpurun_out/r04/b_7p5_s$s.err; tail -c 1500 gpurun_out/r04/b_7p5_s$s.json; echo
done
