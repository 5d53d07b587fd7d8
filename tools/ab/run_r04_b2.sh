mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_gpu_block_cg.py -x -q -k small 2>&1 | grep -E "AssertionError|passed|failed|assert " | cut -c1-300 > gpurun_out/r04/t_block5.log; cat gpurun_out/r04/t_block5.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_s16 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload 7.5nm --steps 5 --warmup 1 --x-block 16 --no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_s16.out 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r04/prof_s16/*/*kernel_stats.csv | head -1); head -25 $f | cut -c1-200
