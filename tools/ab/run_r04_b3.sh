mkdir -p gpurun_out/r04
./tools/probe_mfma_f64_4x4x4 > gpurun_out/r04/probe_4x4x4.txt 2>&1; head -20 gpurun_out/r04/probe_4x4x4.txt
timeout -k 10 500 python -m pytest tests/test_gpu_block_cg.py -x -q 2>&1 | grep -E "AssertionError|passed|failed|assert |Error" | cut -c1-300 > gpurun_out/r04/t_block6.log; cat gpurun_out/r04/t_block6.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_s16b --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload 7.5nm --steps 5 --warmup 1 --x-block 16 --no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_s16b.out 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r04/prof_s16b/*/*kernel_stats.csv | head -1); grep xtb $f | cut -d, -f1-4 | cut -c1-40,200-
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r04/prof_s16b.out | head -2
