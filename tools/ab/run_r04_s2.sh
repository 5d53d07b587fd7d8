mkdir -p gpurun_out/r04
( time timeout -k 10 900 python -u -m pytest tests/test_gpu_scale.py -x -q -k tile20 ) > gpurun_out/r04/t_s1.log 2>&1; tail -6 gpurun_out/r04/t_s1.log | cut -c1-300
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
for sd in 0 2; do
DKMC_XTB_SIDE=$sd timeout -k 10 300 python bench.py --workload 7.5nm --steps 20 --warmup 3 $B > gpurun_out/r04/s2_7p5_side$sd.json 2> gpurun_out/r04/s2_7p5_side$sd.err
python3 - $sd <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r04/s2_7p5_side%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('side',sys.argv[1], d['ms_per_step'], d['split_ms'], d['per_step']['cg_iters_X'], d['roofline']['avg_launch_us'], d['roofline'].get('row_kernel_us'))
PY
done
