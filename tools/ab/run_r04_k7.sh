mkdir -p gpurun_out/r04
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q -k "K_blocked or crossbar_log or CB_edge or potential_fields or kmc_time_vs" > gpurun_out/r04/t_k7.log 2>&1; tail -5 gpurun_out/r04/t_k7.log | cut -c1-300
B="--no-cpu-baseline --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
for kb in 1 2; do
timeout -k 10 300 python bench.py --workload 2.5nm --steps 2 --warmup 1 --scale-points crossbar_10nm_5pitch --k-blocked $kb $B > gpurun_out/r04/k7_xbar_kb$kb.json 2> gpurun_out/r04/k7_xbar_kb$kb.err
python3 - $kb <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r04/k7_xbar_kb%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
v=d['scale_points']['crossbar_10nm_5pitch']
r=v.get('roofline_K_cg') or {}
print('k_blocked',sys.argv[1], v.get('ms_per_step'), v.get('split_ms'), v.get('per_step'), v.get('vs_reference_log'), r.get('us_per_iteration'), r.get('kernel'), r.get('frac'))
PY
tail -2 gpurun_out/r04/k7_xbar_kb$kb.err | cut -c1-300
done
