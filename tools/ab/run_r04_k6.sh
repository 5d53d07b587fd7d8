# A/B: K-CG on the blocked form against the CSR positions (same binary): crossbar (solve_current = 0) and the 85 k device at the log's tolerance
mkdir -p gpurun_out/r04
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "K_blocked or crossbar or CB_edge or potential_fields" > gpurun_out/r04/t_k6.log 2>&1; tail -4 gpurun_out/r04/t_k6.log | cut -c1-300
B="--no-cpu-baseline --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
for kb in 0 1; do
timeout -k 10 300 python bench.py --workload 2.5nm --steps 2 --warmup 1 --scale-points crossbar_10nm_5pitch --k-blocked $kb $B > gpurun_out/r04/k6_xbar_kb$kb.json 2> gpurun_out/r04/k6_xbar_kb$kb.err
python3 - $kb <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r04/k6_xbar_kb%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
v=d['scale_points']['crossbar_10nm_5pitch']
print('k_blocked',sys.argv[1], v.get('ms_per_step'), v.get('split_ms'), v.get('per_step'), v.get('vs_reference_log'), json.dumps(v.get('roofline_K_cg'))[:600])
PY
done
