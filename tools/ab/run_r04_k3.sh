mkdir -p gpurun_out/r04
timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -x -q -k "CB_edge or potential or crossbar or background or log" > gpurun_out/r04/t_k3.log 2>&1; tail -4 gpurun_out/r04/t_k3.log | cut -c1-300
bash tools/ab/run_r04_k2.sh
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/prof_xbar.out') if l.startswith('{')][-1])
v=d['scale_points']['crossbar_10nm_5pitch']
print(v.get('ms_per_step'), v.get('split_ms'), v.get('per_step'), v.get('vs_reference_log'))
PY
