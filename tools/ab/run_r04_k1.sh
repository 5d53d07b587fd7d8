mkdir -p gpurun_out/r04
B="--no-cpu-baseline --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q -v > gpurun_out/r04/t_k1.log 2>&1; tail -4 gpurun_out/r04/t_k1.log | cut -c1-300
timeout -k 10 300 python bench.py --workload 7.5nm --steps 10 --warmup 2 --scale-points crossbar_10nm_5pitch,tile:20:nocurrent $B > gpurun_out/r04/k1_7p5.json 2> gpurun_out/r04/k1_7p5.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/k1_7p5.json').read().strip().splitlines()[-1])
print('7.5nm', d['ms_per_step'], d['split_ms'], d['per_step'])
for k,v in d['scale_points'].items():
    print(k, v.get('ms_per_step'), v.get('split_ms'), v.get('per_step'), v.get('vs_reference_log'), json.dumps(v.get('roofline_K_cg'))[:400])
PY
tail -3 gpurun_out/r04/k1_7p5.err | cut -c1-300
