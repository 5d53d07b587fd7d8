mkdir -p gpurun_out/r04
( time timeout -k 10 1100 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_cmd.json 2> gpurun_out/r04/bench_driver_cmd.err ) 2> gpurun_out/r04/bench_driver_cmd.time
tail -3 gpurun_out/r04/bench_driver_cmd.time; tail -5 gpurun_out/r04/bench_driver_cmd.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_driver_cmd.json').read().strip().splitlines()[-1])
def show(k,v,ind=0):
    if isinstance(v,dict):
        print(' '*ind+k+':')
        for kk,vv in v.items(): show(kk,vv,ind+2)
    else:
        sv=str(v); print(' '*ind+k+': '+(sv if len(sv)<160 else sv[:160]+'...'))
for k,v in d.items(): show(k,v)
PY
