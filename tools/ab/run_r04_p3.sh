# the default N > 1 workload (tile:10) with two ranks sharing ONE GPU through the self-launcher; exchange by peer writes (the host transport's
# all-gather of 2 x 11.9 MB per sweep through pinned memory + gloo would dominate)
mkdir -p gpurun_out/r04
( time DKMC_BENCH_BACKEND=gloo DKMC_BENCH_SINGLE_DEVICE=1 timeout -k 10 900 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-replicas --peer-exchange > gpurun_out/r04/p3_2rank_t10_peer.json 2> gpurun_out/r04/p3_2rank_t10_peer.err ) 2> gpurun_out/r04/p3.time
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/p3_2rank_t10_peer.json').read().strip().splitlines()[-1])
print(d.get('error'), d['value'], d['ms_per_step'], d['n_gpus'], d['config'].get('comm_ranks'), d['config'].get('exchange'), d['per_step'].get('cg_iters_X'), json.dumps(d['sharding'])[:700])
print(json.dumps(d.get('single_gpu_reference'))[:500], d.get('strong_scaling_speedup'))
PY
tail -3 gpurun_out/r04/p3.time; tail -2 gpurun_out/r04/p3_2rank_t10_peer.err | cut -c1-200
