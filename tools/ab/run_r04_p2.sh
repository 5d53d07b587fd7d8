# two ranks sharing ONE GPU through the self-launcher (gloo process group, host transport): the block loop's exchange as the transport's
# all-gather and as the peer-write exchange
mkdir -p gpurun_out/r04
for v in ag peer; do
  X=""; [ $v = peer ] && X="--peer-exchange"
  DKMC_BENCH_BACKEND=gloo DKMC_BENCH_SINGLE_DEVICE=1 timeout -k 10 400 python3 bench.py --gpus 2 --workload tile:5 --steps 3 --warmup 1 --no-replicas $X > gpurun_out/r04/p2_2rank_t5_$v.json 2> gpurun_out/r04/p2_2rank_t5_$v.err
  python3 - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r04/p2_2rank_t5_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d.get('error'), d['value'], d['ms_per_step'], d['n_gpus'], d['config'].get('comm_ranks'), d['config'].get('exchange'), d['config'].get('transport'), d['per_step'].get('cg_iters_X'), json.dumps(d['sharding'])[:600])
PY
  tail -2 gpurun_out/r04/p2_2rank_t5_$v.err | cut -c1-200
done
