mkdir -p gpurun_out/r04
B="--no-cpu-baseline --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r04/prof_xbar
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_xbar --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload 2.5nm --steps 1 --warmup 0 --scale-points crossbar_10nm_5pitch $B > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_xbar.out 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r04/prof_xbar/*/*kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(r['Name'][:40].ljust(42), r['Calls'].rjust(7), '%10.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/prof_xbar.out') if l.startswith('{')][-1])
v=d['scale_points']['crossbar_10nm_5pitch']
print(v.get('ms_per_step'), v.get('split_ms'), v.get('per_step'), v.get('vs_reference_log'))
PY
