# round 5: per-kernel times of one block-CG sweep at tile:10 (rocprofv3 kernel trace of one short bench run) -- the rest of the sweep beside k_xtb_apply
mkdir -p gpurun_out/r05
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order --no-scaling-model"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05/prof_tail --output-format csv -- python3 $R/bench.py --workload tile:10 --steps 2 --warmup 1 $B > $R/gpurun_out/r05/prof_tail.out 2>&1
cd $R
f=$(ls gpurun_out/r05/prof_tail/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/r05/prof_tail_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(r['Name'][:44].ljust(46), r['Calls'].rjust(7), '%10.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
tail -c 400 gpurun_out/r05/prof_tail.out | head -c 300
rm -rf gpurun_out/r05/prof_tail
