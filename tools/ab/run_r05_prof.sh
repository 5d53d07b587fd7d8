# round 5: rocprofv3 kernel-trace summaries of the commands profiles/README.md cites (one `gpurun -- 'bash tools/ab/run_r05_prof.sh'` call)
mkdir -p gpurun_out/r05
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order --no-scaling-model"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05/prof_t10 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload tile:10 --steps 3 --warmup 1 $B > $GRAFT_REPO_ROOT/gpurun_out/r05/prof_t10.out 2>&1 &&
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05/prof_7p5 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload 7.5nm --steps 10 --warmup 2 $B > $GRAFT_REPO_ROOT/gpurun_out/r05/prof_7p5.out 2>&1 &&
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05/prof_t20 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload 2.5nm --steps 1 --warmup 0 --no-cpu-baseline --scale-points tile:20:nocurrent --no-scaling-model --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order > $GRAFT_REPO_ROOT/gpurun_out/r05/prof_t20.out 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_t10 prof_7p5 prof_t20; do f=$(ls gpurun_out/r05/$d/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/r05/${d}_kernel_stats.csv; echo $d; python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r['Name'][:44].ljust(46), r['Calls'].rjust(7), '%10.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
done
rm -rf gpurun_out/r05/prof_t10 gpurun_out/r05/prof_7p5 gpurun_out/r05/prof_t20
