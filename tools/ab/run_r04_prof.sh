mkdir -p gpurun_out/r04
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_t10 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload tile:10 --steps 2 --warmup 1 $B > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_t10.out 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_7p5 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload 7.5nm --steps 10 --warmup 2 $B > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_7p5.out 2>&1
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r04/pmc_list.txt 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_t10 prof_7p5; do f=$(ls gpurun_out/r04/$d/*/*kernel_stats.csv | head -1); echo $d; python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r['Name'][:40].ljust(42), r['Calls'].rjust(7), '%10.1f us'%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
done
grep -i "mfma" gpurun_out/r04/pmc_list.txt | head -20
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r04/prof_t10.out | head -1
