# rocprofv3 counter pass for the matrix instructions of k_xtb_apply (tile:5, one step): own --pmc run, kernel trace only
mkdir -p gpurun_out/r04
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order"
cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/pmc_$n -- python3 $GRAFT_REPO_ROOT/bench.py --workload tile:5 --steps 1 --warmup 0 $B > $GRAFT_REPO_ROOT/gpurun_out/r04/pmc_$n.out 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for f in glob.glob('gpurun_out/r04/pmc_SQ_*/*/*counter_collection.csv'):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('void k_xtb_apply'):
            a = acc[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, (n, v) in acc.items(): out[k] = {"launches": n, "per_launch": v / n}
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/r04/pmc_mfma_summary.json', 'w'), indent=1)
PY
