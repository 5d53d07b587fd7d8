# round 5: where k_xtb_apply's wave cycles go -- SQ counter passes (each its own rocprofv3 --pmc run with the kernel trace only) at tile:5, one step
mkdir -p gpurun_out/r05
B="--no-cpu-baseline --scale-points none --no-alt --no-pmc --no-cpp-host --no-log-tolerance --no-device --no-reference-order --no-scaling-model"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05/pmcq_$i -- python3 $GRAFT_REPO_ROOT/bench.py --workload tile:5 --steps 1 --warmup 0 $B > $GRAFT_REPO_ROOT/gpurun_out/r05/pmcq_$i.out 2>&1 || echo "pass $i failed"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for f in glob.glob('gpurun_out/r05/pmcq_*/*/*counter_collection.csv'):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('void k_xtb_apply'):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        top = max(v); w = [x for x in v if x > 0.5 * top]
        out[k] = {"working_launches": len(w), "per_working_launch": sum(w) / max(len(w), 1)}
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/r05/pmc_apply_wave_cycles.json', 'w'), indent=1)
PY
rm -rf gpurun_out/r05/pmcq_*
