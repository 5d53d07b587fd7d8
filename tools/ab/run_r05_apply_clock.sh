# round 5: the shader clock k_xtb_apply runs at -- GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) / kernel duration per dispatch, for the product
# kernel (round-4 form = <.., 8> and product form = <.., 0>, widths 16 and 8) and the measurement variants 2 (no stream), 4 (no LDS), 7 (matrix instructions alone)
mkdir -p gpurun_out/r05      # (the library must have been built with DKMC_MEASURE_VARIANTS=1 python __graft_entry__.py before the gpurun call)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r05/pmcclk -- python3 $R/tools/ab/apply_clock.py > $R/gpurun_out/r05/pmcclk.out 2>&1 || echo "pmc pass failed"
cd $R
python3 - <<'PY'
import csv, glob, json, collections
dur = {}
for f in glob.glob('gpurun_out/r05/pmcclk/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[r['Dispatch_Id']] = (r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
acc = collections.defaultdict(list)
for f in glob.glob('gpurun_out/r05/pmcclk/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if r['Kernel_Name'].startswith('void k_xtb_apply') and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            name, ns = dur.get(r['Dispatch_Id'], (None, 0))
            if ns > 1000000:
                acc[r['Kernel_Name'].split('(')[0]].append((float(r['Counter_Value']), ns))
out = {}
for k, v in acc.items():
    out[k] = [{"us": round(ns / 1e3, 1), "busy_cycles_sum": c, "GHz_if_8_xcd": round(c / 8 / ns, 3), "GHz_if_1": round(c / ns, 3)} for c, ns in v]
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/r05/pmc_apply_clock.json', 'w'), indent=1)
PY
