mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r04/prof_kl
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/prof_kl --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ab/kcg_levels.py > $GRAFT_REPO_ROOT/gpurun_out/r04/prof_kl.out 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r04/prof_kl/*/*kernel_trace.csv | head -1); python3 - "$f" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'][:28],int(r['Grid_Size_X']) if 'Grid_Size_X' in r else 0) for r in rows if 'k_k' in r['Kernel_Name']]
ks.sort()
# group durations by (name, grid)
d=collections.defaultdict(list)
for a,b,n,gx in ks: d[(n,gx)].append((b-a)/1e3)
for k,v in sorted(d.items()):
    if len(v)>50: print(k, len(v), 'median %.1f us'%sorted(v)[len(v)//2], 'min %.1f'%min(v))
PY
