#!/bin/bash
# same-box A/B of library variants on the K solve: tools/ab/run_k.sh variant1.so variant2.so ...
cp devicekmc_amd/libdevicekmc_hip.so tools/ab/orig.so
for rep in 1 2; do for v in "$@"; do
  cp tools/ab/$v devicekmc_amd/libdevicekmc_hip.so
  python bench.py --workload tile:10 --scale-points none --no-pmc --no-cpp-host --no-alt --no-cpu-baseline --steps 1 --warmup 0 > gpurun_out/ab_tmp.log 2>&1 || { tail -5 gpurun_out/ab_tmp.log; cp tools/ab/orig.so devicekmc_amd/libdevicekmc_hip.so; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_tmp.log').read().strip().splitlines()[-1]);r=d['roofline_K_cg'];print('$v rep$rep', d['split_ms']['potential'], r['us_per_iteration'], r['iterations_timed'], d['ms_per_step'])" | tee -a gpurun_out/ab.log
done; done
cp tools/ab/orig.so devicekmc_amd/libdevicekmc_hip.so
