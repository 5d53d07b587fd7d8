#!/bin/bash
# same-box A/B of library variants: tools/ab/run.sh "<bench args>" variant1.so variant2.so ... (each variant twice, alternating)
args="$1"; shift
cp devicekmc_amd/libdevicekmc_hip.so tools/ab/orig.so
for rep in 1 2; do for v in "$@"; do
  cp tools/ab/$v devicekmc_amd/libdevicekmc_hip.so
  python bench.py $args > gpurun_out/ab_tmp.log 2>&1 || { tail -5 gpurun_out/ab_tmp.log; cp tools/ab/orig.so devicekmc_amd/libdevicekmc_hip.so; exit 1; }
  python3 -c "
import json;d=json.loads(open('gpurun_out/ab_tmp.log').read().strip().splitlines()[-1]);r=d['roofline'];print('$v rep$rep', d['value'], d['ms_per_step'], r['avg_launch_us'], r['row_kernel_us'], d['per_step']['cg_iters_X'])" | tee -a gpurun_out/ab.log
done; done
cp tools/ab/orig.so devicekmc_amd/libdevicekmc_hip.so
