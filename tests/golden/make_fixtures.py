#!/usr/bin/env python3
"""Build the committed golden fixtures from DATA files the reference ships.

Run in the build container only (it reads /root/reference, which does not exist on
the GPU box).  Everything written here is data: site lists (element code + xyz),
a CSR sparsity dump and numbers printed by the reference's own runs.  No reference
source text is copied.

Sources (all under /root/reference/structures):
  single_devices/test_2.5nm/reordered_device_2.5.xyz          9 399 sites  (configs[0])
  single_devices/timing_7.5nm/reordered_device_7.5.xyz        85 071 sites (configs[1])
  single_devices/timing_2.5nm/fullmatrix_assembly/csr*_step#0 X sparsity dumped by the
        reference CUDA path (dump_csr_matrix_txt, iterative_solvers_gpu.cu:142-169)
  single_devices/timing_7.5nm/output_noguess.txt              19 logged supersteps of the
        reference CUDA path (KMC time, Current [uA])
  crossbars/timing_10nm_5pitch/{reordered_crossbar_10_5_initial.xyz,output_initial.txt}
        110 813 sites, 13 logged supersteps (KMC time only, solve_current = 0)

Element codes follow the reference's ELEMENT enum (utils.h:37-44).
"""
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference/structures"
OUT = os.path.dirname(os.path.abspath(__file__))

ELEMENT = {"d": 0, "Od": 1, "V": 2, "O": 3, "Hf": 4, "Ni": 5, "Ti": 6, "Pt": 7, "N": 8}


def read_xyz(path):
    with open(path) as f:
        n = int(f.readline().split()[0])
        f.readline()
        elem = np.empty(n, dtype=np.int8)
        xyz = np.empty((n, 3), dtype=np.float64)
        for i in range(n):
            t = f.readline().split()
            elem[i] = ELEMENT[t[0]]
            xyz[i] = (float(t[1]), float(t[2]), float(t[3]))
    return elem, xyz


def save_structure(src, dst, meta):
    elem, xyz = read_xyz(os.path.join(REF, src))
    np.savez_compressed(os.path.join(OUT, dst), element=elem, xyz=xyz,
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print(dst, len(elem), "sites")


def parse_log(path, keys):
    steps = []
    cur = None
    with open(path) as f:
        for line in f:
            m = re.match(r"KMC step count: (\d+)", line)
            if m:
                cur = {"step": int(m.group(1))}
                steps.append(cur)
                continue
            for k in keys:
                if line.startswith(k + ":") or line.startswith(k + " is:"):
                    cur[k] = float(line.split(":")[1])
    return steps


def main():
    save_structure("single_devices/test_2.5nm/reordered_device_2.5.xyz", "device_2.5nm.npz",
                   dict(lattice=[108.975570, 25.575000, 25.575000], num_atoms_first_layer=144,
                        num_layers_contact=10, num_atoms_contact=144, rnd_seed=4))
    save_structure("single_devices/timing_7.5nm/reordered_device_7.5.xyz", "device_7.5nm.npz",
                   dict(lattice=[108.984050, 76.725000, 76.725000], num_atoms_first_layer=1296,
                        num_layers_contact=10, num_atoms_contact=12960, rnd_seed=5))
    save_structure("crossbars/timing_10nm_5pitch/reordered_crossbar_10_5_initial.xyz",
                   "crossbar_10nm_5pitch.npz",
                   dict(lattice=[108.98, 102.3, 102.3], num_atoms_first_layer=144,
                        num_layers_contact=10, num_atoms_contact=11520, rnd_seed=5))

    d = os.path.join(REF, "single_devices/timing_2.5nm/fullmatrix_assembly")
    row_ptr = np.loadtxt(os.path.join(d, "csrRowPtr_step#0.txt"), dtype=np.int64).astype(np.int32)
    col = np.loadtxt(os.path.join(d, "csrColIndices_step#0.txt"), dtype=np.int64).astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "x_pattern_2.5nm_step0.npz"), row_ptr=row_ptr, col_idx=col)
    print("x pattern", len(row_ptr) - 1, "rows", len(col), "nnz")

    logs = {
        "timing_7.5nm/output_noguess.txt": {
            "note": "reference CUDA path, 85 071 sites, V=5, rnd_seed=5, solve_current=1, heating off",
            "steps": parse_log(os.path.join(REF, "single_devices/timing_7.5nm/output_noguess.txt"),
                               ["KMC time", "Current [uA]"]),
        },
        "crossbars/timing_10nm_5pitch/output_initial.txt": {
            "note": "reference CUDA path, 110 813 sites, V=1, rnd_seed=5, solve_current=0",
            "steps": parse_log(os.path.join(REF, "crossbars/timing_10nm_5pitch/output_initial.txt"),
                               ["KMC time"]),
        },
        "BASELINE.md#2 (reference CPU path run during the survey)": {
            "note": "test_2.5nm, 9 399 sites, V=5, rnd_seed=4, dense-LU CPU path; 6 significant digits as printed",
            "steps": [
                {"step": 0, "Current [uA]": 0.607957, "KMC time": 6.3193e-14,
                 "Charged vacancies": 88, "Uncharged vacancies": 12},
                {"step": 1, "Current [uA]": 0.616689, "KMC time": 1.12225e-10},
            ],
        },
    }
    with open(os.path.join(OUT, "reference_logs.json"), "w") as f:
        json.dump(logs, f, indent=1)
    print("logs written")


if __name__ == "__main__":
    sys.exit(main())
