"""The C-ABI library loads and exports exactly what include/devicekmc_hip.h (the drop-in surface) and include/devicekmc_hip_debug.h (test and
measurement aids, not part of the surface) declare (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(headers=("devicekmc_hip.h", "devicekmc_hip_debug.h")):
    names = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(dkmc_\w+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from devicekmc_amd import lib
    L = lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), n
    assert set(names) == set(lib.SYMBOLS), set(names) ^ set(lib.SYMBOLS)
    # the drop-in header carries no test / timing hooks
    surface = _declared(("devicekmc_hip.h",))
    assert not [n for n in surface if "debug" in n or "check" in n or "_time_" in n], surface


def test_struct_layout_matches_header():
    from devicekmc_amd import lib
    src = open(os.path.join(ROOT, "include", "devicekmc_hip.h")).read()
    body = src[src.index("typedef struct dkmc_gpubuf {"):src.index("} dkmc_gpubuf;")]
    fields = []
    for line in body.splitlines()[1:]:
        line = line.split("/*")[0].strip().rstrip(";")
        if not line:
            continue
        typ, rest = line.split(None, 1)
        for f in rest.split(","):
            fields.append(f.strip().lstrip("*"))
    assert fields == [f[0] for f in lib.dkmc_gpubuf._fields_]
    assert ctypes.sizeof(lib.dkmc_gpubuf) == 32 * 8 + 7 * 4 + 4      # 32 pointers, 7 ints, tail padding


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from devicekmc_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        lib.load()


def test_shim_library_exports_the_reference_names():
    """libdevicekmc_shim.so exports the reference's own unmangled entry points (gpu_solvers.h:36-208) on top of the C ABI."""
    import subprocess
    import __graft_entry__ as g
    g.build()
    shim = os.path.join(ROOT, "devicekmc_amd", "libdevicekmc_shim.so")
    out = subprocess.run(["nm", "-D", "--defined-only", shim], capture_output=True, text=True, check=True).stdout
    names = {l.split()[-1] for l in out.splitlines() if l.strip()}
    for n in ("get_gpu_info", "set_gpu", "copytoConstMemory", "initialize_sparsity", "update_CB_edge_gpu_sparse", "update_charge_gpu",
              "background_potential_gpu_sparse", "poisson_gridless_gpu", "solve_sparse_CG_Jacobi", "execute_kmc_step_gpu",
              "update_power_gpu_sparse", "update_power_gpu_split", "update_temperatureglobal_gpu"):
        assert n in names, n
