import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu)")


@pytest.fixture(autouse=True)
def _library_defaults():
    """Every test starts from the library's default switches (a test that leaves one flipped must not change what the next one checks).
    Only when the HIP library is already in the process: CPU tests never load it here."""
    from devicekmc_amd import lib
    L = lib._lib
    if L is not None:
        L.dkmc_set_current_warm_start(1); L.dkmc_set_x_block(16); L.dkmc_set_x_format(1); L.dkmc_set_x_aux(2)
        L.dkmc_set_cg_tolerance(1e-6); L.dkmc_set_cb_edge_domain(0); L.dkmc_set_tcache_budget(-1); L.dkmc_set_pair_cutoff(6.5)
        L.dkmc_set_k_blocked(1); L.dkmc_set_profiling(0); L.dkmc_set_x_aux_warm(0); L.dkmc_set_x_slab(1); L.dkmc_set_k_slab(1)
        L.dkmc_set_x_apply_form(0); L.dkmc_set_x_items(0); L.dkmc_set_x_poly(8)
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def cell_2p5():
    from devicekmc_amd import structure
    return structure.load_structure(os.path.join(GOLDEN, "device_2.5nm.npz"))


@pytest.fixture(scope="session")
def dev_7p5():
    from devicekmc_amd import structure
    return structure.load_structure(os.path.join(GOLDEN, "device_7.5nm.npz"))


@pytest.fixture(scope="session")
def ref_logs():
    import json
    with open(os.path.join(GOLDEN, "reference_logs.json")) as f:
        return json.load(f)


def params_7p5():
    from devicekmc_amd import params
    return params.KMCParameters(rnd_seed=5, lattice=(108.984050, 76.725000, 76.725000), num_atoms_first_layer=1296,
                                num_atoms_contact=12960, A=76.725e-10 * 76.725e-10)
