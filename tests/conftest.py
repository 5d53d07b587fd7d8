import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu)")


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
