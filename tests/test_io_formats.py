"""Snapshot / log formats against the reference's own shipped files."""
import json
import os
import re

import numpy as np

from devicekmc_amd import io as kio


def test_snapshot_roundtrip(tmp_path, cell_2p5):
    p = str(tmp_path / "snapshot_0.xyz")
    n = 500
    pot = np.linspace(-2.5, 2.5, n); pw = np.zeros(n)
    kio.write_snapshot(p, cell_2p5.element[:n], cell_2p5.x[:n], cell_2p5.y[:n], cell_2p5.z[:n], pot, pw, full_precision=True)
    s = kio.read_xyz(p)
    assert np.array_equal(s.element, cell_2p5.element[:n])
    assert np.array_equal(s.x, cell_2p5.x[:n]) and np.array_equal(s.z, cell_2p5.z[:n])       # bit-exact restart
    kio.write_snapshot(p, cell_2p5.element[:n], cell_2p5.x[:n], cell_2p5.y[:n], cell_2p5.z[:n], pot, pw)
    lines = open(p).read().splitlines()
    assert lines[0] == str(n) and lines[1] == ""
    # default precision: the reference prints 6 significant digits, e.g. "N   -21.0212   0   0   -2.5   0"
    assert lines[2].split() == ["N", "-21.0212", "0", "0", "-2.5", "0"]


def test_step_log_matches_reference_layout(ref_logs):
    """Same block layout and key order as structures/single_devices/timing_7.5nm/output_noguess.txt."""
    log = kio.StepLog()
    log.bias_header(5.0, "Results_5.000000")
    log.step(0, 5, 2.05754e-14, {"Current [uA]": 11.8834, "Z - calculation time - charge [s]": 0.000136599,
                                 "Z - calculation time - dissipated power [s]": 3.69078,
                                 "Z - calculation time - kmc events [s]": 0.245656,
                                 "Z - calculation time - potential from boundaries [s]": 1.16237,
                                 "Z - calculation time - potential from charges [s]": 0.000513108}, t_superstep=6.43505)
    t = log.text().splitlines()
    assert t[:4] == ["--------------------------------", "Applied Voltage = 5 V", "--------------------------------", "Created folder: Results_5.000000"]
    assert t[5:9] == ["KMC step count: 0", "V_vcm: 5", "KMC time is: 2.05754e-14", "Current [uA]: 11.8834"]
    assert t[9] == "Z - calculation time - charge [s]: 0.000136599"
    assert t[13] == "Z - calculation time - potential from charges [s]: 0.000513108"
    # a step closes with 38 dashes and no newline (kmc_main.cpp:278); the next step's 14 dashes (:177) run on in the same line
    log.step(1, 5, 1.8113e-13, {"Current [uA]": 11.8869}, t_superstep=4.82018)
    t = log.text().splitlines()
    assert t[15] == "-" * 52 and t[16] == "KMC step count: 1" and log.text().endswith("-" * 38)
    # the parser used for the golden fixture reads it back
    assert float(t[7].split(":")[1]) == ref_logs["timing_7.5nm/output_noguess.txt"]["steps"][0]["KMC time"]


def test_restart_sidecar_and_rng_position(tmp_path, cell_2p5):
    """f2: the snapshot stays the reference's xyz; the sidecar brings back charge, potentials, T_bg and the KMC stream position."""
    from devicekmc_amd.rng import StdMT19937
    n = 300
    rng = np.random.default_rng(0)
    g = StdMT19937(7)
    g.uniform_batch(11); g.skip(5); g.uniform()
    state = dict(site_charge=rng.integers(-2, 3, n).astype(np.int32), site_potential_boundary=rng.standard_normal(n),
                 site_potential_charge=rng.standard_normal(n), site_power=rng.random(n), site_temperature=np.full(n, 300.0),
                 site_CB_edge=rng.standard_normal(n), atom_virtual_potentials=rng.standard_normal(n // 2),
                 T_bg=301.25, kmc_time=1.5e-12, kmc_step_count=3, rnd_seed_kmc=g.seed, kmc_rng_raw_draws=g.n_raw)
    path = str(tmp_path / "snapshot_3.xyz")
    kio.write_restart(path, cell_2p5.element[:n], cell_2p5.x[:n], cell_2p5.y[:n], cell_2p5.z[:n], state)
    s, st = kio.read_restart(path)
    assert np.array_equal(s.element, cell_2p5.element[:n]) and np.array_equal(s.y, cell_2p5.y[:n])
    for k in ("site_charge", "site_potential_boundary", "site_potential_charge", "site_power", "atom_virtual_potentials", "site_CB_edge"):
        assert np.array_equal(st[k], state[k]), k
    assert st["T_bg"] == 301.25 and st["kmc_time"] == 1.5e-12 and st["kmc_step_count"] == 3
    assert g.n_raw == 2 * (11 + 5 + 1)
    g2 = StdMT19937.at_position(st["rnd_seed_kmc"], st["kmc_rng_raw_draws"])
    assert np.array_equal(g2.uniform_batch(9), g.uniform_batch(9))
    # a snapshot of the reference itself (no sidecar) still loads
    os.remove(path + ".state.npz")
    s2, st2 = kio.read_restart(path)
    assert st2 is None and s2.N == n
