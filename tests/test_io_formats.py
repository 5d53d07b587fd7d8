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
    # the parser used for the golden fixture reads it back
    assert float(t[7].split(":")[1]) == ref_logs["timing_7.5nm/output_noguess.txt"]["steps"][0]["KMC time"]
