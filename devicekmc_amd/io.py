"""Snapshot / log formats of the reference driver, host side.

* Snapshots: ``Device::writeSnapshot`` (Device.cpp:236-252) writes ``N``, a blank line, then per site
  ``element   x   y   z   potential   power`` with iostream's default 6 significant digits; ``restart = 1`` reloads such a
  file as the site list through ``read_xyz`` (utils.cpp:72-98, kmc_main.cpp:65-80: only element and xyz are read back).
  ``write_snapshot(..., full_precision=True)`` keeps 17 digits so that a restart reproduces the coordinates bit for bit.
* Restart state: the reference's restart drops everything but element and position, so a restarted run
  cannot continue the same event sequence.  ``write_restart`` / ``read_restart`` keep the snapshot in the reference's own xyz
  format (17 digits) and put what it drops into a sidecar ``<snapshot>.state.npz``: site_charge, both potentials, power,
  temperature, the warm-start vector of the current solve, T_bg, the KMC time and the position of the KMC random stream.
* Step log: the ``output.txt`` block per superstep (kmc_main.cpp:177-278): ``KMC step count``, ``V_vcm``, ``KMC time is``,
  then the result map in key order (std::map), so the reference's timing_boxplot.py / plotting scripts can read it.
"""
import numpy as np

from .params import ELEMENT_NAMES
from .structure import Structure

_NAME_TO_ELEMENT = {n: i for i, n in enumerate(ELEMENT_NAMES)}


def _g6(v: float) -> str:
    """operator<< of a double with the default precision (6 significant digits, %g)."""
    return "%g" % v


def write_snapshot(path, element, x, y, z, potential, power, full_precision=False):
    fmt = (lambda v: repr(float(v))) if full_precision else _g6
    with open(path, "w") as f:
        f.write("%d\n\n" % len(element))
        for i in range(len(element)):
            f.write("%s   %s   %s   %s   %s   %s\n" % (ELEMENT_NAMES[int(element[i])], fmt(x[i]), fmt(y[i]), fmt(z[i]),
                                                    fmt(potential[i]), fmt(power[i])))


def read_xyz(path) -> Structure:
    """utils.cpp:72-98: first line N, second line skipped, then ``element x y z [ignored...]``."""
    with open(path) as f:
        n = int(f.readline().split()[0])
        f.readline()
        el = np.empty(n, dtype=np.int32); xyz = np.empty((n, 3))
        for i in range(n):
            t = f.readline().split()
            if t[0] not in _NAME_TO_ELEMENT:
                raise ValueError("Error: Unknown element type in update_element!: " + t[0])
            el[i] = _NAME_TO_ELEMENT[t[0]]
            xyz[i] = (float(t[1]), float(t[2]), float(t[3]))
    return Structure(el, xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), {})


def write_restart(path, element, x, y, z, state):
    """Snapshot in the reference's xyz format (full precision) + sidecar with the state the format drops.
    state: dict with site_charge, site_potential_boundary, site_potential_charge, site_power, site_temperature (arrays of N),
    atom_virtual_potentials (array), T_bg, kmc_time, kmc_step_count, rnd_seed_kmc, kmc_rng_raw_draws (scalars)."""
    pot = np.asarray(state["site_potential_boundary"]) + np.asarray(state["site_potential_charge"])
    write_snapshot(path, element, x, y, z, pot, state["site_power"], full_precision=True)
    np.savez(path + ".state.npz", **{k: np.asarray(v) for k, v in state.items()})


def read_restart(path):
    """(Structure, state dict).  The Structure is what read_xyz gives (kmc_main.cpp:65-80); state is None when the sidecar is missing
    (a snapshot written by the reference itself)."""
    import os
    s = read_xyz(path)
    side = path + ".state.npz"
    if not os.path.exists(side):
        return s, None
    with np.load(side) as f:
        state = {k: (f[k].item() if f[k].ndim == 0 else f[k].copy()) for k in f.files}
    if len(state["site_charge"]) != s.N:
        raise ValueError("restart sidecar %s does not belong to %s" % (side, path))
    return s, state


def write_host_bundle(path, structure, p, Vd, element, neigh, nn, layer):
    """Raw binary input of the C++ host driver devicekmc_amd/host/kmc_superstep (the reference's parser and xyz readers are out of
    scope, so the driver takes the prepared device: header, parameters, layers, metals, site arrays, neighbour index)."""
    import struct
    N = structure.N
    N_atom = int(((element != 0) & (element != 1)).sum())
    with open(path, "wb") as f:
        f.write(struct.pack("8i", N, nn, N_atom, len(p.layers), len(p.metals), p.num_atoms_first_layer, p.num_layers_contact, int(p.pbc)))
        f.write(struct.pack("16d", Vd, p.freq, p.sigma, p.k, p.nn_dist, p.high_G, p.low_G, p.m_e, p.V0, p.background_temp,
                            p.dissipation_constant, p.t_ox, p.A, p.c_p, float(p.rnd_seed_kmc), 0.0))
        f.write(np.asarray(p.lattice, dtype=np.float64).tobytes())
        for l in p.layers:
            f.write(struct.pack("4d", l.E_gen_0, l.E_rec_1, l.E_diff_2, l.E_diff_3))
        f.write(np.asarray(p.metals, dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(element, dtype=np.int32).tobytes()); f.write(np.ascontiguousarray(layer, dtype=np.int32).tobytes())
        for a in (structure.x, structure.y, structure.z):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(neigh, dtype=np.int32).tobytes())


class StepLog:
    """Accumulates the per-superstep block of output.txt (kmc_main.cpp:177-278), byte for byte: a step opens with 14 dashes and a newline
    (:177) and closes with 38 dashes and NO newline (:278), so the next step's opening runs on in the same line -- the 52-dash lines of
    the reference's logs (structures/single_devices/timing_7.5nm/output_noguess.txt)."""

    def __init__(self):
        self.buf = []

    def bias_header(self, Vd, folder):
        self.buf.append("--------------------------------\nApplied Voltage = %s V\n--------------------------------\nCreated folder: %s\n" % (_g6(Vd), folder))

    def step(self, kmc_step_count, V_vcm, kmc_time, result_map, t_fields=None, t_log=None, t_superstep=None):
        out = ["--------------\n", "KMC step count: %d\n" % kmc_step_count, "V_vcm: %s\n" % _g6(V_vcm), "KMC time is: %s\n" % _g6(kmc_time)]
        for key in sorted(result_map):                       # std::map iterates in key order
            out.append("%s: %s\n" % (key, _g6(result_map[key])))
        if t_fields is not None:
            out.append("Z - calculation time - all fields [s]: %s\n" % _g6(t_fields))
        if t_log is not None:
            out.append("Z - calculation time - logging results [s]: %s\n" % _g6(t_log))
        if t_superstep is not None:
            out.append("Z - calculation time - KMC superstep [s]: %s\n" % _g6(t_superstep))
        out.append("--------------------------------------")
        self.buf.append("".join(out))

    def text(self):
        return "".join(self.buf)
