"""Host-side construction of the inputs the hot path consumes.

This mirrors what the reference's ``Device`` / ``KMCProcess`` constructors hand to ``GPUBuffers``
(Device.cpp:17-96, KMCProcess.cpp:17-64): site list in the order left contact, oxide, interstitials,
right contact; padded neighbour index (ascending j per row, -1 padding, ``nn`` = global maximum);
layer id per site; initial vacancies.  It is setup code, outside the per-step path.
"""
import json
from dataclasses import dataclass

import numpy as np

from .params import DEFECT, OXYGEN_DEFECT, O_EL, VACANCY, KMCParameters
from .rng import StdMT19937


@dataclass
class Structure:
    element: np.ndarray     # int32 [N]   ELEMENT enum
    x: np.ndarray           # float64 [N]
    y: np.ndarray
    z: np.ndarray
    meta: dict

    @property
    def N(self) -> int:
        return int(self.element.shape[0])


def load_structure(path: str) -> Structure:
    d = np.load(path)
    meta = json.loads(bytes(d["meta"]).decode()) if "meta" in d.files else {}
    xyz = d["xyz"]
    return Structure(d["element"].astype(np.int32), xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy(), meta)


def tile_structure(s: Structure, k: int, period_y: float, period_z: float, n_contact_total: int) -> Structure:
    """Tile the cell k x k in y,z keeping the reference's site order:
    left contact . oxide . interstitials . right contact (reorder_boundary.py:113-122).
    ``n_contact_total`` = number of sites in each contact block of the single cell."""
    if k == 1:
        return s
    N = s.N
    is_int = s.element == DEFECT
    n_int = int(is_int.sum())
    first_int = int(np.argmax(is_int))
    assert is_int[first_int:first_int + n_int].all(), "interstitials must be contiguous"
    blocks = [(0, n_contact_total), (n_contact_total, first_int), (first_int, first_int + n_int), (first_int + n_int, N)]
    el, xs, ys, zs = [], [], [], []
    for lo, hi in blocks:
        for a in range(k):
            for b in range(k):
                el.append(s.element[lo:hi])
                xs.append(s.x[lo:hi])
                ys.append(s.y[lo:hi] + a * period_y)
                zs.append(s.z[lo:hi] + b * period_z)
    meta = dict(s.meta)
    meta["tiling"] = k
    return Structure(np.concatenate(el), np.concatenate(xs), np.concatenate(ys), np.concatenate(zs), meta)


def site_distance(x1, y1, z1, x2, y2, z2, lattice, pbc):
    """gpu_solvers.h:225-257 (site_dist_gpu), vectorised."""
    if pbc:
        dx = x1 - x2
        fy = (y1 - y2) / lattice[1]
        fy = fy - np.round(fy)
        fz = (z1 - z2) / lattice[2]
        fz = fz - np.round(fz)
        dy, dz = fy * lattice[1], fz * lattice[2]
        return np.sqrt(dx * dx + dy * dy + dz * dz)
    dx, dy, dz = x2 - x1, y2 - y1, z2 - z1
    return np.sqrt(dx * dx + dy * dy + dz * dz)


def build_neighbor_index(s: Structure, lattice, pbc: bool, nn_dist: float):
    """Padded neighbour index equal to Device.cpp:98-136 + :69-80, built in O(N log N).
    Returns (neigh_idx int32 [N, nn], nn)."""
    from scipy.spatial import cKDTree

    N = s.N
    pts = np.stack([s.x, s.y, s.z], axis=1)
    if pbc:
        big = 10.0 * (np.ptp(s.x) + 1.0) + 100.0
        w = np.stack([s.x - s.x.min(), np.mod(s.y, lattice[1]), np.mod(s.z, lattice[2])], axis=1)
        tree = cKDTree(w, boxsize=[big, lattice[1], lattice[2]])
    else:
        tree = cKDTree(pts)
    pairs = tree.query_pairs(nn_dist * (1 + 1e-9) + 1e-9, output_type="ndarray")
    i, j = pairs[:, 0], pairs[:, 1]
    d = site_distance(s.x[i], s.y[i], s.z[i], s.x[j], s.y[j], s.z[j], lattice, pbc)
    keep = d < nn_dist
    i, j = i[keep], j[keep]
    src = np.concatenate([i, j])
    dst = np.concatenate([j, i])
    order = np.lexsort((dst, src))
    src, dst = src[order], dst[order]
    counts = np.bincount(src, minlength=N)
    nn = int(counts.max())
    start = np.concatenate([[0], np.cumsum(counts)[:-1]])
    slot = np.arange(src.shape[0]) - start[src]
    neigh = np.full((N, nn), -1, dtype=np.int32)
    neigh[src, slot] = dst
    return neigh, nn


def site_layers(x: np.ndarray, layers) -> np.ndarray:
    """KMCProcess.cpp:34-50: the last layer whose [start_x, end_x] contains the site."""
    layer = np.full(x.shape[0], -1, dtype=np.int32)
    for j, L in enumerate(layers):
        inside = (L.start_x <= x) & (x <= L.end_x)
        layer[inside] = j
    if (layer < 0).any():
        raise ValueError("Site #%d is not inside the device!" % int(np.argmax(layer < 0)))
    return layer


def make_substoichiometric(element: np.ndarray, concentration: float, rng: StdMT19937) -> int:
    """Device.cpp:202-233: convert int(c * #O) oxygen atoms, drawing positions over the atom list;
    consumes exactly the random numbers the reference would."""
    atom_ind = np.nonzero((element != DEFECT) & (element != OXYGEN_DEFECT))[0]
    n_atom = atom_ind.shape[0]
    num_O = int((element == O_EL).sum())
    todo = int(concentration * num_O)
    added = todo
    while todo > 0:
        probe = rng.copy()
        u = probe.uniform_batch(max(64, 4 * todo))
        used = 0
        for r in u:
            used += 1
            loc = int(r * n_atom)
            if element[atom_ind[loc]] == O_EL:
                element[atom_ind[loc]] = VACANCY
                todo -= 1
                if todo == 0:
                    break
        rng.skip(used)
    return added


def prepare_device(s: Structure, p: KMCParameters, neigh_nn=None):
    """Everything GPUBuffers is constructed from (kmc_main.cpp:88-121), as numpy arrays.
    neigh_nn: optional precomputed (neigh_idx, nn), e.g. from the HIP cell-list builder."""
    element = s.element.copy()
    if p.pristine:
        rng = StdMT19937(p.rnd_seed)
        make_substoichiometric(element, p.initial_vacancy_concentration, rng)
    neigh, nn = neigh_nn if neigh_nn is not None else build_neighbor_index(s, p.lattice, p.pbc, p.nn_dist)
    layer = site_layers(s.x, p.layers)
    return element, neigh, nn, layer
