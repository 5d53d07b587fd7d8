"""Host-side mirror of the reference's operator interface for the hot path.

Same names and argument meaning as the reference's C++ host classes, so the parity tests read like
calls into the reference:

  GPUBuffers            gpu_buffers.h:12-162     device-pointer bag (+ sync_HostToGPU / sync_GPUToHost ...)
  Device                Device.h:61-231          setLaplacePotential / updateCharge / updatePotential /
                                                 updatePower / updateTemperature (the USE_CUDA branches of
                                                 potential_solver.cpp, current_solver.cpp, heat_solver.cpp)
  KMCProcess            KMCProcess.h:13-43       executeKMCStep (KMCProcess.cpp:259-373, USE_CUDA branch)

Device memory is owned by torch tensors (plumbing); all computation is done by the HIP library
through the C ABI of include/devicekmc_hip.h.  Nothing here computes on the CPU.
"""
import ctypes as C
import time

import numpy as np
import torch

from . import lib as _lib
from .lib import check, dkmc_gpubuf
from .params import KMCParameters
from .rng import StdMT19937
from .structure import Structure, prepare_device


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class GPUBuffers:
    """gpu_buffers.h:12-162.  Arrays live in torch tensors on ``device``; ``self.c`` is the C struct."""

    F64 = ["site_power", "site_potential_boundary", "site_potential_charge", "site_temperature", "site_CB_edge",
           "atom_power", "atom_CB_edge", "site_x", "site_y", "site_z", "atom_x", "atom_y", "atom_z"]
    I32 = ["site_charge", "atom_charge", "site_element", "atom_element", "site_layer"]

    def __init__(self, layers, site_layer, freq, N, N_atom, site_x, site_y, site_z, nn, sigma, k, lattice, neigh_idx,
                 metals, device="cuda:0"):
        L = _lib.load()
        self.dev = torch.device(device)
        self.N_, self.N_atom_, self.nn_ = int(N), int(N_atom), int(nn)
        self.num_metal_types_ = len(metals)
        self.t = {}
        for n in self.F64:
            self.t[n] = torch.zeros(N, dtype=torch.float64, device=self.dev)
        for n in self.I32:
            self.t[n] = torch.zeros(N, dtype=torch.int32, device=self.dev)
        self.t["atom_virtual_potentials"] = torch.zeros(N_atom + 2, dtype=torch.float64, device=self.dev)
        self.t["neigh_idx"] = torch.as_tensor(np.ascontiguousarray(neigh_idx, dtype=np.int32).reshape(-1)).to(self.dev)
        self.t["site_layer"].copy_(torch.as_tensor(np.ascontiguousarray(site_layer, dtype=np.int32)))
        for n, a in (("site_x", site_x), ("site_y", site_y), ("site_z", site_z)):
            self.t[n].copy_(torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)))
        self.t["metal_types"] = torch.as_tensor(np.asarray(metals, dtype=np.int32)).to(self.dev)
        for n, v in (("sigma", [sigma]), ("k", [k]), ("freq", [freq]), ("lattice", list(lattice)), ("T_bg", [0.0])):
            self.t[n] = torch.tensor(v, dtype=torch.float64, device=self.dev)
        self.c = dkmc_gpubuf()
        for n in dkmc_gpubuf._PTRS:
            if n in self.t:
                setattr(self.c, n, self.t[n].data_ptr())
        self.c.N_, self.c.nn_, self.c.N_atom_, self.c.num_metal_types_ = self.N_, self.nn_, self.N_atom_, self.num_metal_types_
        # copytoConstMemory (gpu_buffers.h:102)
        E = [np.array([getattr(l, f) for l in layers], dtype=np.float64) for f in ("E_gen_0", "E_rec_1", "E_diff_2", "E_diff_3")]
        check(L.dkmc_copy_to_const_memory(_np_ptr(E[0]), _np_ptr(E[1]), _np_ptr(E[2]), _np_ptr(E[3]), len(layers)))
        torch.cuda.synchronize(self.dev)

    def __del__(self):
        # the pattern arrays of initialize_sparsity are the only memory of this object that torch does not own
        try:
            if self.__dict__.get("c") is not None and _lib._lib is not None:
                _lib._lib.dkmc_free_sparsity(C.byref(self.c))
        except Exception:
            pass

    def __getattr__(self, name):
        t = self.__dict__.get("t", {})
        if name in t:
            return t[name]
        raise AttributeError(name)

    # gpu_buffers.cpp:10-32
    def sync_HostToGPU(self, device):
        def put(name, arr, dt):
            self.t[name].copy_(torch.as_tensor(np.ascontiguousarray(arr, dtype=dt)))
        put("site_element", device.site_element, np.int32)
        put("site_charge", device.site_charge, np.int32)
        put("site_power", device.site_power, np.float64)
        put("site_CB_edge", device.site_CB_edge, np.float64)
        put("site_potential_boundary", device.site_potential_boundary, np.float64)
        put("site_potential_charge", device.site_potential_charge, np.float64)
        put("site_temperature", device.site_temperature, np.float64)
        self.t["T_bg"].fill_(float(device.T_bg))
        torch.cuda.synchronize(self.dev)

    # gpu_buffers.cpp:34-55
    def sync_GPUToHost(self, device):
        torch.cuda.synchronize(self.dev)
        device.site_element = self.t["site_element"].cpu().numpy()
        device.site_charge = self.t["site_charge"].cpu().numpy()
        device.site_power = self.t["site_power"].cpu().numpy()
        device.site_CB_edge = self.t["site_CB_edge"].cpu().numpy()
        device.site_potential_boundary = self.t["site_potential_boundary"].cpu().numpy()
        device.site_potential_charge = self.t["site_potential_charge"].cpu().numpy()
        device.site_temperature = self.t["site_temperature"].cpu().numpy()
        device.T_bg = float(self.t["T_bg"].item())

    def copy_power_fromGPU(self):
        return self.t["site_power"].cpu().numpy()

    def copy_charge_toGPU(self, charge):
        self.t["site_charge"].copy_(torch.as_tensor(np.ascontiguousarray(charge, dtype=np.int32)))

    def copy_Tbg_toGPU(self, T_bg):
        self.t["T_bg"].fill_(float(T_bg))


def build_neighbor_index_gpu(x, y, z, lattice, pbc, nn_dist, device="cuda:0"):
    """Device.cpp:98-136 + :69-80 on the GPU (cell list, dkmc_build_neighbor_index).  Returns (neigh int32 [N, nn], nn)."""
    L = _lib.load()
    dev = torch.device(device)
    _use_stream(dev)
    tx, ty, tz = (torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(dev) for a in (x, y, z))
    lat = np.ascontiguousarray(lattice, dtype=np.float64)
    nn = C.c_int(0)
    check(L.dkmc_build_neighbor_index(len(x), _ptr(tx), _ptr(ty), _ptr(tz), _np_ptr(lat), int(pbc), nn_dist, C.byref(nn), None))
    out = torch.empty(len(x) * nn.value, dtype=torch.int32, device=dev)
    check(L.dkmc_build_neighbor_index(len(x), _ptr(tx), _ptr(ty), _ptr(tz), _np_ptr(lat), int(pbc), nn_dist, C.byref(nn), _ptr(out)))
    return out.cpu().numpy().reshape(len(x), nn.value), nn.value


def _use_stream(dev):
    check(_lib.load().dkmc_set_stream(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))


class Device:
    """Device.h:61-231 (USE_CUDA build): host copy of the site fields + the thin callers of the GPU path."""

    def __init__(self, structure: Structure, p: KMCParameters, gpu_neighbors=None):
        """gpu_neighbors: device string (e.g. "cuda:0") to build the neighbour index with the HIP cell list instead of
        the host k-d tree (same result; minutes faster at 1e6 sites)."""
        self.p = p
        self.N = structure.N
        self.site_x, self.site_y, self.site_z = structure.x, structure.y, structure.z
        neigh = None
        if gpu_neighbors:
            neigh = build_neighbor_index_gpu(structure.x, structure.y, structure.z, p.lattice, p.pbc, p.nn_dist, gpu_neighbors)
        self.site_element, self.neigh_idx, self.max_num_neighbors, self.site_layer = prepare_device(structure, p, neigh)
        self.lattice, self.pbc, self.nn_dist, self.sigma, self.k = p.lattice, p.pbc, p.nn_dist, p.sigma, p.k
        self.T_bg = p.background_temp
        self.site_charge = np.zeros(self.N, dtype=np.int32)
        self.site_CB_edge = np.zeros(self.N)
        self.site_potential_boundary = np.zeros(self.N)
        self.site_potential_charge = np.zeros(self.N)
        self.site_power = np.zeros(self.N)
        self.site_temperature = np.full(self.N, self.T_bg)
        self.N_atom = int(((self.site_element != 0) & (self.site_element != 1)).sum())
        self.imacro = 0.0

    def make_gpubuf(self, device="cuda:0") -> GPUBuffers:
        """kmc_main.cpp:116-121: GPUBuffers ctor + sync_HostToGPU + initialize_sparsity."""
        p = self.p
        g = GPUBuffers(p.layers, self.site_layer, p.freq, self.N, self.N_atom, self.site_x, self.site_y, self.site_z,
                       self.max_num_neighbors, self.sigma, self.k, self.lattice, self.neigh_idx, list(p.metals), device)
        g.sync_HostToGPU(self)
        _use_stream(g.dev)
        _lib.load().dkmc_set_cg_tolerance(p.cg_tol)
        check(_lib.load().dkmc_initialize_sparsity(C.byref(g.c), int(p.pbc), p.nn_dist, p.num_atoms_first_layer))
        return g

    # potential_solver.cpp:4-19
    def setLaplacePotential(self, gpubuf: GPUBuffers, p: KMCParameters, Vd: float):
        n = p.num_atoms_first_layer
        gpubuf.sync_HostToGPU(self)
        _use_stream(gpubuf.dev)
        _lib.load().dkmc_set_cg_tolerance(p.cg_tol)            # the parameters are authoritative for every solve (the engine's tolerance is process-wide)
        _lib.load().dkmc_set_cb_edge_domain(1 if p.cb_edge_domain == "atoms" else 0)
        check(_lib.load().dkmc_update_CB_edge_gpu_sparse(C.byref(gpubuf.c), self.N, n, n, Vd, int(self.pbc), p.high_G, p.low_G,
                                                         self.nn_dist, len(p.metals)))
        gpubuf.sync_GPUToHost(self)

    # potential_solver.cpp:142-160
    def updateCharge(self, gpubuf: GPUBuffers, metals=None):
        t0 = time.perf_counter()
        _use_stream(gpubuf.dev)
        check(_lib.load().dkmc_update_charge_gpu(_ptr(gpubuf.site_element), _ptr(gpubuf.site_charge), _ptr(gpubuf.neigh_idx),
                                                 gpubuf.N_, gpubuf.nn_, _ptr(gpubuf.metal_types), gpubuf.num_metal_types_))
        return {"Z - calculation time - charge [s]": time.perf_counter() - t0}

    # potential_solver.cpp:232-285
    def updatePotential(self, gpubuf: GPUBuffers, p: KMCParameters, Vd: float, kmc_step_count: int = 0, sync=False):
        L = _lib.load()
        n = p.num_atoms_first_layer
        _use_stream(gpubuf.dev)
        L.dkmc_set_cg_tolerance(p.cg_tol)
        t0 = time.perf_counter()
        check(L.dkmc_background_potential_gpu_sparse(C.byref(gpubuf.c), self.N, n, n, Vd, int(self.pbc), p.high_G, p.low_G,
                                                     self.nn_dist, len(p.metals), kmc_step_count))
        if sync:
            torch.cuda.synchronize(gpubuf.dev)
        t1 = time.perf_counter()
        check(L.dkmc_poisson_gridless_gpu(p.num_atoms_contact, int(self.pbc), gpubuf.N_, _ptr(gpubuf.lattice), _ptr(gpubuf.sigma),
                                          _ptr(gpubuf.k), _ptr(gpubuf.site_x), _ptr(gpubuf.site_y), _ptr(gpubuf.site_z),
                                          _ptr(gpubuf.site_charge), _ptr(gpubuf.site_potential_charge)))
        if sync:
            torch.cuda.synchronize(gpubuf.dev)
        t2 = time.perf_counter()
        return {"Z - calculation time - potential from boundaries [s]": t1 - t0,
                "Z - calculation time - potential from charges [s]": t2 - t1}

    # current_solver.cpp:4-47 (constants :8-17)
    def updatePower(self, gpubuf: GPUBuffers, p: KMCParameters, Vd: float):
        t0 = time.perf_counter()
        _use_stream(gpubuf.dev)
        imacro = C.c_double(0.0)
        n = p.num_atoms_first_layer
        _lib.load().dkmc_set_cg_tolerance(p.cg_tol)
        check(_lib.load().dkmc_update_power_gpu_sparse(C.byref(gpubuf.c), n, n, p.num_layers_contact, Vd, int(self.pbc),
                                                       p.X_high_G, p.X_low_G, p.X_loop_G, p.G0, p.X_tol, self.nn_dist, p.m_e, p.V0,
                                                       len(p.metals), C.byref(imacro), int(p.solve_heating_local),
                                                       int(p.solve_heating_global), 1.0))
        self.imacro = imacro.value
        return {"Current [uA]": self.imacro * 1e6, "Z - calculation time - dissipated power [s]": time.perf_counter() - t0}

    # heat_solver.cpp:250-312 (global branch; computed on the device instead of copy_power_fromGPU + host sum)
    def updateTemperature(self, gpubuf: GPUBuffers, p: KMCParameters, step_time: float):
        result = {}
        if p.solve_heating_global:
            _use_stream(gpubuf.dev)
            P = C.c_double(0.0)
            check(_lib.load().dkmc_update_temperature_global_analytic(_ptr(gpubuf.site_power), _ptr(gpubuf.T_bg), gpubuf.N_, step_time,
                                                                      p.dissipation_constant, p.t_ox, p.A, p.c_p, C.byref(P)))
            self.T_bg = float(gpubuf.T_bg.item())
            result["Global temperature [K]"] = self.T_bg
            result["Total dissipated power [mW]"] = P.value * 1e3
        elif p.solve_heating_local:
            # heat_solver.cpp:286-308; the dense host inverses of the reference are replaced by sparse solves on the device
            _use_stream(gpubuf.dev)
            ns, steady, iters, T = C.c_int(0), C.c_int(0), C.c_int(0), C.c_double(0.0)
            check(_lib.load().dkmc_update_temperature_local(C.byref(gpubuf.c), step_time, p.delta_t, p.tau, p.background_temp,
                                                            p.k_th_interface, p.k_th_vacancies, self.nn_dist, p.num_atoms_contact,
                                                            C.byref(ns), C.byref(steady), C.byref(iters), C.byref(T)))
            self.T_bg = T.value
            self.last_heat_solves, self.last_heat_steady, self.last_heat_cg_iters = ns.value, bool(steady.value), iters.value
            result["Global temperature [K]"] = self.T_bg
        return result

    # heat_solver.cpp:5-37
    def get_num_in_contacts(self, num_atoms_contact: int, contact_name: str) -> int:
        el, N = self.site_element, self.N
        nondef = np.flatnonzero(el != 0)                      # positions of the non-DEFECT sites
        if contact_name == "left":
            return 0 if num_atoms_contact <= 0 else int(nondef[num_atoms_contact - 1]) + 1
        return 0 if num_atoms_contact <= 0 else N - int(nondef[len(nondef) - num_atoms_contact])

    # heat_solver.cpp:40-246 (kmc_main.cpp:90-94: once, when solve_heating_local is set)
    def constructLaplacian(self, gpubuf: GPUBuffers, p: KMCParameters):
        _use_stream(gpubuf.dev)
        N_metals = int(np.isin(self.site_element, list(p.metals)).sum())                      # Device.cpp:45-50
        self.N_left_tot = self.get_num_in_contacts(p.num_atoms_contact, "left")
        self.N_right_tot = self.get_num_in_contacts(N_metals - p.num_atoms_contact, "right")
        self.N_interface = self.N - self.N_left_tot - self.N_right_tot
        gamma = 1.0 / (p.delta * ((p.k_th_interface / p.k_th_metal) + 1.0))
        check(_lib.load().dkmc_construct_laplacian(C.byref(gpubuf.c), self.N_left_tot, self.N_right_tot, gamma))


class KMCProcess:
    """KMCProcess.h:13-43: owns the KMC random stream and the per-site layer ids."""

    def __init__(self, device: Device, freq: float):
        self.freq = freq
        self.random_generator = StdMT19937(device.p.rnd_seed_kmc)
        self.layers = device.p.layers
        self.site_layer = device.site_layer
        self.batch = 64            # events worth of random numbers handed to the device per launch
        self.last_event_log = None

    # KMCProcess.cpp:259-373 (USE_CUDA branch) -> execute_kmc_step_gpu (kmc_events.cu:146-365)
    def executeKMCStep(self, gpubuf: GPUBuffers, device: Device, want_log=False):
        L = _lib.load()
        t0 = time.perf_counter()
        _use_stream(gpubuf.dev)
        g = gpubuf
        total_events, logs, resume = 0, [], 0
        while True:
            probe = self.random_generator.copy()
            u = np.ascontiguousarray(probe.uniform_batch(2 * self.batch))
            n_ev, exhausted, et = C.c_int(0), C.c_int(0), C.c_double(0.0)
            log = np.zeros((self.batch, 4), dtype=np.int32) if want_log else None
            check(L.dkmc_execute_kmc_step_gpu(device.N, device.max_num_neighbors, _ptr(g.neigh_idx), _ptr(g.site_layer), _ptr(g.lattice),
                                              int(device.pbc), _ptr(g.T_bg), _ptr(g.freq), _ptr(g.sigma), _ptr(g.k),
                                              _ptr(g.site_x), _ptr(g.site_y), _ptr(g.site_z), _ptr(g.site_potential_boundary),
                                              _ptr(g.site_potential_charge), _ptr(g.site_temperature), _ptr(g.site_element),
                                              _ptr(g.site_charge), _np_ptr(u), len(u), resume,
                                              C.byref(n_ev), C.byref(exhausted), _np_ptr(log) if want_log else None, C.byref(et)))
            self.random_generator.skip(2 * n_ev.value)       # exactly the numbers the reference would have drawn
            total_events += n_ev.value
            if want_log:
                logs.append(log[:n_ev.value])
            if not exhausted.value:
                break
            resume = 1
        if want_log:
            self.last_event_log = np.concatenate(logs) if logs else np.zeros((0, 4), dtype=np.int32)
        self.last_n_events = total_events
        return {"Z - calculation time - kmc events [s]": time.perf_counter() - t0}, et.value


def save_restart(path, device: Device, sim: KMCProcess, gpubuf: GPUBuffers, kmc_time=0.0, kmc_step_count=0):
    """Device::writeSnapshot (Device.cpp:236-252) + the state it drops (io.write_restart): after load_restart the run continues with
    the same event sequence, bit for bit.  The start vector of the next current solve is part of that state in both modes of
    dkmc_set_current_warm_start: gpubuf.atom_virtual_potentials (mode 0 reads it) and the library's private unscaled copy of the last
    solution (mode 1, the default; dkmc_get_current_warm_vector) together with the solutions of the block-CG's auxiliary columns
    (dkmc_get_current_warm_aux)."""
    from . import io
    L = _lib.load()
    gpubuf.sync_GPUToHost(device)
    n = C.c_int(0)
    check(L.dkmc_get_current_warm_vector(C.byref(gpubuf.c), None, 0, C.byref(n)))
    warm = np.zeros(n.value)
    if n.value:
        check(L.dkmc_get_current_warm_vector(C.byref(gpubuf.c), _np_ptr(warm), n.value, C.byref(n)))
    na = C.c_longlong(0)
    check(L.dkmc_get_current_warm_aux(C.byref(gpubuf.c), None, 0, C.byref(na)))
    warm_aux = np.zeros(na.value)
    if na.value:
        check(L.dkmc_get_current_warm_aux(C.byref(gpubuf.c), _np_ptr(warm_aux), na.value, C.byref(na)))
    state = dict(site_charge=device.site_charge, site_potential_boundary=device.site_potential_boundary,
                 site_potential_charge=device.site_potential_charge, site_power=device.site_power,
                 site_temperature=device.site_temperature, site_CB_edge=device.site_CB_edge,
                 atom_virtual_potentials=gpubuf.atom_virtual_potentials.cpu().numpy(), current_warm_vector=warm, current_warm_aux=warm_aux,
                 T_bg=float(device.T_bg), kmc_time=float(kmc_time), kmc_step_count=int(kmc_step_count),
                 rnd_seed_kmc=int(sim.random_generator.seed), kmc_rng_raw_draws=int(sim.random_generator.n_raw),
                 current_warm_start=int(L.dkmc_get_current_warm_start()), N_atom_buffer=int(gpubuf.N_atom_))
    io.write_restart(path, device.site_element, device.site_x, device.site_y, device.site_z, state)


def load_restart(path, p: KMCParameters, device="cuda:0", gpu_neighbors=None):
    """kmc_main.cpp:65-80 (restart = 1: the snapshot is the site list, no substoichiometry pass) + the sidecar state.
    Returns (Device, KMCProcess, GPUBuffers, state)."""
    import copy as _copy
    from . import io
    s, state = io.read_restart(path)
    p = _copy.copy(p); p.pristine = False
    dev = Device(s, p, gpu_neighbors=gpu_neighbors)
    sim = KMCProcess(dev, p.freq)
    gb = dev.make_gpubuf(device)
    if state is not None:
        dev.site_charge = np.asarray(state["site_charge"], dtype=np.int32)
        dev.site_potential_boundary = np.asarray(state["site_potential_boundary"], dtype=np.float64)
        dev.site_potential_charge = np.asarray(state["site_potential_charge"], dtype=np.float64)
        dev.site_power = np.asarray(state["site_power"], dtype=np.float64)
        dev.site_temperature = np.asarray(state["site_temperature"], dtype=np.float64)
        dev.site_CB_edge = np.asarray(state["site_CB_edge"], dtype=np.float64)
        dev.T_bg = float(state["T_bg"])
        gb.sync_HostToGPU(dev)
        warm = np.asarray(state["current_warm_vector"], dtype=np.float64) if "current_warm_vector" in state else np.zeros(0)
        if int(state.get("current_warm_start", 0)) != 0 and "current_warm_vector" not in state:
            raise ValueError("restart sidecar was written in warm-start mode 1 by a version that did not save the private start vector: "
                             "a bit-identical continuation is not possible")
        check(_lib.load().dkmc_set_current_warm_vector(C.byref(gb.c), _np_ptr(warm) if len(warm) else None, len(warm)))
        waux = np.asarray(state["current_warm_aux"], dtype=np.float64) if "current_warm_aux" in state else np.zeros(0)
        check(_lib.load().dkmc_set_current_warm_aux(C.byref(gb.c), _np_ptr(waux) if len(waux) else None, len(waux)))
        # the start vector of the next current solve: entries [0, Na + 1) of the saved buffer are read (Na = atoms of the snapshot).  The
        # buffer of the saved run was sized for ITS initial atom count, this one for the snapshot's: the used part must fit both.
        m = np.asarray(state["atom_virtual_potentials"], dtype=np.float64)
        need = dev.N_atom + 1
        if len(m) < need or gb.atom_virtual_potentials.numel() < need:
            raise ValueError("restart sidecar holds %d virtual potentials, the snapshot needs %d" % (len(m), need))
        n = min(len(m), gb.atom_virtual_potentials.numel())
        gb.atom_virtual_potentials.zero_()
        gb.atom_virtual_potentials[:n].copy_(torch.as_tensor(m[:n]))
        sim.random_generator = StdMT19937.at_position(int(state["rnd_seed_kmc"]), int(state["kmc_rng_raw_draws"]))
    return dev, sim, gb, state


def get_stats():
    s = _lib.load().dkmc_get_stats().contents
    return {f[0]: getattr(s, f[0]) for f in s._fields_}


def get_last_X():
    """CSR of the last update_power call (dump_csr_matrix_txt twin)."""
    L = _lib.load()
    rows, nnz = C.c_int(0), C.c_longlong(0)
    check(L.dkmc_get_last_X(C.byref(rows), C.byref(nnz), None, None, None))
    rp = np.empty(rows.value + 1, dtype=np.int32); ci = np.empty(nnz.value, dtype=np.int32); data = np.empty(nnz.value)
    check(L.dkmc_get_last_X(C.byref(rows), C.byref(nnz), _np_ptr(rp), _np_ptr(ci), _np_ptr(data)))
    return rp, ci, data
