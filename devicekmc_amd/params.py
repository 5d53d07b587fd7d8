"""Simulation parameters for the KMC superstep hot path.

Mirrors the fields of the reference's ``KMCParameters`` (input_parser.h:9-126) that the hot path
reads, the derived quantities of ``set_expression_parameters`` (input_parser.cpp:391-398), the
per-call constants ``Device::updatePower`` derives (current_solver.cpp:8-17) and the compile-time
layer table (structure_input.h:8-50).  Defaults are the values of
``structures/single_devices/test_2.5nm/parameters.txt``.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

# ELEMENT enum, utils.h:37-44
DEFECT, OXYGEN_DEFECT, VACANCY, O_EL, Hf_EL, Ni_EL, Ti_EL, Pt_EL, N_EL, NULL_ELEMENT = range(10)
ELEMENT_NAMES = ["d", "Od", "V", "O", "Hf", "Ni", "Ti", "Pt", "N", "NULL"]
# EVENTTYPE enum, utils.h:53-60
VACANCY_GENERATION, VACANCY_RECOMBINATION, VACANCY_DIFFUSION, ION_DIFFUSION, NULL_EVENT = range(5)


@dataclass
class Layer:
    """structure_input.h:12-50 / utils.h:63-72"""
    type: str
    E_gen_0: float
    E_rec_1: float
    E_diff_2: float
    E_diff_3: float
    start_x: float
    end_x: float


def default_layers() -> List[Layer]:
    return [
        Layer("contact", 0.0, 0.0, 0.0, 0.76, -22.0, 0.0),
        Layer("interface", 3.93, 0.0, 1.09, 0.76, 0.0, 3.0),
        Layer("oxide", 3.93, 0.0, 1.09, 0.76, 3.0, 48.1431),
        Layer("interface", 1.66, 0.0, 1.09, 0.76, 48.1431, 52.643100),
        Layer("contact", 1.73, 0.0, 0.0, 2.8, 52.643100, 90.0),
    ]


@dataclass
class KMCParameters:
    rnd_seed: int = 4                       # Device RNG (substoichiometry), parameters.txt
    rnd_seed_kmc: int = 1                   # KMC RNG, structure_input.h:8
    lattice: Tuple[float, float, float] = (108.975570, 25.575000, 25.575000)
    pristine: bool = True
    initial_vacancy_concentration: float = 0.05
    freq: float = 10e13                     # attempt_frequency
    nn_dist: float = 3.5
    pbc: bool = False
    num_atoms_first_layer: int = 144
    num_layers_contact: int = 10
    num_atoms_contact: int = 144
    metals: Tuple[int, ...] = (Ti_EL, N_EL)
    solve_potential: bool = True
    solve_current: bool = True
    solve_heating_global: bool = False
    solve_heating_local: bool = False
    perturb_structure: bool = True
    G_coeff: float = 1.0
    sigma: float = 3.5e-10
    epsilon: float = 23.0
    m_r: float = 0.85
    V0: float = 1.6
    background_temp: float = 300.0
    t_ox: float = 52.6838e-10
    A: float = 25.575000e-10 * 25.575000e-10
    c_p: float = 1.92
    dissipation_constant: float = 1e-13
    small_step: float = 1e-17
    Rs: float = 1e-16
    q: float = 1.60217663e-19
    m_0: float = 9.11e-31
    layers: List[Layer] = field(default_factory=default_layers)
    cg_tol: float = 1e-6                    # iterative_solvers_gpu.cu:322
    # Domain of the CB-edge system (tools/pin_current_constants.py).  "sites" = update_CB_edge_gpu_sparse as in the
    # snapshot's source (potential_solver_gpu.cu:595-694: every site, interstitials included); "atoms" = interstitial sites (DEFECT /
    # OXYGEN_DEFECT) carry no link -- the revision that wrote the reference's CSR dump and its current log (`log_revision()`).
    cb_edge_domain: str = "sites"
    # local temperature model (parameters.txt:76-91)
    k_th_metal: float = 29.0
    k_th_non_vacancy: float = 0.5
    k_th_vacancies: float = 5.0
    delta_t: float = 1e-13
    delta: float = 1.0
    power_adjustment_term: float = 1.0
    L_char: float = 3.5e-10

    # input_parser.cpp:391-398
    @property
    def high_G(self) -> float:
        return self.G_coeff * 1

    @property
    def low_G(self) -> float:
        return self.G_coeff * 1e-8

    @property
    def k(self) -> float:
        return 8.987552e9 / self.epsilon

    @property
    def m_e(self) -> float:
        return self.m_r * self.m_0

    @property
    def k_th_interface(self) -> float:       # input_parser.cpp:395
        return self.k_th_non_vacancy + (self.k_th_vacancies - self.k_th_non_vacancy) * self.initial_vacancy_concentration

    @property
    def tau(self) -> float:                  # input_parser.cpp:396
        return self.k_th_interface / (self.L_char * self.L_char * self.c_p * 1e6)

    # current_solver.cpp:8-17
    @property
    def X_loop_G(self) -> float:
        return self.high_G * 10000000

    @property
    def X_high_G(self) -> float:
        return self.high_G * 100000

    @property
    def X_low_G(self) -> float:
        return self.low_G

    @property
    def G0(self) -> float:
        return 2 * 3.8612e-5 * 1e-5

    @property
    def X_tol(self) -> float:
        return self.q * 0.01

    def log_revision(self) -> "KMCParameters":
        """The settings under which the reference's own artefacts are reproduced: CG tolerance 1e-12 ("used to be 1e-12",
        iterative_solvers_gpu.cu:322) and the CB edge solved on atoms only."""
        import copy
        p = copy.deepcopy(self)
        p.cg_tol = 1e-12
        p.cb_edge_domain = "atoms"
        return p

    def for_tiling(self, k: int) -> "KMCParameters":
        """Parameters for the shipped 2.5 nm cell tiled k x k laterally."""
        import copy
        p = copy.deepcopy(self)
        p.lattice = (self.lattice[0], self.lattice[1] * k, self.lattice[2] * k)
        p.num_atoms_first_layer = self.num_atoms_first_layer * k * k
        p.num_atoms_contact = self.num_atoms_contact * k * k
        p.A = self.A * k * k
        return p
