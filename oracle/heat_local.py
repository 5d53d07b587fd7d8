"""CPU restatement of the reference's LOCAL temperature model (dense, numpy).

TEST INFRASTRUCTURE ONLY (see oracle/README.md): the checker for tests/test_gpu_parity.py, never the product path.

PARITY UNPINNED: the reference ships no fixture, log line or printed number for this model (every shipped parameters.txt
has solve_heating_local = 0) and its sources cannot be built here.  The functions below restate
heat_solver.cpp line by line, dense inverse included, and are what the HIP path (csrc/heat.hip, sparse CG) is compared to.

Functions follow, in order:
  get_num_in_contacts                    heat_solver.cpp:5-37
  construct_laplacian                    heat_solver.cpp:40-246
  update_local_temperature               heat_solver.cpp:354-437
  update_local_temperature_steady_state  heat_solver.cpp:441-513
  update_temperature_local               heat_solver.cpp:286-308 (the local branch of Device::updateTemperature)
"""
import numpy as np

DEFECT, VACANCY = 0, 2          # utils.h:37-44
T_1 = 50.0                      # Device.h:117


def get_num_in_contacts(site_element, num_atoms_contact, contact_name):
    """heat_solver.cpp:5-37: number of SITES spanned by the first / last `num_atoms_contact` non-DEFECT sites."""
    N = len(site_element)
    if contact_name == "left":
        i = count = 0
        while i < num_atoms_contact:
            if site_element[count] != DEFECT:
                i += 1
            count += 1
        return count
    i = count = N
    while i > N - num_atoms_contact:
        if site_element[count - 1] != DEFECT:
            i -= 1
        count -= 1
    return N - count


class LocalHeatOracle:
    def __init__(self, site_element, neigh_idx, metals, num_atoms_contact, nn_dist, delta, delta_t, tau, k_th_interface, k_th_metal):
        """construct_laplacian, heat_solver.cpp:40-246.  neigh_idx: padded neighbour index (-1 = no neighbour)."""
        el = np.asarray(site_element)
        N = len(el)
        self.N, self.nn_dist = N, nn_dist
        is_metal = np.isin(el, list(metals))
        N_metals = int(is_metal.sum())                                                  # Device.cpp:45-50
        self.N_left_tot = get_num_in_contacts(el, num_atoms_contact, "left")            # :44
        self.N_right_tot = get_num_in_contacts(el, N_metals - num_atoms_contact, "right")   # :45
        self.N_interface = Ni = N - self.N_left_tot - self.N_right_tot                  # :46
        gamma = 1.0 / (delta * ((k_th_interface / k_th_metal) + 1.0))                   # :86
        step_time = delta_t * tau                                                       # :87
        self.gamma = gamma
        idx = np.full(N, -1, dtype=np.int64)                                            # :90-103
        idx[self.N_left_tot:N - self.N_right_tot] = np.arange(Ni)
        self.index_mapping = idx
        L = np.zeros((Ni, Ni))
        for i in range(self.N_left_tot, N - self.N_right_tot):                          # :106-139
            ii = idx[i]
            for j in neigh_idx[i]:
                if j < 0:
                    continue
                jj = idx[j]
                if i != j and jj != -1:
                    L[ii, jj] = 1.0
                if is_metal[j]:
                    L[ii, ii] = -gamma                                                  # boundary atom iff connected to a metallic site
        offsum = L.sum(axis=1) - np.diag(L)                                             # :142-153: L[i,i] += -sum_{j != i} L[i,j]
        L[np.diag_indices(Ni)] += -offsum
        self.L = L
        self.laplacian = np.linalg.inv(np.eye(Ni) - step_time * L)                      # :156-170, :184/:190-198
        self.laplacian_ss = np.linalg.inv(L)                                            # :172-181, :185/:206-216

    def _p_transfer(self, site_element, background_temp, k_th_interface, k_th_vacancies):
        # :369-370 / :455-456 (the names are the reference's: vacancies use k_th_interface)
        pv = 1.0 / ((self.nn_dist * 1e-10 * k_th_interface) * (T_1 - background_temp))
        pn = 1.0 / ((self.nn_dist * 1e-10 * k_th_vacancies) * (T_1 - background_temp))
        el = np.asarray(site_element)[self.N_left_tot:self.N - self.N_right_tot]
        return np.where(el == VACANCY, pv, pn)

    def _finish(self, site_temperature, num_atoms_contact):
        # :423-432 / :499-508
        return float(site_temperature[num_atoms_contact:self.N - num_atoms_contact].sum() / (self.N - 2 * num_atoms_contact))

    def update_local_temperature(self, site_temperature, site_power, site_element, background_temp, t, tau,
                                 k_th_interface, k_th_vacancies, num_atoms_contact):
        """heat_solver.cpp:354-437; site_temperature is updated in place, returns T_bg."""
        lo, hi = self.N_left_tot, self.N - self.N_right_tot
        T_0 = background_temp
        step_time = t * tau
        c = self._p_transfer(site_element, background_temp, k_th_interface, k_th_vacancies)
        T_vec = (site_temperature[lo:hi] - T_0) / (T_1 - T_0)                           # :373-384
        T_transf = self.laplacian @ (T_vec + site_power[lo:hi] * c * step_time)         # :387-420
        site_temperature[lo:hi] = T_transf * (T_1 - T_0) + T_0
        return self._finish(site_temperature, num_atoms_contact)

    def update_local_temperature_steady_state(self, site_temperature, site_power, site_element, background_temp,
                                              k_th_interface, k_th_vacancies, num_atoms_contact):
        """heat_solver.cpp:441-513."""
        lo, hi = self.N_left_tot, self.N - self.N_right_tot
        T_0 = background_temp
        c = self._p_transfer(site_element, background_temp, k_th_interface, k_th_vacancies)
        T_transf = self.laplacian_ss @ (site_power[lo:hi] * c)                          # :459-492
        site_temperature[lo:hi] = -T_transf * (T_1 - T_0) + T_0
        return self._finish(site_temperature, num_atoms_contact)

    def update_temperature_local(self, site_temperature, site_power, site_element, step_time, background_temp, delta_t, tau,
                                 k_th_interface, k_th_vacancies, num_atoms_contact):
        """heat_solver.cpp:286-308.  Returns (T_bg, number of solves, steady-state flag)."""
        if step_time > 1e3 * delta_t:
            return self.update_local_temperature_steady_state(site_temperature, site_power, site_element, background_temp,
                                                              k_th_interface, k_th_vacancies, num_atoms_contact), 1, 1
        n = 0
        T_bg = None
        for _ in range(int(step_time / delta_t) + 1):                                  # i = 0 .. int(step_time / delta_t), inclusive
            T_bg = self.update_local_temperature(site_temperature, site_power, site_element, background_temp, delta_t, tau,
                                                 k_th_interface, k_th_vacancies, num_atoms_contact)
            n += 1
        return T_bg, n, 0
