// events.hip -- KMC event table and residence-time event loop.
// Replaces execute_kmc_step_gpu / build_event_list / zero_out_events (kmc_events.cu:34-365).
//
// The reference runs, per executed event, a full thrust::inclusive_scan over N*nn rates, a binary
// search, ~11 one-element memcpys, a full zero_out_events pass and 4 memsets, with a host round trip
// between each.  Here the table is built once per step together with a 4-level sum
// tree (slot -> site row -> 64 rows -> 4096 rows -> total), and ONE resident workgroup then runs the
// whole event loop on the device: tree descent to pick the slot, execution, invalidation of the
// O(nn^2) slots that touch the two sites, and repair of the O(nn) tree nodes above them.  Every tree
// node is a fixed-order sum of its children, so selection is run-to-run deterministic; it equals the
// sequential-prefix definition of the host engine (utils.h:91-99, KMCProcess.cpp:303-311) except for
// draws that land within rounding distance of a bucket edge.
#include "common.h"

struct LayerEnergies { double gen[DKMC_MAX_LAYERS], rec[DKMC_MAX_LAYERS], vdiff[DKMC_MAX_LAYERS], odiff[DKMC_MAX_LAYERS]; };
// per-layer zero-field energies in constant memory, like the reference (kmc_events.cu:10-13); a kernel-argument struct
// indexed by layer[j] would be demoted to scratch memory
__constant__ LayerEnergies c_layerE;

struct EvOut {                 // device -> host mailbox
    double event_time, psum_last;
    int n_events, exhausted, bad, n_charged;
};

// rate of slot (i, j): kmc_events.cu:52-122.  Returns the event type, P through *prob.
__device__ __forceinline__ int slot_rate(int i, int j, int N, const int *__restrict__ layer, double laty, double latz, int pbc,
                                         double T_bg, double freq, double sigma, double kk,
                                         const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                         const double *__restrict__ pb, const double *__restrict__ pc,
                                         const int *__restrict__ element, const int *__restrict__ charge,
                                         int ei, int qi, double xi, double yi, double zi, double phii,
                                         double *prob)
{
    int type = EV_NULL; double P = 0.0;
    if (j >= 0 && j < N) {
        const int ej = element[j];
        const bool gen = (ei == DEFECT && ej == O_EL), rec = (ei == OXYGEN_DEFECT && ej == VACANCY);
        const bool vdf = (ei == VACANCY && ej == O_EL), idf = (ei == OXYGEN_DEFECT && ej == DEFECT);
        if (gen || rec || vdf || idf) {
            const double dist = 1e-10 * site_dist(xi, yi, zi, x[j], y[j], z[j], laty, latz, pbc);
            const double dphi = phii - (pb[j] + pc[j]);
            const int qj = charge[j], lj = layer[j];
            double E0, En;
            if (gen) { En = 2 * dphi; E0 = c_layerE.gen[lj]; type = EV_GEN; }
            else if (rec) {
                const double self = v_solve(dist, 2, sigma, kk);
                const int cs = qi - qj;
                En = cs * (dphi + (cs / 2) * self);            // integer division (kmc_events.cu:77)
                E0 = c_layerE.rec[lj]; type = EV_REC;
            } else if (vdf) {
                const double self = (qi != 0) ? v_solve(dist, qi, sigma, kk) : 0.0;
                En = (qi - qj) * (dphi + self);
                E0 = c_layerE.vdiff[lj]; type = EV_VDIFF;             // layer of j (kmc_events.cu:98)
            } else {
                const double self = (qi != 0) ? v_solve(dist, 2, sigma, kk) : 0.0;
                En = (qi - qj) * (dphi - self);
                E0 = c_layerE.odiff[lj]; type = EV_IDIFF;
            }
            const double EA = E0 - En - 0;
            P = exp(-1 * EA / (DKMC_KB * T_bg)) * freq;
        }
    }
    *prob = P;
    return type;
}

// fixed-order sum of one row of the table (all lanes get the result)
__device__ __forceinline__ double row_sum(const double *__restrict__ row, int nn, int lane)
{
    double s = 0.0;
    for (int c = lane; c < nn; c += WAVE) s += row[c];
    return wave_sum_all(s);
}
// fixed-order sum of up to 64 consecutive entries starting at base (entries >= n count as 0)
__device__ __forceinline__ double group_sum(const double *__restrict__ v, int base, int n, int lane)
{
    const int k = base + lane;
    return wave_sum_all(k < n ? v[k] : 0.0);
}

// one wave per site row: rates of its nn slots + row sum
__global__ __launch_bounds__(256) void k_ev_build(int N, int nn, const int *__restrict__ neigh, const int *__restrict__ layer,
                                                  const double *__restrict__ lattice, int pbc, const double *__restrict__ T_bg_p,
                                                  const double *__restrict__ freq_p, const double *__restrict__ sigma_p,
                                                  const double *__restrict__ k_p, const double *__restrict__ x,
                                                  const double *__restrict__ y, const double *__restrict__ z,
                                                  const double *__restrict__ pb, const double *__restrict__ pc,
                                                  const int *__restrict__ element, const int *__restrict__ charge,
                                                  int *__restrict__ ev_type, double *__restrict__ ev_prob, double *__restrict__ rowsum)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const double laty = lattice[1], latz = lattice[2], T_bg = *T_bg_p, freq = *freq_p, sigma = *sigma_p, kk = *k_p;
    const int ei = element[i], qi = charge[i];
    const double xi = x[i], yi = y[i], zi = z[i], phii = pb[i] + pc[i];
    double s = 0.0;
    for (int c = lane; c < nn; c += WAVE) {
        const size_t idx = (size_t)i * nn + c;
        double P;
        const int type = slot_rate(i, neigh[idx], N, layer, laty, latz, pbc, T_bg, freq, sigma, kk, x, y, z, pb, pc, element, charge,
                                   ei, qi, xi, yi, zi, phii, &P);
        ev_prob[idx] = P;
        if (ev_type) ev_type[idx] = type;
        s += P;
    }
    s = wave_sum_all(s);
    if (lane == 0 && rowsum) rowsum[i] = s;
}

// one wave per tree node: out[k] = sum of in[64k .. 64k+63]
__global__ __launch_bounds__(256) void k_ev_level(int n_in, const double *__restrict__ in, int n_out, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= n_out) return;
    const double s = group_sum(in, k * 64, n_in, lane);
    if (lane == 0) out[k] = s;
}

// wave-level search: first entry (index order) of v[base .. base+count) whose inclusive prefix exceeds
// `target`; returns its index relative to base (or the last positive entry if rounding leaves none) and
// the prefix before it through *before.  All lanes of the wave must call; result is wave-uniform.
__device__ __forceinline__ int wave_pick(const double *__restrict__ v, int base, int count, int limit, double target, int lane, double *before)
{
    double carry = 0.0; int last_pos = -1; double last_before = 0.0;
    for (int c0 = 0; c0 < count; c0 += WAVE) {
        const int k = base + c0 + lane;
        const double val = (c0 + lane < count && k < limit) ? v[k] : 0.0;
        const double inc = wave_scan_incl(val, lane) + carry;
        const unsigned long long hit = __ballot(inc > target);
        if (hit) {
            const int first = __ffsll((long long)hit) - 1;
            const double prev = __shfl(inc, first > 0 ? first - 1 : 0, WAVE);
            *before = first > 0 ? prev : carry;
            return c0 + first;
        }
        const unsigned long long pos = __ballot(val > 0.0);
        if (pos) {
            const int lastl = 63 - __clzll((long long)pos);
            const double prev = __shfl(inc, lastl > 0 ? lastl - 1 : 0, WAVE);
            last_pos = c0 + lastl; last_before = lastl > 0 ? prev : carry;
        }
        carry = __shfl(inc, 63, WAVE);
    }
    *before = last_before;
    return last_pos;
}

#define EVL_NT 1024
// The whole event loop of one KMC step (kmc_events.cu:210-349) in one resident workgroup.
__global__ __launch_bounds__(EVL_NT) void k_ev_loop(int N, int nn, const int *__restrict__ neigh, double *__restrict__ ev_prob,
                                                    double *__restrict__ rowsum, double *__restrict__ g2, double *__restrict__ g3,
                                                    int ng2, int ng3, int *__restrict__ element, int *__restrict__ charge,
                                                    const double *__restrict__ uniform, int n_uniform,
                                                    const double *__restrict__ freq_p, EvOut *__restrict__ out,
                                                    int *__restrict__ evlog, int max_log, const int *__restrict__ ncharged_p)
{
    __shared__ int sh_i, sh_j, sh_stop;
    __shared__ double sh_psum;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const double inv_freq = 1 / (*freq_p);
    int n_events = 0, exhausted = 0, bad = 0;
    double event_time = 0.0, psum_last = 0.0;
    for (;;) {
        if (2 * n_events + 1 >= n_uniform) { exhausted = 1; break; }
        // ---- select ----
        if (w == 0) {
            double s = 0.0;
            for (int k = lane; k < ng3; k += WAVE) s += g3[k];
            const double Psum = wave_sum_all(s);
            const double number = uniform[2 * n_events] * Psum;
            double before;
            int k3 = wave_pick(g3, 0, ng3, ng3, number, lane, &before);
            int i_sel = -1, j_sel = -1;
            if (k3 >= 0) {
                double rem = number - before;
                int k2 = wave_pick(g2, k3 * 64, 64, ng2, rem, lane, &before);
                if (k2 >= 0) {
                    rem -= before; k2 += k3 * 64;
                    int r = wave_pick(rowsum, k2 * 64, 64, N, rem, lane, &before);
                    if (r >= 0) {
                        rem -= before; r += k2 * 64;
                        const int sl = wave_pick(ev_prob + (size_t)r * nn, 0, nn, nn, rem, lane, &before);
                        if (sl >= 0) { i_sel = r; j_sel = neigh[(size_t)r * nn + sl];
                            if (lane == 0 && evlog && n_events < max_log) evlog[4 * n_events] = r * nn + sl; }
                    }
                }
            }
            if (lane == 0) { sh_i = i_sel; sh_j = j_sel; sh_psum = Psum; sh_stop = (i_sel < 0 || j_sel < 0); }
        }
        __syncthreads();
        if (sh_stop) { bad = 1; event_time = INFINITY; break; }     // no positive rate left (reference: undefined)
        const int i = sh_i, j = sh_j;
        const double Psum = sh_psum;
        // ---- execute (kmc_events.cu:249-320) ----
        if (tid == 0) {
            const int ei = element[i], ej = element[j], qi = charge[i], qj = charge[j];
            int type = EV_NULL;
            if (ei == DEFECT && ej == O_EL) { type = EV_GEN; element[i] = OXYGEN_DEFECT; element[j] = VACANCY; charge[i] = -2; charge[j] = 2; }
            else if (ei == OXYGEN_DEFECT && ej == VACANCY) { type = EV_REC; element[i] = DEFECT; element[j] = O_EL; charge[i] = 0; charge[j] = 0; }
            else if (ei == VACANCY && ej == O_EL) { type = EV_VDIFF; element[i] = ej; element[j] = ei; charge[i] = qj; charge[j] = qi; }
            else if (ei == OXYGEN_DEFECT && ej == DEFECT) { type = EV_IDIFF; element[i] = ej; element[j] = ei; charge[i] = qj; charge[j] = qi; }
            if (evlog && n_events < max_log) { evlog[4 * n_events + 1] = i; evlog[4 * n_events + 2] = j; evlog[4 * n_events + 3] = type; }
        }
        // ---- invalidate every slot whose row or target is i or j (zero_out_events + memsets) ----
        for (int c = tid; c < 2 * nn; c += EVL_NT) { const int r = c < nn ? i : j; ev_prob[(size_t)r * nn + (c < nn ? c : c - nn)] = 0.0; }
        for (int c = tid; c < 2 * nn * nn; c += EVL_NT) {
            const int a = c / nn, s = c - a * nn;
            const int n = neigh[(size_t)(a < nn ? i : j) * nn + (a < nn ? a : a - nn)];
            if (n < 0) continue;
            const int t = neigh[(size_t)n * nn + s];
            if (t == i || t == j) ev_prob[(size_t)n * nn + s] = 0.0;
        }
        __syncthreads();
        // ---- repair the sum tree above the touched rows: i, j and their neighbours ----
        const int n_aff = 2 + 2 * nn;
        for (int a = w; a < n_aff; a += EVL_NT / 64) {
            const int r = a == 0 ? i : a == 1 ? j : neigh[(size_t)(a - 2 < nn ? i : j) * nn + (a - 2 < nn ? a - 2 : a - 2 - nn)];
            if (r < 0) continue;
            const double s = row_sum(ev_prob + (size_t)r * nn, nn, lane);
            if (lane == 0) rowsum[r] = s;
        }
        __syncthreads();
        for (int a = w; a < n_aff; a += EVL_NT / 64) {
            const int r = a == 0 ? i : a == 1 ? j : neigh[(size_t)(a - 2 < nn ? i : j) * nn + (a - 2 < nn ? a - 2 : a - 2 - nn)];
            if (r < 0) continue;
            const int k2 = r >> 6;
            const double s = group_sum(rowsum, k2 * 64, N, lane);
            if (lane == 0) g2[k2] = s;
        }
        __syncthreads();
        for (int a = w; a < n_aff; a += EVL_NT / 64) {
            const int r = a == 0 ? i : a == 1 ? j : neigh[(size_t)(a - 2 < nn ? i : j) * nn + (a - 2 < nn ? a - 2 : a - 2 - nn)];
            if (r < 0) continue;
            const int k3 = r >> 12;
            const double s = group_sum(g2, k3 * 64, ng2, lane);
            if (lane == 0) g3[k3] = s;
        }
        __syncthreads();
        // ---- waiting time (kmc_events.cu:348): assigned, not accumulated ----
        event_time = -log(uniform[2 * n_events + 1]) / Psum;
        psum_last = Psum;
        ++n_events;
        if (!(event_time < inv_freq)) break;
    }
    if (tid == 0) { out->event_time = event_time; out->psum_last = psum_last; out->n_events = n_events; out->exhausted = exhausted; out->bad = bad;
                    out->n_charged = ncharged_p ? *ncharged_p : -1; }     // charged-site count of the last pair sum rides along in the mailbox
}

// called by dkmc_copy_to_const_memory (engine.hip)
int events_upload_layers()
{
    Engine &e = eng(); LayerEnergies L;
    for (int i = 0; i < DKMC_MAX_LAYERS; ++i) { L.gen[i] = e.E_gen[i]; L.rec[i] = e.E_rec[i]; L.vdiff[i] = e.E_Vdiff[i]; L.odiff[i] = e.E_Odiff[i]; }
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_layerE), &L, sizeof(L)));
    return 0;
}

extern "C" int dkmc_build_event_list(int N, int nn, const int *neigh, const int *layer, const double *lattice, int pbc,
                                     const double *T_bg, const double *freq, const double *sigma, const double *k,
                                     const double *x, const double *y, const double *z, const double *pb, const double *pc,
                                     const int *element, const int *charge, int *ev_type, double *ev_prob)
{
    hipLaunchKernelGGL(k_ev_build, dim3((N + 3) / 4), dim3(256), 0, eng().stream, N, nn, neigh, layer, lattice, pbc, T_bg, freq, sigma, k,
                       x, y, z, pb, pc, element, charge, ev_type, ev_prob, (double *)nullptr);
    KCHK();
    return 0;
}

extern "C" int dkmc_execute_kmc_step_gpu(int N, int nn, const int *neigh, const int *layer, const double *lattice, int pbc,
                                         const double *T_bg, const double *freq, const double *sigma, const double *k,
                                         const double *x, const double *y, const double *z, const double *pb, const double *pc,
                                         const double *temperature, int *element, int *charge,
                                         const double *h_uniform, int n_uniform, int resume,
                                         int *n_events_out, int *exhausted_out, int *h_event_log, double *event_time_out)
{
    (void)temperature;     // unused by the reference kernel as well (Ekin = 0, kmc_events.cu:65)
    Engine &e = eng(); hipStream_t st = e.stream;
    const size_t total = (size_t)N * nn;
    const int ng2 = (N + 63) / 64, ng3 = (ng2 + 63) / 64;
    const int max_log = n_uniform / 2;
    double *ev_prob = (double *)scratch(S_EV_PROB, total * 8);
    double *rowsum = (double *)scratch(S_EV_ROWSUM, (size_t)N * 8);
    double *g2 = (double *)scratch(S_EV_G2, (size_t)ng2 * 8), *g3 = (double *)scratch(S_EV_G3, (size_t)ng3 * 8);
    double *uni = (double *)scratch(S_EV_UNI, (size_t)(n_uniform > 0 ? n_uniform : 1) * 8);
    int *evlog = (int *)scratch(S_EV_LOG, (size_t)(max_log > 0 ? max_log : 1) * 16);
    EvOut *out = (EvOut *)scratch(S_EV_CTRL, sizeof(EvOut));
    if (!ev_prob || !rowsum || !g2 || !g3 || !uni || !evlog || !out) return e.err_code;
    if (n_uniform > 0) HIPCHK(hipMemcpyAsync(uni, h_uniform, (size_t)n_uniform * 8, hipMemcpyHostToDevice, st));
    if (!resume) {
        hipLaunchKernelGGL(k_ev_build, dim3((N + 3) / 4), dim3(256), 0, st, N, nn, neigh, layer, lattice, pbc, T_bg, freq, sigma, k,
                           x, y, z, pb, pc, element, charge, (int *)nullptr, ev_prob, rowsum);
        hipLaunchKernelGGL(k_ev_level, dim3((ng2 + 3) / 4), dim3(256), 0, st, N, rowsum, ng2, g2);
        hipLaunchKernelGGL(k_ev_level, dim3((ng3 + 3) / 4), dim3(256), 0, st, ng2, g2, ng3, g3);
    }
    hipLaunchKernelGGL(k_ev_loop, dim3(1), dim3(EVL_NT), 0, st, N, nn, neigh, ev_prob, rowsum, g2, g3, ng2, ng3, element, charge,
                       uni, n_uniform, freq, out, h_event_log ? evlog : (int *)nullptr, max_log, (const int *)e.buf[S_PW_CNT]);
    KCHK();
    EvOut h;
    HIPCHK(hipMemcpyAsync(&h, out, sizeof(EvOut), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (h_event_log && h.n_events > 0) HIPCHK(hipMemcpy(h_event_log, evlog, (size_t)h.n_events * 16, hipMemcpyDeviceToHost));
    e.stats.n_events = h.n_events; e.stats.psum_last = h.psum_last;
    if (h.n_charged >= 0) e.stats.n_charged = h.n_charged;
    if (n_events_out) *n_events_out = h.n_events;
    if (exhausted_out) *exhausted_out = h.exhausted;
    if (event_time_out) *event_time_out = h.event_time;
    if (h.bad) return dkmc_fail(7, "execute_kmc_step: no event with a positive rate", __FILE__, __LINE__);
    return 0;
}
