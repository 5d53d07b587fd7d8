// engine.hip -- process-wide engine state, GPUBuffers twin, small utilities.
#include "common.h"
#include <stdlib.h>

int events_upload_layers();      // events.hip
void xstate_reset(const void *key);   // current.hip

static Engine g_engine;
Engine &eng() { return g_engine; }

int dkmc_fail(int code, const char *what, const char *file, int line)
{
    Engine &e = eng();
    snprintf(e.err, sizeof(e.err), "%s (%s:%d)", what, file, line);
    e.err_code = code ? code : -1;
    fprintf(stderr, "devicekmc_hip: %s\n", e.err);   // the reference prints and carries on (utils.h:145-153)
    return e.err_code;
}

void *scratch(int slot, size_t bytes)
{
    Engine &e = eng();
    if (bytes == 0) bytes = 16;
    if (e.bufsz[slot] < bytes) {
        if (e.buf[slot]) { (void)hipStreamSynchronize(e.stream); (void)hipFree(e.buf[slot]); }
        size_t want = bytes + bytes / 8 + 256;      // slack so slowly growing sizes do not reallocate each step
        hipError_t rc = hipMalloc(&e.buf[slot], want);
        if (rc != hipSuccess) {
            char msg[160]; snprintf(msg, sizeof(msg), "hipMalloc(%zu bytes) failed in scratch(slot %d): %s", want, slot, hipGetErrorString(rc));
            dkmc_fail(2, msg, __FILE__, __LINE__); e.buf[slot] = nullptr; e.bufsz[slot] = 0; return nullptr;
        }
        e.bufsz[slot] = want;
    }
    return e.buf[slot];
}

extern "C" {

const char *dkmc_last_error(void) { return eng().err; }
void dkmc_clear_error(void) { eng().err[0] = 0; eng().err_code = 0; }
const dkmc_stats *dkmc_get_stats(void) { return &eng().stats; }
void dkmc_set_cg_tolerance(double tol) { eng().cg_tol = tol; }
void dkmc_set_current_warm_start(int mode) { eng().current_warm_start = mode; }
int dkmc_get_current_warm_start(void) { return eng().current_warm_start; }
void dkmc_set_profiling(int on) { eng().profiling = on; }
void dkmc_set_x_format(int tiled) { eng().x_format = tiled ? 1 : 0; }
int dkmc_get_x_format(void) { return eng().x_format; }
void dkmc_set_tcache_budget(long long bytes) { eng().tcache_budget = bytes; }
void dkmc_set_pair_cutoff(double x_cut) { eng().pair_cut = x_cut > 0.0 ? x_cut : 0.0; }
void dkmc_set_k_slab(int on) { eng().k_slab = on ? 1 : 0; }
int dkmc_get_k_slab(void) { return eng().k_slab; }
void dkmc_set_x_aux_warm(int on) { eng().x_aux_warm = on ? 1 : 0; }
int dkmc_get_x_aux_warm(void) { return eng().x_aux_warm; }
void dkmc_set_x_items(int kc) { eng().x_items_kc = kc > 0 ? (kc < 256 ? kc : 256) : 0; }
void dkmc_set_x_poly(int degree) { eng().x_poly = degree < 0 ? 0 : (degree > 16 ? 16 : degree); }
int dkmc_get_x_poly(void) { return eng().x_poly; }
void dkmc_set_x_apply_form(int form) { eng().x_apply_form = form == 1 ? 1 : 0; }
int dkmc_get_x_apply_form(void) { return eng().x_apply_form; }
void dkmc_set_x_slab(int on) { eng().x_slab = on ? 1 : 0; }
int dkmc_get_x_slab(void) { return eng().x_slab; }
void dkmc_set_x_block(int s) { eng().x_block = s < 1 ? 1 : (s > 16 ? 16 : s); }
int dkmc_get_x_block(void) { return eng().x_block; }
void dkmc_set_x_aux(int mode) { eng().x_aux = mode < 0 ? 0 : (mode > 3 ? 2 : mode); }
int dkmc_get_x_aux(void) { return eng().x_aux; }
void dkmc_set_k_blocked(int on) { eng().k_blocked = on ? 1 : 0; }
int dkmc_get_k_blocked(void) { return eng().k_blocked; }
void dkmc_set_cb_edge_domain(int atoms_only) { eng().cb_edge_domain = atoms_only ? 1 : 0; }

int dkmc_get_gpu_info(char *gpu_string, int capacity, int dev)
{
    hipDeviceProp_t prop;
    HIPCHK(hipSetDevice(dev));
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    snprintf(gpu_string, capacity, "%s", prop.name);
    return 0;
}

int dkmc_set_gpu(int dev) { HIPCHK(hipSetDevice(dev)); eng().device = dev; return 0; }
int dkmc_set_stream(void *s) { eng().stream = (hipStream_t)s; return 0; }
int dkmc_synchronize(void) { HIPCHK(hipStreamSynchronize(eng().stream)); return 0; }

int dkmc_copy_to_const_memory(const double *E_gen, const double *E_rec, const double *E_Vdiff, const double *E_Odiff, int nl)
{
    Engine &e = eng();
    if (nl > DKMC_MAX_LAYERS) return dkmc_fail(3, "more than 5 layers (MAX_NUM_LAYERS, kmc_events.cu:7)", __FILE__, __LINE__);
    e.num_layers = nl;
    for (int i = 0; i < nl; ++i) { e.E_gen[i] = E_gen[i]; e.E_rec[i] = E_rec[i]; e.E_Vdiff[i] = E_Vdiff[i]; e.E_Odiff[i] = E_Odiff[i]; }
    return events_upload_layers();
}

// ---- GPUBuffers twin -----------------------------------------------------------------------------
#define ALLOC(field, n, T) HIPCHK(hipMalloc((void **)&buf->field, (size_t)(n) * sizeof(T)))
int dkmc_gpubuf_create(dkmc_gpubuf *buf, int N, int N_atom, int nn, int nmt,
                       const int *h_layer, const double *hx, const double *hy, const double *hz,
                       const int *h_neigh, const int *h_metals, double freq, double sigma, double k, const double *h_lattice)
{
    memset(buf, 0, sizeof(*buf));
    buf->N_ = N; buf->N_atom_ = N_atom; buf->nn_ = nn; buf->num_metal_types_ = nmt;
    ALLOC(site_layer, N, int); ALLOC(site_element, N, int); ALLOC(metal_types, nmt > 0 ? nmt : 1, int);
    ALLOC(site_x, N, double); ALLOC(site_y, N, double); ALLOC(site_z, N, double);
    ALLOC(site_power, N, double); ALLOC(site_CB_edge, N, double);
    ALLOC(site_potential_boundary, N, double); ALLOC(site_potential_charge, N, double);
    ALLOC(site_temperature, N, double); ALLOC(site_charge, N, int);
    ALLOC(neigh_idx, (size_t)N * nn, int);
    ALLOC(T_bg, 1, double); ALLOC(sigma, 1, double); ALLOC(k, 1, double); ALLOC(lattice, 3, double); ALLOC(freq, 1, double);
    ALLOC(atom_element, N, int); ALLOC(atom_x, N, double); ALLOC(atom_y, N, double); ALLOC(atom_z, N, double);
    ALLOC(atom_power, N, double); ALLOC(atom_CB_edge, N, double); ALLOC(atom_charge, N, int);
    ALLOC(atom_virtual_potentials, N_atom + 2, double);
    HIPCHK(hipMemset(buf->atom_virtual_potentials, 0, (size_t)(N_atom + 2) * sizeof(double)));
    HIPCHK(hipMemset(buf->site_power, 0, (size_t)N * sizeof(double)));
    HIPCHK(hipMemcpy(buf->site_layer, h_layer, (size_t)N * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(buf->site_x, hx, (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(buf->site_y, hy, (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(buf->site_z, hz, (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    if (nmt > 0) HIPCHK(hipMemcpy(buf->metal_types, h_metals, (size_t)nmt * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(buf->sigma, &sigma, sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(buf->k, &k, sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(buf->freq, &freq, sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(buf->lattice, h_lattice, 3 * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(buf->neigh_idx, h_neigh, (size_t)N * nn * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}
#undef ALLOC

int dkmc_gpubuf_free(dkmc_gpubuf *b)
{
    xstate_reset(b->site_x);
    void *ptrs[] = { b->site_charge, b->site_power, b->site_potential_boundary, b->site_potential_charge, b->site_temperature,
                     b->site_CB_edge, b->T_bg, b->atom_power, b->atom_CB_edge, b->atom_virtual_potentials, b->atom_charge,
                     b->site_element, b->atom_element, b->site_x, b->site_y, b->site_z, b->atom_x, b->atom_y, b->atom_z,
                     b->metal_types, b->sigma, b->k, b->lattice, b->freq, b->neigh_idx, b->site_layer,
                     b->Device_row_ptr_d, b->Device_col_indices_d, b->contact_left_row_ptr, b->contact_left_col_indices,
                     b->contact_right_row_ptr, b->contact_right_col_indices };
    for (void *p : ptrs) if (p) (void)hipFree(p);
    memset(b, 0, sizeof(*b));
    return 0;
}

int dkmc_gpubuf_sync_host_to_gpu(dkmc_gpubuf *b, const int *el, const int *q, const double *pw, const double *cb,
                                 const double *pb, const double *pc, const double *T, const double *acb, double T_bg)
{
    const size_t N = b->N_;
    HIPCHK(hipStreamSynchronize(eng().stream));
    HIPCHK(hipMemcpy(b->site_element, el, N * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->site_charge, q, N * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->site_power, pw, N * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->site_CB_edge, cb, N * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->site_potential_boundary, pb, N * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->site_potential_charge, pc, N * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->site_temperature, T, N * sizeof(double), hipMemcpyHostToDevice));
    if (acb) HIPCHK(hipMemcpy(b->atom_CB_edge, acb, (size_t)b->N_atom_ * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->T_bg, &T_bg, sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int dkmc_gpubuf_sync_gpu_to_host(const dkmc_gpubuf *b, int *el, int *q, double *pw, double *cb,
                                 double *pb, double *pc, double *T, double *acb, double *T_bg)
{
    const size_t N = b->N_;
    HIPCHK(hipStreamSynchronize(eng().stream));
    HIPCHK(hipMemcpy(el, b->site_element, N * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(q, b->site_charge, N * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pw, b->site_power, N * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cb, b->site_CB_edge, N * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pb, b->site_potential_boundary, N * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(pc, b->site_potential_charge, N * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(T, b->site_temperature, N * sizeof(double), hipMemcpyDeviceToHost));
    if (acb) HIPCHK(hipMemcpy(acb, b->atom_CB_edge, (size_t)b->N_atom_ * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(T_bg, b->T_bg, sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int dkmc_copy_power_from_gpu(const dkmc_gpubuf *b, double *h_power)
{
    HIPCHK(hipStreamSynchronize(eng().stream));
    HIPCHK(hipMemcpy(h_power, b->site_power, (size_t)b->N_ * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}
int dkmc_copy_charge_to_gpu(dkmc_gpubuf *b, const int *h_charge)
{
    HIPCHK(hipStreamSynchronize(eng().stream));
    HIPCHK(hipMemcpy(b->site_charge, h_charge, (size_t)b->N_ * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}
int dkmc_copy_Tbg_to_gpu(dkmc_gpubuf *b, double T_bg)
{
    HIPCHK(hipStreamSynchronize(eng().stream));
    HIPCHK(hipMemcpy(b->T_bg, &T_bg, sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

} // extern "C"
