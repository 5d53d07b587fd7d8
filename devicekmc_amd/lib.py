"""ctypes binding of libdevicekmc_hip.so (C ABI: include/devicekmc_hip.h).

There is no CPU fallback: if the HIP library is missing or fails to load, importing raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdevicekmc_hip.so")

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)
vp = C.c_void_p


class dkmc_gpubuf(C.Structure):
    """Mirror of struct dkmc_gpubuf (public members of GPUBuffers, gpu_buffers.h:16-52)."""
    _PTRS = ["site_charge", "site_power", "site_potential_boundary", "site_potential_charge", "site_temperature",
             "site_CB_edge", "T_bg", "atom_power", "atom_CB_edge", "atom_virtual_potentials", "atom_charge",
             "site_element", "atom_element", "site_x", "site_y", "site_z", "atom_x", "atom_y", "atom_z",
             "metal_types", "sigma", "k", "lattice", "freq", "neigh_idx", "site_layer",
             "Device_row_ptr_d", "Device_col_indices_d", "contact_left_row_ptr", "contact_left_col_indices",
             "contact_right_row_ptr", "contact_right_col_indices"]
    _INTS = ["Device_nnz", "contact_left_nnz", "contact_right_nnz", "num_metal_types_", "N_", "nn_", "N_atom_"]
    _fields_ = [(n, vp) for n in _PTRS] + [(n, C.c_int) for n in _INTS]


class dkmc_stats(C.Structure):
    _fields_ = [("cg_iters_K", C.c_int), ("cg_iters_CB", C.c_int), ("cg_iters_X", C.c_int),
                ("cg_rr_K", C.c_double), ("cg_rr_CB", C.c_double), ("cg_rr_X", C.c_double),
                ("n_events", C.c_int), ("n_charged", C.c_int), ("N_atom", C.c_int),
                ("X_nnz", C.c_longlong), ("psum_last", C.c_double),
                ("spmv_long_ms", C.c_double), ("spmv_short_ms", C.c_double),
                ("spmv_long_launches", C.c_int), ("spmv_short_launches", C.c_int),
                ("spmv_long_nnz", C.c_longlong), ("spmv_short_nnz", C.c_longlong),
                ("spmv_long_rows", C.c_int), ("spmv_short_rows", C.c_int),
                ("spmv_segments", C.c_int), ("spmv_pad", C.c_int), ("spmv_segment_entries", C.c_longlong),
                ("comm_ranks", C.c_int), ("comm_local_segments", C.c_int), ("comm_count_per_rank", C.c_longlong),
                ("comm_ms", C.c_double), ("comm_launches", C.c_int), ("comm_pad", C.c_int),
                ("spmv_tiles", C.c_int), ("spmv_pad2", C.c_int), ("spmv_tile_entries", C.c_longlong),
                ("xt_subblocks", C.c_longlong), ("xt_local_subblocks", C.c_longlong), ("xt_items", C.c_int), ("xt_kc", C.c_int),
                ("xt_sparse_nnz", C.c_longlong), ("xt_ns", C.c_int), ("xt_split_launch", C.c_int),
                ("kcg_ms", C.c_double), ("kcg_iters_timed", C.c_int), ("kcg_blocked", C.c_int), ("pair_ms", C.c_double), ("pair_evaluated", C.c_longlong), ("pair_tested", C.c_longlong), ("xt_records", C.c_longlong), ("tcache_bytes", C.c_longlong), ("kcg_bytes", C.c_longlong), ("xb_aux", C.c_int), ("xb_pad", C.c_int), ("xb_width", C.c_int), ("xb_fallback", C.c_int)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)


# every symbol include/devicekmc_hip.h and include/devicekmc_hip_debug.h declare: name -> (restype, argtypes)
_I, _D = C.c_int, C.c_double
SYMBOLS = {
    "dkmc_last_error": (C.c_char_p, []),
    "dkmc_clear_error": (None, []),
    "dkmc_get_stats": (C.POINTER(dkmc_stats), []),
    "dkmc_get_gpu_info": (_I, [C.c_char_p, _I, _I]),
    "dkmc_set_gpu": (_I, [_I]),
    "dkmc_set_stream": (_I, [vp]),
    "dkmc_synchronize": (_I, []),
    "dkmc_set_cg_tolerance": (None, [_D]),
    "dkmc_set_cb_edge_domain": (None, [_I]),
    "dkmc_set_x_block": (None, [_I]),
    "dkmc_get_x_block": (_I, []),
    "dkmc_set_k_slab": (None, [_I]),
    "dkmc_get_k_slab": (_I, []),
    "dkmc_xt_tile_census": (_I, [C.POINTER(C.c_longlong)]),
    "dkmc_set_x_items": (None, [_I]),
    "dkmc_set_x_poly": (None, [_I]),
    "dkmc_get_x_poly": (_I, []),
    "dkmc_set_x_apply_form": (None, [_I]),
    "dkmc_get_x_apply_form": (_I, []),
    "dkmc_set_x_slab": (None, [_I]),
    "dkmc_get_x_slab": (_I, []),
    "dkmc_set_x_aux_warm": (None, [_I]),
    "dkmc_get_x_aux_warm": (_I, []),
    "dkmc_set_x_aux": (None, [_I]),
    "dkmc_get_x_aux": (_I, []),
    "dkmc_set_k_blocked": (None, [_I]),
    "dkmc_get_k_blocked": (_I, []),
    "dkmc_set_pair_cutoff": (None, [_D]),
    "dkmc_reset_pair_sum_cache": (None, []),
    "dkmc_set_tcache_budget": (None, [C.c_longlong]),
    "dkmc_set_current_warm_start": (None, [_I]),
    "dkmc_get_current_warm_start": (_I, []),
    "dkmc_get_current_warm_vector": (_I, [C.POINTER(dkmc_gpubuf), vp, _I, c_int_p]),
    "dkmc_set_current_warm_vector": (_I, [C.POINTER(dkmc_gpubuf), vp, _I]),
    "dkmc_get_current_warm_aux": (_I, [C.POINTER(dkmc_gpubuf), vp, C.c_longlong, C.POINTER(C.c_longlong)]),
    "dkmc_set_current_warm_aux": (_I, [C.POINTER(dkmc_gpubuf), vp, C.c_longlong]),
    "dkmc_set_profiling": (None, [_I]),
    "dkmc_set_x_format": (None, [_I]),
    "dkmc_get_x_format": (_I, []),
    "dkmc_gpubuf_create": (_I, [C.POINTER(dkmc_gpubuf), _I, _I, _I, _I, vp, vp, vp, vp, vp, vp, _D, _D, _D, vp]),
    "dkmc_gpubuf_free": (_I, [C.POINTER(dkmc_gpubuf)]),
    "dkmc_gpubuf_sync_host_to_gpu": (_I, [C.POINTER(dkmc_gpubuf), vp, vp, vp, vp, vp, vp, vp, vp, _D]),
    "dkmc_gpubuf_sync_gpu_to_host": (_I, [C.POINTER(dkmc_gpubuf), vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dkmc_copy_power_from_gpu": (_I, [C.POINTER(dkmc_gpubuf), vp]),
    "dkmc_copy_charge_to_gpu": (_I, [C.POINTER(dkmc_gpubuf), vp]),
    "dkmc_copy_Tbg_to_gpu": (_I, [C.POINTER(dkmc_gpubuf), _D]),
    "dkmc_copy_to_const_memory": (_I, [vp, vp, vp, vp, _I]),
    "dkmc_build_neighbor_index": (_I, [_I, vp, vp, vp, vp, _I, _D, c_int_p, vp]),
    "dkmc_initialize_sparsity": (_I, [C.POINTER(dkmc_gpubuf), _I, _D, _I]),
    "dkmc_free_sparsity": (_I, [C.POINTER(dkmc_gpubuf)]),
    "dkmc_update_charge_gpu": (_I, [vp, vp, vp, _I, _I, vp, _I]),
    "dkmc_update_CB_edge_gpu_sparse": (_I, [C.POINTER(dkmc_gpubuf), _I, _I, _I, _D, _I, _D, _D, _D, _I]),
    "dkmc_background_potential_gpu_sparse": (_I, [C.POINTER(dkmc_gpubuf), _I, _I, _I, _D, _I, _D, _D, _D, _I, _I]),
    "dkmc_poisson_gridless_gpu": (_I, [_I, _I, _I, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dkmc_solve_sparse_CG_Jacobi": (_I, [vp, vp, vp, _I, _I, vp, vp, c_int_p, c_dbl_p]),
    "dkmc_solve_sparse_CG_splitmatrix": (_I, [vp, _I, vp, vp, vp, _I, _I, vp, _I, vp, vp, _D, c_int_p, c_dbl_p]),
    "dkmc_execute_kmc_step_gpu": (_I, [_I, _I, vp, vp, vp, _I, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                       vp, _I, _I, c_int_p, c_int_p, vp, c_dbl_p]),
    "dkmc_build_event_list": (_I, [_I, _I, vp, vp, vp, _I, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dkmc_update_power_gpu_sparse": (_I, [C.POINTER(dkmc_gpubuf), _I, _I, _I, _D, _I, _D, _D, _D, _D, _D, _D, _D, _D, _I,
                                          c_dbl_p, _I, _I, _D]),
    "dkmc_get_last_X": (_I, [c_int_p, C.POINTER(C.c_longlong), vp, vp, vp]),
    "dkmc_update_temperatureglobal_gpu": (_I, [vp, vp, _I, _D, _D, _D, _D, _D]),
    "dkmc_update_temperature_global_analytic": (_I, [vp, vp, _I, _D, _D, _D, _D, _D, c_dbl_p]),
    "dkmc_set_heat_cg_tolerance": (None, [_D]),
    "dkmc_construct_laplacian": (_I, [C.POINTER(dkmc_gpubuf), _I, _I, _D]),
    "dkmc_update_temperature_local": (_I, [C.POINTER(dkmc_gpubuf), _D, _D, _D, _D, _D, _D, _D, _I,
                                           C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), c_dbl_p]),
    "dkmc_xt_time_share": (_I, [_I, _I, _I, c_dbl_p, c_dbl_p, c_int_p, C.POINTER(C.c_longlong)]),
    "dkmc_xt_check_shares": (_I, [_I, c_dbl_p, c_dbl_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), c_int_p]),
    "dkmc_xtb_check_product": (_I, [_I, c_dbl_p, c_dbl_p]),
    "dkmc_xtb_time_apply": (_I, [_I, _I, _I, c_dbl_p]),
    "dkmc_kcg_emulate_slabs": (_I, [C.POINTER(dkmc_gpubuf), _I, _I, _I, _D, _D, _D, _I, _I, _I, _I, c_dbl_p, c_int_p, c_int_p, c_dbl_p, C.POINTER(C.c_longlong)]),
    "dkmc_xtb_emulate_slabs": (_I, [_I, _I, _D, _I, _I, c_dbl_p, c_int_p, c_int_p, c_dbl_p, C.POINTER(C.c_longlong), c_int_p]),
    "dkmc_debug_inject_fault": (None, [_I, _I]),
    "dkmc_debug_step_stop_word": (_I, [_I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dkmc_comm_unique_id": (_I, [C.c_char_p]),
    "dkmc_comm_init_rccl": (_I, [_I, _I, C.c_char_p]),
    "dkmc_comm_peer_prepare": (_I, [C.c_size_t, C.c_char_p]),
    "dkmc_comm_peer_attach": (_I, [C.c_char_p]),
    "dkmc_comm_peer_detach": (_I, []),
    "dkmc_comm_peer_info": (_I, [C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "dkmc_comm_init_host": (_I, [_I, _I, ALLGATHER_FN, vp]),
    "dkmc_comm_allgather_host": (_I, [vp, C.c_size_t]),
    "dkmc_comm_info": (_I, [C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "dkmc_comm_destroy": (_I, []),
}

_lib = None


def load():
    """Load the HIP library; raises if it is missing (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        # torch bundles its own HIP runtime (soname libamdhip64.so.7); it has to be in the process before
        # this library so that both resolve to ONE runtime (two runtimes in one process cannot share memory)
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class DeviceKMCError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = load().dkmc_last_error().decode()
        load().dkmc_clear_error()
        raise DeviceKMCError("devicekmc_hip error %d: %s" % (rc, msg))
