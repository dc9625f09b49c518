"""ctypes loader for libcapital_amd.so - the product's only compute backend.

Fails loudly if the HIP library is missing: there is no CPU fallback by design."""
import ctypes as C
import os

import torch  # noqa: F401  - MUST load first: torch bundles its own libamdhip64/librccl with the same sonames as
              # /opt/rocm; loading ours first leaves the process with a HIP runtime that sees no device.

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcapital_amd.so")

_lib = None

i64 = C.c_int64
dbl = C.c_double
ptr = C.c_void_p
cint = C.c_int

# name -> (restype, argtypes); mirrors include/capital_amd.h one to one
SIGNATURES = {
    "cap_status_string": (C.c_char_p, [cint]),
    "cap_device_info": (cint, [C.c_char_p, cint, C.POINTER(cint), C.POINTER(i64)]),
    "cap_dgemm": (cint, [cint, cint, i64, i64, i64, dbl, ptr, i64, ptr, i64, dbl, ptr, i64, ptr]),
    "cap_dsyrk": (cint, [cint, cint, i64, i64, dbl, ptr, i64, dbl, ptr, i64, ptr]),
    "cap_dtrmm": (cint, [cint, cint, cint, cint, i64, i64, dbl, ptr, i64, ptr, i64, ptr, ptr]),
    "cap_dtrmm_work_size": (i64, [cint, i64, i64]),
    "cap_dtrsm": (cint, [cint, cint, cint, i64, i64, dbl, ptr, i64, ptr, i64, ptr, ptr]),
    "cap_dtrsm_work_size": (i64, [cint, i64, i64]),
    "cap_dpotrf": (cint, [cint, i64, ptr, i64, ptr, ptr, ptr]),
    "cap_dpotrf_work_size": (i64, [i64]),
    "cap_dtrtri": (cint, [cint, i64, ptr, i64, ptr, ptr]),
    "cap_dtrtri_work_size": (i64, [i64]),
    "cap_desc_create": (cint, [C.POINTER(ptr), i64, i64, i64, i64]),
    "cap_desc_create_view": (cint, [C.POINTER(ptr), i64, i64, i64, i64, ptr, i64]),
    "cap_desc_create_bc": (cint, [C.POINTER(ptr), i64, i64, i64, cint, cint, cint, cint, ptr, i64]),
    "cap_desc_set_position": (cint, [ptr, i64, i64]),
    "cap_desc_import_host_global": (cint, [ptr, ptr, i64, ptr]),
    "cap_desc_export_host_global": (cint, [ptr, ptr, i64, ptr]),
    "cap_cholinv_factor_desc": (cint, [ptr, ptr, ptr]),
    "cap_cholinv_get_R_desc": (cint, [ptr, ptr, ptr]),
    "cap_cholinv_get_Rinv_desc": (cint, [ptr, ptr, ptr]),
    "cap_dist2d_factor_desc": (cint, [ptr, ptr, ptr]),
    "cap_dist2d_get_R_desc": (cint, [ptr, ptr, ptr]),
    "cap_dist2d_get_Rinv_desc": (cint, [ptr, ptr, ptr]),
    "cap_desc_destroy": (cint, [ptr]),
    "cap_desc_data": (ptr, [ptr]),
    "cap_desc_get": (i64, [ptr, cint]),
    "cap_desc_import_host": (cint, [ptr, ptr, i64, ptr]),
    "cap_desc_export_host": (cint, [ptr, ptr, i64, ptr]),
    "cap_fill_symmetric": (cint, [ptr, i64, i64, i64, i64, i64, cint, ptr]),
    "cap_fill_random": (cint, [ptr, i64, i64, i64, i64, i64, i64, i64, i64, ptr]),
    "cap_copy_window": (cint, [ptr, cint, i64, i64, i64, ptr, cint, i64, i64, i64, i64, i64, cint, cint, ptr]),
    "cap_cyclic_import": (cint, [ptr, i64, ptr, i64, i64, i64, i64, i64, i64, i64, ptr]),
    "cap_cyclic_export": (cint, [ptr, i64, ptr, i64, i64, i64, i64, i64, i64, i64, ptr]),
    "cap_remove_triangle": (cint, [ptr, i64, i64, i64, i64, i64, i64, cint, ptr]),
    "cap_cholesky_residual_terms": (cint, [ptr, i64, ptr, i64, i64, ptr, ptr, ptr]),
    "cap_sumsq": (cint, [ptr, i64, i64, i64, cint, cint, ptr, ptr]),
    "cap_comm_unique_id": (cint, [ptr]),
    "cap_comm_create": (cint, [C.POINTER(ptr), ptr, cint, cint, ptr]),
    "cap_comm_create_self": (cint, [C.POINTER(ptr)]),
    "cap_comm_create_callbacks": (cint, [C.POINTER(ptr), cint, cint, ptr, ptr, ptr, ptr]),
    "cap_comm_split": (cint, [ptr, cint, cint, C.POINTER(ptr)]),
    "cap_comm_dup": (cint, [ptr, C.POINTER(ptr)]),
    "cap_comm_backend": (cint, [ptr]),
    "cap_comm_query": (cint, [ptr, C.POINTER(cint), C.POINTER(cint), C.POINTER(cint)]),
    "cap_comm_reduce_sum": (cint, [ptr, ptr, i64, cint, ptr]),
    "cap_topo_coords": (cint, [cint, cint, cint, cint, C.POINTER(cint), C.POINTER(cint), C.POINTER(cint), C.POINTER(cint)]),
    "cap_topo_create": (cint, [C.POINTER(ptr), cint, ptr, cint, cint, cint]),
    "cap_topo_create_from": (cint, [C.POINTER(ptr), cint, ptr, cint, cint, cint, C.POINTER(ptr), cint]),
    "cap_topo_destroy": (cint, [ptr]),
    "cap_topo_comm": (ptr, [ptr, cint]),
    "cap_topo_get": (cint, [ptr, cint]),
    "cap_comm_destroy": (cint, [ptr]),
    "cap_comm_rank": (cint, [ptr]),
    "cap_comm_size": (cint, [ptr]),
    "cap_comm_allreduce_sum": (cint, [ptr, ptr, i64, ptr]),
    "cap_comm_bcast": (cint, [ptr, ptr, i64, cint, ptr]),
    "cap_comm_allgather": (cint, [ptr, ptr, ptr, i64, ptr]),
    "cap_comm_barrier": (cint, [ptr, ptr]),
    "cap_comm_alltoallv": (cint, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr]),
    "cap_comm_exchange": (cint, [ptr, ptr, ptr, i64, cint, ptr]),
    "cap_comm_set_alltoallv_callback": (cint, [ptr, ptr]),
    "cap_redist_plan_create": (cint, [C.POINTER(ptr), i64, i64, ptr, cint, cint]),
    "cap_redist_plan_destroy": (cint, [ptr]),
    "cap_redist_get": (i64, [ptr, cint]),
    "cap_redist_message_elems": (i64, [i64, i64, cint, cint, cint, cint, cint, cint]),
    "cap_redistribute_cyclic_to_bc": (cint, [ptr, ptr, i64, ptr, i64, ptr]),
    "cap_redistribute_bc_to_cyclic": (cint, [ptr, ptr, i64, ptr, i64, ptr]),
    "cap_cholinv_plan_create": (cint, [C.POINTER(ptr), i64, cint, i64, i64, C.c_char, ptr]),
    "cap_cholinv_plan_destroy": (cint, [ptr]),
    "cap_cholinv_factor": (cint, [ptr, ptr, i64, ptr]),
    "cap_cholinv_get_R": (cint, [ptr, ptr, i64, ptr]),
    "cap_cholinv_get_Rinv": (cint, [ptr, ptr, i64, ptr]),
    "cap_cholinv_R_ptr": (ptr, [ptr, C.POINTER(i64)]),
    "cap_cholinv_Rinv_ptr": (ptr, [ptr, C.POINTER(i64)]),
    "cap_cholinv_info": (cint, [ptr, ptr, C.POINTER(i64)]),
    "cap_cholinv_set_option": (cint, [ptr, C.c_char_p, i64]),
    "cap_cholinv_get_option": (i64, [ptr, C.c_char_p]),
    "cap_cholinv_profile": (cint, [ptr, C.POINTER(i64), C.POINTER(dbl), C.POINTER(dbl)]),
    "cap_cholinv_profile_launches": (cint, [ptr, C.POINTER(dbl), C.POINTER(dbl), i64, C.POINTER(i64)]),
    "cap_dist_plan_create": (cint, [C.POINTER(ptr), i64, i64, ptr]),
    "cap_dist_plan_destroy": (cint, [ptr]),
    "cap_dist_local_cols": (i64, [ptr]),
    "cap_dist_factor": (cint, [ptr, ptr, i64, ptr]),
    "cap_dist_R_ptr": (ptr, [ptr, C.POINTER(i64)]),
    "cap_dist_info": (cint, [ptr, ptr, C.POINTER(i64)]),
    "cap_dist_get_R": (cint, [ptr, ptr, i64, ptr]),
    "cap_dist_get_Rinv": (cint, [ptr, ptr, i64, ptr]),
    "cap_dist_Rinv_ptr": (ptr, [ptr, C.POINTER(i64)]),
    "cap_dist_set_option": (cint, [ptr, C.c_char_p, i64]),
    "cap_dist_get_option": (i64, [ptr, C.c_char_p]),
    "cap_dist_profile": (cint, [ptr, C.POINTER(i64), C.POINTER(dbl), C.POINTER(dbl)]),
    "cap_dist_profile_streams": (cint, [ptr, C.POINTER(dbl)]),
    "cap_dist_profile_launches": (cint, [ptr, C.POINTER(dbl), C.POINTER(dbl), i64, C.POINTER(i64)]),
    "cap_dist_progress": (cint, [ptr, C.POINTER(i64)]),
    "cap_dist_profile_inverse": (cint, [ptr, C.POINTER(dbl)]),
    "cap_fill_symmetric_bc": (cint, [ptr, i64, i64, i64, cint, cint, cint, ptr]),
    "cap_bc_owner": (cint, [i64, cint]),
    "cap_bc_local_block": (i64, [i64, cint]),
    "cap_bc_num_local_cols": (i64, [i64, i64, cint, cint]),
    "cap_dist2d_plan_create": (cint, [C.POINTER(ptr), i64, i64, ptr, cint, ptr, ptr]),
    "cap_dist2d_plan_destroy": (cint, [ptr]),
    "cap_dist2d_get": (i64, [ptr, cint]),
    "cap_dist2d_factor": (cint, [ptr, ptr, i64, ptr]),
    "cap_dist2d_R_ptr": (ptr, [ptr, C.POINTER(i64)]),
    "cap_dist2d_get_R": (cint, [ptr, ptr, i64, ptr]),
    "cap_dist2d_info": (cint, [ptr, ptr, C.POINTER(i64)]),
    "cap_dist2d_set_option": (cint, [ptr, C.c_char_p, i64]),
    "cap_dist2d_get_Rinv": (cint, [ptr, ptr, i64, ptr]),
    "cap_dist2d_Rinv_ptr": (ptr, [ptr, C.POINTER(i64)]),
    "cap_bc2d_local_extent": (i64, [i64, i64, cint, cint, cint, cint, cint]),
    "cap_bc2d_rows_le": (cint, [cint, cint, cint, cint, cint, cint]),
    "cap_fill_symmetric_bc2d": (cint, [ptr, i64, i64, i64, cint, cint, cint, cint, cint, ptr]),
    "cap_summa_plan_create": (cint, [C.POINTER(ptr), ptr, i64, i64, i64, cint]),
    "cap_summa_plan_destroy": (cint, [ptr]),
    "cap_summa_local_dims": (None, [ptr, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "cap_summa_dgemm": (cint, [ptr, dbl, ptr, i64, ptr, i64, dbl, ptr, i64, ptr]),
    "cap_util_transpose": (cint, [ptr, ptr, ptr, i64, ptr]),
    "cap_summa_dtrmm": (cint, [ptr, cint, cint, cint, cint, dbl, ptr, i64, cint, ptr, i64, ptr]),
    "cap_summa_dsyrk": (cint, [ptr, cint, cint, dbl, ptr, i64, dbl, ptr, i64, cint, ptr]),
    "cap_mpchol_plan_create": (cint, [C.POINTER(ptr), i64, i64]),
    "cap_mpchol_plan_destroy": (cint, [ptr]),
    "cap_mpchol_factor": (cint, [ptr, ptr, i64, ptr]),
    "cap_mpchol_info": (cint, [ptr, ptr, C.POINTER(i64)]),
    "cap_mpchol_solve": (cint, [ptr, ptr, i64, ptr, i64, ptr, i64, i64, cint, dbl, C.POINTER(cint), C.POINTER(dbl), ptr]),
    "cap_mpchol_R32_ptr": (ptr, [ptr, C.POINTER(i64)]),
    "cap_mpchol_set_option": (cint, [ptr, C.c_char_p, i64]),
    "cap_mpchol_profile": (cint, [ptr, C.POINTER(i64), C.POINTER(dbl), C.POINTER(dbl), C.POINTER(dbl)]),
    "cap_chain_fallbacks": (i64, []),
    "cap_chain_inject_timeouts": (cint, [cint]),
    "cap_bf16_update": (cint, [cint, i64, i64, i64, C.c_float, ptr, i64, ptr, i64, ptr, i64, cint, cint, ptr]),
    "cap_dmp_plan_create": (cint, [C.POINTER(ptr), i64, i64, i64, ptr]),
    "cap_dmp_plan_destroy": (cint, [ptr]),
    "cap_dmp_local_cols": (i64, [ptr]),
    "cap_dmp_factor": (cint, [ptr, ptr, i64, ptr]),
    "cap_dmp_info": (cint, [ptr, ptr, C.POINTER(i64)]),
    "cap_dmp_solve": (cint, [ptr, ptr, i64, ptr, i64, ptr, i64, i64, cint, dbl, C.POINTER(cint), C.POINTER(dbl), ptr]),
    "cap_dmp_R32_ptr": (ptr, [ptr, C.POINTER(i64)]),
    "cap_cacqr_plan_create": (cint, [C.POINTER(ptr), i64, i64, cint, ptr]),
    "cap_cacqr_plan_create_grid": (cint, [C.POINTER(ptr), i64, i64, cint, ptr]),
    "cap_cacqr_local_cols": (i64, [ptr]),
    "cap_cacqr_R_piece": (cint, [ptr, ptr, i64, ptr]),
    "cap_cacqr_plan_destroy": (cint, [ptr]),
    "cap_cacqr_factor": (cint, [ptr, ptr, i64, ptr]),
    "cap_cacqr_Q_ptr": (ptr, [ptr, C.POINTER(i64)]),
    "cap_cacqr_R_ptr": (ptr, [ptr, C.POINTER(i64)]),
    "cap_cacqr_info": (cint, [ptr, ptr, C.POINTER(i64)]),
}


class CapitalError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle with typed signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CapitalError(
            "libcapital_amd.so not found at %s - build it with `python -m capital_amd.build` "
            "(there is no CPU fallback)" % LIB_PATH)
    h = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            f = getattr(h, name)
        except AttributeError:
            missing.append(name)
            continue
        f.restype = res
        f.argtypes = args
    if missing:
        raise CapitalError("libcapital_amd.so lacks symbols declared in include/capital_amd.h: %s" % missing)
    _lib = h
    return h


def check(status, what=""):
    if status != 0:
        msg = lib().cap_status_string(status)
        raise CapitalError("%s failed: status %d (%s)" % (what or "capital_amd call", status,
                                                          msg.decode() if msg else "?"))


def check_info(status, what=""):
    """Status of a *_info query: CAP_OK, or CAP_ERR_NOT_SPD when the info word it returns is non-zero (the caller reads the word);
    anything else means the query itself failed and the info word was NOT written - never to be read as "info = 0"."""
    if status not in (0, 3):
        check(status, what)
