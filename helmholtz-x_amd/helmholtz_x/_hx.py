"""ctypes binding of libhx.so (include/hx.h).  There is NO fallback: if the library is missing, or a
call fails, an exception is raised -- the product path never routes through a CPU implementation."""
import ctypes as C
import os

import torch  # first: libhx.so then binds to the HIP runtime torch already loaded (same SONAME)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libhx.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libhx.so not found at %s -- build it with `make -C helmholtz-x_amd/csrc` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

_vp, _i32, _i64, _int = C.c_void_p, C.c_int32, C.c_int64, C.c_int
_dp = C.POINTER(C.c_double)

HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p)                    # hx_halo_fn
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p)      # hx_allreduce_fn
HALO_END_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)                                        # hx_halo_end_fn
HALO_BEGIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p))  # hx_halo_begin_fn
COMM_ID_BYTES = 128
ERR_BREAKDOWN = -5

SIGNATURES = {
    "hx_version": (C.c_int, []),
    "hx_last_error": (C.c_char_p, []),
    "hx_device_info": (C.c_int, [C.c_char_p, _int, C.POINTER(_int), C.POINTER(C.c_size_t)]),
    "hx_zaxpy": (C.c_int, [_i64, _dp, _vp, _vp, _vp]),
    "hx_zscal": (C.c_int, [_i64, _dp, _vp, _vp]),
    "hx_zlincomb": (C.c_int, [_i64, _int, _dp, C.POINTER(_vp), _vp, _vp]),
    "hx_zscatter_axpy": (C.c_int, [_i64, _dp, _vp, _vp, _vp, _vp]),
    "hx_csr_locate_slots": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hx_zgather": (C.c_int, [_i64, _vp, _vp, _vp, _vp]),
    "hx_panel_dot_ws": (C.c_size_t, [_int]),
    "hx_zpanel_dot": (C.c_int, [_i64, _int, _vp, _i64, _vp, _vp, _vp, _vp]),
    "hx_zpanel_dot_c64": (C.c_int, [_i64, _int, _vp, _i64, _vp, _vp, _vp, _vp]),
    "hx_zpanel_sub": (C.c_int, [_i64, _int, _vp, _i64, _vp, _vp, _vp]),
    "hx_zpanel_sub_c64": (C.c_int, [_i64, _int, _vp, _i64, _vp, _vp, _vp]),
    "hx_zpanel_mul": (C.c_int, [_i64, _int, _int, _vp, _i64, _vp, _vp, _i64, _vp]),
    "hx_zsparse_dots": (C.c_int, [_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hx_zsparse_axpys": (C.c_int, [_int, _vp, _vp, _vp, _vp, _dp, _vp, _int, _vp]),
    "hx_zthreshold_pack": (C.c_int, [_i64, _vp, C.c_double, _i32, _vp, _vp, _vp, _vp]),
    "hx_csr_spmv_z": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _dp, _dp, _int, _vp]),
    "hx_sell_create": (C.c_int, [C.POINTER(_vp), _i32, _vp, _vp, _vp, _int, _int, _vp]),
    "hx_sell_index_bits": (C.c_int, [_vp]),
    "hx_sell_wide_slices": (C.c_int, [_vp]),
    "hx_sell_stencil_slices": (C.c_int, [_vp]),
    "hx_sell_stencil_entries": (_i64, [_vp]),
    "hx_sell_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i64)]),
    "hx_sell_tail_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i32)]),
    "hx_sell_spmv_z": (C.c_int, [_vp, _vp, _vp, _dp, _dp, _vp]),
    "hx_sell_spmv_c64": (C.c_int, [_vp, _vp, _vp, C.c_float, C.c_float, _vp]),
    "hx_sell_smooth_z": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_double, _vp, C.c_double, _vp]),
    "hx_sell_smooth_c64": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, _vp, C.c_float, _vp]),
    "hx_sell_destroy": (C.c_int, [_vp]),
    "hx_profile_sell_spmv": (C.c_int, [_int, _i32]),
    "hx_profile_sell_skipped": (_i64, [_int]),
    "hx_profile_sell_spmv_read": (C.c_int, [C.POINTER(_i64), C.POINTER(C.c_double)]),
    "hx_profile_sell_read": (C.c_int, [_int, C.POINTER(_i64), C.POINTER(C.c_double)]),
    "hx_build_v2e": (C.c_int, [_i32, _i32, _int, _vp, _vp, _vp, _vp]),
    "hx_p1_pattern_count": (C.c_int, [_i32, _i32, _int, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    "hx_p1_pattern_fill": (C.c_int, [_i32, _int, _vp, _vp, _vp, _vp, _vp]),
    "hx_assemble_p1_stiff_mass": (C.c_int, [_int, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    "hx_assemble_p1_robin": (C.c_int, [_int, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    "hx_csr_apply_dirichlet": (C.c_int, [_i32, _vp, _vp, _vp, _vp, C.c_double, _vp]),
    "hx_assemble_p1_load": (C.c_int, [_int, _int, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _dp, _vp, _i32, _vp, _vp]),
    "hx_assemble_p1_load_rows": (C.c_int, [_int, _int, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dp, _vp, _vp, _vp]),
    "hx_assemble_p2_stiff_mass": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    "hx_assemble_p2_robin": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    "hx_assemble_p2_load": (C.c_int, [_int, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dp, _int, _vp, _vp, _vp]),
    "hx_assemble_p2_load_cells": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _dp, _int, _vp, _vp, _vp]),
    "hx_p1_cell_volumes": (C.c_int, [_int, _i32, _vp, _vp, _vp, _vp]),
    "hx_p1_locate_points": (C.c_int, [_int, _i32, _vp, _vp, _int, _vp, C.c_double, _vp, _vp]),
    "hx_sptrsv_create": (C.c_int, [C.POINTER(_vp), _i32, _int, _vp, _vp, _vp, _vp]),
    "hx_sptrsv_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i64)]),
    "hx_sptrsv_solve": (C.c_int, [_vp, _vp, _vp, _vp]),
    "hx_sptrsv_destroy": (C.c_int, [_vp]),
    "hx_splu_analyse": (C.c_int, [C.POINTER(_vp), _i32, _vp, _vp, _vp, _int, _int, _int]),
    "hx_splu_info": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(C.c_double)]),
    "hx_splu_symbolic": (C.c_int, [_vp, C.c_char_p, _vp, _i64, C.POINTER(_i64)]),
    "hx_splu_factor": (C.c_int, [_vp, _vp, _vp]),
    "hx_splu_solve": (C.c_int, [_vp, _vp, _vp, _int, _vp]),
    "hx_splu_destroy": (C.c_int, [_vp]),
    "hx_csr_jacobi_z": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _vp]),
    "hx_csr_spmv_c64": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp]),
    "hx_csr_jacobi_c64": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp]),
    "hx_zdiag_scale": (C.c_int, [_i64, C.c_double, _vp, _vp, _vp, _vp]),
    "hx_amg_create": (C.c_int, [C.POINTER(_vp), _int, _int, _int, C.c_double]),
    "hx_amg_set_level": (C.c_int, [_vp, _int, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i64, _vp,
                                   _vp, _vp]),
    "hx_amg_set_smoother": (C.c_int, [_vp, _int, C.c_double]),
    "hx_amg_set_real_transfers": (C.c_int, [_vp, _int]),
    "hx_amg_set_coarse": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp]),
    "hx_amg_set_level_c": (C.c_int, [_vp, _int, _i32, _i64, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i64, _vp,
                                     _vp, _vp]),
    "hx_amg_set_coarse_c": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp]),
    "hx_amg_enable_graph": (C.c_int, [_vp, _int]),
    "hx_amg_graph_active": (C.c_int, [_vp]),
    "hx_amg_vcycle": (C.c_int, [_vp, _vp, _vp, _vp]),
    "hx_amg_destroy": (C.c_int, [_vp]),
    "hx_gmres_ws": (C.c_size_t, [_i64, _int]),
    "hx_gmres_ws_recycling": (C.c_size_t, [_i64, _int]),
    "hx_gmres": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _int, _int, _vp,
                           C.POINTER(_int), C.POINTER(C.c_double), _vp]),
    "hx_cocg_ws": (C.c_size_t, [_i64]),
    "hx_cocg_backward_error": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "hx_amg_next_solve_uses_x": (C.c_int, [_vp]),
    "hx_cocg": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _int, _vp,
                          C.POINTER(_int), C.POINTER(C.c_double), _vp]),
    "hx_cocg_sharded": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _int, _vp, _vp, _vp,
                                  C.POINTER(_int), C.POINTER(C.c_double), _vp]),
    "hx_amg_set_level_sell": (C.c_int, [_vp, _int, _vp]),
    "hx_amg_enable_recycling": (C.c_int, [_vp, _i64, _int, _int]),
    "hx_gmres_hessenberg": (C.c_int, [_vp, C.POINTER(_int), _vp]),
    "hx_amg_set_recycle": (C.c_int, [_vp, _int, _vp, _vp, _vp, _vp]),
    "hx_amg_set_sharded": (C.c_int, [_vp, _int, _vp, HALO_FN, ALLREDUCE_FN, _vp]),
    "hx_amg_set_level_deep": (C.c_int, [_vp, _int, _int, _i32, _i64, _i32, _i64, _vp, _vp, _vp]),
    "hx_amg_set_halo_overlap": (C.c_int, [_vp, HALO_BEGIN_FN, HALO_END_FN]),
    "hx_amg_set_prolongator_ghost_rows": (C.c_int, [_vp, C.c_int]),
    "hx_block_width": (C.c_int, []),
    "hx_zblock_pack": (C.c_int, [_i64, _int, _vp, _i64, _vp, _vp]),
    "hx_zblock_unpack": (C.c_int, [_i64, _int, _vp, _vp, _i64, _vp]),
    "hx_sell_spmm_z": (C.c_int, [_vp, _vp, _vp, _dp, _dp, _vp]),
    "hx_cocg_block_ws": (C.c_size_t, [_i64]),
    "hx_cocg_block": (C.c_int, [_vp, _i32, _vp, _int, _vp, _vp, C.c_double, _int, _vp, C.POINTER(_int), C.POINTER(C.c_double), _vp]),
    "hx_zsmall_gemv": (C.c_int, [_int, _int, _vp, _vp, _vp, _vp]),
    "hx_zroll": (C.c_int, [_i64, _i64, _vp, _vp, _vp]),
    "hx_zdense_inverse": (C.c_int, [_int, _vp, _vp]),
    "hx_pool_set_sharing": (C.c_int, [_int]),
    "hx_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "hx_get_option": (C.c_int, [C.c_char_p, C.c_char_p, _int]),
    "hx_pool_trim": (C.c_int, []),
    "hx_pool_ordering_stats": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hx_pool_stats": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                C.POINTER(C.c_int64), C.c_int]),
    "hx_comm_halo_begin": (C.c_int, [_vp, _int, _vp, _int, _vp, C.POINTER(_vp)]),
    "hx_comm_halo_end": (C.c_int, [_vp, _int, _vp]),
    "hx_comm_overlapped": (_i64, [_vp]),
    "hx_comm_stream_switches": (_i64, [_vp]),
    "hx_comm_unique_id": (C.c_int, [_vp, _int]),
    "hx_comm_create": (C.c_int, [C.POINTER(_vp), _int, _int, C.c_char_p, _int]),
    "hx_comm_destroy": (C.c_int, [_vp]),
    "hx_comm_library": (C.c_char_p, []),
    "hx_comm_set_plan": (C.c_int, [_vp, _int, _int, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    "hx_comm_exchange": (C.c_int, [_vp, _int, _vp, _vp, _int, _vp]),
    "hx_comm_halo": (C.c_int, [_vp, _int, _vp, _int, _vp]),
    "hx_comm_allreduce": (C.c_int, [_vp, _vp, _i64, _int, _int, _vp]),
    "hx_comm_allgather": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "hx_comm_counters": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), _int]),
    "hx_shape_derivative_facets": (C.c_int, [_int, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _int, _vp, _vp, _vp]),
    "hx_sa_strength_diag": (C.c_int, [_i32, _vp, _vp, _vp, C.c_double, _vp, _vp]),
    "hx_sa_power_step": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hx_gmres_sharded": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double, _int, _int, _vp, _vp, _vp,
                                   C.POINTER(_int), C.POINTER(C.c_double), _vp]),
    "hx_csr_diagonal_z": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp]),
    "hx_sa_filter_count": (C.c_int, [_i32, _vp, _vp, _vp, C.c_double, C.c_double, _vp, _vp, C.POINTER(_i64), _vp]),
    "hx_sa_filter_fill": (C.c_int, [_i32, _vp, _vp, _vp, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp]),
    "hx_sa_power_rho": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _int, C.POINTER(C.c_double), _vp]),
    "hx_sa_prolongator_count": (C.c_int, [_i32, _vp, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    "hx_sa_prolongator_fill": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp, _vp]),
    "hx_csr_transpose_d": (C.c_int, [_i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "hx_spgemm_symbolic": (C.c_int, [C.POINTER(_vp), _i32, _vp, _vp, _i32, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    "hx_spgemm_numeric": (C.c_int, [_vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _int, _vp, _vp, _vp]),
    "hx_spgemm_destroy": (C.c_int, [_vp]),
    "hx_host_aggregate": (C.c_int, [_i32, _vp, _vp, _vp, _vp, C.POINTER(_i32)]),
    "hx_host_aggregate_f32": (C.c_int, [_i32, _vp, _vp, _vp, _vp, C.POINTER(_i32)]),
    "hx_host_sa_filter": (C.c_int, [_i32, _vp, _vp, _vp, _int, C.c_double, C.c_double, _vp, _vp, _vp]),
    "hx_host_galerkin_rap": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int]),
    "hx_host_csr_diagonal": (C.c_int, [_i32, _vp, _vp, _vp, _vp]),
    "hx_host_power_rho": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _int, C.POINTER(C.c_double)]),
    "hx_host_sa_prolongator": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, C.c_double, _vp, _vp, _vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(lib, _name)      # AttributeError here = the library does not export what hx.h declares
    _f.restype = _res
    _f.argtypes = _args


class HxError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise HxError("libhx error %d: %s" % (rc, lib.hx_last_error().decode(errors="replace")))


def require_gpu():
    if not torch.cuda.is_available():
        raise HxError("no MI355X visible: the helmholtz_x hot path runs on the GPU only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The caller's current stream as the C-ABI's hx_stream.  (torch.cuda.current_stream() builds a Stream object per call --
    8.6 us, 1850 calls in a fixed-point call of the reference's mesh = 11 % of it; the raw query is what torch's own compiled
    code uses.)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def set_option(name, value=None):
    """Run-time option of libhx (include/hx.h: "Run-time options"); value None removes the override."""
    check(lib.hx_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name, default=None):
    buf = C.create_string_buffer(256)
    return buf.value.decode() if lib.hx_get_option(name.encode(), buf, 256) == 0 else default


def device_empty(shape, dtype, device=None):
    """torch.empty on the GPU; when torch's allocator runs out of memory the blocks libhx keeps cached (which torch cannot see)
    are returned to the driver and the allocation is tried once more."""
    device = device if device is not None else require_gpu()
    try:
        return torch.empty(shape, dtype=dtype, device=device)
    except torch.cuda.OutOfMemoryError:
        check(lib.hx_pool_trim())
        torch.cuda.empty_cache()
        return torch.empty(shape, dtype=dtype, device=device)


_SHARING = None          # ranks of the process group that use THIS rank's device (count_device_sharing)


def count_device_sharing(group=None):
    """How many ranks of ``group`` run on the device of this rank: every rank publishes (host, PCI domain:bus:device of its
    current device) and counts its equals.  Collective; tells the device-buffer pool (hx_pool_set_sharing) -- a launcher that
    masks one GPU per rank (ROCR_VISIBLE_DEVICES) leaves LOCAL_WORLD_SIZE / visible devices = 8 although nobody shares."""
    global _SHARING
    import socket
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and torch.cuda.is_available()):
        return 1
    p = torch.cuda.get_device_properties(torch.cuda.current_device())
    mine = (socket.gethostname(), getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", torch.cuda.current_device()),
            getattr(p, "pci_device_id", 0))
    everyone = [None] * dist.get_world_size(group)
    dist.all_gather_object(everyone, mine, group=group)
    _SHARING = max(1, sum(1 for e in everyone if tuple(e) == mine))
    check(lib.hx_pool_set_sharing(_SHARING))
    return _SHARING


def shares_device():
    """More ranks than GPUs (rehearsals on a one-GPU box): caches are trimmed when solver objects are destroyed."""
    import os
    if _SHARING is not None:
        return _SHARING > 1
    try:
        return int(os.environ.get("LOCAL_WORLD_SIZE", "1")) > max(torch.cuda.device_count(), 1)
    except ValueError:
        return False


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def z2(z):
    z = complex(z)
    return (C.c_double * 2)(z.real, z.imag)


def nptr(a):
    """Host address of a contiguous numpy array."""
    return a.ctypes.data_as(C.c_void_p)
