"""ctypes binding of libtsx (include/tsx.h).  Plumbing only: the product is the HIP library.

The library must exist in-tree (tenstream_amd/lib/libtsx.so, built by __graft_entry__.build()
or `make -C tenstream_amd/csrc`); there is no Python/CPU fallback -- a missing library or a
missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TSX_LIB: another build of the library (A/B scripts: scripts/ab_lib.sh); the default is the in-tree one
LIB_PATH = os.environ.get("TSX_LIB") or os.path.join(_HERE, "lib", "libtsx.so")

TSX_HOST, TSX_DEVICE = 0, 1
TSX_SOLVER_3_10, TSX_SOLVER_8_16 = 310, 816
TSX_PC_NONE, TSX_PC_COLUMN, TSX_PC_ZEBRA = 0, 1, 2
TSX_ERR_NO_DEVICE = 2


class TsxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libtsx error {code}: {msg}")
        self.code = code


class Grid(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "solver_id", "Nz", "xm", "ym", "xs", "ys", "glob_xm", "glob_ym", "rank", "nranks",
        "neigh_w", "neigh_e", "neigh_s", "neigh_n", "device", "force_halo")]


class KspOpts(C.Structure):
    _fields_ = [("rtol", C.c_double), ("atol", C.c_double), ("dtol", C.c_double), ("maxit", C.c_int32),
                ("pc", C.c_int32), ("pc_sweeps", C.c_int32), ("check_every", C.c_int32), ("fp32_directions", C.c_int32),
                ("pc_coeff_fp16", C.c_int32), ("skip_complete_initial_run", C.c_int32), ("explicit_solver", C.c_int32),
                ("accept_incomplete_solve", C.c_int32), ("initial_guess_zero", C.c_int32)]


class KspResult(C.Structure):
    _fields_ = [("reason", C.c_int32), ("niter", C.c_int32), ("rnorm0", C.c_double), ("rnorm", C.c_double),
                ("res_hist", C.c_double * 100), ("nhist", C.c_int32), ("solve_ms", C.c_float),
                ("import_ms", C.c_float), ("export_ms", C.c_float)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)),
                         C.POINTER(C.c_size_t), C.POINTER(C.c_int))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)

# every symbol include/tsx.h declares (tests check the .so exports exactly these)
SYMBOLS = (
    "tsx_last_error", "tsx_version", "tsx_device_count", "tsx_abi_sizes", "tsx_create", "tsx_destroy", "tsx_default_ksp_opts",
    "tsx_determine_ksp_tolerances", "tsx_set_stream", "tsx_comm_unique_id", "tsx_comm_init", "tsx_comm_set_callbacks", "tsx_comm_peer_export", "tsx_comm_peer_attach", "tsx_comm_peer_selftest", "tsx_comm_peer_disable", "tsx_comm_peer_set_fences", "tsx_comm_peer_reset",
    "tsx_diff_set_coeffs", "tsx_lut_set_diffuse", "tsx_lut_load_diffuse_mmap4", "tsx_diff_set_optprop",
    "tsx_diff_get_coeffs", "tsx_pprts_set_angles", "tsx_lut_set_direct", "tsx_lut_load_direct_mmap4", "tsx_pprts_set_optprop", "tsx_pprts_set_optical_properties", "tsx_pprts_solve",
    "tsx_pprts_zero_guess", "tsx_pprts_get_result", "tsx_pprts_get_field", "tsx_diff_apply", "tsx_diff_solve", "tsx_diff_pc_apply", "tsx_bench_kernel", "tsx_algorithmic_bytes",
    "tsx_probe_copy_bandwidth", "tsx_opp_get_coeff", "tsx_opp_get_info", "tsx_pprts_select_solution", "tsx_dedup_info", "tsx_pc_info", "tsx_flow_info", "tsx_pprts_set_direct_tolerances",
    "tsx_probe_bandwidth", "tsx_diff_apply_r", "tsx_diff_solve_r", "tsx_dir_set_coeffs", "tsx_dir_solve", "tsx_setup_b_solar", "tsx_setup_b_thermal",
    "tsx_debug_code_read", "tsx_log_enable", "tsx_log_get", "tsx_pool_stats",
)

PEER_BLOB_BYTES = 192   # TSX_PEER_BLOB_BYTES
_lib = None


def load():
    """dlopen libtsx.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tenstream_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, ip, dp = C.c_void_p, C.c_int, C.POINTER(C.c_double)
    lib.tsx_last_error.restype = C.c_char_p
    lib.tsx_version.restype = ip
    lib.tsx_device_count.restype = ip
    lib.tsx_create.argtypes = [C.POINTER(Grid), C.POINTER(vp)]
    lib.tsx_destroy.argtypes = [vp]
    lib.tsx_default_ksp_opts.argtypes = [C.POINTER(KspOpts)]
    lib.tsx_default_ksp_opts.restype = None
    lib.tsx_determine_ksp_tolerances.argtypes = [vp, C.c_double, dp, dp, C.POINTER(C.c_int32)]
    lib.tsx_set_stream.argtypes = [vp, vp]
    lib.tsx_comm_unique_id.argtypes = [vp]
    lib.tsx_comm_init.argtypes = [vp, vp]
    lib.tsx_comm_set_callbacks.argtypes = [vp, EXCHANGE_FN, ALLREDUCE_FN, vp]
    lib.tsx_comm_peer_export.argtypes = [vp, vp]
    lib.tsx_comm_peer_attach.argtypes = [vp, vp]
    lib.tsx_comm_peer_selftest.argtypes = [vp, ip, dp]
    lib.tsx_comm_peer_disable.argtypes = [vp]
    lib.tsx_comm_peer_set_fences.argtypes = [vp, ip]
    lib.tsx_comm_peer_reset.argtypes = [vp]
    lib.tsx_diff_set_coeffs.argtypes = [vp, vp, ip, vp, vp, vp, vp, ip]
    lib.tsx_lut_set_diffuse.argtypes = [vp, vp, C.c_int32, C.c_int64, C.c_int32, vp, vp, ip]
    lib.tsx_lut_load_diffuse_mmap4.argtypes = [vp, C.c_char_p]
    lib.tsx_diff_set_optprop.argtypes = [vp, vp, vp, vp, vp, C.c_double, vp, vp, vp, vp, ip]
    lib.tsx_diff_get_coeffs.argtypes = [vp, vp, ip]
    lib.tsx_pprts_set_angles.argtypes = [vp, C.c_double, C.c_double]
    lib.tsx_lut_set_direct.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, vp, vp, ip]
    lib.tsx_lut_load_direct_mmap4.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.tsx_pprts_set_optprop.argtypes = [vp, vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp, vp, vp, vp, vp, vp, vp, ip]
    lib.tsx_pprts_set_optical_properties.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_double, C.c_double, ip, ip]
    lib.tsx_pprts_solve.argtypes = [vp, C.c_double, ip, C.POINTER(KspOpts), C.POINTER(KspResult)]
    lib.tsx_pprts_zero_guess.argtypes = [vp]
    lib.tsx_pprts_select_solution.argtypes = [vp, C.c_int32]
    lib.tsx_pprts_get_result.argtypes = [vp, vp, vp, vp, vp, ip]
    lib.tsx_pprts_get_field.argtypes = [vp, ip, vp, ip]
    lib.tsx_diff_apply.argtypes = [vp, vp, vp, ip]
    lib.tsx_diff_solve.argtypes = [vp, vp, vp, ip, C.POINTER(KspOpts), C.POINTER(KspResult)]
    lib.tsx_diff_pc_apply.argtypes = [vp, vp, vp, ip, ip, ip, ip]
    lib.tsx_bench_kernel.argtypes = [vp, ip, ip, C.POINTER(C.c_float)]
    lib.tsx_algorithmic_bytes.argtypes = [vp, ip, dp]
    lib.tsx_probe_copy_bandwidth.argtypes = [vp, C.c_size_t, ip, dp]
    lib.tsx_probe_bandwidth.argtypes = [vp, C.c_size_t, ip, dp]
    lib.tsx_log_enable.argtypes = [vp, ip]
    lib.tsx_pool_stats.argtypes = [ip, C.POINTER(C.c_int64)]
    lib.tsx_log_get.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.POINTER(C.c_int64), dp]
    lib.tsx_debug_code_read.argtypes = [ip, ip, C.c_longlong, C.c_longlong, vp, C.POINTER(C.c_ulonglong)]
    lib.tsx_opp_get_coeff.argtypes = [vp] + [C.c_float] * 6 + [ip, ip, ip, ip, vp]
    lib.tsx_dedup_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.tsx_pprts_set_direct_tolerances.argtypes = [vp, C.c_double, C.c_double, C.c_int32]
    lib.tsx_pc_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    if hasattr(lib, "tsx_flow_info"):   # (an older build loaded through TSX_LIB for an A/B run has none)
        lib.tsx_flow_info.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.tsx_opp_get_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp]
    lib.tsx_diff_apply_r.argtypes = [vp, vp, vp, ip, ip]
    lib.tsx_diff_solve_r.argtypes = [vp, vp, vp, ip, ip, C.POINTER(KspOpts), C.POINTER(KspResult)]
    lib.tsx_dir_set_coeffs.argtypes = [vp, vp, vp, ip, vp, vp, vp, vp, C.c_double, C.c_double, ip]
    lib.tsx_dir_solve.argtypes = [vp, C.c_double, vp, ip, ip, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_int32), dp,
                                  C.POINTER(C.c_int32)]
    lib.tsx_setup_b_solar.argtypes = [vp, vp, vp, vp, ip, ip]
    lib.tsx_setup_b_thermal.argtypes = [vp, vp, vp, vp, vp, C.c_double, C.c_double, vp, ip, ip]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise TsxError(rc, load().tsx_last_error().decode())
