"""ctypes binding of include/fmmbem.h (libfmmbem_hip.so).

The library is hand-written HIP for gfx950 plus host C++; it is loaded from this directory (built
in-tree by `make -C csrc` / __graft_entry__.build()).  There is no fallback: if the shared object
is missing, importing this module raises, and if no GPU is visible every execute returns
FMMBEM_ERR_NO_DEVICE.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMMBEM_LIB") or os.path.join(_HERE, "libfmmbem_hip.so")   # FMMBEM_LIB: A/B builds of the same ABI

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_TREE, ERR_UNSUPPORTED, ERR_IO = range(8)
PMAX = 16
MAX_QUAD = 79          # FMMBEM_MAX_QUAD
KERNEL_LAPLACE_BEM, KERNEL_STOKES_BEM = 0, 1
EVAL_FMM, EVAL_LOCAL, EVAL_BLOCK_DIAGONAL = 0, 1, 2
L2L_COMPLETE, L2L_REFERENCE = 0, 1
BC_POTENTIAL, BC_NORMAL_DERIV = 0, 1


class Options(C.Structure):
    """fmmbem_options"""
    _fields_ = [("kernel", C.c_int32), ("p_max", C.c_int32), ("quad_k", C.c_int32), ("theta", C.c_double),
                ("ncrit", C.c_uint32), ("sparse_local", C.c_int32), ("host_only", C.c_int32),
                ("device", C.c_int32), ("shard_rank", C.c_int32), ("shard_world", C.c_int32),
                ("quad_k_fine", C.c_int32), ("evaluator", C.c_int32), ("mu", C.c_double),
                ("shard_upward", C.c_int32), ("l2l_rule", C.c_int32), ("near_stream_fraction", C.c_double), ("n_devices", C.c_int32), ("devices", C.c_int32 * 8)]


class Stats(C.Structure):
    """fmmbem_stats"""
    _fields_ = ([(n, C.c_int64) for n in (
        "n_panels", "n_boxes", "n_leaves", "n_levels", "near_nnz", "near_nnz_total", "p2p_pairs", "m2l_pairs",
        "m2l_pairs_owned", "m2m_ops", "l2l_ops", "p2m_leaves", "l2p_leaves", "m2l_classes", "owned_leaf_begin",
        "owned_leaf_end", "owned_row_begin", "owned_row_end", "near_bytes")] +
        [("expansions_active", C.c_int32), ("last_p", C.c_int32)] +
        [(n, C.c_double) for n in ("build_host_ms", "build_assemble_ms", "ms_total", "ms_gather", "ms_near",
                                   "ms_scatter", "ms_p2m", "ms_m2m", "ms_mh", "ms_m2l", "ms_l2l", "ms_l2p")] +
        [("timed_executes", C.c_int64), ("l2l_reference_omitted", C.c_int64), ("m2l_items", C.c_int64),
         ("m2l_passes", C.c_int64), ("near_side_entries", C.c_int64), ("m2l_kernel", C.c_int32), ("expansion_slots", C.c_int32),
         ("rot_nop_orders", C.c_int64), ("tree_coder_levels", C.c_int32), ("n_devices", C.c_int32), ("geometry_shared", C.c_int32), ("near_recomputed_pairs", C.c_int64)])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class SolverOpts(C.Structure):
    """fmmbem_solver_options"""
    _fields_ = [("residual", C.c_double), ("max_iters", C.c_int32), ("restart", C.c_int32), ("max_p", C.c_int32),
                ("p_min", C.c_int32), ("variable_p", C.c_int32), ("relax_type", C.c_int32), ("order_rule", C.c_int32),
                ("flexible", C.c_int32), ("initial_p", C.c_int32)]


class Preconditioner(C.Structure):
    """fmmbem_preconditioner"""
    _fields_ = [("kind", C.c_int32), ("reciprocals", C.c_void_p), ("inner_plan", C.c_void_p), ("inner", SolverOpts)]


class SolverLog(C.Structure):
    """fmmbem_solver_log"""
    _fields_ = [("iterations", C.c_int32), ("residual", C.c_double), ("seconds", C.c_double), ("capacity", C.c_int32),
                ("p", C.POINTER(C.c_int32)), ("resid", C.POINTER(C.c_double))]


PC_IDENTITY, PC_DIAGONAL, PC_INNER_PLAN = 0, 1, 2
ORDER_GMRES, ORDER_GMRES_STOKES, ORDER_FGMRES, ORDER_FGMRES_STOKES = 0, 1, 2, 3


class FmmBemError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("fmmbem status %d (%s): %s" % (status, _status_string(status), text))
        self.status = status


_lib = None

# every symbol include/fmmbem.h declares (tests check that the library exports all of them)
SYMBOLS = (
    "fmmbem_options_default", "fmmbem_plan_create", "fmmbem_plan_destroy", "fmmbem_plan_execute",
    "fmmbem_mgs_column_device", "fmmbem_mgs_scratch_doubles", "fmmbem_plan_execute_device", "fmmbem_plan_exchange_doubles", "fmmbem_plan_exchange_counts", "fmmbem_plan_upward_device",
    "fmmbem_plan_downward_device", "fmmbem_plan_near_split_device", "fmmbem_plan_near_device", "fmmbem_plan_set_result_slices", "fmmbem_plan_shard_rows",
    "fmmbem_plan_assemble_slices_device", "fmmbem_plan_set_timing", "fmmbem_plan_set_graphs", "fmmbem_plan_stats",
    "fmmbem_plan_get_perm", "fmmbem_plan_get_boxes", "fmmbem_plan_get_pairs", "fmmbem_plan_get_near_row",
    "fmmbem_plan_get_expansions", "fmmbem_plan_get_diagonal", "fmmbem_kernel_entries", "fmmbem_mesh_unit_sphere", "fmmbem_mesh_red_blood_cell", "fmmbem_mesh_red_blood_cells", "fmmbem_mesh_read_msh",
    "fmmbem_mesh_read_vert_face", "fmmbem_mesh_write_vert_face", "fmmbem_quadrature", "fmmbem_status_string", "fmmbem_last_error",
    "fmmbem_version", "fmmbem_plan_create_like", "fmmbem_host_register", "fmmbem_host_unregister", "fmmbem_solver_options_default", "fmmbem_gmres_device", "fmmbem_gmres",
    "fmmbem_ops_create", "fmmbem_ops_destroy", "fmmbem_ops_slots", "fmmbem_ops_p2m", "fmmbem_ops_m2m", "fmmbem_ops_m2l", "fmmbem_ops_l2l", "fmmbem_ops_l2p",
)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `make -C %s` (hipcc --offload-arch=gfx950). "
                          "There is no CPU fallback." % (LIB_PATH, os.path.join(_HERE, "csrc")))
    # One HIP runtime per process: torch ships its own libamdhip64.so (SONAME libamdhip64.so.7). Loading
    # torch FIRST makes our NEEDED "libamdhip64.so.7" resolve to that already-loaded copy, so device
    # pointers of torch tensors and torch streams are valid inside this library.  (Loaded the other way
    # round the process ends up with two runtimes and torch then reports "No HIP GPUs are available".)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64p = C.c_void_p, C.c_int, C.POINTER(C.c_int64)
    L.fmmbem_options_default.argtypes = [C.POINTER(Options)]
    L.fmmbem_options_default.restype = None
    L.fmmbem_plan_create.argtypes = [C.POINTER(Options), C.c_size_t, vp, vp, C.POINTER(vp)]
    L.fmmbem_plan_destroy.argtypes = [vp]
    L.fmmbem_plan_destroy.restype = None
    L.fmmbem_plan_execute.argtypes = [vp, i32, vp, vp]
    L.fmmbem_plan_execute_device.argtypes = [vp, i32, vp, vp, vp]
    L.fmmbem_plan_near_device.argtypes = [vp, vp, vp, vp]
    L.fmmbem_plan_near_split_device.argtypes = [vp, vp, vp]
    L.fmmbem_plan_exchange_doubles.argtypes = [vp, i32, C.POINTER(C.c_size_t)]
    L.fmmbem_plan_exchange_counts.argtypes = [vp, i32, vp, vp]
    L.fmmbem_mgs_column_device.argtypes = [C.c_int64, vp, vp, C.c_int64, i32, vp, vp, vp, vp]
    L.fmmbem_mgs_scratch_doubles.argtypes = [i32]
    L.fmmbem_plan_upward_device.argtypes = [vp, i32, vp, vp, vp]
    L.fmmbem_plan_downward_device.argtypes = [vp, i32, vp, vp, vp]
    L.fmmbem_plan_set_result_slices.argtypes = [vp, i32]
    L.fmmbem_plan_shard_rows.argtypes = [vp, vp]
    L.fmmbem_plan_assemble_slices_device.argtypes = [vp, vp, C.c_size_t, vp, vp]
    L.fmmbem_plan_set_timing.argtypes = [vp, i32]
    L.fmmbem_plan_set_graphs.argtypes = [vp, i32]
    L.fmmbem_plan_stats.argtypes = [vp, C.POINTER(Stats)]
    L.fmmbem_plan_get_perm.argtypes = [vp, vp]
    L.fmmbem_plan_get_boxes.argtypes = [vp] * 8
    L.fmmbem_plan_get_pairs.argtypes = [vp, i32, vp, i64p]
    L.fmmbem_plan_get_near_row.argtypes = [vp, C.c_int64, vp, vp, i64p]
    L.fmmbem_plan_get_expansions.argtypes = [vp, i32, i32, vp]
    L.fmmbem_plan_get_diagonal.argtypes = [vp, vp]
    L.fmmbem_kernel_entries.argtypes = [C.POINTER(Options), C.c_size_t, vp, vp, vp, vp]
    L.fmmbem_mesh_unit_sphere.argtypes = [i32, vp, C.POINTER(C.c_size_t)]
    L.fmmbem_mesh_red_blood_cell.argtypes = [i32, vp, C.POINTER(C.c_size_t)]
    L.fmmbem_mesh_red_blood_cells.argtypes = [i32, i32, vp, vp, C.POINTER(C.c_size_t)]
    L.fmmbem_mesh_read_msh.argtypes = [C.c_char_p, vp, C.POINTER(C.c_size_t)]
    L.fmmbem_mesh_read_vert_face.argtypes = [C.c_char_p, C.c_char_p, vp, C.POINTER(C.c_size_t)]
    L.fmmbem_mesh_write_vert_face.argtypes = [C.c_char_p, C.c_char_p, vp, C.c_size_t]
    L.fmmbem_quadrature.argtypes = [i32, vp, vp, C.POINTER(i32)]
    L.fmmbem_status_string.argtypes = [i32]
    L.fmmbem_status_string.restype = C.c_char_p
    L.fmmbem_last_error.restype = C.c_char_p
    L.fmmbem_version.restype = i32
    L.fmmbem_host_register.argtypes = [vp, C.c_size_t]
    L.fmmbem_host_unregister.argtypes = [vp]
    L.fmmbem_solver_options_default.argtypes = [C.POINTER(SolverOpts)]
    L.fmmbem_solver_options_default.restype = None
    L.fmmbem_gmres_device.argtypes = [vp, C.POINTER(SolverOpts), vp, vp, C.POINTER(Preconditioner), C.POINTER(SolverLog), vp]
    L.fmmbem_plan_create_like.argtypes = [vp, vp, C.POINTER(vp)]
    L.fmmbem_gmres.argtypes = [vp, C.POINTER(SolverOpts), vp, vp, C.POINTER(Preconditioner), C.POINTER(SolverLog)]
    L.fmmbem_ops_create.argtypes = [C.POINTER(Options), C.POINTER(vp)]
    L.fmmbem_ops_destroy.argtypes = [vp]
    L.fmmbem_ops_destroy.restype = None
    L.fmmbem_ops_slots.argtypes = [vp]
    L.fmmbem_ops_p2m.argtypes = [vp, i32, C.c_size_t, vp, vp, vp, vp, vp]
    for fn in (L.fmmbem_ops_m2m, L.fmmbem_ops_m2l, L.fmmbem_ops_l2l):
        fn.argtypes = [vp, i32, i32, vp, vp, vp]
    L.fmmbem_ops_l2p.argtypes = [vp, i32, vp, vp, C.c_size_t, vp, vp, vp]
    _lib = L
    return L


def _status_string(status):
    return lib().fmmbem_status_string(status).decode()


def check(status):
    if status != OK:
        raise FmmBemError(status, lib().fmmbem_last_error().decode())
