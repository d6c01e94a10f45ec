"""Host-side mirror of the reference's operator interface for the FMM matvec path.

Reference surface (C++ templates) -> here:
    FMMOptions                              include/FMMOptions.hpp:9-60        -> FMMOptions
    LaplaceSphericalBEM(p, k) / set_p(p)    kernel/LaplaceSphericalBEM.hpp:131-140 -> LaplaceSphericalBEM
    FMM_plan<K>(K, panels, opts)            include/FMM_plan.hpp:34-43         -> FMM_plan(K, panels, opts)
    plan.execute(charges) -> results        include/FMM_plan.hpp:75-90         -> FMM_plan.execute
    plan.kernel() / plan.options()          include/FMM_plan.hpp:66-71, 94-96  -> .kernel() / .options()
Same names, same argument meaning; errors are exceptions (FmmBemError) instead of exit()/printf.
All arithmetic happens in libfmmbem_hip.so on the GPU; this file only marshals arrays.
"""
import ctypes as C
import os

import numpy as np

from . import _capi


class FMMOptions:
    """include/FMMOptions.hpp:9-60 (fields the hot path reads)."""

    def __init__(self):
        self.lazy_evaluation = True     # FMMOptions.hpp:41
        self.local_evaluation = False
        self.sparse_local = True        # examples/LaplaceBEM.cpp:81 forces it
        self.block_diagonal = False
        self.theta = 0.5                # DefaultMAC(0.5), FMMOptions.hpp:45
        self.ncrit = 64                 # NCRIT_, FMMOptions.hpp:46
        # not in the reference: True applies exactly the L2L edges the reference's lazy evaluator queues, which on
        # adaptive trees leaves some boxes without their ancestors' far field (include/fmmbem.h, fmmbem_l2l_rule)
        self.reference_l2l = False
        # not in the reference: share of the near-field pairs kept as a matrix; < 1: the rest is recomputed every matvec by the
        # same kernel between its streamed items (include/fmmbem.h, fmmbem_options.near_stream_fraction)
        self.near_stream_fraction = 1.0

    def set_mac_theta(self, theta):     # FMMOptions.hpp:50-52
        self.theta = float(theta)

    def set_max_per_box(self, ncrit):   # FMMOptions.hpp:58-60
        self.ncrit = int(ncrit)

    def max_per_box(self):
        return self.ncrit


class _SingleOperators:
    """The translation operators one at a time -- P2M, M2M, M2L, L2L, L2P of kernel/KernelSkeleton.hpp:62-212 as the BEM kernels
    define them (kernel/LaplaceSphericalBEM.hpp:307-476, kernel/StokesSphericalBEM.hpp:391-530) -- run by the device kernels of
    the matvec on a two-box plan (fmmbem_ops_*, csrc/ops.hip).  Argument order and the "+=" into the last argument are the
    reference's; an expansion is a complex array (slots, p (p + 1) / 2), slots = 2 (Laplace: G, dG/dn) or 4 (Stokes: the stokeslet
    group M[0]); panels are (n, 3, 3) vertex arrays with one boundary-condition flag each (None: all 0)."""
    _ops = None
    device = 0

    def _handle(self):
        if self._ops is None or self._ops[1] != (self.K, getattr(self, "Mu", None), self.device):
            self._close()
            o = _capi.Options()
            _capi.lib().fmmbem_options_default(C.byref(o))
            o.p_max, o.quad_k, o.device = _capi.PMAX, self.K, int(self.device)
            if isinstance(self, StokesSphericalBEM):
                o.kernel, o.mu = _capi.KERNEL_STOKES_BEM, self.Mu
            h = C.c_void_p()
            _capi.check(_capi.lib().fmmbem_ops_create(C.byref(o), C.byref(h)))
            self._ops = (h, (self.K, getattr(self, "Mu", None), self.device))
        return self._ops[0]

    def _close(self):
        if self._ops is not None:
            _capi.lib().fmmbem_ops_destroy(self._ops[0])
            self._ops = None

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass

    def slots(self):
        """expansions per box that P2M writes and L2P reads (multipole_type's size)"""
        return 4 if isinstance(self, StokesSphericalBEM) else 2

    def init_multipole(self, slots=None):
        """a zeroed multipole_type / local_type at the current order (init_multipole / init_local, LaplaceSphericalBEM.hpp:143-156)"""
        return np.zeros((self.slots() if slots is None else slots, self.P * (self.P + 1) // 2), dtype=np.complex128)

    init_local = init_multipole

    def _expansion(self, E, writable):
        S = self.P * (self.P + 1) // 2
        if not (isinstance(E, np.ndarray) and E.dtype == np.complex128 and E.ndim == 2 and E.shape[1] == S and E.flags.c_contiguous):
            raise ValueError("an expansion is a C-contiguous complex128 array (slots, p (p + 1) / 2) at the kernel's current p")
        if writable and not E.flags.writeable:
            raise ValueError("the target expansion must be writable")
        return E

    @staticmethod
    def _panels(panels, bc):
        v = np.ascontiguousarray(panels, dtype=np.float64).reshape(-1, 9)
        b = None
        if bc is not None:
            b = np.ascontiguousarray(bc, dtype=np.uint8)
            if b.shape != (len(v),):
                raise ValueError("one boundary-condition flag per panel")
        return v, b

    def P2M(self, sources, charges, center, M, bc=None):
        """M += the moments of the sources about center (P2M(source, charge, center, M), vectorised over the sources)"""
        v, b = self._panels(sources, bc)
        M = self._expansion(M, True)
        if M.shape[0] != self.slots():
            raise ValueError("P2M writes %d expansions" % self.slots())
        q = np.ascontiguousarray(charges, dtype=np.float64).reshape(-1)
        if q.size != len(v) * (3 if isinstance(self, StokesSphericalBEM) else 1):
            raise ValueError("one charge per source")
        c = np.ascontiguousarray(center, dtype=np.float64)
        vp = C.c_void_p
        _capi.check(_capi.lib().fmmbem_ops_p2m(self._handle(), self.P, len(v), v.ctypes.data_as(vp), None if b is None else b.ctypes.data_as(vp),
                                                q.ctypes.data_as(vp), c.ctypes.data_as(vp), M.ctypes.data_as(vp)))

    def _shift(self, fn, source, target, translation):
        source, target = self._expansion(source, False), self._expansion(target, True)
        if source.shape != target.shape:
            raise ValueError("source and target expansions differ in shape")
        t = np.ascontiguousarray(translation, dtype=np.float64)
        vp = C.c_void_p
        _capi.check(fn(self._handle(), self.P, source.shape[0], source.ctypes.data_as(vp), target.ctypes.data_as(vp), t.ctypes.data_as(vp)))

    def M2M(self, Msource, Mtarget, translation):
        """Mtarget += the source multipole moved by translation = centre(target) - centre(source)"""
        self._shift(_capi.lib().fmmbem_ops_m2m, Msource, Mtarget, translation)

    def M2L(self, Msource, Ltarget, translation):
        self._shift(_capi.lib().fmmbem_ops_m2l, Msource, Ltarget, translation)

    def L2L(self, Lsource, Ltarget, translation):
        self._shift(_capi.lib().fmmbem_ops_l2l, Lsource, Ltarget, translation)

    def L2P(self, L, center, targets, result, bc=None):
        """result += the local expansion about center evaluated at the targets' centroids (L2P(L, center, target, result))"""
        v, b = self._panels(targets, bc)
        L = self._expansion(L, False)
        if L.shape[0] != self.slots():
            raise ValueError("L2P reads %d expansions" % self.slots())
        dof = 3 if isinstance(self, StokesSphericalBEM) else 1
        if not (isinstance(result, np.ndarray) and result.dtype == np.float64 and result.size == len(v) * dof and result.flags.c_contiguous):
            raise ValueError("result must be a C-contiguous float64 array with one value (Stokes: three) per target")
        c = np.ascontiguousarray(center, dtype=np.float64)
        vp = C.c_void_p
        _capi.check(_capi.lib().fmmbem_ops_l2p(self._handle(), self.P, L.ctypes.data_as(vp), c.ctypes.data_as(vp), len(v), v.ctypes.data_as(vp),
                                                None if b is None else b.ctypes.data_as(vp), result.ctypes.data_as(vp)))


class LaplaceSphericalBEM(_SingleOperators):
    """Kernel descriptor: expansion order p and Gauss rule key k (kernel/LaplaceSphericalBEM.hpp:131)."""
    POTENTIAL, NORMAL_DERIV = _capi.BC_POTENTIAL, _capi.BC_NORMAL_DERIV

    def __init__(self, p=5, k=3):
        if not 1 <= int(p) <= _capi.PMAX:
            raise ValueError("p must be in 1..%d" % _capi.PMAX)
        self.P = int(p)
        self.K = int(k)

    def set_p(self, p):                 # kernel/LaplaceSphericalBEM.hpp:137-140
        if not 1 <= int(p) <= _capi.PMAX:
            raise ValueError("p must be in 1..%d" % _capi.PMAX)
        self.P = int(p)


class StokesSphericalBEM(_SingleOperators):
    """Kernel descriptor of kernel/StokesSphericalBEM.hpp:131-141: order p, Gauss key k, viscosity mu and the
    near-regime rule K_fine (ctor default 25; examples/StokesBEM.cpp:128-129,218 uses 19).  Only the VELOCITY
    boundary condition (stokeslet single layer, the operator the solve uses) is built; charges and results are
    Vec<3,double> per panel, i.e. arrays of shape (N, 3)."""
    VELOCITY, TRACTION = 0, 1

    def __init__(self, p=5, k=3, mu=1e-3):
        if not 1 <= int(p) <= _capi.PMAX:
            raise ValueError("p must be in 1..%d" % _capi.PMAX)
        self.P, self.K, self.Mu, self.K_fine = int(p), int(k), float(mu), 25

    def set_p(self, p):
        if not 1 <= int(p) <= _capi.PMAX:
            raise ValueError("p must be in 1..%d" % _capi.PMAX)
        self.P = int(p)

    def set_Kfine(self, k):
        self.K_fine = int(k)


def unit_sphere(recursions, center=(0.0, 0.0, 0.0)):
    """Triangulation::UnitSphere (examples/BEM/Triangulation.hpp:105-121) -> (N, 3, 3) vertex array."""
    n = C.c_size_t(0)
    _capi.check(_capi.lib().fmmbem_mesh_unit_sphere(recursions, None, C.byref(n)))
    v = np.empty((n.value, 3, 3), dtype=np.float64)
    _capi.check(_capi.lib().fmmbem_mesh_unit_sphere(recursions, v.ctypes.data_as(C.c_void_p), C.byref(n)))
    if any(center):
        v += np.asarray(center, dtype=np.float64)
    return v


def quadrature(key):
    """Triangle Gauss rule of examples/BEM/GaussQuadrature.hpp: (barycentric points (n, 3), weights (n,))."""
    pts, w, n = np.empty((_capi.MAX_QUAD, 3)), np.empty(_capi.MAX_QUAD), C.c_int(0)
    _capi.check(_capi.lib().fmmbem_quadrature(int(key), pts.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), C.byref(n)))
    return pts[:n.value].copy(), w[:n.value].copy()


def red_blood_cell(recursions):
    """Triangulation::RedBloodCell, identity rotation, zero shift (examples/BEM/Triangulation.hpp:184-255)."""
    n = C.c_size_t(0)
    _capi.check(_capi.lib().fmmbem_mesh_red_blood_cell(recursions, None, C.byref(n)))
    v = np.empty((n.value, 3, 3), dtype=np.float64)
    _capi.check(_capi.lib().fmmbem_mesh_red_blood_cell(recursions, v.ctypes.data_as(C.c_void_p), C.byref(n)))
    return v


def red_blood_cells(recursions, cells, placement=None):
    """Triangulation::MultipleRedBloodCell (examples/BEM/Triangulation.hpp:260-321).  placement: (cells, 6) rows of
    alpha, beta, gamma, shift x, y, z; None: the reference's own drand48-driven orientations and offsets."""
    pl = None
    if placement is not None:
        placement = np.ascontiguousarray(placement, dtype=np.float64)
        if placement.shape != (cells, 6):
            raise ValueError("placement must be (cells, 6)")
        pl = placement.ctypes.data_as(C.c_void_p)
    n = C.c_size_t(0)
    _capi.check(_capi.lib().fmmbem_mesh_red_blood_cells(recursions, cells, pl, None, C.byref(n)))
    v = np.empty((n.value, 3, 3), dtype=np.float64)
    _capi.check(_capi.lib().fmmbem_mesh_red_blood_cells(recursions, cells, pl, v.ctypes.data_as(C.c_void_p), C.byref(n)))
    return v


def kernel_entries(K, targets, sources, target_bc=None, device=0):
    """K(target_i, source_i) for n panel pairs -- Kernel::operator() of kernel/LaplaceSphericalBEM.hpp:273-297 /
    kernel/StokesSphericalBEM.hpp:377-389, evaluated on the device.  targets, sources: (n, 3, 3) vertices;
    target_bc: n flags (the target's flag selects G vs dG/dn).  Returns (n,) or (n, 3, 3)."""
    t = np.ascontiguousarray(targets, dtype=np.float64).reshape(-1, 9)
    s = np.ascontiguousarray(sources, dtype=np.float64).reshape(-1, 9)
    if t.shape != s.shape:
        raise ValueError("one source per target")
    o = _capi.Options()
    _capi.lib().fmmbem_options_default(C.byref(o))
    o.quad_k, o.device = K.K, int(device)
    stokes = isinstance(K, StokesSphericalBEM)
    if stokes:
        o.kernel, o.mu, o.quad_k_fine = _capi.KERNEL_STOKES_BEM, K.Mu, K.K_fine
    bcp = None
    if target_bc is not None:
        target_bc = np.ascontiguousarray(target_bc, dtype=np.uint8)
        if target_bc.shape != (len(t),):
            raise ValueError("target_bc must have one flag per pair")
        bcp = target_bc.ctypes.data_as(C.c_void_p)
    out = np.empty((len(t), 3, 3) if stokes else len(t))
    _capi.check(_capi.lib().fmmbem_kernel_entries(C.byref(o), len(t), t.ctypes.data_as(C.c_void_p), bcp,
                                                  s.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    return out


def _read(fn, *paths):
    n = C.c_size_t(0)
    args = [os.fsencode(p) for p in paths]
    _capi.check(fn(*args, None, C.byref(n)))
    v = np.empty((n.value, 3, 3), dtype=np.float64)
    _capi.check(fn(*args, v.ctypes.data_as(C.c_void_p), C.byref(n)))
    return v


def read_msh(path):
    """MeshIO::readMsh (examples/BEM/MshReader.hpp:18-94): gmsh v2 ASCII triangles -> (N, 3, 3)."""
    return _read(_capi.lib().fmmbem_mesh_read_msh, path)


def read_vert_face(vert_path, face_path):
    """MeshIO::ReadVertFace (examples/BEM/VertFaceReader.hpp:17-76) -> (N, 3, 3)."""
    return _read(_capi.lib().fmmbem_mesh_read_vert_face, vert_path, face_path)


def write_vert_face(vert_path, face_path, panels):
    """The .vert/.face dump of the generators (examples/BEM/Triangulation.hpp:124-134), readable by read_vert_face."""
    v = np.ascontiguousarray(panels, dtype=np.float64).reshape(-1, 9)
    _capi.check(_capi.lib().fmmbem_mesh_write_vert_face(os.fsencode(vert_path), os.fsencode(face_path),
                                                        v.ctypes.data_as(C.c_void_p), len(v)))


class FMM_plan:
    """FMM_plan<LaplaceSphericalBEM> (include/FMM_plan.hpp:16-128).

    panels: (N, 3, 3) triangle vertices (the arguments of Panel(p0, p1, p2)); bc: N boundary flags.
    p_max: largest order later set through kernel().set_p (defaults to K.P, or 16 if the caller
    intends to relax p upward).
    """

    def __init__(self, K, panels, opts=None, bc=None, p_max=None, device=0, shard=(0, 1), host_only=False,
                 shard_upward=False, devices=None, replicate_upward=False):
        opts = opts if opts is not None else FMMOptions()
        # executor/make_executor.hpp:24-60: lazy_evaluation wins, then local_evaluation, then block_diagonal; the
        # non-lazy upward/interact/downward evaluators compute the same operator as the lazy ones
        evaluator = _capi.EVAL_FMM
        if not opts.lazy_evaluation:
            if opts.local_evaluation:
                evaluator = _capi.EVAL_LOCAL
            elif opts.block_diagonal:
                evaluator = _capi.EVAL_BLOCK_DIAGONAL
        self._K = K
        self._opts = opts
        v = np.ascontiguousarray(panels, dtype=np.float64).reshape(-1, 9)
        self.n = v.shape[0]
        o = _capi.Options()
        _capi.lib().fmmbem_options_default(C.byref(o))
        o.p_max = int(p_max if p_max is not None else K.P)
        o.quad_k = K.K
        o.theta = opts.theta
        o.ncrit = opts.ncrit
        o.sparse_local = 1 if opts.sparse_local else 0
        o.host_only = 1 if host_only else 0
        o.evaluator = evaluator
        o.l2l_rule = _capi.L2L_REFERENCE if getattr(opts, "reference_l2l", False) else _capi.L2L_COMPLETE
        o.near_stream_fraction = float(getattr(opts, "near_stream_fraction", 1.0))
        o.device = int(device)
        self.device = int(device)
        if devices is not None and len(devices) > 1:
            # one plan over several devices of this process (fmmbem_options.n_devices): vectors live on devices[0]
            if len(devices) > 8:
                raise ValueError("at most 8 devices per plan")
            o.n_devices = len(devices)
            for i, dv in enumerate(devices):
                o.devices[i] = int(dv)
            o.device = self.device = int(devices[0])
        self.dof = 1
        if isinstance(K, StokesSphericalBEM):
            o.kernel = _capi.KERNEL_STOKES_BEM
            o.mu = K.Mu
            o.quad_k_fine = K.K_fine
            self.dof = 3
        o.shard_rank, o.shard_world = int(shard[0]), int(shard[1])
        self._shard_world = int(shard[1])
        # shard_upward: False / True (1: all-gather of the multipoles) / 2 (all-to-all of the ones each receiver reads)
        o.shard_upward = (2 if shard_upward == 2 else 1) if (shard_upward and int(shard[1]) > 1) else 0
        if o.n_devices > 1:                    # inside a multi-device plan: owners + selective exchange, or every device repeats the upward pass
            o.shard_upward = 0 if replicate_upward else 2
        self.shard_upward = bool(o.shard_upward)
        self.exchange_mode = int(o.shard_upward)
        self.p_max = o.p_max
        bcp = None
        if bc is not None:
            bc = np.ascontiguousarray(bc, dtype=np.uint8)
            if bc.shape != (self.n,):
                raise ValueError("bc must have one flag per panel")
            bcp = bc.ctypes.data_as(C.c_void_p)
        h = C.c_void_p()
        _capi.check(_capi.lib().fmmbem_plan_create(C.byref(o), self.n, v.ctypes.data_as(C.c_void_p), bcp, C.byref(h)))
        self._h = h

    def like(self, bc, K=None):
        """A plan of the same panels and options with other boundary-condition flags (fmmbem_plan_create_like): shares this
        plan's tree, lists and tables; builds only the near-matrix values, P2M moments and expansions.  The drivers'
        right-hand-side plan (examples/LaplaceBEM.cpp:218-232).  K: the kernel object the new plan reads its order from
        (default: this plan's)."""
        import copy
        other = copy.copy(self)
        other._K = K if K is not None else self._K
        bcp = None
        if bc is not None:
            bc = np.ascontiguousarray(bc, dtype=np.uint8)
            if bc.shape != (self.n,):
                raise ValueError("bc must have one flag per panel")
            bcp = bc.ctypes.data_as(C.c_void_p)
        h = C.c_void_p()
        other._h = None
        _capi.check(_capi.lib().fmmbem_plan_create_like(self._h, bcp, C.byref(h)))
        other._h = h
        return other

    # ---- reference surface ----
    def kernel(self):
        return self._K

    def options(self):
        return self._opts

    def execute(self, charges):
        """results = plan.execute(charges) at the kernel's current p. numpy in, numpy out (host)."""
        x = np.ascontiguousarray(charges, dtype=np.float64)
        want = (self.n,) if self.dof == 1 else (self.n, self.dof)
        if x.shape != want:
            raise ValueError("charges must have shape %r" % (want,))
        y = np.empty(want)
        _capi.check(_capi.lib().fmmbem_plan_execute(self._h, self._K.P, x.ctypes.data_as(C.c_void_p),
                                                    y.ctypes.data_as(C.c_void_p)))
        return y

    def __call__(self, x, y):
        """Preconditioner-style operator()(x, y) adapter (examples/BEM/Preconditioner.hpp:11-15)."""
        y[...] = self.execute(x)

    # ---- device-resident variant (torch tensors or raw device pointers) ----
    def execute_device(self, x_ptr, y_ptr, stream=0, p=None):
        _capi.check(_capi.lib().fmmbem_plan_execute_device(self._h, self._K.P if p is None else int(p),
                                                           C.c_void_p(x_ptr), C.c_void_p(y_ptr), C.c_void_p(stream)))

    # ---- split execute of a plan that shards the upward pass (include/fmmbem.h) ----
    def exchange_counts(self, p=None):
        """shard_upward = 2: (send, recv) int64 arrays of doubles per peer shard at order p (fmmbem_plan_exchange_counts)."""
        w = self._shard_world
        send, recv = np.zeros(w, dtype=np.int64), np.zeros(w, dtype=np.int64)
        _capi.check(_capi.lib().fmmbem_plan_exchange_counts(self._h, self._K.P if p is None else int(p),
                                                             send.ctypes.data_as(C.c_void_p), recv.ctypes.data_as(C.c_void_p)))
        return send, recv

    def exchange_doubles(self, p=None):
        n = C.c_size_t(0)
        _capi.check(_capi.lib().fmmbem_plan_exchange_doubles(self._h, self._K.P if p is None else int(p), C.byref(n)))
        return n.value

    def upward_device(self, x_ptr, send_ptr, stream=0, p=None):
        _capi.check(_capi.lib().fmmbem_plan_upward_device(self._h, self._K.P if p is None else int(p), C.c_void_p(x_ptr),
                                                          C.c_void_p(send_ptr), C.c_void_p(stream)))

    def downward_device(self, recv_ptr, y_ptr, stream=0, p=None):
        _capi.check(_capi.lib().fmmbem_plan_downward_device(self._h, self._K.P if p is None else int(p), C.c_void_p(recv_ptr),
                                                            C.c_void_p(y_ptr), C.c_void_p(stream)))

    def near_split_device(self, y_ptr, stream=0):
        _capi.check(_capi.lib().fmmbem_plan_near_split_device(self._h, C.c_void_p(y_ptr), C.c_void_p(stream)))

    # ---- results as tree-order slices (multi-GPU all-gather instead of all-reduce, include/fmmbem.h) ----
    def set_result_slices(self, on=True):
        _capi.check(_capi.lib().fmmbem_plan_set_result_slices(self._h, 1 if on else 0))

    def shard_rows(self, world):
        cut = np.empty(world + 1, dtype=np.int64)
        _capi.check(_capi.lib().fmmbem_plan_shard_rows(self._h, cut.ctypes.data_as(C.c_void_p)))
        return cut

    def assemble_slices_device(self, slices_ptr, chunk_doubles, y_ptr, stream=0):
        _capi.check(_capi.lib().fmmbem_plan_assemble_slices_device(self._h, C.c_void_p(slices_ptr), int(chunk_doubles),
                                                                  C.c_void_p(y_ptr), C.c_void_p(stream)))

    def near_device(self, x_ptr, y_ptr, stream=0):
        _capi.check(_capi.lib().fmmbem_plan_near_device(self._h, C.c_void_p(x_ptr), C.c_void_p(y_ptr), C.c_void_p(stream)))

    def execute_torch(self, x, out=None, p=None):
        """x: float64 CUDA tensor (N,), ORIGINAL panel order. Runs on torch's current stream."""
        import torch
        if x.dtype != torch.float64 or not x.is_cuda or not x.is_contiguous() or x.numel() != self.n * self.dof:
            raise ValueError("x must be a contiguous float64 CUDA tensor with dof values per panel")
        if x.device.index != self.device:
            raise ValueError("x lives on cuda:%s but the plan was built on device %d" % (x.device.index, self.device))
        if out is None:
            out = torch.empty_like(x)
        self.execute_device(x.data_ptr(), out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream, p)
        return out

    # ---- introspection ----
    def set_timing(self, on=True):
        """True / 1: HIP events around every stage; 2: around the near-field kernel only (an event record costs ~5 us of
        stream time, 85 us per fully instrumented matvec at N = 1M); False / 0: off."""
        _capi.check(_capi.lib().fmmbem_plan_set_timing(self._h, 2 if on == 2 else (1 if on else 0)))

    def set_graphs(self, on=True):
        """Replay each order's launch chain as a hipGraph from its second execute on (fmmbem_plan_set_graphs)."""
        _capi.check(_capi.lib().fmmbem_plan_set_graphs(self._h, 1 if on else 0))

    def stats(self):
        s = _capi.Stats()
        _capi.check(_capi.lib().fmmbem_plan_stats(self._h, C.byref(s)))
        return s.as_dict()

    def perm(self):
        out = np.empty(self.n, dtype=np.uint32)
        _capi.check(_capi.lib().fmmbem_plan_get_perm(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def boxes(self):
        nb = self.stats()["n_boxes"]
        d = dict(center=np.empty((nb, 3)), side=np.empty(nb))
        for k in ("level", "leaf", "parent", "bb", "be"):
            d[k] = np.empty(nb, dtype=np.int32)
        _capi.check(_capi.lib().fmmbem_plan_get_boxes(self._h, *[d[k].ctypes.data_as(C.c_void_p) for k in
                                                                 ("center", "side", "level", "leaf", "parent", "bb", "be")]))
        return d

    def pairs(self, which):
        idx = {"p2p": 0, "m2l": 1, "m2m": 2, "l2l": 3, "m2l_work": 4, "m2l_items": 5, "m2l_items_long": 6}[which]
        n = C.c_int64(0)
        _capi.check(_capi.lib().fmmbem_plan_get_pairs(self._h, idx, None, C.byref(n)))
        out = np.empty((n.value, 2), dtype=np.int32)
        _capi.check(_capi.lib().fmmbem_plan_get_pairs(self._h, idx, out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out

    def near_row(self, row, values=True):
        n = C.c_int64(0)
        _capi.check(_capi.lib().fmmbem_plan_get_near_row(self._h, row, None, None, C.byref(n)))
        cols = np.empty(n.value, dtype=np.uint32)
        vals = np.empty(n.value) if values else None
        _capi.check(_capi.lib().fmmbem_plan_get_near_row(
            self._h, row, cols.ctypes.data_as(C.c_void_p),
            vals.ctypes.data_as(C.c_void_p) if values else None, C.byref(n)))
        return cols, vals

    def diagonal(self):
        """K(s,s) of every unknown, original order (the entries Preconditioners::Diagonal inverts)."""
        out = np.empty(self.n * self.dof)
        _capi.check(_capi.lib().fmmbem_plan_get_diagonal(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def expansions(self, which, p=None):
        p = self._K.P if p is None else p
        nb = self.stats()["n_boxes"]
        out = np.empty((nb, self.stats()["expansion_slots"], p * (p + 1) // 2), dtype=np.complex128)   # Stokes: 8, 11 with TRACTION targets
        _capi.check(_capi.lib().fmmbem_plan_get_expansions(self._h, 0 if which == "M" else 1, p,
                                                           out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            _capi.lib().fmmbem_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
