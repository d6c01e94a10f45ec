"""The caller of the hot path: restarted GMRES with per-iteration relaxation of the FMM order p.

Mirrors examples/BEM/SolverOptions.hpp:11-39 (SolverOptions, predict_p) and examples/BEM/GMRES.hpp:143-252
(GMRES with modified Gram-Schmidt and Givens rotations, p chosen from the current residual before every
matvec).  Krylov vectors live in HBM as torch tensors; the matvec is FMM_plan.execute_torch (HIP kernels);
per iteration ONE device->host transfer brings the new Hessenberg column to the host for the O(R) Givens
update.  The vector algebra (dot/axpy/norm) is torch plumbing around the operator, not part of the hot path.
"""
import math
import os

import torch


class SolverOptions:
    """examples/BEM/SolverOptions.hpp:11-39."""
    SIMONCINI, BOURAS = 0, 1

    def __init__(self, residual=1e-5, max_iters=500, max_p=16, restart=None, variable_p=True, p_min=5):
        self.residual = residual
        self.max_iters = max_iters
        self.restart = max_iters if restart is None else restart     # LaplaceBEM.cpp:162-163
        self.max_p = max_p
        self.p_min = p_min
        self.variable_p = variable_p
        self.relax_type = SolverOptions.BOURAS

    def predict_p(self, eps):
        """SolverOptions.hpp:25-38: Bouras-Fraysse  p = min(ceil(-log2(min(tol / min(eps, 1), 1))), max_p)."""
        if not self.variable_p:
            return self.max_p
        if eps <= 0.0:
            # the C++ code divides by zero here: alpha = inf, nu saturates at 1 (Bouras) and the order is 0, which the
            # callers raise to their floor (GMRES.hpp:195 max(1u, .)); Simoncini's -log2(0) = inf saturates at max_p
            return 0 if self.relax_type == SolverOptions.BOURAS else self.max_p
        if self.relax_type == SolverOptions.BOURAS:
            alpha = 1.0 / min(eps, 1.0)
            nu = min(alpha * self.residual, 1.0)
            return min(int(math.ceil(-math.log2(nu))) & 0xFFFFFFFF, self.max_p)     # (unsigned)ceil(...)
        return min(int(math.ceil(-math.log2(eps))) & 0xFFFFFFFF, self.max_p)


def _apply(execute, z, out):
    """w = A z into a reusable buffer when the operator can write in place (FMM_plan.execute_torch, ShardedFMM.execute)."""
    try:
        return execute(z, out=out)
    except TypeError:
        out.copy_(execute(z))
        return out


class _Mgs:
    """Buffers of the fused orthogonalisation (csrc/krylov.hip, fmmbem_mgs_column_device): the column of H on the device and
    the reduction scratch, zeroed once."""

    def __init__(self, restart, like):
        from . import _capi
        self.lib = _capi.lib()
        self.h = torch.empty(restart + 2, dtype=like.dtype, device=like.device)
        self.scratch = torch.zeros(int(self.lib.fmmbem_mgs_scratch_doubles(restart + 1)), dtype=like.dtype, device=like.device)


def _mgs_column(w, V, i, mgs):
    """h_k = <w, V_k>, w -= h_k V_k for k <= i; h_{i+1} = |w|; V_{i+1} = w / h_{i+1}.  Returns the i + 2 numbers on the host.
    Device vectors go through ONE library call (i + 3 launches); anything else (the CPU tests drive this solver with the
    oracle as the operator) takes the same steps as torch calls."""
    if mgs is not None and w.is_cuda and w.is_contiguous() and V.is_contiguous():
        from . import _capi
        _capi.check(mgs.lib.fmmbem_mgs_column_device(w.numel(), w.data_ptr(), V.data_ptr(), V.stride(0), i + 1, mgs.h.data_ptr(),
                                                      V[i + 1].data_ptr(), mgs.scratch.data_ptr(),
                                                      torch.cuda.current_stream(w.device).cuda_stream))
        return mgs.h[:i + 2].tolist()
    hs = []
    for k in range(i + 1):
        hk = torch.dot(w, V[k])
        hs.append(hk)
        w.addcmul_(V[k], hk, value=-1.0)                  # w -= hk * V[k]
    hn = torch.linalg.vector_norm(w)
    hs.append(hn)
    torch.div(w, hn, out=V[i + 1])
    return torch.stack(hs).tolist()


def _generate_plane_rotation(dx, dy):          # GMRES.hpp:88-105
    if dy == 0.0:
        return 1.0, 0.0
    if abs(dy) > abs(dx):
        tmp = dx / dy
        sn = 1.0 / math.sqrt(1.0 + tmp * tmp)
        return tmp * sn, sn
    tmp = dy / dx
    cs = 1.0 / math.sqrt(1.0 + tmp * tmp)
    return cs, tmp * cs


def gmres(MV, x, b, opts, M=None, log=None, stokes=False):
    """GMRES(MV, x, b, opts[, M]) of examples/BEM/GMRES.hpp:143-252; stokes=True: the order rule of
    examples/BEM/GMRES_Stokes.hpp:229, p = max(p_min, predict_p - 1), vectors = N x 3 values flattened.

    MV: object with execute_torch(tensor)->tensor (or execute) and kernel().set_p(p);
    x, b: float64 CUDA tensors (x is updated in place and returned); M: optional callable z = M(v).
    log: optional list receiving (iteration, p, |residual|) per inner iteration.
    Returns (x, iterations, |residual|)."""
    execute = getattr(MV, "execute_torch", None) or MV.execute
    K = MV.kernel()
    R, n = opts.restart, x.numel()
    V = torch.empty((R + 1, n), dtype=x.dtype, device=x.device)
    wbuf = torch.empty(n, dtype=x.dtype, device=x.device)
    mgs = _Mgs(R, x) if (x.is_cuda and os.environ.get("FMMBEM_FUSED_MGS", "1") != "0") else None
    H = [[0.0] * R for _ in range(R + 1)]
    cs, sn, s = [0.0] * R, [0.0] * R, [0.0] * (R + 1)
    normb = float(torch.linalg.vector_norm(b))
    it, resid = 0, 0.0
    if normb == 0.0:                                      # b = 0: the reference divides by zero and stops on NaN; x = x0 is returned
        return x, 0, 0.0
    while True:                                           # outer (restart) loop, :166
        w = execute(x)                                    # at the kernel's current p
        w = w - b
        beta = float(torch.linalg.vector_norm(w))
        if beta == 0.0:                                   # x already solves the system (e.g. on a restart after exact convergence)
            return x, it, 0.0
        V[0] = w * (-1.0 / beta)
        s[0] = beta
        i = -1
        resid = s[0] / normb
        while True:                                       # inner loop, :186
            i += 1
            it += 1
            p = max(opts.p_min, opts.predict_p(abs(resid)) - 1) if stokes else max(1, opts.predict_p(abs(resid)))   # :195
            K.set_p(p)
            z = V[i] if M is None else M(V[i])
            w = _apply(execute, z, wbuf)
            col = _mgs_column(w, V, i, mgs)               # modified Gram-Schmidt, :203-212; the one sync of the iteration
            for k in range(i + 2):
                H[k][i] = col[k]
            for k in range(i):                            # PlaneRotation, :108-117
                t = cs[k] * H[k][i] + sn[k] * H[k + 1][i]
                H[k + 1][i] = -sn[k] * H[k][i] + cs[k] * H[k + 1][i]
                H[k][i] = t
            cs[i], sn[i] = _generate_plane_rotation(H[i][i], H[i + 1][i])
            t = cs[i] * H[i][i] + sn[i] * H[i + 1][i]
            H[i + 1][i] = -sn[i] * H[i][i] + cs[i] * H[i + 1][i]
            H[i][i] = t
            s[i + 1] = -sn[i] * s[i]
            s[i] = cs[i] * s[i]
            resid = s[i + 1] / normb
            if log is not None:
                log.append((it, p, abs(resid)))
            if abs(resid) < opts.residual:
                break
            if not (i + 1 < R and i + 1 <= opts.max_iters and abs(resid) > opts.residual):
                break
        y = s[:i + 1]
        for j in range(i, -1, -1):                        # back substitution, :228-234
            y[j] /= H[j][j]
            for k in range(j - 1, -1, -1):
                y[k] -= H[k][j] * y[j]
        for j in range(i + 1):                            # x += y_j M(V_j), :237-241
            x += y[j] * (V[j] if M is None else M(V[j]))
        if not (abs(resid) > opts.residual and it < opts.max_iters):
            break
    return x, it, abs(resid)


def fgmres(MV, x, b, opts, M, log=None, stokes=False):
    """FGMRES(MV, x, b, opts, M, context) of examples/BEM/GMRES.hpp:276-380: flexible GMRES, the preconditioned
    vectors Z_j = M(V_j) are kept and the solution is updated from them (:318-320, :368-371).
    stokes=True: the order rule of examples/BEM/GMRES_Stokes.hpp:373, p = max(5, predict_p)."""
    execute = getattr(MV, "execute_torch", None) or MV.execute
    K = MV.kernel()
    R, n = opts.restart, x.numel()
    V = torch.empty((R + 1, n), dtype=x.dtype, device=x.device)
    wbuf = torch.empty(n, dtype=x.dtype, device=x.device)
    Z = torch.empty((R, n), dtype=x.dtype, device=x.device)
    mgs = _Mgs(R, x) if (x.is_cuda and os.environ.get("FMMBEM_FUSED_MGS", "1") != "0") else None
    H = [[0.0] * R for _ in range(R + 1)]
    cs, sn, s = [0.0] * R, [0.0] * R, [0.0] * (R + 1)
    normb = float(torch.linalg.vector_norm(b))
    it, resid = 0, 0.0
    if normb == 0.0:
        return x, 0, 0.0
    while True:
        w = execute(x) - b
        beta = float(torch.linalg.vector_norm(w))
        if beta == 0.0:
            return x, it, 0.0
        V[0] = w * (-1.0 / beta)
        s[0] = beta
        i = -1
        resid = s[0] / normb
        while True:
            i += 1
            it += 1
            p = max(5 if stokes else 1, opts.predict_p(abs(resid)))        # :324 (the product has no order 0)
            K.set_p(p)
            Z[i] = M(V[i])
            w = _apply(execute, Z[i], wbuf)
            col = _mgs_column(w, V, i, mgs)               # modified Gram-Schmidt, :203-212; the one sync of the iteration
            for k in range(i + 2):
                H[k][i] = col[k]
            for k in range(i):
                t = cs[k] * H[k][i] + sn[k] * H[k + 1][i]
                H[k + 1][i] = -sn[k] * H[k][i] + cs[k] * H[k + 1][i]
                H[k][i] = t
            cs[i], sn[i] = _generate_plane_rotation(H[i][i], H[i + 1][i])
            t = cs[i] * H[i][i] + sn[i] * H[i + 1][i]
            H[i + 1][i] = -sn[i] * H[i][i] + cs[i] * H[i + 1][i]
            H[i][i] = t
            s[i + 1] = -sn[i] * s[i]
            s[i] = cs[i] * s[i]
            resid = s[i + 1] / normb
            if log is not None:
                log.append((it, p, abs(resid)))
            if abs(resid) < opts.residual:
                break
            if not (i + 1 < R and i + 1 <= opts.max_iters and abs(resid) > opts.residual):
                break
        y = s[:i + 1]
        for j in range(i, -1, -1):
            y[j] /= H[j][j]
            for k in range(j - 1, -1, -1):
                y[k] -= H[k][j] * y[j]
        for j in range(i + 1):
            x += y[j] * Z[j]
        if not (abs(resid) > opts.residual and it < opts.max_iters):
            break
    return x, it, abs(resid)


def _c_options(opts, stokes, flexible, initial_p):
    from . import _capi
    o = _capi.SolverOpts()
    o.residual, o.max_iters, o.restart, o.max_p, o.p_min = opts.residual, opts.max_iters, opts.restart, opts.max_p, opts.p_min
    o.variable_p, o.relax_type = int(bool(opts.variable_p)), opts.relax_type
    o.order_rule = ((_capi.ORDER_FGMRES_STOKES if stokes else _capi.ORDER_FGMRES) if flexible else
                    (_capi.ORDER_GMRES_STOKES if stokes else _capi.ORDER_GMRES))
    o.flexible, o.initial_p = int(bool(flexible)), int(initial_p)
    return o


def gmres_capi(MV, x, b, opts, M=None, log=None, stokes=False, flexible=False):
    """The same solve through the C ABI's device-resident solver (include/fmmbem.h fmmbem_gmres_device; csrc/krylov.hip):
    what a C or C++ caller of the library gets.  MV: an FMM_plan; x, b: float64 CUDA tensors (x updated in place);
    M: None, a Diagonal, or a LocalInnerSolver / BlockDiagonal.  Returns (x, iterations, |residual|, seconds)."""
    import ctypes as C
    from . import _capi
    o = _c_options(opts, stokes, flexible, MV.kernel().P)
    pc = None
    if M is not None:
        pc = _capi.Preconditioner()
        if isinstance(M, Diagonal):
            pc.kind, pc.reciprocals = _capi.PC_DIAGONAL, M.recip.data_ptr()
        elif isinstance(M, _InnerSolver):
            pc.kind, pc.inner_plan = _capi.PC_INNER_PLAN, M.plan._h
            pc.inner = _c_options(M.options, False, False, M.options.max_p)
        else:
            raise TypeError("gmres_capi: M must be None, Diagonal, LocalInnerSolver or BlockDiagonal")
    cap = max(1, opts.max_iters + opts.restart + 2)
    ps, rs = (C.c_int32 * cap)(), (C.c_double * cap)()
    lg = _capi.SolverLog()
    lg.capacity, lg.p, lg.resid = cap, ps, rs
    _capi.check(_capi.lib().fmmbem_gmres_device(MV._h, C.byref(o), x.data_ptr(), b.data_ptr(), C.byref(pc) if pc is not None else None,
                                                C.byref(lg), torch.cuda.current_stream(x.device).cuda_stream))
    if log is not None:
        log.extend((k + 1, int(ps[k]), float(rs[k])) for k in range(min(lg.iterations, cap)))
    return x, lg.iterations, lg.residual, lg.seconds


class Diagonal:
    """Preconditioners::Diagonal (examples/BEM/Preconditioner.hpp:19-42): y = x / K(s,s), panel by panel."""

    def __init__(self, plan, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.recip = (1.0 / torch.from_numpy(plan.diagonal())).to(dev)

    def __call__(self, v):
        return v * self.recip


class _InnerSolver:
    """Preconditioner = a few GMRES steps on a near-field-only operator (examples/BEM/LocalPC.hpp:26-59,
    BlockDiagonalPC.hpp:16-60): options.residual = 1e-1, variable_p = false, max_iters = 1, restart 50."""

    def __init__(self, fb, K, panels, fmm_opts, bc=None, device=0):
        self.plan = fb.FMM_plan(K, panels, fmm_opts, bc=bc, device=device)
        self.options = SolverOptions(residual=1e-1, max_iters=1, max_p=K.P, restart=50, variable_p=False)

    def __call__(self, v):
        y = torch.zeros_like(v)
        gmres(self.plan, y, v.contiguous(), self.options)
        return y


class LocalInnerSolver(_InnerSolver):
    """Preconditioners::LocalInnerSolver -- FMMOptions of local_options(), LocalPC.hpp:7-16."""

    def __init__(self, fb, K, panels, bc=None, device=0):
        o = fb.FMMOptions()
        o.local_evaluation, o.lazy_evaluation, o.sparse_local = True, False, True
        o.set_mac_theta(0.5)
        super().__init__(fb, K, panels, o, bc, device)


class BlockDiagonal(_InnerSolver):
    """Preconditioners::BlockDiagonal -- FMMOptions of BlockDiagonal::local_options(), BlockDiagonalPC.hpp:53-63."""

    def __init__(self, fb, K, panels, bc=None, device=0):
        o = fb.FMMOptions()
        o.local_evaluation, o.lazy_evaluation, o.sparse_local, o.block_diagonal = False, False, True, True
        o.set_mac_theta(0.5)
        super().__init__(fb, K, panels, o, bc, device)


def laplace_bem_first_kind(fb, panels, p=12, k=3, tol=1e-5, theta=0.5, ncrit=64, max_iters=500, device=0, log=None):
    """The solve of examples/LaplaceBEM.cpp:168-291 (first-kind equation, identity preconditioner):
    b = A_flipped-BC * 1 (:218-232), x0 = 0, GMRES with relaxed p (max_p = p). Returns (x, iterations, residual)."""
    import numpy as np
    opts = fb.FMMOptions()
    opts.set_mac_theta(theta)
    opts.set_max_per_box(ncrit)
    K = fb.LaplaceSphericalBEM(p, k)
    n = len(panels)
    plan = fb.FMM_plan(K, panels, opts, p_max=p, device=device)
    rhs_plan = fb.FMM_plan(fb.LaplaceSphericalBEM(p, k), panels, opts, bc=np.ones(n, dtype=np.uint8), p_max=p, device=device)
    dev = torch.device("cuda", device)
    ones = torch.ones(n, dtype=torch.float64, device=dev)
    b = rhs_plan.execute_torch(ones)
    rhs_plan.close()
    x = torch.zeros(n, dtype=torch.float64, device=dev)
    so = SolverOptions(residual=tol, max_iters=max_iters, max_p=p)
    return gmres(plan, x, b, so, log=log)
