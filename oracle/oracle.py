"""ctypes wrapper around oracle/liboracle.so -- CPU ORACLE, TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (fmm-bem-relaxed_amd/) must never import it: a product path that routes
through the oracle voids every parity claim.

The oracle is a plain-C restatement of the reference's LaplaceSphericalBEM FMM matvec
(see oracle/fmm_oracle.h for the parity status and the per-function reference citations).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so with gcc (make). Building the checker is not using it.
    FMM_ORACLE_FLAVOR=refflags selects liboracle_refflags.so: the same sources compiled with the reference's flags
    (oracle/Makefile) -- what bench.py's cpu_baseline leg times, in a process of its own."""
    refflags = os.environ.get("FMM_ORACLE_FLAVOR") == "refflags"
    so = os.path.join(_HERE, "liboracle_refflags.so" if refflags else "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["refflags"] if refflags else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
        L.orc_unit_sphere.restype = C.c_long
        L.orc_unit_sphere.argtypes = [i32, vp]
        L.orc_create.restype = vp
        L.orc_create.argtypes = [i32, vp, vp, i32, dbl, C.c_uint]
        L.orc_create_eval.restype = vp
        L.orc_create_eval.argtypes = [i32, vp, vp, i32, dbl, C.c_uint, i32]
        L.orc_complete_l2l.restype = None
        L.orc_complete_l2l.argtypes = [vp]
        L.orc_destroy.argtypes = [vp]
        L.orc_stats.argtypes = [vp, vp]
        L.orc_build_near.argtypes = [vp]
        L.orc_build_near.restype = i32
        L.orc_matvec.argtypes = [vp, i32, vp, vp, i32, vp]
        L.orc_matvec.restype = i32
        L.orc_direct.argtypes = [vp, vp, vp, i32, i32]
        L.orc_direct_rows.argtypes = [vp, vp, vp, i32, vp]
        L.orc_stokes_direct_rows.argtypes = [vp, vp, vp, i32, vp]
        L.orc_near_only.argtypes = [vp, vp, vp]
        L.orc_get_perm.argtypes = [vp, vp]
        L.orc_get_boxes.argtypes = [vp] * 10
        L.orc_get_pairs.argtypes = [vp, i32, vp]
        L.orc_get_pairs.restype = i32
        L.orc_get_list.argtypes = [vp, i32, vp]
        L.orc_get_list.restype = i32
        L.orc_near_nnz.argtypes = [vp]
        L.orc_near_nnz.restype = i64
        L.orc_get_near.argtypes = [vp, vp, vp, vp]
        L.orc_get_expansions.argtypes = [vp, i32, i32, vp]
        L.orc_get_panels.argtypes = [vp, vp, vp, vp, vp]
        L.orc_kernel_entries.argtypes = [vp, i32, vp, vp, vp]
        L.orc_tables_create.restype = vp
        L.orc_tables_create.argtypes = [i32]
        L.orc_tables_destroy.argtypes = [vp]
        L.orc_get_tables.argtypes = [vp, vp, vp, vp]
        L.orc_eval_multipole.argtypes = [vp, dbl, dbl, dbl, vp, vp]
        L.orc_eval_local.argtypes = [vp, dbl, dbl, dbl, vp, vp]
        L.orc_cart2sph.argtypes = [vp, vp, vp, vp]
        L.orc_m2m.argtypes = [vp, vp, vp, vp]
        L.orc_m2l.argtypes = [vp, vp, vp, vp]
        L.orc_l2l.argtypes = [vp, vp, vp, vp]
        L.orc_single_p2m.argtypes = [i32, i32, i32, dbl, i32, vp, vp, vp, vp, vp]
        L.orc_single_p2m.restype = i32
        L.orc_single_l2p.argtypes = [i32, i32, i32, dbl, vp, vp, i32, vp, vp, vp]
        L.orc_single_l2p.restype = i32
        L.orc_semi_analytical.argtypes = [vp, vp, vp, vp, vp, vp, i32]
        L.orc_quadrature.argtypes = [i32, vp, vp]
        L.orc_quadrature.restype = i32
        L.orc_num_threads.restype = i32
        L.orc_set_num_threads.argtypes = [i32]
        L.orc_stokes_config.argtypes = [vp, dbl, i32]
        L.orc_stokes_config.restype = i32
        L.orc_stokes_build_near.argtypes = [vp]
        L.orc_stokes_build_near.restype = i32
        L.orc_stokes_matvec.argtypes = [vp, i32, vp, vp, i32, vp]
        L.orc_stokes_matvec.restype = i32
        L.orc_stokes_direct.argtypes = [vp, vp, vp, i32, i32]
        L.orc_stokes_get_near.argtypes = [vp, vp]
        L.orc_stokes_get_expansions.argtypes = [vp, i32, i32, vp]
        L.orc_stokes_entries.argtypes = [vp, i32, vp, vp, vp]
        L.orc_red_blood_cell_map.argtypes = [C.c_long, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def num_threads():
    """OpenMP threads the oracle's parallel stages use."""
    return lib().orc_num_threads()


def set_num_threads(n):
    """OpenMP threads of the oracle's parallel stages from now on (torchrun hands every rank OMP_NUM_THREADS=1)."""
    lib().orc_set_num_threads(int(n))


def unit_sphere(recursions, center=(0.0, 0.0, 0.0)):
    """Triangulation::UnitSphere (examples/BEM/Triangulation.hpp:105-121): (N, 3, 3) vertices."""
    n = lib().orc_unit_sphere(recursions, None)
    v = np.empty((n, 3, 3), dtype=np.float64)
    lib().orc_unit_sphere(recursions, _p(v))
    if any(center):
        v += np.asarray(center, dtype=np.float64)
    return v


def semi_analytical(y0, y1, y2, x, same=False):
    """AnalyticalIntegral::SemiAnalytical (examples/BEM/SemiAnalytical.hpp:148-203): (int 1/r, int d(1/r)/dn) over the
    flat triangle y0 y1 y2 seen from x."""
    a = [np.ascontiguousarray(t, dtype=np.float64) for t in (y0, y1, y2, x)]
    G, dG = C.c_double(0.0), C.c_double(0.0)
    lib().orc_semi_analytical(C.byref(G), C.byref(dG), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), 1 if same else 0)
    return G.value, dG.value


def quadrature(key):
    pts = np.zeros((80, 3))            # ORC_MAXK
    w = np.zeros(80)
    n = lib().orc_quadrature(key, _p(pts), _p(w))
    if n < 0:
        raise ValueError("invalid quadrature key %d" % key)
    return pts[:n].copy(), w[:n].copy()


STAT_NAMES = ("n", "boxes", "leaves", "levels", "near_nnz", "m2l_pairs", "m2m_ops", "l2l_ops",
              "p2m_leaves", "l2p_leaves", "p2p_pairs", "l2l_skipped", "nq")
STAGES = ("init", "near", "p2m", "m2m", "m2l", "l2l", "l2p", "total")
FAITHFUL = 1


class Oracle:
    """One FMM_plan<LaplaceSphericalBEM>-equivalent on the CPU (include/FMM_plan.hpp:34-90)."""

    def __init__(self, vertices, bc=None, K=3, theta=0.5, ncrit=64, evaluator=0, complete_l2l=False):
        """evaluator: 0 FMM, 1 local only (EvalLocalSparse), 2 block diagonal (EvalDiagonalSparse).
        complete_l2l: replace the reference's L2L list (which omits stats()['l2l_skipped'] edges on adaptive trees)
        by the complete one -- not a reference rule, see tree.c:orc_complete_l2l."""
        v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 9)
        self.n = v.shape[0]
        if bc is None:
            bc = np.zeros(self.n, dtype=np.uint8)
        bc = np.ascontiguousarray(bc, dtype=np.uint8)
        self._h = lib().orc_create_eval(self.n, _p(v), _p(bc), K, theta, ncrit, evaluator)
        if not self._h:
            raise ValueError("orc_create failed (bad quadrature key or empty input)")
        if complete_l2l:
            lib().orc_complete_l2l(self._h)
        self.bc = bc

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self):
        out = np.zeros(16, dtype=np.int64)
        lib().orc_stats(self._h, _p(out))
        return dict(zip(STAT_NAMES, out.tolist()))

    def build_near(self):
        if lib().orc_build_near(self._h):
            raise MemoryError("near matrix")

    def matvec(self, x, p, faithful=False, return_times=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.n)
        st = np.zeros(8)
        rc = lib().orc_matvec(self._h, p, _p(x), _p(y), FAITHFUL if faithful else 0, _p(st))
        if rc:
            raise RuntimeError("orc_matvec rc=%d" % rc)
        return (y, dict(zip(STAGES, st.tolist()))) if return_times else y

    def direct(self, x, rows=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.n)
        b, e = (0, self.n) if rows is None else rows
        lib().orc_direct(self._h, _p(x), _p(y), b, e)
        return y if rows is None else y[b:e]

    def direct_rows(self, x, rows):
        """Direct sum on a list of target rows (original panel order): the seeded whole-vector sample of the accuracy gate."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        y = np.zeros(len(rows))
        lib().orc_direct_rows(self._h, _p(x), _p(y), len(rows), _p(rows))
        return y

    def near_only(self, x):
        self.build_near()
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.n)
        lib().orc_near_only(self._h, _p(x), _p(y))
        return y

    def perm(self):
        out = np.empty(self.n, dtype=np.uint32)
        lib().orc_get_perm(self._h, _p(out))
        return out

    def boxes(self):
        nb = self.stats()["boxes"]
        d = dict(center=np.empty((nb, 3)), side=np.empty(nb))
        for k in ("level", "leaf", "parent", "cb", "ce", "bb", "be"):
            d[k] = np.empty(nb, dtype=np.int32)
        lib().orc_get_boxes(self._h, *[_p(d[k]) for k in
                                       ("center", "side", "level", "leaf", "parent", "cb", "ce", "bb", "be")])
        return d

    def pairs(self, which):
        idx = {"p2p": 0, "m2l": 1, "m2m": 2, "l2l": 3}[which]
        n = lib().orc_get_pairs(self._h, idx, None)
        out = np.empty((n, 2), dtype=np.int32)
        lib().orc_get_pairs(self._h, idx, _p(out))
        return out

    def leaf_list(self, which):
        idx = {"p2m": 0, "l2p": 1}[which]
        n = lib().orc_get_list(self._h, idx, None)
        out = np.empty(n, dtype=np.int32)
        lib().orc_get_list(self._h, idx, _p(out))
        return out

    def near_csr(self):
        self.build_near()
        nnz = lib().orc_near_nnz(self._h)
        rp = np.empty(self.n + 1, dtype=np.int64)
        col = np.empty(nnz, dtype=np.uint32)
        val = np.empty(nnz)
        lib().orc_get_near(self._h, _p(rp), _p(col), _p(val))
        return rp, col, val

    def expansions(self, p, which):
        """M or L of the last matvec at order p: complex array [boxes, 2, p(p+1)/2]."""
        nb = self.stats()["boxes"]
        out = np.empty((nb, 2, p * (p + 1) // 2), dtype=np.complex128)
        lib().orc_get_expansions(self._h, p, 0 if which == "M" else 1, _p(out))
        return out

    def panels(self):
        nq = self.stats()["nq"]
        c, nrm, a, q = np.empty((self.n, 3)), np.empty((self.n, 3)), np.empty(self.n), np.empty((self.n, nq, 3))
        lib().orc_get_panels(self._h, _p(c), _p(nrm), _p(a), _p(q))
        return dict(center=c, normal=nrm, area=a, quad=q)

    def kernel_entries(self, ti, sj):
        ti = np.ascontiguousarray(ti, dtype=np.int32)
        sj = np.ascontiguousarray(sj, dtype=np.int32)
        out = np.empty(len(ti))
        lib().orc_kernel_entries(self._h, len(ti), _p(ti), _p(sj), _p(out))
        return out


class Tables:
    """LaplaceSpherical precomputed tables + single-operator entry points (kernel/LaplaceSpherical.hpp)."""

    def __init__(self, P):
        self.P = P
        self._h = lib().orc_tables_create(P)
        if not self._h:
            raise ValueError("bad P")

    def __del__(self):
        try:
            lib().orc_tables_destroy(self._h)
        except Exception:
            pass

    def arrays(self):
        P = self.P
        pre, A, Cn = np.empty(4 * P * P), np.empty(4 * P * P), np.empty(P ** 4, dtype=np.complex128)
        lib().orc_get_tables(self._h, _p(pre), _p(A), _p(Cn))
        return pre, A, Cn

    def eval_multipole(self, rho, alpha, beta):
        Y = np.zeros(4 * self.P ** 2, dtype=np.complex128)
        Yt = np.zeros_like(Y)
        lib().orc_eval_multipole(self._h, rho, alpha, beta, _p(Y), _p(Yt))
        return Y[:self.P ** 2], Yt[:self.P ** 2]

    def eval_local(self, rho, alpha, beta):
        Y = np.zeros(4 * self.P ** 2, dtype=np.complex128)
        Yt = np.zeros_like(Y)
        lib().orc_eval_local(self._h, rho, alpha, beta, _p(Y), _p(Yt))
        return Y, Yt

    def _op(self, fn, src, tr):
        src = np.ascontiguousarray(src, dtype=np.complex128)
        dst = np.zeros_like(src)
        tr = np.ascontiguousarray(tr, dtype=np.float64)
        fn(self._h, _p(src), _p(dst), _p(tr))
        return dst

    def m2m(self, Ms, tr):
        return self._op(lib().orc_m2m, Ms, tr)

    def m2l(self, Ms, tr):
        return self._op(lib().orc_m2l, Ms, tr)

    def l2l(self, Ls, tr):
        return self._op(lib().orc_l2l, Ls, tr)


def single_p2m(kernel, P, K, verts, bc, charges, center, mu=1.0):
    """KernelSkeleton::P2M of the BEM kernels on caller-supplied panels: the expansions [2 or 4][S] of sources about `center`
    (kernel/LaplaceSphericalBEM.hpp:307-352; kernel/StokesSphericalBEM.hpp:391-432, VELOCITY sources)."""
    verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 9)
    n = len(verts)
    bc = np.zeros(n, dtype=np.uint8) if bc is None else np.ascontiguousarray(bc, dtype=np.uint8)
    charges = np.ascontiguousarray(charges, dtype=np.float64)
    M = np.zeros((4 if kernel else 2, P * (P + 1) // 2), dtype=np.complex128)
    center = np.ascontiguousarray(center, dtype=np.float64)
    if lib().orc_single_p2m(kernel, P, K, mu, n, _p(verts), _p(bc), _p(charges), _p(center), _p(M)) != 0:
        raise ValueError("bad P or K")
    return M


def single_l2p(kernel, P, K, L, center, verts, bc, mu=1.0):
    """KernelSkeleton::L2P of the BEM kernels (kernel/LaplaceSphericalBEM.hpp:448-476; StokesSphericalBEM.hpp:512-522)."""
    verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 9)
    n = len(verts)
    bc = np.zeros(n, dtype=np.uint8) if bc is None else np.ascontiguousarray(bc, dtype=np.uint8)
    L = np.ascontiguousarray(L, dtype=np.complex128)
    out = np.zeros(n * (3 if kernel else 1))
    center = np.ascontiguousarray(center, dtype=np.float64)
    if lib().orc_single_l2p(kernel, P, K, mu, _p(L), _p(center), n, _p(verts), _p(bc), _p(out)) != 0:
        raise ValueError("bad P or K")
    return out


def cart2sph(d):
    d = np.ascontiguousarray(d, dtype=np.float64)
    r, t, p = C.c_double(), C.c_double(), C.c_double()
    lib().orc_cart2sph(C.byref(r), C.byref(t), C.byref(p), _p(d))
    return r.value, t.value, p.value


def red_blood_cell(recursions):
    """Triangulation::RedBloodCell (examples/BEM/Triangulation.hpp:210-255), identity rotation, zero shift."""
    v = unit_sphere(recursions)
    lib().orc_red_blood_cell_map(len(v), _p(v))
    return v


class StokesOracle(Oracle):
    """FMM_plan<StokesSphericalBEM>-equivalent on the CPU, velocity boundary condition only
    (kernel/StokesSphericalBEM.hpp:260-375, 391-432, 512-528; StokesSpherical.hpp:318-401)."""

    def __init__(self, vertices, K=4, K_fine=19, mu=1e-3, theta=0.5, ncrit=64, evaluator=0, complete_l2l=False, bc=None):
        """bc: per-panel flags, 0 VELOCITY / 1 TRACTION.  The target's flag picks the integral of a near-matrix entry
        (StokesSphericalBEM.hpp:377-389); the far field is the velocity branch only, so matvec() refuses traction panels
        unless the evaluator is near-field-only (1, 2)."""
        super().__init__(vertices, bc=bc, K=K, theta=theta, ncrit=ncrit, evaluator=evaluator, complete_l2l=complete_l2l)
        if lib().orc_stokes_config(self._h, mu, K_fine):
            raise ValueError("invalid K_fine")
        self.mu = mu

    def matvec(self, x, p, faithful=False, return_times=False):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.n, 3)
        y = np.empty((self.n, 3))
        st = np.zeros(8)
        rc = lib().orc_stokes_matvec(self._h, p, _p(x), _p(y), FAITHFUL if faithful else 0, _p(st))
        if rc:
            raise RuntimeError("orc_stokes_matvec rc=%d" % rc)
        return (y, dict(zip(STAGES, st.tolist()))) if return_times else y

    def direct(self, x, rows=None):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.n, 3)
        y = np.zeros((self.n, 3))
        b, e = (0, self.n) if rows is None else rows
        lib().orc_stokes_direct(self._h, _p(x), _p(y), b, e)
        return y if rows is None else y[b:e]

    def direct_rows(self, x, rows):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.n, 3)
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        y = np.zeros((len(rows), 3))
        lib().orc_stokes_direct_rows(self._h, _p(x), _p(y), len(rows), _p(rows))
        return y

    def near_csr(self):
        if lib().orc_stokes_build_near(self._h):
            raise MemoryError("near matrix")
        nnz = lib().orc_near_nnz(self._h)
        rp = np.empty(self.n + 1, dtype=np.int64)
        col = np.empty(nnz, dtype=np.uint32)
        lib().orc_get_near(self._h, _p(rp), _p(col), None)
        val = np.empty((nnz, 3, 3))
        lib().orc_stokes_get_near(self._h, _p(val))
        return rp, col, val

    def expansions(self, p, which):
        nb = self.stats()["boxes"]
        out = np.empty((nb, 8, p * (p + 1) // 2), dtype=np.complex128)
        lib().orc_stokes_get_expansions(self._h, p, 0 if which == "M" else 1, _p(out))
        return out

    def kernel_entries(self, ti, sj):
        ti = np.ascontiguousarray(ti, dtype=np.int32)
        sj = np.ascontiguousarray(sj, dtype=np.int32)
        out = np.empty((len(ti), 3, 3))
        lib().orc_stokes_entries(self._h, len(ti), _p(ti), _p(sj), _p(out))
        return out
