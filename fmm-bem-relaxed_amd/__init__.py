"""fmm-bem-relaxed_amd: MI355X-native FMM matvec behind the reference's FMM_plan / kernel surface.

Only the hot path lives here: csrc/ (hand-written HIP for gfx950 + the C ABI of include/fmmbem.h)
and this thin host-side mirror of the reference's operator interface.
"""
from ._capi import FmmBemError, LIB_PATH, Options, PMAX, SYMBOLS, lib  # noqa: F401
from .plan import (FMM_plan, FMMOptions, LaplaceSphericalBEM, StokesSphericalBEM, kernel_entries, quadrature, read_msh,  # noqa: F401
                   read_vert_face, red_blood_cell, red_blood_cells, unit_sphere, write_vert_face)


def __getattr__(name):
    # torch is only needed for the multi-GPU wrapper; import it lazily
    if name == "ShardedFMM":
        from .distributed import ShardedFMM
        return ShardedFMM
    if name in ("SolverOptions", "gmres", "gmres_capi", "fgmres", "LocalInnerSolver", "BlockDiagonal", "Diagonal", "laplace_bem_first_kind"):
        from . import solver
        return getattr(solver, name)
    raise AttributeError(name)
