"""examples/LaplaceBEM.py mirrors the reference driver examples/LaplaceBEM.cpp.  The
reference run `LaplaceBEM -recursions 6 -p 12 -theta 0.5` recorded in SURVEY.md section 8d config 5: 6 iterations with
p = 12,3,2,1,1 at the default tolerance; exterior-point error 6.2e-4 at convergence."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_first_kind_sphere_run_matches_recorded_reference_output():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "LaplaceBEM.py"), "-recursions", "6", "-p", "12",
                        "-theta", "0.5"], capture_output=True, text=True, check=True)
    out = r.stdout
    assert "N = 8192" in out and "max-p = 12" in out and "1st-kind equation being solved" in out
    ps = [int(ln.rsplit(":", 1)[1]) for ln in out.splitlines() if ln.startswith("it: ")]
    assert ps[:5] == [12, 3, 2, 1, 1] and len(ps) == 6
    err = float([ln for ln in out.splitlines() if ln.startswith("external phi")][0].rsplit(":", 1)[1])
    assert abs(err - 6.2e-4) < 0.3e-4
    rel = float([ln for ln in out.splitlines() if ln.startswith("relative error")][0].split(":")[1])
    assert rel < 5e-3


@pytest.mark.gpu
def test_mesh_and_preconditioner_flags(tmp_path):
    # a sphere written as gmsh, read back through -mesh (winding swapped by the reader as the reference does)
    import numpy as np
    import fmm_bem_relaxed_amd as fb
    v = fb.unit_sphere(4)[:, [0, 2, 1], :]                     # pre-swap so that the reader's swap restores the outward normals
    msh = tmp_path / "sphere.msh"
    with open(msh, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % (3 * len(v)))
        for i, p in enumerate(v.reshape(-1, 3)):
            f.write("%d %.17g %.17g %.17g\n" % (i + 1, p[0], p[1], p[2]))
        f.write("$EndNodes\n$Elements\n%d\n" % len(v))
        for e in range(len(v)):
            f.write("%d 2 2 0 1 %d %d %d\n" % (e + 1, 3 * e + 1, 3 * e + 2, 3 * e + 3))
        f.write("$EndElements\n")
    for flags in (["-mesh", str(msh), "-p", "8"], ["-recursions", "4", "-p", "8", "-local"],
                  ["-recursions", "4", "-p", "8", "-diagonal"], ["-recursions", "4", "-p", "8", "-fgmres", "-fixed_p"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "LaplaceBEM.py")] + flags,
                           capture_output=True, text=True, check=True)
        rel = float([ln for ln in r.stdout.splitlines() if ln.startswith("relative error")][0].split(":")[1])
        assert rel < 3e-2, (flags, r.stdout[-400:])


@pytest.mark.gpu
def test_stokes_driver_sphere_drag():
    """examples/StokesBEM.py mirrors examples/StokesBEM.cpp: flow past the unit sphere, drag against 6 pi mu; orders follow
    GMRES_Stokes.hpp:229 (max(p_min, predict_p - 1)): first iteration at p - 1, never below p_min."""
    for flags, first_p in ((["-recursions", "4", "-p", "10"], 9), (["-recursions", "4", "-p", "10", "-local"], 10),
                           (["-recursions", "4", "-p", "8", "-diagonal", "-fixed_p"], 8)):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "StokesBEM.py")] + flags,
                           capture_output=True, text=True, check=True)
        out = r.stdout
        ps = [int(ln.rsplit(":", 1)[1]) for ln in out.splitlines() if ln.startswith("it: ")]
        assert ps[0] == first_p and min(ps) >= 5, (flags, ps)
        err = float([ln for ln in out.splitlines() if ln.startswith("error on a sphere")][0].split(":")[1])
        assert err < 2e-2, (flags, out[-600:])
        assert "Area error" in out and "POINTWISE ERRORS" in out
