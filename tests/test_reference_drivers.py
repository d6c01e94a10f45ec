"""The reference's OWN driver programs -- /root/reference/examples/LaplaceBEM.cpp and StokesBEM.cpp, read in place, not a line
changed -- built against the product: `include/fmmbem/compat/` (forwarding headers under the reference's file names) first on the
include path, the reference's examples/BEM behind it for the solver, preconditioner, mesh and timing headers the drivers pull in,
`-lfmmbem_hip`.  This is the switch a user of the reference makes.

CPU: they compile and link (build container only: the GPU box has no reference tree).  GPU: the binaries built in the container
by `make -C oracle ref` (oracle/_ref/, which travels) run and reproduce the reference's recorded output (SURVEY.md section 8d
config 5; tests/golden/reference_known_answers.json) and this repository's Python drivers."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference"
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def _compile(tmp_path, name):
    exe = str(tmp_path / name)
    lib = os.path.join(ROOT, "fmm-bem-relaxed_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I" + os.path.join(ROOT, "include", "fmmbem", "compat"),
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(REF, "examples", "BEM"),
                           os.path.join(REF, "examples", name + ".cpp"), "-o", exe, "-L" + lib, "-lfmmbem_hip",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "examples")), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", ["LaplaceBEM", "StokesBEM"])
def test_reference_drivers_compile_unmodified_against_the_adapter(tmp_path, gpu_available, name):
    exe = _compile(tmp_path, name)
    if not gpu_available:                                  # the driver does not catch: the adapter's exception ends it, loudly
        r = subprocess.run([exe, "-recursions", "3", "-p", "6"], capture_output=True, text=True)
        assert r.returncode != 0 and "no HIP device" in r.stderr


def _lines(out, prefix):
    return [ln for ln in out.splitlines() if ln.strip().startswith(prefix)]


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "LaplaceBEM_ref")), reason="oracle/_ref/LaplaceBEM_ref not built (make -C oracle ref)")
def test_reference_laplace_driver_reproduces_its_recorded_output():
    """`LaplaceBEM -recursions 6 -p 12 -theta 0.5`: 6 iterations, p = 12, 3, 2, 1, 1 printed, exterior-point error 6.2e-4 -- what the
    unmodified reference printed for the survey, now with the matvec on the GPU; and the -diagonal / -second_kind variants run."""
    exe = os.path.join(REFDIR, "LaplaceBEM_ref")
    r = subprocess.run([exe, "-recursions", "6", "-p", "12", "-theta", "0.5"], capture_output=True, text=True, check=True)
    out = r.stdout
    assert "N = 8192" in out and "1st-kind equation being solved" in out and "Solver: GMRES" in out
    ps = [int(ln.rsplit(":", 1)[1]) for ln in _lines(out, "it: ")]
    assert ps == [12, 3, 2, 1, 1]                               # the converged iteration is not printed (GMRES.hpp:216-219)
    assert "after 6 iterations" in _lines(out, "Final residual")[0]
    err = float(_lines(out, "external phi")[0].rsplit(":", 1)[1])
    assert abs(err - 6.2e-4) < 0.3e-4
    assert float(_lines(out, "relative error")[0].split(":")[1]) < 5e-3
    # the Python driver on the same problem prints the same schedule and the same error
    py = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "LaplaceBEM.py"), "-recursions", "6", "-p", "12", "-theta", "0.5"],
                        capture_output=True, text=True, check=True).stdout
    perr = float(_lines(py, "external phi")[0].rsplit(":", 1)[1])
    assert abs(perr - err) <= 1e-3 * err
    r = subprocess.run([exe, "-recursions", "5", "-p", "10", "-diagonal"], capture_output=True, text=True, check=True)
    assert "Preconditioner: Diagonal" in r.stdout and float(_lines(r.stdout, "relative error")[0].split(":")[1]) < 2e-2


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "StokesBEM_ref")), reason="oracle/_ref/StokesBEM_ref not built (make -C oracle ref)")
def test_reference_stokes_driver_runs_on_the_adapter():
    """`StokesBEM -recursions 4 -p 10`: flow past the unit sphere through the reference's own GMRES_Stokes.hpp above the GPU plan,
    beside examples/StokesBEM.py on the same problem: the same right-hand-side check, the same order schedule (first iteration at
    p - 1, never below p_min) and the same residual history to the printed digits while the two Arnoldi processes agree (ten
    iterations; then summation order shows), the same area error.  The reference's drag line is NOT compared: its loop
    (StokesBEM.cpp:343-352) never advances `i`, so its Fx is the first panel's traction times the total area."""
    exe = os.path.join(REFDIR, "StokesBEM_ref")
    out = subprocess.run([exe, "-recursions", "4", "-p", "10"], capture_output=True, text=True, check=True).stdout
    py = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "StokesBEM.py"), "-recursions", "4", "-p", "10"],
                        capture_output=True, text=True, check=True).stdout

    def history(text):
        its = [ln.split() for ln in _lines(text, "it: ")]
        return [int(t[-1]) for t in its], [float(t[3].rstrip(",")) for t in its]
    (pc, rc), (pp, rp) = history(out), history(py)
    assert pc[0] == 9 and min(pc) >= 5 and pc[:12] == pp[:12]
    assert all(abs(a - b) <= 2e-3 * b for a, b in zip(rc[:10], rp[:10]))
    assert abs(len(pc) - len(pp)) <= 2
    for tag in ("rhs error", "Area error"):
        a = float(_lines(out, tag)[0].split(":")[1]); b = float(_lines(py, tag)[0].split(":")[1])
        assert abs(a - b) <= 1e-4 * abs(b), tag
    assert float(_lines(py, "error on a sphere")[0].split(":")[1]) < 2e-2        # the drag, summed over the panels
