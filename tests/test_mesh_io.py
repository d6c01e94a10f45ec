"""Mesh input/output of the reference's drivers behind the C ABI: MeshIO::readMsh
(examples/BEM/MshReader.hpp:18-94), MeshIO::ReadVertFace (VertFaceReader.hpp:17-76), the .vert/.face dump and
Triangulation::RedBloodCell (Triangulation.hpp:124-134, 184-255).  tests/golden/tetra_mixed.msh is a hand-written
gmsh 2.2 file: 4 nodes, one point element, one line element and four triangles (one with three tags)."""
import os

import numpy as np
import pytest

from conftest import ROOT


def test_read_msh_triangles_only_winding_swapped(fb):
    v = fb.read_msh(os.path.join(ROOT, "tests", "golden", "tetra_mixed.msh"))
    nodes = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
    tri = np.array([[1, 2, 3], [1, 2, 4], [1, 3, 4], [2, 3, 4]]) - 1          # as written in the file
    assert v.shape == (4, 3, 3)
    # MshReader.hpp:89: triangle_type(nodes[v1], nodes[v3], nodes[v2])
    assert np.array_equal(v, nodes[tri[:, [0, 2, 1]]])


def test_vert_face_round_trip_is_exact(fb, tmp_path):
    v = fb.unit_sphere(3)
    vp, fp = str(tmp_path / "s.vert"), str(tmp_path / "s.face")
    fb.write_vert_face(vp, fp, v)
    assert open(fp).readline().strip() == str(len(v)) and open(vp).readline().strip() == str(3 * len(v))
    assert np.array_equal(fb.read_vert_face(vp, fp), v)
    # a shared-vertex file the way an external mesher writes it
    open(vp, "w").write("4\n0 0 0\n1 0 0\n0 1 0\n0 0 1\n")
    open(fp, "w").write("2\n1 2 3\n1 3 4\n")
    w = fb.read_vert_face(vp, fp)
    assert np.array_equal(w[1], [[0, 0, 0], [0, 1, 0], [0, 0, 1]])             # winding kept (VertFaceReader.hpp:73)


def test_red_blood_cell_matches_oracle_map(fb, oracle_mod):
    for r in (2, 4):
        assert np.array_equal(fb.red_blood_cell(r), oracle_mod.red_blood_cell(r))
    v = fb.red_blood_cell(3)
    assert v.shape == (128, 3, 3) and abs(np.abs(v[..., 0]).max() - 3.91) < 1e-12


def test_errors_are_codes(fb, tmp_path):
    with pytest.raises(fb.FmmBemError) as e:
        fb.read_msh(str(tmp_path / "missing.msh"))
    assert e.value.status == 7
    bad = tmp_path / "bad.msh"
    bad.write_text("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n2\n1 0 0 0\n")
    with pytest.raises(fb.FmmBemError) as e:
        fb.read_msh(str(bad))
    assert e.value.status == 7
    (tmp_path / "v").write_text("1\n0 0 0\n")
    (tmp_path / "f").write_text("1\n1 2 3\n")
    with pytest.raises(fb.FmmBemError):
        fb.read_vert_face(str(tmp_path / "v"), str(tmp_path / "f"))           # vertex number out of range


def test_multiple_red_blood_cells(fb):
    """Triangulation::MultipleRedBloodCell (examples/BEM/Triangulation.hpp:260-321): cells of 2*4^r panels, each a rigid
    motion of the single cell; the first cell of the reference's own sequence is rotated but not shifted (:273-279)."""
    one = fb.red_blood_cell(3)
    place = np.array([[0, 0, 0, 0, 0, 0], [0.3, -0.2, 1.1, 5.0, 9.0, -2.0]])
    v = fb.red_blood_cells(3, 2, place)
    assert v.shape == (2 * len(one), 3, 3) and np.array_equal(v[:len(one)], one)
    a, b, g = place[1, :3]
    ca, cb, cg, sa, sb, sg = np.cos(a), np.cos(b), np.cos(g), np.sin(a), np.sin(b), np.sin(g)
    M = np.array([[cb * cg, -cb * sg, sb], [ca * sg + cg * sa * sb, ca * cg - sa * sb * sg, -cb * sa],
                  [sa * sg - ca * cg * sb, cg * sa + ca * sb * sg, ca * cb]])           # RotationMatrix, :142-163
    assert np.allclose(M @ M.T, np.eye(3), atol=1e-15)
    assert np.allclose(v[len(one):], one @ M.T + place[1, 3:], rtol=0, atol=1e-14)
    w = fb.red_blood_cells(3, 3)                        # the reference's drand48-driven placement: deterministic
    assert np.array_equal(w, fb.red_blood_cells(3, 3)) and w.shape == (3 * len(one), 3, 3)
    d = lambda c: np.linalg.norm(c.reshape(-1, 3), axis=1)
    assert np.allclose(np.sort(d(w[:len(one)])), np.sort(d(one)), atol=1e-13)          # cell 0: rotation only
    cen = [w[i * len(one):(i + 1) * len(one)].reshape(-1, 3).mean(axis=0) for i in range(3)]
    assert cen[1][1] - cen[0][1] >= 2 * 3.91 - 1e-9 and cen[2][1] - cen[1][1] >= 2 * 3.91 - 1e-9   # cells do not overlap
    with pytest.raises(fb.FmmBemError):
        fb.red_blood_cells(3, 0)
