// mesh_io.cpp -- host-side mesh input/output of the reference's drivers behind the C ABI (include/fmmbem.h):
//   MeshIO::readMsh        examples/BEM/MshReader.hpp:18-94      gmsh v2 ASCII, triangles only, winding swapped
//   MeshIO::ReadVertFace   examples/BEM/VertFaceReader.hpp:17-76 .vert / .face pair, 1-indexed faces
//   the .vert/.face writer examples/BEM/Triangulation.hpp:124-134, 243-254
//   RedBloodCell           examples/BEM/Triangulation.hpp:184-255 (identity rotation, zero shift)
// No device code; errors are status codes (the reference reads past a missing file silently).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/fmmbem.h"
#include "host_plan.hpp"

namespace fmmbem {
int fail(int status, const std::string& msg);          // plan.hip: records fmmbem_last_error()
}
using fmmbem::fail;

namespace {

// MeshIO::split (examples/BEM/MeshIO.hpp:11-30) splits on single white-space characters; the readers only use
// atoi/atof on the fields, for which plain stream extraction gives the same values.
std::vector<std::string> fields(const std::string& line) {
  std::vector<std::string> v;
  std::istringstream ss(line);
  std::string f;
  while (ss >> f) v.push_back(f);
  return v;
}

int emit(const std::vector<double>& tri, double* vertices, size_t* n_panels) {
  const size_t n = tri.size() / 9;
  if (vertices) {
    if (*n_panels < n) return fail(FMMBEM_ERR_INVALID, "vertices buffer too small for the mesh");
    for (size_t i = 0; i < tri.size(); ++i) vertices[i] = tri[i];
  }
  *n_panels = n;
  return FMMBEM_OK;
}

}  // namespace

extern "C" {

int fmmbem_mesh_read_msh(const char* path, double* vertices, size_t* n_panels) {
  if (!path || !n_panels) return fail(FMMBEM_ERR_INVALID, "null argument");
  std::ifstream mesh(path);
  if (!mesh) return fail(FMMBEM_ERR_IO, std::string("cannot open ") + path);
  std::string str;
  bool found = false;
  while (std::getline(mesh, str)) {                      // MshReader.hpp:30-34
    if (!str.empty() && str.back() == '\r') str.pop_back();
    if (str == "$Nodes") { found = true; break; }
  }
  if (!found || !std::getline(mesh, str)) return fail(FMMBEM_ERR_IO, "msh: no $Nodes section");
  const long num_nodes = std::atol(str.c_str());
  if (num_nodes <= 0) return fail(FMMBEM_ERR_IO, "msh: bad node count");
  std::vector<double> nodes(3 * (size_t)num_nodes, 0.0);
  for (long i = 0; i < num_nodes; ++i) {                 // :44-59, nodes[node_no-1]
    if (!std::getline(mesh, str)) return fail(FMMBEM_ERR_IO, "msh: truncated $Nodes");
    const auto f = fields(str);
    if (f.size() < 4) return fail(FMMBEM_ERR_IO, "msh: malformed node line");
    const long id = std::atol(f[0].c_str());
    if (id < 1 || id > num_nodes) return fail(FMMBEM_ERR_IO, "msh: node number out of range");
    for (int k = 0; k < 3; ++k) nodes[3 * (size_t)(id - 1) + k] = std::atof(f[1 + k].c_str());
  }
  std::getline(mesh, str);                               // $EndNodes
  std::getline(mesh, str);                               // $Elements
  if (!std::getline(mesh, str)) return fail(FMMBEM_ERR_IO, "msh: no $Elements section");
  const long num_elements = std::atol(str.c_str());
  if (num_elements < 0) return fail(FMMBEM_ERR_IO, "msh: bad element count");
  // The reference stores triangle number e at elements[e-1] and cuts the vector to (elements - skipped): exact
  // for an all-triangle file.  Here triangles are kept in element-number order and the others dropped, which is
  // the same thing for such a file and well defined for a mixed one.
  std::vector<std::vector<double>> byno((size_t)num_elements);
  for (long i = 0; i < num_elements; ++i) {
    if (!std::getline(mesh, str)) return fail(FMMBEM_ERR_IO, "msh: truncated $Elements");
    const auto f = fields(str);
    if (f.size() < 3) return fail(FMMBEM_ERR_IO, "msh: malformed element line");
    const long no = std::atol(f[0].c_str());
    if (std::atoi(f[1].c_str()) != 2) continue;          // :74-78 non-triangular element
    const int ntags = std::atoi(f[2].c_str());
    if (ntags < 0 || f.size() < (size_t)(3 + ntags + 3)) return fail(FMMBEM_ERR_IO, "msh: malformed triangle line");
    long v[3];
    for (int k = 0; k < 3; ++k) {
      v[k] = std::atol(f[3 + ntags + k].c_str()) - 1;
      if (v[k] < 0 || v[k] >= num_nodes) return fail(FMMBEM_ERR_IO, "msh: vertex number out of range");
    }
    if (no < 1 || no > num_elements) return fail(FMMBEM_ERR_IO, "msh: element number out of range");
    const long order[3] = {v[0], v[2], v[1]};            // :89 triangle_type(nodes[v1], nodes[v3], nodes[v2])
    std::vector<double>& t = byno[(size_t)(no - 1)];
    t.resize(9);
    for (int a = 0; a < 3; ++a)
      for (int k = 0; k < 3; ++k) t[3 * a + k] = nodes[3 * (size_t)order[a] + k];
  }
  std::vector<double> tri;
  for (const auto& t : byno) tri.insert(tri.end(), t.begin(), t.end());
  return emit(tri, vertices, n_panels);
}

int fmmbem_mesh_read_vert_face(const char* vert_path, const char* face_path, double* vertices, size_t* n_panels) {
  if (!vert_path || !face_path || !n_panels) return fail(FMMBEM_ERR_INVALID, "null argument");
  std::ifstream vert(vert_path), face(face_path);
  if (!vert) return fail(FMMBEM_ERR_IO, std::string("cannot open ") + vert_path);
  if (!face) return fail(FMMBEM_ERR_IO, std::string("cannot open ") + face_path);
  std::string str;
  if (!std::getline(vert, str)) return fail(FMMBEM_ERR_IO, "vert: empty file");
  const long nv = std::atol(str.c_str());                // VertFaceReader.hpp:30-33
  if (nv <= 0) return fail(FMMBEM_ERR_IO, "vert: bad vertex count");
  std::vector<double> xyz(3 * (size_t)nv);
  for (long i = 0; i < nv; ++i) {
    if (!std::getline(vert, str)) return fail(FMMBEM_ERR_IO, "vert: truncated");
    const auto f = fields(str);
    if (f.size() < 3) return fail(FMMBEM_ERR_IO, "vert: malformed line");
    for (int k = 0; k < 3; ++k) xyz[3 * (size_t)i + k] = std::atof(f[k].c_str());
  }
  if (!std::getline(face, str)) return fail(FMMBEM_ERR_IO, "face: empty file");
  const long nf = std::atol(str.c_str());                // :54-57
  if (nf < 0) return fail(FMMBEM_ERR_IO, "face: bad face count");
  std::vector<double> tri(9 * (size_t)nf);
  for (long i = 0; i < nf; ++i) {
    if (!std::getline(face, str)) return fail(FMMBEM_ERR_IO, "face: truncated");
    const auto f = fields(str);
    if (f.size() < 3) return fail(FMMBEM_ERR_IO, "face: malformed line");
    for (int a = 0; a < 3; ++a) {
      const long v = std::atol(f[a].c_str()) - 1;        // :67-71, 1-indexed, winding kept
      if (v < 0 || v >= nv) return fail(FMMBEM_ERR_IO, "face: vertex number out of range");
      for (int k = 0; k < 3; ++k) tri[9 * (size_t)i + 3 * a + k] = xyz[3 * (size_t)v + k];
    }
  }
  return emit(tri, vertices, n_panels);
}

int fmmbem_mesh_write_vert_face(const char* vert_path, const char* face_path, const double* vertices, size_t n_panels) {
  if (!vert_path || !face_path || (!vertices && n_panels)) return fail(FMMBEM_ERR_INVALID, "null argument");
  std::ofstream vert(vert_path), face(face_path);
  if (!vert) return fail(FMMBEM_ERR_IO, std::string("cannot open ") + vert_path);
  if (!face) return fail(FMMBEM_ERR_IO, std::string("cannot open ") + face_path);
  // Triangulation.hpp:124-134 writes three vertices per triangle at the stream's default precision and no count
  // lines -- files its own reader cannot read back (ReadVertFace wants the counts first, VertFaceReader.hpp:30,54).
  // The counts are written here, and the coordinates with 17 significant digits so that a round trip is exact.
  vert.precision(17);
  vert << 3 * n_panels << "\n";
  face << n_panels << "\n";
  size_t vnum = 1;
  for (size_t i = 0; i < n_panels; ++i, vnum += 3) {
    for (int a = 0; a < 3; ++a)
      vert << vertices[9 * i + 3 * a] << " " << vertices[9 * i + 3 * a + 1] << " " << vertices[9 * i + 3 * a + 2] << "\n";
    face << vnum << ' ' << vnum + 1 << ' ' << vnum + 2 << "\n";
  }
  return (vert && face) ? FMMBEM_OK : fail(FMMBEM_ERR_IO, "write failed");
}

int fmmbem_quadrature(int key, double* points, double* weights, int* n) {
  if (!n) return fail(FMMBEM_ERR_INVALID, "null argument");
  fmmbem::QuadRule r;
  if (!fmmbem::quad_rule(key, r)) return fail(FMMBEM_ERR_INVALID, "invalid Gauss rule key (valid: 1 3 4 7 13 17 19 25)");
  for (int q = 0; q < r.n; ++q) {
    if (points) for (int k = 0; k < 3; ++k) points[3 * q + k] = r.pts[q][k];
    if (weights) weights[q] = r.w[q];
  }
  *n = r.n;
  return FMMBEM_OK;
}

int fmmbem_mesh_red_blood_cell(int recursions, double* vertices, size_t* n_panels) {
  if (!n_panels) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (recursions < 1 || recursions > 12) return fail(FMMBEM_ERR_INVALID, "recursions outside [1, 12]");
  const size_t n = (size_t)2 << (2 * recursions);
  if (vertices) {
    if (*n_panels < n) return fail(FMMBEM_ERR_INVALID, "vertices buffer too small for the mesh");
    fmmbem::unit_sphere(recursions, vertices);            // same octahedron subdivision, Triangulation.hpp:215-220
    const double r = 3.91, C0 = 0.81, C2 = 7.83, C4 = -4.39;   // ConvertRedBloodCellTriangle, :184-206
    for (size_t i = 0; i < 3 * n; ++i) {
      double* v = vertices + 3 * i;
      const double x = v[0] * r, y = v[1] * r;
      const double rho = std::sqrt(x * x + y * y), ratio = rho / r;
      const int sg = (0 < v[2]) - (v[2] < 0);
      const double z = std::sqrt(1 - ratio * ratio + 1e-12) * (C0 + C2 * ratio * ratio + C4 * ratio * ratio * ratio * ratio) * 0.5 * sg;
      v[0] = x; v[1] = y; v[2] = z;
    }
  }
  *n_panels = n;
  return FMMBEM_OK;
}

// Triangulation::MultipleRedBloodCell (examples/BEM/Triangulation.hpp:260-321): `cells` cells of 2*4^recursions panels,
// cell i rotated by RotationMatrix(alpha, beta, gamma) (:142-163) and shifted by s.  placement = cells x {alpha, beta,
// gamma, sx, sy, sz}, or NULL for the reference's own sequence: it draws from the UNSEEDED drand48 stream (reproduced here
// with a private erand48 state at the libc default), three angles per cell -- as arguments of one call, which g++
// evaluates right to left (gamma first); that order is the one assumption this generator cannot pin without running
// the reference -- then for i > 0: sy += 2*3.91 + 3.91 u, sx = i u + 2.83 u, sz = +-10 u.
int fmmbem_mesh_red_blood_cells(int recursions, int cells, const double* placement, double* vertices, size_t* n_panels) {
  if (!n_panels) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (recursions < 1 || recursions > 12 || cells < 1) return fail(FMMBEM_ERR_INVALID, "recursions outside [1, 12] or cells < 1");
  const size_t per = (size_t)2 << (2 * recursions), n = per * (size_t)cells;
  if (vertices) {
    if (*n_panels < n) return fail(FMMBEM_ERR_INVALID, "vertices buffer too small for the mesh");
    unsigned short st[3] = {0x330E, 0xABCD, 0x1234};
    double sx = 0, sy = 0, sz = 0;
    for (int i = 0; i < cells; ++i) {
      double ang[3], sh[3];
      if (placement) {
        for (int k = 0; k < 3; ++k) { ang[k] = placement[6 * i + k]; sh[k] = placement[6 * i + 3 + k]; }
      } else {
        ang[2] = erand48(st); ang[1] = erand48(st); ang[0] = erand48(st);
        if (i > 0) {
          sy += 2 * 3.91 + 3.91 * erand48(st);
          sx = i * erand48(st) + 2.83 * erand48(st);
          sz = ((i % 2 == 0) ? 1 : -1) * 10 * erand48(st);
        }
        sh[0] = sx; sh[1] = sy; sh[2] = sz;
      }
      double* v = vertices + 9 * per * (size_t)i;
      size_t one = per;
      if (int rc = fmmbem_mesh_red_blood_cell(recursions, v, &one)) return rc;
      const double ca = std::cos(ang[0]), cb = std::cos(ang[1]), cg = std::cos(ang[2]);
      const double sa = std::sin(ang[0]), sb = std::sin(ang[1]), sg = std::sin(ang[2]);
      const double M[3][3] = {{cb * cg, -cb * sg, sb},
                              {ca * sg + cg * sa * sb, ca * cg - sa * sb * sg, -cb * sa},
                              {sa * sg - ca * cg * sb, cg * sa + ca * sb * sg, ca * cb}};
      for (size_t q = 0; q < 3 * per; ++q) {
        double* x = v + 3 * q;
        const double r[3] = {M[0][0] * x[0] + M[0][1] * x[1] + M[0][2] * x[2], M[1][0] * x[0] + M[1][1] * x[1] + M[1][2] * x[2],
                             M[2][0] * x[0] + M[2][1] * x[1] + M[2][2] * x[2]};
        for (int k = 0; k < 3; ++k) x[k] = r[k] + sh[k];
      }
    }
  }
  *n_panels = n;
  return FMMBEM_OK;
}

}  // extern "C"
