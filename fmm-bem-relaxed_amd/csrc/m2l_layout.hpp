// m2l_layout.hpp -- LDS image of the M2L translation table and the lane -> output map, shared by the
// host (which tabulates them once per plan) and the kernel (which needs the strides at compile time).
//
// The rescaled singular harmonics of a translation factor as  Yh[r,c] = Z^c * gh[r,c]  with Z = i e^{i beta} and
// gh[r,c] = G[r,|c|] * (c < 0 ? (-1)^c : 1) REAL (kernels_m2l.hip header).  Only gh goes through the LDS: in the
// inner loop lane (j,k) needs gh[j+n, m-k] for a wave-uniform (n,m), and takes n and n+1 with ONE ds_read_b128.
// The image is column-major in r with an even column stride Rr (8-byte units):
//     copy E:  gh[r,c] at        (c + C0) * Rr + r          pair (r, r+1) is 16-byte aligned for even r
//     copy O:  gh[r,c] at  O0 +  (c + C0) * Rr + r + 1      ...                               for odd  r
// With n even the parity of r = j + n is the parity of j: a lane always reads the same copy, and its 16-byte slot is
//     [cb(j) + j - k*Rr] / 2  +  [(m + C0) * Rr + n] / 2  =  lane constant + wave-uniform term.
// ds_read_b128 services a wavefront in four fixed 16-lane groups (MI355X_MICROARCH.md, LDS table); a group is
// conflict-free iff its 16 slots are distinct mod 16.  The outputs are dealt to the groups so that the lane constant
//     lambda(j,k) = j/2 - k*h (j even),  D + (j+1)/2 - k*h (j odd),   h = Rr/2, D = O0/2 mod 16
// is distinct mod 16 inside every group -- possible for every p <= 16 with the (h, D) below (exhaustive search for
// the smallest h whose largest residue class fits the fewest groups): zero LDS bank conflicts in the hot loop.
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace fmmbem {

constexpr int kM2LHalfStride[17] = {0, 2, 3, 4, 5, 6, 7, 9, 9, 10, 13, 12, 13, 14, 15, 17, 17};   // h, index = p
constexpr int kM2LOddShift[17] = {0, 0, 0, 0, 0, 0, 0, 3, 2, 1, 4, 0, 0, 0, 3, 7, 4};             // D, index = p

constexpr int m2l_rr(int P) { return 2 * kM2LHalfStride[P]; }                 // column stride, doubles (>= 2P+2)
constexpr int m2l_c0(int P) { return 2 * P - 2; }                             // column of c = 0 (c in [-(2P-2), P-1])
constexpr int m2l_ncols(int P) { return 3 * P - 2; }
constexpr int m2l_copy_doubles(int P) { return m2l_ncols(P) * m2l_rr(P); }
constexpr int m2l_odd_base(int P) {          // O0: smallest even offset behind copy E with (O0/2) % 16 == D
  int o = (m2l_copy_doubles(P) + 2 + 1) & ~1;
  while ((o / 2) % 16 != kM2LOddShift[P]) o += 2;
  return o;
}
constexpr int m2l_lds_doubles(int P) { return m2l_odd_base(P) + m2l_copy_doubles(P) + 2; }
constexpr int m2l_entries(int P) { return P * (2 * P + 1); }                  // G[r,a], a <= r < 2P
constexpr int m2l_team(int P) { return (P * (P + 1) / 2 + 63) / 64; }         // wavefronts per target box
constexpr int kM2LMaxThreads = 192;                                           // team of 3 wavefronts at p = 16

// the four 16-lane groups of ds_read_b128 within one wavefront
inline const std::array<std::array<int, 16>, 4>& b128_lane_groups() {
  static const std::array<std::array<int, 16>, 4> g = {{
      {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
      {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
      {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
      {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}}};
  return g;
}

// lane map for order P: entry t (thread within the team) = j | k << 8 | stored_index << 16, or -1.
// Returns false if the dealing fails (never for p <= 16; checked by the caller).
inline bool m2l_lane_map(int P, std::vector<int32_t>& out) {
  const int h = kM2LHalfStride[P], D = kM2LOddShift[P], team = m2l_team(P), bins = 4 * team;
  out.assign(kM2LMaxThreads, -1);
  std::vector<std::vector<int32_t>> cls(16);
  for (int j = 0, idx = 0; j < P; ++j)
    for (int k = 0; k <= j; ++k, ++idx) {
      const int lam = (j & 1) ? D + (j + 1) / 2 - k * h : j / 2 - k * h;
      cls[(lam % 16 + 16) % 16].push_back(j | (k << 8) | (idx << 16));
    }
  std::vector<int> fill(bins, 0);
  // member i of every residue class goes to 16-lane group i: residues inside a group stay distinct, and the outputs
  // occupy the FEWEST groups -- a ds_read_b128 costs one LDS cycle per group that has a live lane, and M2L runs at the
  // LDS limit.  (h, D) above are chosen so that the largest class fits: 2 groups at p = 6, 7; 3 at p = 8; 4 at p = 9,
  // 10; from p = 11 the first wavefront is filled before the second.
  for (int r = 0; r < 16; ++r) {
    if ((int)cls[r].size() > bins) return false;
    for (std::size_t i = 0; i < cls[r].size(); ++i) {
      const int bin = (int)i, wave = bin / 4, grp = bin % 4;
      if (fill[bin] >= 16) return false;
      out[wave * 64 + b128_lane_groups()[grp][fill[bin]++]] = cls[r][i];
    }
  }
  return true;
}

// scatter map: class-table entry e = r(r+1)/2 + a (a <= r < 2P) -> its (up to) four LDS places
// {c = +a, c = -a} x {copy E, copy O}, each encoded as  (index in doubles) << 1 | negate,  -1 = unused.
// c = +a is read only for a <= P-1, c = -a only for 1 <= a <= 2P-2; gh[r,-a] = (-1)^a G[r,a].
inline void m2l_scatter_map(int P, std::vector<int32_t>& out) {
  const int Rr = m2l_rr(P), c0 = m2l_c0(P), o0 = m2l_odd_base(P);
  out.assign((std::size_t)4 * m2l_entries(P), -1);
  for (int r = 0; r < 2 * P; ++r)
    for (int a = 0; a <= r; ++a) {
      int32_t* q = out.data() + 4 * (r * (r + 1) / 2 + a);
      if (a <= P - 1) {
        q[0] = ((a + c0) * Rr + r) << 1;
        q[1] = (o0 + (a + c0) * Rr + r + 1) << 1;
      }
      if (a >= 1 && a <= 2 * P - 2) {
        q[2] = (((-a + c0) * Rr + r) << 1) | (a & 1);
        q[3] = ((o0 + (-a + c0) * Rr + r + 1) << 1) | (a & 1);
      }
    }
}

}  // namespace fmmbem
