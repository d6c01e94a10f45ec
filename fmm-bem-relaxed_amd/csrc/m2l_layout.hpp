// m2l_layout.hpp -- LDS image of the M2L translation table and the lane -> output map, shared by the
// host (which tabulates them once per plan) and the kernel (which needs the row stride at compile time).
//
// In the M2L inner loop lane (j,k) reads Yh[(j+n), m-k] for a wave-uniform (n,m).  With the table stored
// with a CONSTANT row stride R (16-B slots), the slot is  j*R - k + [n*R + m + const]: per lane a constant
// plus a wave-uniform term.  ds_read_b128 services a wavefront in four fixed 16-lane groups
// (MI355X_MICROARCH.md, LDS table); a group is conflict-free iff its 16 slots are distinct mod 16.  So the
// outputs are dealt to the groups such that (j*R - k) mod 16 is distinct inside every group -- possible
// for every p <= 16 with the residues R mod 16 below (found by exhaustive search: the largest residue class
// never exceeds the number of 16-lane groups of the team).  Result: zero LDS bank conflicts in the hot loop
// (the dense n^2+n+m layout measured 2.2 LDS cycles per group).
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace fmmbem {

constexpr int kM2LResidue[17] = {0, 1, 2, 3, 4, 3, 3, 7, 3, 3, 7, 3, 2, 2, 2, 2, 2};   // index = p

constexpr int m2l_stride(int P) {            // smallest R >= 4P-1 with R % 16 == residue(P)
  int r = 4 * P - 1;
  while (r % 16 != kM2LResidue[P]) ++r;
  return r;
}
constexpr int m2l_team(int P) { return (P * (P + 1) / 2 + 63) / 64; }       // wavefronts per target box
constexpr int m2l_col0(int P) { return 2 * P - 1; }                         // slot of column c = 0 within a row
constexpr int m2l_lds_slots(int P) { return 2 * P * m2l_stride(P); }
constexpr int kM2LMaxThreads = 192;                                         // team of 3 wavefronts at p = 16

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
  const int R = m2l_stride(P), team = m2l_team(P), bins = 4 * team;
  out.assign(kM2LMaxThreads, -1);
  std::vector<std::vector<int32_t>> cls(16);
  for (int j = 0, idx = 0; j < P; ++j)
    for (int k = 0; k <= j; ++k, ++idx) cls[((j * R - k) % 16 + 16) % 16].push_back(j | (k << 8) | (idx << 16));
  std::vector<int> fill(bins, 0);
  int next_bin = 0;                          // rotate the starting bin so the bins fill evenly
  for (int r = 0; r < 16; ++r) {
    if ((int)cls[r].size() > bins) return false;
    for (std::size_t i = 0; i < cls[r].size(); ++i) {
      const int bin = (next_bin + (int)i) % bins;
      const int wave = bin / 4, grp = bin % 4;
      if (fill[bin] >= 16) return false;
      out[wave * 64 + b128_lane_groups()[grp][fill[bin]++]] = cls[r][i];
    }
    next_bin = (next_bin + (int)cls[r].size()) % bins;
  }
  return true;
}

// scatter map: linear table index r^2+r+c (r < 2P) -> LDS slot r*R + c + col0
inline void m2l_scatter_map(int P, std::vector<int32_t>& out) {
  const int R = m2l_stride(P), c0 = m2l_col0(P);
  out.assign(4 * P * P, 0);
  for (int r = 0; r < 2 * P; ++r)
    for (int c = -r; c <= r; ++c) out[r * r + r + c] = r * R + c + c0;
}

}  // namespace fmmbem
