// shift_ops.hpp -- M2M and L2L as precomputed sparse operators.
//
// The reference evaluates M2M / L2L with four nested loops per box pair, recomputing
// (-1)^n A[n,m] A[j-n,k-m] / A[j,k], an i^{...} phase (through std::pow(CI,...)) and range tests for every
// term (kernel/LaplaceSpherical.hpp:255-282, 385-410).  None of that depends on the boxes: the set of
// terms of output row (j,k), the source coefficient each term reads (conjugated or not), the harmonic
// Y[n,m] it multiplies and its real factor are fixed by p alone.  They are generated ONCE on the host, in
// the reference's loop order, and stored in ELL form (term i of every row contiguous across rows) so that
// lanes = rows read them coalesced.  Only the harmonics Y depend on the translation (one small table per
// parent/child offset class).
#pragma once
#include <cstdint>
#include <vector>

namespace fmmbem {

struct ShiftOps {
  int P = 0, S = 0;
  // ELL arrays, entry [i * S + row]
  int up_maxlen = 0, down_maxlen = 0;
  std::vector<uint16_t> up_src, up_y;      // src: stored index | 0x8000 if conjugated; y: n^2+n+m of the harmonic
  std::vector<double> up_real;             // real factor, EPS included
  std::vector<int> up_len;                 // [S] terms per row (independent of the order p <= P in use)
  std::vector<uint16_t> down_src, down_y;
  std::vector<double> down_real;
  std::vector<int> down_cnt;               // [(p-1) * S + row] terms usable at order p (terms sorted by n)
};

// A: the Anm table (index n^2+n+m, n < 2*kPmax), EPS-scaled as in LaplaceSpherical::precompute
inline ShiftOps build_shift_ops(int P, const std::vector<double>& A, double eps) {
  ShiftOps o;
  o.P = P;
  o.S = P * (P + 1) / 2;
  struct Term { uint16_t src, y; double real; int n; };
  std::vector<std::vector<Term>> up(o.S), down(o.S);
  auto oe = [](int n) { return (n & 1) ? -1.0 : 1.0; };
  for (int j = 0; j < P; ++j)
    for (int k = 0; k <= j; ++k) {
      const int jk = j * j + j + k, jks = j * (j + 1) / 2 + k;
      // ---- M2M (kernel/LaplaceSpherical.hpp:255-282) ----
      for (int n = 0; n <= j; ++n) {
        for (int m = -n; m <= (k - 1 < n ? k - 1 : n); ++m)
          if (j - n >= k - m) {
            const int jnkm = (j - n) * (j - n) + j - n + k - m, jnkms = (j - n) * (j - n + 1) / 2 + k - m, nm = n * n + n + m;
            const double ph = (m < 0 && (m & 1)) ? -1.0 : 1.0;                       // i^{m-|m|}
            up[jks].push_back({(uint16_t)jnkms, (uint16_t)nm, eps * ph * oe(n) * A[nm] * A[jnkm] / A[jk], n});
          }
        for (int m = k; m <= n; ++m)
          if (j - n >= m - k) {
            const int jnkm = (j - n) * (j - n) + j - n + k - m, jnkms = (j - n) * (j - n + 1) / 2 - k + m, nm = n * n + n + m;
            up[jks].push_back({(uint16_t)(jnkms | 0x8000), (uint16_t)nm, eps * oe(k + n + m) * A[nm] * A[jnkm] / A[jk], n});
          }
      }
      // ---- L2L (kernel/LaplaceSpherical.hpp:385-410) ----
      for (int n = j; n < P; ++n) {
        for (int m = j + k - n; m < 0; ++m) {
          const int jnkm = (n - j) * (n - j) + n - j + m - k, nm = n * n + n - m, nms = n * (n + 1) / 2 - m;
          down[jks].push_back({(uint16_t)(nms | 0x8000), (uint16_t)jnkm, eps * oe(k) * A[jnkm] * A[jk] / A[nm], n});
        }
        for (int m = 0; m <= n; ++m) {
          const int dmk = m - k, a = dmk < 0 ? -dmk : dmk;
          if (n - j >= a) {
            const int jnkm = (n - j) * (n - j) + n - j + m - k, nm = n * n + n + m, nms = n * (n + 1) / 2 + m;
            const double ph = (dmk < 0 && (dmk & 1)) ? -1.0 : 1.0;                   // i^{m-k-|m-k|}
            down[jks].push_back({(uint16_t)nms, (uint16_t)jnkm, eps * ph * A[jnkm] * A[jk] / A[nm], n});
          }
        }
      }
    }
  auto pack = [&](const std::vector<std::vector<Term>>& rows, int& maxlen, std::vector<uint16_t>& src,
                  std::vector<uint16_t>& y, std::vector<double>& real) {
    maxlen = 0;
    for (auto& r : rows) maxlen = (int)r.size() > maxlen ? (int)r.size() : maxlen;
    maxlen = (maxlen + 3) & ~3;                          // kernels consume terms four at a time; padding terms are zeros
    src.assign((size_t)maxlen * o.S, 0);
    y.assign((size_t)maxlen * o.S, 0);
    real.assign((size_t)maxlen * o.S, 0.0);
    for (int row = 0; row < o.S; ++row)
      for (size_t i = 0; i < rows[row].size(); ++i) {
        src[i * o.S + row] = rows[row][i].src;
        y[i * o.S + row] = rows[row][i].y;
        real[i * o.S + row] = rows[row][i].real;
      }
  };
  pack(up, o.up_maxlen, o.up_src, o.up_y, o.up_real);
  pack(down, o.down_maxlen, o.down_src, o.down_y, o.down_real);
  o.up_len.resize(o.S);
  for (int row = 0; row < o.S; ++row) o.up_len[row] = (int)up[row].size();
  o.down_cnt.assign((size_t)P * o.S, 0);
  for (int p = 1; p <= P; ++p)
    for (int row = 0; row < o.S; ++row) {
      int c = 0;
      for (auto& t : down[row]) c += t.n < p;          // terms are generated with n ascending
      o.down_cnt[(size_t)(p - 1) * o.S + row] = c;
    }
  return o;
}

}  // namespace fmmbem
