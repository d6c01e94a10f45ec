// shift_ops.hpp -- M2M and L2L as precomputed sparse operators.
//
// The reference evaluates M2M / L2L with four nested loops per box pair, recomputing
// (-1)^n A[n,m] A[j-n,k-m] / A[j,k], an i^{...} phase (through std::pow(CI,...)) and range tests for every
// term (kernel/LaplaceSpherical.hpp:255-282, 385-410).  None of that depends on the boxes: the set of
// terms of output row (j,k), the source coefficient each term reads (conjugated or not), the harmonic
// Y[n,m] it multiplies and its real factor are fixed by p alone.  They are generated ONCE on the host, in
// the reference's loop order, and stored in ELL form over equal-sized pieces of rows (VOp below) so that
// lanes read them coalesced.  Only the harmonics Y depend on the translation (one small table per
// parent/child offset class).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace fmmbem {

// One operator (up or down) at one order p with its terms dealt to "virtual rows" of at most T terms: output row
// (j,k) of L2L has (p-j)^2 terms and of M2M (j+1)^2 -- 100 down to 1 at p = 10 -- so with lanes = rows a wavefront
// runs 100 steps at 19 % lane use.  Rows are cut into pieces of T terms (reference order kept inside a piece and
// between pieces), pieces are the lanes' work (ELL over pieces, coalesced), and a row sums its pieces in order.
struct VOp {
  int T = 8, V = 0, maxp = 0, S = 0;     // V: virtual rows, padded to a multiple of 64; S = p(p+1)/2
  std::vector<uint16_t> src, y;          // [i * V + v], i < T
  std::vector<double> real;
  std::vector<uint16_t> npiece;          // [row] pieces of the row
  std::vector<uint16_t> piece;           // [k * S + row] -> v, k < maxp
};

struct ShiftOps {
  int P = 0, S = 0;
  std::vector<VOp> up_v, down_v;           // index p - 1
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
  auto deal = [&](const std::vector<std::vector<Term>>& rows, int p, bool prefix) {
    VOp v;
    v.S = p * (p + 1) / 2;
    // terms per piece, same box, N = 1M, M2M + L2L ms with T = 2 / 4 / 8 / 16: p = 1 0.157 / 0.154 / 0.191, p = 3 - / 0.167 / 0.195 / 0.244,
    // p = 5 0.241 / 0.214 / 0.229, p = 6 - / 0.220 / 0.237 / 0.277, p = 7 0.373 / 0.286 / 0.249, p = 10 - / 0.468 / 0.427 / 0.440;
    // 16 above p = 12 keeps the per-wavefront partial-sum array small
    // 12 terms at the middle orders: M2M better at p = 8, 10 (0.147 / 0.211 against 0.166 / 0.220) but worse at p = 7, 9 (0.146 / 0.200
    // against 0.13 / 0.18); L2L worse at p = 8, 10, better at p = 12 (0.298 against 0.330, the one case taken; prefix = the L2L operator)
    v.T = p <= 6 ? 4 : p > 12 ? 16 : (prefix && p == 12) ? 12 : 8;
    struct Piece { int row, begin, cnt, k; };
    std::vector<Piece> pieces;
    v.npiece.assign(v.S, 0);
    for (int row = 0; row < v.S; ++row) {
      int len = 0;
      if (prefix) { for (auto& t : rows[row]) len += t.n < p; } else len = (int)rows[row].size();
      for (int b = 0, k = 0; b < len; b += v.T, ++k) { pieces.push_back({row, b, len - b < v.T ? len - b : v.T, k}); v.npiece[row] = (uint16_t)(k + 1); }
      v.maxp = v.npiece[row] > v.maxp ? v.npiece[row] : v.maxp;
    }
    std::stable_sort(pieces.begin(), pieces.end(), [](const Piece& a, const Piece& b) { return a.cnt > b.cnt; });
    v.V = ((int)pieces.size() + 63) & ~63;
    v.src.assign((size_t)v.T * v.V, 0); v.y.assign((size_t)v.T * v.V, 0); v.real.assign((size_t)v.T * v.V, 0.0);
    v.piece.assign((size_t)(v.maxp ? v.maxp : 1) * v.S, 0);
    for (size_t q = 0; q < pieces.size(); ++q) {
      const Piece& pc = pieces[q];
      v.piece[(size_t)pc.k * v.S + pc.row] = (uint16_t)q;
      for (int i = 0; i < pc.cnt; ++i) {
        const Term& t = rows[pc.row][pc.begin + i];
        v.src[(size_t)i * v.V + q] = t.src; v.y[(size_t)i * v.V + q] = t.y; v.real[(size_t)i * v.V + q] = t.real;
      }
    }
    return v;
  };
  for (int p = 1; p <= P; ++p) { o.up_v.push_back(deal(up, p, false)); o.down_v.push_back(deal(down, p, true)); }
  return o;
}

}  // namespace fmmbem
