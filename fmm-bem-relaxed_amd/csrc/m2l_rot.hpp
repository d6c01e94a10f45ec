// m2l_rot.hpp -- M2L by rotation, axial translation, rotation back: the structure shared by the host (which tabulates
// the constants once per plan) and the kernel (whose fully unrolled code consumes them in exactly this order).
//
// Reference operator: LaplaceSpherical::M2L (kernel/LaplaceSpherical.hpp:296-329), an O(p^4) double sum per pair.  Same
// linear map, factored (the "point and shoot" form of White & Head-Gordon / Greengard & Rokhlin '97), O(p^3):
//     L  =  E(-beta) X^T E(-alpha) conj(X)  .  Tz(rho)  .  X^T E(alpha) conj(X) E(beta)  M
// for the translation vector (rho, alpha, beta) = cart2sph(c_target - c_source), acting on the coefficient vectors of
// every degree n (orders m = -n..n, negative orders by conjugation, as the reference stores them):
//   E(g)   = diag(e^{i m g})                     rotation about z
//   X      = the fixed rotation that takes the y axis to the z axis (by pi/2 about x), in this basis
//            X[m,m'] = i^{3|m| + |m'|} d^n_{m,m'}(pi/2)   (Wigner small-d; verified against sampled harmonics)
//            so that  X^T E(alpha) conj(X)  is the rotation by alpha about y -- no per-translation matrix is needed
//   Tz     = M2L along +z, which couples equal orders only:  L'[j,k] = sum_{n>=k} Tz[j,n,k] rho^{-(j+n+1)} M'[n,k],
//            Tz[j,n,k] = (-1)^{k+j} A[n,k] A[j,k] EPS / A[j+n,0]  (the reference's Cnm entry for m = k, :106-116)
// In real coordinates (a_0; a_m = Re, b_m = Im, m = 1..n) conj(X) and X^T are real and three quarters empty: an entry
// (m, m' >= 1) carries ONE real coefficient  s = d[m,m'] + d[m,-m']  (n+m even, acts on a_m')  or  d[m,m'] - d[m,-m']
// (n+m odd, acts on b_m'), routed by the phase i^{m+3m'} to Re or Im of the output with a sign; X^T has the same
// entries times (-1)^{m+m'}.  n^2+n+1 multiply-adds per degree instead of (2n+1)^2 complex ones.
// At p = 10: 4 x 340 (fixed rotations) + 4 x 180 (z rotations) + 670 (axial translation) + 200 (powers of rho) ~ 3 000
// FMAs per pair against 15 400 for the double sum -- and every lane works on its own (target, source) pair out of
// registers: no LDS image, no bank-conflict layout, no cliff where p(p+1)/2 passes 64.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace fmmbem {

constexpr int kRotPmax = 12;                       // orders the register-resident kernel is instantiated for

// structural zero of the real rotation block of degree n at (output order m, input order mp)
constexpr bool rot_live(int n, int m, int mp) {
  if (mp == 0) return ((n + m) & 1) == 0;          // d[m,0] = 0 for n+m odd
  return !(m == 0 && ((n + mp) & 1));              // d[0,mp] = 0 for n+mp odd
}
constexpr int rot_nnz(int n) {
  int c = 0;
  for (int m = 0; m <= n; ++m)
    for (int mp = 0; mp <= n; ++mp) c += rot_live(n, m, mp) ? 1 : 0;
  return c;
}
constexpr int rot_off(int n) {                     // first coefficient of degree n (degrees stored one after another)
  int c = 0;
  for (int i = 0; i < n; ++i) c += rot_nnz(i);
  return c;
}
constexpr int tz_off(int P, int k) {               // axial block of order k at expansion order P: (P-k) x (P-k), [j-k][n-k]
  int c = rot_off(P);
  for (int i = 0; i < k; ++i) c += (P - i) * (P - i);
  return c;
}
constexpr int rot_table_doubles(int P) { return (tz_off(P, P) + 7) & ~7; }

// d^j_{mp,m}(pi/2): 2^-j sqrt((j+mp)!(j-mp)!/((j+m)!(j-m)!)) sum_s (-1)^{mp-m+s} C(j+m,s) C(j-m,mp-m+s); the sum is an
// exact integer (|terms| sum to C(2j, j-mp) <= C(30,15))
inline double wigner_d_half(int j, int mp, int m) {
  auto binom = [](int n, int k) -> long long {
    if (k < 0 || k > n) return 0;
    long long r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return r;
  };
  long long tot = 0;
  const int s0 = m - mp > 0 ? m - mp : 0, s1 = j + m < j - mp ? j + m : j - mp;
  for (int s = s0; s <= s1; ++s) tot += (((mp - m + s) & 1) ? -1 : 1) * binom(j + m, s) * binom(j - m, mp - m + s);
  long double f = 1.0L;                           // (j+mp)!(j-mp)! / ((j+m)!(j-m)!)
  for (int i = 2; i <= j + mp; ++i) f *= i;
  for (int i = 2; i <= j - mp; ++i) f *= i;
  for (int i = 2; i <= j + m; ++i) f /= i;
  for (int i = 2; i <= j - m; ++i) f /= i;
  return (double)((long double)tot * std::sqrt(f) * std::ldexp(1.0L, -j));
}

// constants of order P in the order the kernel reads them
inline void build_rot_table(int P, std::vector<double>& out) {
  out.assign((size_t)rot_table_doubles(P), 0.0);
  size_t at = 0;
  for (int n = 0; n < P; ++n)
    for (int m = 0; m <= n; ++m)
      for (int mp = 0; mp <= n; ++mp) {
        if (!rot_live(n, m, mp)) continue;
        if (mp == 0) out[at++] = wigner_d_half(n, m, 0);
        else if (((n + m) & 1) == 0) out[at++] = wigner_d_half(n, m, mp) + wigner_d_half(n, m, -mp);
        else out[at++] = wigner_d_half(n, m, mp) - wigner_d_half(n, m, -mp);
      }
  auto fact = [](int k) { long double f = 1; for (int i = 2; i <= k; ++i) f *= i; return f; };
  for (int k = 0; k < P; ++k)
    for (int j = k; j < P; ++j)
      for (int n = k; n < P; ++n) {
        // (-1)^{k+j} A[n,k] A[j,k] EPS / A[j+n,0] with A[n,m] = (-1)^n / (EPS sqrt((n-m)!(n+m)!)): the EPS cancel
        const int sgn = ((k + j + n + j + j + n) & 1) ? -1 : 1;
        out[at++] = (double)(sgn * fact(j + n) / std::sqrt(fact(n - k) * fact(n + k) * fact(j - k) * fact(j + k)));
      }
}

// ---- the constants as ONE stream in the order the kernel consumes them, signs folded -------------------------------------
//     [conj(X) rotation][X^T rotation][axial translation][conj(X) rotation][X^T rotation]
// rotation entry (n, m, mp), rows m = 0..n, live mp = 0..n: the coefficient above times -1 where the routing phase i^kk has
// kk >= 2, and for X^T times (-1)^{m+mp}: the kernel's FMA is always  acc += c * src.  Axial translation: Tz[j,n,k] for
// k, then j >= k, then n >= k.  The kernel reads the stream sixteen at a time (one 8-byte load per lane, lane & 15) and
// multiplies by lane k of every 16-lane row (v_fmac_f64_dpp row_newbcast:k).
constexpr int kRotGroup = 16;
#ifndef FMMBEM_ROT_AHEAD
#define FMMBEM_ROT_AHEAD 8
#endif
constexpr int kRotAhead = FMMBEM_ROT_AHEAD;          // groups in flight (kernels_m2l_rot.hip: 8)
constexpr int rot_kk(int n, int m, int mp) {        // routing phase of entry (m, mp): 0 +Re, 1 +Im, 2 -Re, 3 -Im
  return ((mp == 0 ? m : m + 3 * mp) + ((mp != 0 && ((n + m) & 1)) ? 1 : 0)) & 3;
}
// position of a constant in the stream: stage 0..4 = rotation, rotation, axial, rotation, rotation.
// ORDER OF CONSUMPTION, rotation segment: degree by degree; within a degree FIRST the output rows m with n + m even -- they read
// the real parts a[n, .] only -- THEN the rows with n + m odd, which read the imaginary parts b[n, .] only: once the first class
// is through, the degree's a[] are dead and their registers take the second class's outputs (peak of a degree 3n + 2 doubles
// instead of 4n + 3: 20 VGPRs at n = 9, where the kernel sat 30-odd registers over the 256 it has and moved the excess through
// AGPRs).  Within a class the rows two at a time, and for a pair of rows the input orders mp ascending with the two rows' entries
// side by side -- (r0, mp), (r1, mp), (r0, mp + 1), ... -- so that consecutive FMAs feed FOUR accumulators (Re and Im of two
// outputs), not two: a lone wavefront issues an FP64 FMA every 3.7 ns into two alternating accumulators and every 2.95 ns into
// four (tools/microbench/dpp_chain.hip).  Every output still adds its own terms in ascending mp: the same bits in any order of rows.
constexpr int rot_class_count(int n, int cls) {      // rows of class cls (0: n + m even, 1: n + m odd) of degree n
  const int first = (n + cls) & 1;                  // smallest m of the class
  return first > n ? 0 : (n - first) / 2 + 1;
}
constexpr int rot_class_row(int n, int cls, int i) { return ((n + cls) & 1) + 2 * i; }
constexpr int rot_index(int n, int m, int mp) {     // within one rotation segment (degrees 1..P-1)
  int c = rot_off(n) - 1;
  for (int cls = 0; cls < 2; ++cls)
    for (int i0 = 0; i0 < rot_class_count(n, cls); i0 += 2)
      for (int q = 0; q <= n; ++q)
        for (int i = i0; i < i0 + 2 && i < rot_class_count(n, cls); ++i) {
          const int r = rot_class_row(n, cls, i);
          if (r == m && q == mp) return c;
          c += rot_live(n, r, q) ? 1 : 0;
        }
  return c;
}
constexpr int rot_plain_index(int n, int m, int mp) {   // in build_rot_table's table: row by row
  int c = rot_off(n);
  for (int i = 0; i <= n; ++i)
    for (int q = 0; q <= n; ++q) {
      if (i == m && q == mp) return c;
      c += rot_live(n, i, q) ? 1 : 0;
    }
  return c;
}
// The same factorisation serves the two shifts of the tree passes (reference: LaplaceSpherical::M2M :245-285, L2L :378-411):
// in the frame where the translation vector is the z axis only the m = 0 harmonic of the shift survives, so
//   M2M   M'[j,k] = sum_{n=k..j}   Tm[j,n,k] rho^(j-n) M[n,k],   Tm[j,n,k] = (-1)^(j-n) a(j-n,0) a(n,k) / a(j,k)
//   L2L   L'[j,k] = sum_{n=j..P-1} Tl[j,n,k] rho^(n-j) L[n,k],   Tl[j,n,k] = a(n-j,0) a(j,k) / a(n,k)
// with a(n,m) = (-1)^n / sqrt((n-m)!(n+m)!) (the reference's Anm without its 1/EPS, which cancels against the EPS the
// operators multiply in), (rho, alpha, beta) = cart2sph(c_parent - c_child) for M2M and of (c_child - c_parent) for L2L, and
// the rotations of M2L on either side (multipole and local coefficients transform alike).  Checked against the oracle's
// operators to 5e-16 (tests/test_rot_operators.py).
enum RotOp { kRotM2L = 0, kRotM2M = 1, kRotL2L = 2 };

// axial block of order k: which (j, n) it holds, row by row (j), and where entry (j, n) sits in it
constexpr int axial_row_begin(int P, int op, int k, int j) { (void)P; return op == kRotL2L ? j : k; }
constexpr int axial_row_end(int P, int op, int k, int j) { (void)k; return op == kRotM2M ? j + 1 : P; }      // one past the last n
constexpr int axial_block_len(int P, int op, int k) {
  int c = 0;
  for (int j = k; j < P; ++j) c += axial_row_end(P, op, k, j) - axial_row_begin(P, op, k, j);
  return c;
}
constexpr int axial_len(int P, int op) {
  int c = 0;
  for (int k = 0; k < P; ++k) c += axial_block_len(P, op, k);
  return c;
}
// ORDER OF CONSUMPTION, axial block of order k: the output rows j two at a time, n ascending with the two rows' entries side by
// side (as in the rotation segments: four accumulators in turn; two for k = 0, where the imaginary parts are zero)
constexpr int axial_index(int P, int op, int k, int j, int n) {
  int c = 0;
  for (int i = 0; i < k; ++i) c += axial_block_len(P, op, i);
  for (int j0 = k; j0 < P; j0 += 2)
    for (int nn = k; nn < P; ++nn)
      for (int r = j0; r <= (j0 + 1 < P ? j0 + 1 : P - 1); ++r) {
        if (nn < axial_row_begin(P, op, k, r) || nn >= axial_row_end(P, op, k, r)) continue;
        if (r == j && nn == n) return c;
        ++c;
      }
  return c;
}
constexpr int tz_index(int P, int k, int j, int n) { return axial_index(P, kRotM2L, k, j, n); }
constexpr int rot_stage_base(int P, int stage, int op = kRotM2L) {
  const int R = rot_off(P) - 1, T = axial_len(P, op);
  return stage == 0 ? 0 : stage == 1 ? R : stage == 2 ? 2 * R : stage == 3 ? 2 * R + T : 3 * R + T;
}
constexpr int rot_stream_len(int P, int op = kRotM2L) { return 4 * (rot_off(P) - 1) + axial_len(P, op); }   // degree 0 is the identity
constexpr int rot_stream_doubles(int P, int op = kRotM2L) { return ((rot_stream_len(P, op) + kRotGroup - 1) / kRotGroup + 1) * kRotGroup; }

inline void build_rot_stream(int P, std::vector<double>& out, int op = kRotM2L) {
  std::vector<double> plain;
  build_rot_table(P, plain);                        // [rotation coefficients, unsigned][Tz of M2L]
  out.assign((size_t)rot_stream_doubles(P, op), 0.0);
  size_t at = 0;                                    // base of the segment being written; entries go where the index functions say
  auto rotation = [&](bool back) {
    for (int n = 1; n < P; ++n)
      for (int m = 0; m <= n; ++m)
        for (int mp = 0; mp <= n; ++mp) {
          if (!rot_live(n, m, mp)) continue;
          const bool neg = (rot_kk(n, m, mp) >= 2) != (back && ((m + mp) & 1));
          const double c = plain[(size_t)rot_plain_index(n, m, mp)];
          out[at + (size_t)rot_index(n, m, mp)] = neg ? -c : c;
        }
    at += (size_t)(rot_off(P) - 1);
  };
  auto fact = [](int k) { long double f = 1; for (int i = 2; i <= k; ++i) f *= i; return f; };
  auto a = [&](int n, int m) { return ((n & 1) ? -1.0L : 1.0L) / std::sqrt(fact(n - m) * fact(n + m)); };
  auto axial = [&]() {
    for (int k = 0; k < P; ++k)
      for (int j = k; j < P; ++j)
        for (int n = axial_row_begin(P, op, k, j); n < axial_row_end(P, op, k, j); ++n)
          out[at + (size_t)axial_index(P, op, k, j, n)] =
              op == kRotM2L ? plain[(size_t)tz_off(P, k) + (size_t)(j - k) * (P - k) + (n - k)]      // build_rot_table: [j-k][n-k]
              : op == kRotM2M ? (double)((((j - n) & 1) ? -1.0L : 1.0L) * a(j - n, 0) * a(n, k) / a(j, k))
                              : (double)(a(n - j, 0) * a(j, k) / a(n, k));
    at += (size_t)axial_len(P, op);
  };
  rotation(false); rotation(true); axial(); rotation(false); rotation(true);
}


}  // namespace fmmbem
