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
// ORDER OF CONSUMPTION, rotation segment: degree by degree; within a degree the output rows m two at a time, and for a pair of
// rows the input orders mp ascending with the two rows' entries side by side -- (m0, mp), (m0 + 1, mp), (m0, mp + 1), ... --
// so that consecutive FMAs feed FOUR accumulators (Re and Im of two outputs), not two: a lone wavefront issues an FP64 FMA
// every 3.7 ns into two alternating accumulators and every 2.95 ns into four (tools/microbench/dpp_chain.hip).  Every output
// still adds its own terms in ascending mp: the same bits as the row-by-row order.
constexpr int rot_index(int n, int m, int mp) {     // within one rotation segment (degrees 1..P-1)
  int c = rot_off(n) - 1;
  for (int m0 = 0; m0 <= n; m0 += 2)
    for (int q = 0; q <= n; ++q)
      for (int r = m0; r <= (m0 + 1 <= n ? m0 + 1 : n); ++r) {
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


// ====================================================================================================================
// Split form (csrc/kernels_m2l_rot2.hip; M2L only): one (target, source) pair on TWO lanes of a wavefront, the even degrees
// on one ("E"), the odd degrees on the other ("O"), so that a lane holds half the coefficients and two wavefronts fit a SIMD
// at p = 9 ... 12.  Degrees are taken in pairs q = (2q, 2q+1); a lane's data are SLOTS (q, t), t = 0 .. 2q+1:
//     O lane:  slot (q, t) = coefficient (n = 2q+1, m = t)
//     E lane:  slot (q, t) = coefficient (n = 2q,   m = t-1)        (t = 0: padding, zero)
// With that shift by one in the order, the real rotation block of the EVEN degree 2q is a sub-pattern of the block of the ODD
// degree 2q+1: for mp >= 2 the routing phase rot_kk(2q, m-1, mp-1) = rot_kk(2q+1, m, mp), the a / b choice of the source
// depends on n + m only, and the column mp_E = 0 is live exactly where the skeleton reads a.  So BOTH lanes run the program
// of the odd degrees only -- the same instruction reads the same slots -- and differ in the constants, which come from two
// streams through the DPP row broadcast (rows 0, 2 of a wavefront are E lanes, rows 1, 3 O lanes; the partner is lane ^ 16).
// The z rotations are local (slot t turns by e^{i t g} on O lanes, e^{i (t-1) g} on E lanes); only the axial translation
// couples the degrees: every lane sums over ITS degrees for the outputs of both parities and the partners swap the halves
// that belong to the other -- one exchange per pass.
// ====================================================================================================================
constexpr int rot2_pairs(int P) { return (P + 1) / 2; }                  // degree pairs; an odd P leaves the last O degree empty
constexpr int rot2_sidx(int q, int t) { return q * (q + 1) + t; }        // slot index: sum_{q' < q} (2 q' + 2) + t
constexpr int rot2_nslots(int P) { return rot2_pairs(P) * (rot2_pairs(P) + 1); }
constexpr int rot2_qmin(int t) { return t / 2; }                          // first degree pair that has slot order t
// one rotation segment of the skeleton: the live entries of the odd degrees 1, 3, ..., in the order of fixed_rotation
constexpr int rot2_rot_len(int P) {
  int c = 0;
  for (int q = 0; q < rot2_pairs(P); ++q) c += rot_nnz(2 * q + 1);
  return c;
}
constexpr int rot2_rot_index(int q, int m, int mp) {
  int c = 0;
  for (int i = 0; i < q; ++i) c += rot_nnz(2 * i + 1);
  const int n = 2 * q + 1;
  for (int i = 0; i <= n; ++i)
    for (int j = 0; j <= n; ++j) {
      if (i == m && j == mp) return c;
      c += rot_live(n, i, j) ? 1 : 0;
    }
  return c;
}
// axial segment.  At slot order t the O lanes translate order k = t, the E lanes order k = t - 1, each from ITS degrees (input
// pairs qi >= qmin(t)).  OWN rows: the output degrees of the lane's parity, pairs qo >= qmin(t).  OTHER rows: the output degrees of
// the partner's parity -- on an E lane the odd degrees 2 qo + 1 >= t - 1, i.e. qo >= qmin_other(t) = (t - 1) / 2, one pair more
// than the own rows when t is even (the O lanes' constants of that row are zero).  Per t: for qo from qmin_other(t): the own
// row (if qo >= qmin(t)), then the other row; a row = the inputs qi = qmin(t) .. Q - 1.
constexpr int rot2_qmin_other(int t) { return t ? (t - 1) / 2 : 0; }
constexpr int rot2_axial_rows(int P, int t) { return (rot2_pairs(P) - rot2_qmin(t)) + (rot2_pairs(P) - rot2_qmin_other(t)); }
constexpr int rot2_axial_len(int P) {
  int c = 0;
  for (int t = 0; t < 2 * rot2_pairs(P); ++t) c += rot2_axial_rows(P, t) * (rot2_pairs(P) - rot2_qmin(t));
  return c;
}
constexpr int rot2_axial_index(int P, int t, int qo, int other, int qi) {
  int c = 0;
  for (int i = 0; i < t; ++i) c += rot2_axial_rows(P, i) * (rot2_pairs(P) - rot2_qmin(i));
  const int w = rot2_pairs(P) - rot2_qmin(t);
  int row = 0;
  for (int q = rot2_qmin_other(t); q < qo; ++q) row += (q >= rot2_qmin(t) ? 1 : 0) + 1;
  if (other && qo >= rot2_qmin(t)) row += 1;
  return c + row * w + (qi - rot2_qmin(t));
}
constexpr int rot2_stage_base(int P, int stage) {
  const int R = rot2_rot_len(P), T = rot2_axial_len(P);
  return stage == 0 ? 0 : stage == 1 ? R : stage == 2 ? 2 * R : stage == 3 ? 2 * R + T : 3 * R + T;
}
constexpr int rot2_stream_len(int P) { return 4 * rot2_rot_len(P) + rot2_axial_len(P); }
// layout: groups of sixteen positions; group g holds 16 E constants then 16 O constants: [g][parity][k]
constexpr int rot2_stream_doubles(int P) { return ((rot2_stream_len(P) + kRotGroup - 1) / kRotGroup + 1) * 2 * kRotGroup; }

// op: the axial operator in the middle -- M2L, or one of the two shifts of the tree passes (same skeleton, the entries outside
// the shift's triangle are zeros: the split kernel runs one instruction stream for all three)
inline void build_rot2_stream(int P, std::vector<double>& out, int op = kRotM2L) {
  std::vector<double> plain;
  build_rot_table(P, plain);                        // [rotation coefficients of degrees 0 .. P-1, unsigned][Tz of M2L]
  out.assign((size_t)rot2_stream_doubles(P), 0.0);
  size_t at = 0;
  auto put = [&](double e, double o) {
    const size_t g = at / kRotGroup, k = at % kRotGroup;
    out[g * 2 * kRotGroup + k] = e;
    out[g * 2 * kRotGroup + kRotGroup + k] = o;
    ++at;
  };
  // coefficient of the real rotation block of degree n at (m, mp), with the sign the kernel's "acc += c * src" wants
  auto coef = [&](int n, int m, int mp, bool back) -> double {
    if (n < 0 || n >= P || m < 0 || mp < 0 || m > n || mp > n || !rot_live(n, m, mp)) return 0.0;
    if (n == 0) return 1.0;
    size_t ci = (size_t)rot_off(n);
    for (int i = 0; i <= n; ++i)
      for (int j = 0; j <= n; ++j) {
        if (i == m && j == mp) {
          const bool neg = (rot_kk(n, m, mp) >= 2) != (back && ((m + mp) & 1));
          return neg ? -plain[ci] : plain[ci];
        }
        ci += rot_live(n, i, j) ? 1 : 0;
      }
    return 0.0;
  };
  auto rotation = [&](bool back) {
    for (int q = 0; q < rot2_pairs(P); ++q) {
      const int n = 2 * q + 1;                       // the skeleton: the odd degree of the pair (present or not)
      for (int m = 0; m <= n; ++m)
        for (int mp = 0; mp <= n; ++mp) {
          if (!rot_live(n, m, mp)) continue;
          put(coef(2 * q, m - 1, mp - 1, back), coef(n, m, mp, back));
        }
    }
  };
  // Tz[j, n, k] of M2L (build_rot_table's second part: k, then j >= k, then n >= k), zero outside its range; for the shifts
  // Tm / Tl of build_rot_stream, zero outside the triangle
  auto fact = [](int k) { long double f = 1; for (int i = 2; i <= k; ++i) f *= i; return f; };
  auto an = [&](int n, int m) { return ((n & 1) ? -1.0L : 1.0L) / std::sqrt(fact(n - m) * fact(n + m)); };
  auto tz = [&](int j, int n, int k) -> double {
    if (k < 0 || j < k || n < k || j >= P || n >= P) return 0.0;
    if (op == kRotM2L) return plain[(size_t)tz_off(P, k) + (size_t)(j - k) * (P - k) + (n - k)];
    if (n < axial_row_begin(P, op, k, j) || n >= axial_row_end(P, op, k, j)) return 0.0;
    return op == kRotM2M ? (double)((((j - n) & 1) ? -1.0L : 1.0L) * an(j - n, 0) * an(n, k) / an(j, k))
                         : (double)(an(n - j, 0) * an(j, k) / an(n, k));
  };
  auto axial = [&]() {
    const int Q = rot2_pairs(P);
    for (int t = 0; t < 2 * Q; ++t)
      for (int qo = rot2_qmin_other(t); qo < Q; ++qo)
        for (int other = qo >= rot2_qmin(t) ? 0 : 1; other < 2; ++other)
          for (int qi = rot2_qmin(t); qi < Q; ++qi) {
            // O lane: inputs n = 2 qi + 1 at order k = t; E lane: inputs n = 2 qi at order k = t - 1.
            // own: the output degree of the lane's parity; other: of the partner's parity
            const double o = other ? tz(2 * qo, 2 * qi + 1, t) : tz(2 * qo + 1, 2 * qi + 1, t);
            const double e = other ? tz(2 * qo + 1, 2 * qi, t - 1) : tz(2 * qo, 2 * qi, t - 1);
            put(e, o);
          }
  };
  rotation(false); rotation(true); axial(); rotation(false); rotation(true);
}

}  // namespace fmmbem
