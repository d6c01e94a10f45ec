// kernels_far.hip -- gfx950 kernels of the far field except M2L (kernels_m2l.hip); the expansion order p
// is a launch argument (the solver's per-iteration relaxation of p).
//
//   p2m        LaplaceSphericalBEM::P2M            kernel/LaplaceSphericalBEM.hpp:307-352
//   m2m_level  LaplaceSpherical::M2M               kernel/LaplaceSpherical.hpp:245-285
//   mh_prep    source-side rescaling of M for M2L  (folds Anm and the i^{|k-m|-|k|-|m|} phase of Cnm,
//                                                   kernel/LaplaceSpherical.hpp:106-116)
//   (m2l       lives in kernels_m2l.hip)
//   l2l_level  LaplaceSpherical::L2L               kernel/LaplaceSpherical.hpp:378-411
//   l2p        LaplaceSphericalBEM::L2P            kernel/LaplaceSphericalBEM.hpp:448-476
//
// Index conventions are the reference's: full harmonic index nm = n^2+n+m (m in [-n,n]); stored
// coefficient index nms = n(n+1)/2+m (m >= 0, negative orders by conjugation).  M and L hold exactly
// the reference's coefficients (EPS-scaled Anm and all), so they can be compared box by box.
//
// M2L algebra.  With A = Anm, the reference sums
//     L[j,k] += sum_{n,m} Mt[n,m] * i^{|k-m|-|k|-|m|} (-1)^j A[n,m] A[j,k] / A[j+n,m-k] * EPS * Y[j+n,m-k]
// (Mt[n,-m] = conj(M[n,m])).  Because i^{|k-m|-|k|-|m|} = i^{-|k|} * i^{-|m|} * i^{|m-k|}, this is
//     L[j,k] += i^{-k} (-1)^j A[j,k] * sum_{n,m} Mh[n,m] * Yh[j+n, m-k]
// with Mh[n,m] = i^{-|m|} A[n,m] Mt[n,m]  (per source box, computed once per matvec by mh_prep) and
//      Yh[r,c] = i^{|c|} EPS Y[r,c] / A[r,c] (per translation vector, tabulated once per plan).
// The inner loop is then a plain complex correlation: one wavefront per target box, lanes over the
// (j,k) outputs, Yh staged in LDS, Mh read through the scalar cache (wave-uniform), L accumulated
// in registers over the whole source list and written once -- no atomics, fixed summation order.
#include "device_launch.hpp"
#define FMMBEM_INLINE __attribute__((always_inline))

#include <mutex>
#include <type_traits>

namespace fmmbem {

namespace {

constexpr int kWave = 64;
constexpr int kPmaxDev = 16, kSmax = kPmaxDev * (kPmaxDev + 1) / 2;
constexpr double kEps = 1e-12;                         // kernel/LaplaceSpherical.hpp:30

// stored index -> (j,k), up to p = 16
struct JK { unsigned char j[136], k[136]; };
__constant__ JK kJK;
__constant__ double kRecip[40];                        // 1/k, k = 1..39 (Legendre recurrence denominators)
JK make_jk() {
  JK t;
  int i = 0;
  for (int j = 0; j < 16; ++j)
    for (int k = 0; k <= j; ++k, ++i) { t.j[i] = (unsigned char)j; t.k[i] = (unsigned char)k; }
  return t;
}

__device__ inline double2 cmul(double2 a, double2 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ inline double2 mul_i_pow(double2 a, int q) {              // a * i^q
  switch (q & 3) {
    case 0: return a;
    case 1: return {-a.y, a.x};
    case 2: return {-a.x, -a.y};
    default: return {a.y, -a.x};
  }
}

// Spherical coordinates of d as the reference's cart2sph produces them
// (kernel/LaplaceSpherical.hpp:528-541): rho = |d| + EPS, alpha = acos(z/rho); the azimuth is kept as
// (cos beta, sin beta) -- atan(y/x) (+pi for x<0) has exactly that cosine/sine -- with the two
// degenerate branches (|x|+|y| < EPS -> beta = 0; |x| < EPS -> beta = +-pi/2).
struct Sph { double rho, ca, sa, cb, sb; };
__device__ inline Sph cart2sph(double dx, double dy, double dz) {
  Sph s;
  s.rho = sqrt(dx * dx + dy * dy + dz * dz) + kEps;
  s.ca = dz / s.rho;
  s.sa = sqrt((1.0 - s.ca) * (1.0 + s.ca));
  if (fabs(dx) + fabs(dy) < kEps) { s.cb = 1; s.sb = 0; }
  else if (fabs(dx) < kEps) { s.cb = 0; s.sb = dy > 0 ? 1.0 : -1.0; }
  else { const double h = 1.0 / sqrt(dx * dx + dy * dy); s.cb = dx * h; s.sb = dy * h; }
  return s;
}

// Per-step constants of the harmonic recurrences, in the m-major order the loops below visit (n,m):
// pref = sqrt((n-m)!/(n+m)!), and the Legendre step P_{n+1}^m = c1 x P_n^m - c2 P_{n-1}^m with
// c1 = (2n+1)/(n-m+1), c2 = (n+m)/(n-m+1)  (c1 = 2m+1, c2 = 0 on the diagonal n = m).
// Kept in LDS: fetched as wave-uniform scalar loads from global memory they cost one SMEM round trip
// per step (measured: 51 % of the P2M wave cycles were s_waitcnt).
__device__ inline void fill_step_tables(const DevicePlan& d, int P, int lane, double* sPref, double* sC1, double* sC2) {
  int t = 0;
  for (int m = 0; m < P; ++m)
    for (int n = m; n < P; ++n, ++t)
      if ((t & (kWave - 1)) == lane) {
        sPref[t] = d.tabPref[n * n + n + m];
        sC1[t] = n == m ? (double)(2 * m + 1) : (double)(2 * n + 1) * kRecip[n - m + 1];
        sC2[t] = n == m ? 0.0 : (double)(n + m) * kRecip[n - m + 1];
      }
}

// ---------------------------------------------------------------------------------------------
// P2M: one wavefront per source leaf (workgroups stride over the leaves).  Lane = one quadrature point of
// one panel; every lane runs the Legendre / rho^n / e^{-im beta} recurrences of evalMultipole(rho,alpha,-beta)
// (kernel/LaplaceSpherical.hpp:455-488) for its point and drops its weighted harmonic for coefficient row r
// into an LDS tile [row][lane]; after kP2MBand rows the tile is reduced along the lanes (two lanes per row,
// conflict-free 16-B reads) and added to the leaf's accumulator.  No per-coefficient wave shuffles.
// ---------------------------------------------------------------------------------------------
constexpr int kP2MBand = 8;                           // rows per LDS tile: 8 lanes reduce one row
constexpr int kP2MCols = 32;                          // tile columns: lanes l and l+32 are added in registers first

// value of lane l+32 (for l < 32) without touching the LDS: v_permlane32_swap exchanges the upper half of one
// operand with the lower half of the other (gfx950)
__device__ __forceinline__ double from_upper_half(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[1], (int)a[1]);
}

constexpr int kP2MWaves = 4;                          // independent leaves per workgroup
struct D4 { double x, y, z, w; };
typedef __attribute__((address_space(4))) D4 ConstD4;  // wave-uniform constants through the scalar cache

template <int slot>
__global__ __launch_bounds__(kP2MWaves * kWave) void p2m_kernel(DevicePlan d, const int P, const int wmode, const int out_slot) {
  __shared__ double2 tile_all[kP2MWaves][kP2MBand][kP2MCols + 1];
  __shared__ double2 acc_all[kP2MWaves][kSmax];
  __shared__ int rowidx_all[kP2MWaves][kP2MBand];
  const int S = P * (P + 1) / 2;
  // per-step recurrence constants: wave-uniform, fetched ONE STEP AHEAD with scalar loads (fetched at the point of
  // use they cost an SMEM round trip per step; kept in LDS they cost three LDS reads per step of a kernel that
  // SQ counters show at 83 % LDS issue)
  const ConstD4* steptab = reinterpret_cast<const ConstD4*>(reinterpret_cast<uintptr_t>(d.tabStep + (size_t)(P - 1) * (kSmax + 1) * 4));
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  double2 (*tile)[kP2MCols + 1] = tile_all[wave];
  double2* acc = acc_all[wave];
  int* rowidx = rowidx_all[wave];
  const int64_t N = d.n;
  const int nq = d.nq;
  for (int li = blockIdx.x * kP2MWaves + wave; li < d.n_p2m; li += gridDim.x * kP2MWaves) {
    const int leaf = d.p2m_leaf[li];
    const int box = d.leaf_box[leaf];
    const int row0 = d.leaf_row0[leaf], npts = d.leaf_nrows[leaf] * nq;
    const double c0 = d.box_center[3 * box], c1 = d.box_center[3 * box + 1], c2 = d.box_center[3 * box + 2];
    for (int i = lane; i < S; i += kWave) acc[i] = {0, 0};
    int inband = 0;
    auto flush = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      constexpr int kPer = kWave / kP2MBand, kCols = kP2MCols / kPer;   // lanes per row, columns per lane
      const int r = lane & (kP2MBand - 1), part = lane / kP2MBand;
      double sr = 0, si = 0;
      if (r < inband) {
        const double2* rowp = &tile[r][part * kCols];
#pragma unroll
        for (int k = 0; k < kCols; ++k) { sr += rowp[k].x; si += rowp[k].y; }
      }
#pragma unroll
      for (int off = kP2MBand; off < kWave; off <<= 1) {
        sr += __shfl_xor(sr, off, kWave);
        si += __shfl_xor(si, off, kWave);
      }
      if (part == 0 && r < inband) { double2& a = acc[rowidx[r]]; a.x += sr; a.y += si; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      inband = 0;
    };
    for (int chunk = 0; chunk < npts; chunk += kWave) {
      const int pt = chunk + lane;
      const int panel = pt / nq, q = pt - panel * nq;
      const int64_t i = row0 + panel;
      // wmode 0: Laplace, charge * w * Area for panels whose BC feeds this slot (LaplaceSphericalBEM.hpp:323-344)
      // wmode 1..3: Stokes f_e * w * Area; wmode 4: (f . x_q) * w * Area with the ABSOLUTE quadrature point
      // (kernel/StokesSphericalBEM.hpp:417-431)
      // wmode 5..11 (slot = 1, the gradient branch): the seven dipole potentials of the Stokes double layer without stored
      // gradient records -- the moment is  wq (u . grad)(rho^n Ynm)  with
      //   5..7   Psi_k:    u = g (the panel's density),  wq = n_k w Area      (expansion slots 4..6, p2m_apply_kernel<3>)
      //   8      Psi_0:    u = g,                         wq = (c . n) w Area  (slot 7; y.n is constant on a flat panel)
      //   9..11  Theta_i:  u = n,                         wq = g_i w Area      (slots 8..10)
      const bool live = pt < npts && (wmode ? true : d.bc[i] == slot);
      double wq = 0.0;
      double n0 = live ? d.nx[i] : 0, n1 = live ? d.ny[i] : 0, n2 = live ? d.nz[i] : 0;      // the direction u of the gradient branch
      if (live) {
        const double aw = d.area[i] * d.qw[q];
        if (wmode == 0) wq = d.xt[i] * aw;
        else if (wmode < 4) wq = d.xt[3 * i + (wmode - 1)] * aw;
        else if (wmode == 4) wq = (d.xt[3 * i] * d.quad[(q * 3 + 0) * N + i] + d.xt[3 * i + 1] * d.quad[(q * 3 + 1) * N + i] +
                                   d.xt[3 * i + 2] * d.quad[(q * 3 + 2) * N + i]) * aw;
        else if (wmode >= 9) wq = d.xt[3 * i + (wmode - 9)] * aw;
        else {
          wq = (wmode == 8 ? d.cx[i] * n0 + d.cy[i] * n1 + d.cz[i] * n2 : wmode == 5 ? n0 : wmode == 6 ? n1 : n2) * aw;
          n0 = d.xt[3 * i]; n1 = d.xt[3 * i + 1]; n2 = d.xt[3 * i + 2];
        }
      }
      const Sph s = cart2sph(live ? d.quad[(q * 3 + 0) * N + i] - c0 : 0.3, live ? d.quad[(q * 3 + 1) * N + i] - c1 : 0.4,
                             live ? d.quad[(q * 3 + 2) * N + i] - c2 : 0.5);
      double pn = 1, rhom = 1, er = 1, ei = 0, fact = 1;
      int step = 0;
      D4 cnext = {steptab[0].x, steptab[0].y, steptab[0].z, 0};
#pragma nounroll
      for (int m = 0; m < P; ++m) {
        double p = pn, p1 = p, rhon = rhom;
#pragma nounroll
        for (int n = m; n < P; ++n, ++step) {
          const D4 cst = cnext;
          cnext = {steptab[step + 1].x, steptab[step + 1].y, steptab[step + 1].z, 0};     // row S is zeros
          const double pref = cst.x;
          // Ynm[n,m] at (rho, alpha, -beta): rho^n P_n^m(cos a) pref e^{-i m beta}
          const double mag = rhon * p * pref;
          const double yr = mag * er, yi = -mag * ei;
          const double pcur = p;
          const double pnext = cst.y * s.ca * pcur - cst.z * p1;   // Legendre recurrence, P_{n+1}^m
          double vr, vi;
          if (slot == 0) {                            // source BC POTENTIAL: G moments (LaplaceSphericalBEM.hpp:326)
            vr = wq * yr; vi = wq * yi;
          } else {                                    // NORMAL_DERIV: (n . grad)(rho^n Ynm) (:331-343)
            double tmag;                              // YnmTheta magnitude
            if (n == m) tmag = rhon * (pnext - (m + 1) * s.ca * pcur) / s.sa * pref;
            else tmag = rhon * ((n - m + 1) * pnext - (n + 1) * s.ca * pcur) / s.sa * pref;
            const double tr = tmag * er, ti = -tmag * ei;
            const double rho = s.rho, sa = s.sa, ca = s.ca, cb = s.cb, sb = s.sb;
            const double brr = (double)n / rho * yr, bri = (double)n / rho * yi;       // d/d rho
            const double ber = (double)m * yi, bei = -(double)m * yr;                  // -i m Ynm
            const double gxr = sa * cb * brr + ca * cb / rho * tr - sb / rho / sa * ber;
            const double gxi = sa * cb * bri + ca * cb / rho * ti - sb / rho / sa * bei;
            const double gyr = sa * sb * brr + ca * sb / rho * tr + cb / rho / sa * ber;
            const double gyi = sa * sb * bri + ca * sb / rho * ti + cb / rho / sa * bei;
            const double gzr = ca * brr - sa / rho * tr;
            const double gzi = ca * bri - sa / rho * ti;
            vr = wq * (n0 * gxr + n1 * gyr + n2 * gzr);
            vi = wq * (n0 * gxi + n1 * gyi + n2 * gzi);
          }
          // the P2M kernel is LDS-issue bound (SQ counters: LDS 83 %, VALU 36 %): halve the tile traffic in registers
          vr += from_upper_half(vr);
          vi += from_upper_half(vi);
          if (lane < kP2MCols) tile[inband][lane] = {vr, vi};
          if (lane == 0) rowidx[inband] = n * (n + 1) / 2 + m;
          if (++inband == kP2MBand) flush();
          p1 = pcur; p = pnext;
          rhon *= s.rho;
        }
        pn = -pn * fact * s.sa;
        rhom *= s.rho;
        const double nr = er * s.cb - ei * s.sb, ni = er * s.sb + ei * s.cb;
        er = nr; ei = ni;
        fact += 2;
      }
      if (inband) flush();
    }
    double2* M = d.M + ((size_t)box * d.nslots + out_slot) * d.s_max;
    for (int i = lane; i < S; i += kWave) M[i] = acc[i];
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// P2M as a stored operator (DevicePlan::p2m_tab).  p2m_table (once per plan): thread = panel of a source leaf; for each of
// its quadrature points the recurrences of p2m_kernel above, the weighted harmonic added to the panel's own record.
// p2m_apply (every matvec): wavefront = leaf, lane = coefficient, a loop over the leaf's panels -- coalesced reads of
// 16 S(p) bytes per panel, two FMAs per coefficient; no recurrences, no LDS, no cross-lane reduction.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync();     // below

// Packed Laplace records (DevicePlan::p2m_packed): where coefficient (n, m) of a record lives
__device__ __forceinline__ void coef_nm(int idx, int& n, int& m) {      // idx = n (n + 1) / 2 + m
  n = 0;
  while ((n + 1) * (n + 2) / 2 <= idx) ++n;
  m = idx - n * (n + 1) / 2;
}
__device__ __forceinline__ int packed_cpos(int n, int m) { return n * (n - 1) / 2 + m - 1; }      // m >= 1

template <int NT>
__global__ __launch_bounds__(kWave) void p2m_table_kernel(DevicePlan d, double2* __restrict__ tab) {
  const int P = d.p_max, SM = d.p2m_stride;          // record stride of the table
  const ConstD4* steptab = reinterpret_cast<const ConstD4*>(reinterpret_cast<uintptr_t>(d.tabStep + (size_t)(P - 1) * (kSmax + 1) * 4));
  const int64_t N = d.n;
  const int nq = d.nq;
  for (int li = blockIdx.x; li < d.n_p2m; li += gridDim.x) {
    const int leaf = d.p2m_leaf[li], box = d.leaf_box[leaf];
    const int row0 = d.leaf_row0[leaf], nrows = d.leaf_nrows[leaf];
    const double c0 = d.box_center[3 * box], c1 = d.box_center[3 * box + 1], c2 = d.box_center[3 * box + 2];
    for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
      const int64_t i = row0 + r;
      double2* out = tab + (size_t)(i - d.p2m_tab_row0) * NT * SM;
      const bool deriv = (NT == 1 && d.bc[i] != 0) || NT == 3;   // Laplace NORMAL_DERIV panel: (n . grad)(rho^n Ynm) moments (LaplaceSphericalBEM.hpp:331-343);
                                                                  // NT == 3: the three components of that gradient, one record each (Stokes double layer, below)
      const double n0 = d.nx[i], n1 = d.ny[i], n2 = d.nz[i];
      for (int q = 0; q < nq; ++q) {
        const double qx = d.quad[(q * 3 + 0) * N + i], qy = d.quad[(q * 3 + 1) * N + i], qz = d.quad[(q * 3 + 2) * N + i];
        const double aw = d.area[i] * d.qw[q];
        const Sph s = cart2sph(qx - c0, qy - c1, qz - c2);
        double pn = 1, rhom = 1, er = 1, ei = 0, fact = 1;
        int step = 0;
#pragma nounroll
        for (int m = 0; m < P; ++m) {
          double p = pn, p1 = p, rhon = rhom;
#pragma nounroll
          for (int n = m; n < P; ++n, ++step) {
            const double pref = steptab[step].x;
            const double mag = rhon * p * pref;
            const double yr = mag * er, yi = -mag * ei;                  // Ynm at (rho, alpha, -beta)
            const double pcur = p;
            const double pnext = steptab[step].y * s.ca * pcur - steptab[step].z * p1;
            double vr = yr, vi = yi;
            if (deriv) {
              double tmag;
              if (n == m) tmag = rhon * (pnext - (m + 1) * s.ca * pcur) / s.sa * pref;
              else tmag = rhon * ((n - m + 1) * pnext - (n + 1) * s.ca * pcur) / s.sa * pref;
              const double tr = tmag * er, ti = -tmag * ei;
              const double rho = s.rho, sa = s.sa, ca = s.ca, cb = s.cb, sb = s.sb;
              const double brr = (double)n / rho * yr, bri = (double)n / rho * yi;
              const double ber = (double)m * yi, bei = -(double)m * yr;
              const double gxr = sa * cb * brr + ca * cb / rho * tr - sb / rho / sa * ber;
              const double gxi = sa * cb * bri + ca * cb / rho * ti - sb / rho / sa * bei;
              const double gyr = sa * sb * brr + ca * sb / rho * tr + cb / rho / sa * ber;
              const double gyi = sa * sb * bri + ca * sb / rho * ti + cb / rho / sa * bei;
              const double gzr = ca * brr - sa / rho * tr;
              const double gzi = ca * bri - sa / rho * ti;
              vr = n0 * gxr + n1 * gyr + n2 * gzr;
              vi = n0 * gxi + n1 * gyi + n2 * gzi;
              if constexpr (NT == 3) {
                const int gidx = n * (n + 1) / 2 + m;
                const double gr[3] = {gxr, gyr, gzr}, gi[3] = {gxi, gyi, gzi};
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                  double2 acc = q ? out[(size_t)e * SM + gidx] : double2{0, 0};
                  acc.x += aw * gr[e]; acc.y += aw * gi[e];
                  out[(size_t)e * SM + gidx] = acc;
                }
              }
            }
            const int idx = n * (n + 1) / 2 + m;
            const double wt[4] = {aw, aw * qx, aw * qy, aw * qz};      // Stokes: moments of 1 and of the ABSOLUTE point (StokesSphericalBEM.hpp:417-431)
            if constexpr (NT == 1) {
              if (d.p2m_packed) {                      // packed record: complex (n, m >= 1), then the reals (n, 0) -- vi is identically zero there
                if (m) {
                  double2* at = out + packed_cpos(n, m);
                  double2 acc = q ? *at : double2{0, 0};
                  acc.x += aw * vr; acc.y += aw * vi;
                  *at = acc;
                } else {
                  double* at = reinterpret_cast<double*>(out + d.p2m_real_off) + n;
                  *at = (q ? *at : 0.0) + aw * vr;
                }
              } else {
                double2 acc = q ? out[idx] : double2{0, 0};
                acc.x += aw * vr; acc.y += aw * vi;
                out[idx] = acc;
              }
            } else if constexpr (NT != 3) {
#pragma unroll
              for (int e = 0; e < NT; ++e) {
                double2 acc = q ? out[(size_t)e * SM + idx] : double2{0, 0};
                acc.x += wt[e] * vr; acc.y += wt[e] * vi;
                out[(size_t)e * SM + idx] = acc;
              }
            }
            p1 = pcur; p = pnext;
            rhon *= s.rho;
          }
          pn = -pn * fact * s.sa;
          rhom *= s.rho;
          const double nr = er * s.cb - ei * s.sb, ni = er * s.sb + ei * s.cb;
          er = nr; ei = ni;
          fact += 2;
        }
      }
    }
  }
}

// The same table with the quadrature points INNERMOST (rules of up to four points): the recurrences of a panel's K points run side
// by side, a coefficient's K contributions are added in registers in the order q = 0 .. K - 1 -- the order the kernel above adds them
// in memory: the same record up to the compiler's choice of fused multiply-adds (measured: identical, or different by <= 4e-16
// relative) -- and every record entry is written once instead of read and rewritten K times (Stokes config 4: the table 11.3 -> 2 ms,
// the plan 0.061 -> 0.046 s; the stride between two threads' records is 3 KB, every one of those accesses a line of its own).
template <int NT, int NQ>
__global__ __launch_bounds__(kWave) void p2m_table_points_kernel(DevicePlan d, double2* __restrict__ tab) {
  const int P = d.p_max, SM = d.p2m_stride;
  const ConstD4* steptab = reinterpret_cast<const ConstD4*>(reinterpret_cast<uintptr_t>(d.tabStep + (size_t)(P - 1) * (kSmax + 1) * 4));
  const int64_t N = d.n;
  for (int li = blockIdx.x; li < d.n_p2m; li += gridDim.x) {
    const int leaf = d.p2m_leaf[li], box = d.leaf_box[leaf];
    const int row0 = d.leaf_row0[leaf], nrows = d.leaf_nrows[leaf];
    const double c0 = d.box_center[3 * box], c1 = d.box_center[3 * box + 1], c2 = d.box_center[3 * box + 2];
    for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
      const int64_t i = row0 + r;
      double2* out = tab + (size_t)(i - d.p2m_tab_row0) * NT * SM;
      const bool deriv = (NT == 1 && d.bc[i] != 0) || NT == 3;
      const double n0 = d.nx[i], n1 = d.ny[i], n2 = d.nz[i];
      Sph s[NQ];
      double qx[NQ], qy[NQ], qz[NQ], aw[NQ], pn[NQ], rhom[NQ], er[NQ], ei[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        qx[q] = d.quad[(q * 3 + 0) * N + i]; qy[q] = d.quad[(q * 3 + 1) * N + i]; qz[q] = d.quad[(q * 3 + 2) * N + i];
        aw[q] = d.area[i] * d.qw[q];
        s[q] = cart2sph(qx[q] - c0, qy[q] - c1, qz[q] - c2);
        pn[q] = 1; rhom[q] = 1; er[q] = 1; ei[q] = 0;
      }
      double fact = 1;
      int step = 0;
#pragma nounroll
      for (int m = 0; m < P; ++m) {
        double p[NQ], p1[NQ], rhon[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { p[q] = pn[q]; p1[q] = p[q]; rhon[q] = rhom[q]; }
#pragma nounroll
        for (int n = m; n < P; ++n, ++step) {
          const double pref = steptab[step].x, cy = steptab[step].y, cz = steptab[step].z;
          double2 acc[NT == 3 ? 3 : NT];
#pragma unroll
          for (int e = 0; e < (NT == 3 ? 3 : NT); ++e) acc[e] = double2{0, 0};
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const double mag = rhon[q] * p[q] * pref;
            const double yr = mag * er[q], yi = -mag * ei[q];
            const double pcur = p[q];
            const double pnext = cy * s[q].ca * pcur - cz * p1[q];
            double vr = yr, vi = yi;
            if (deriv) {
              double tmag;
              if (n == m) tmag = rhon[q] * (pnext - (m + 1) * s[q].ca * pcur) / s[q].sa * pref;
              else tmag = rhon[q] * ((n - m + 1) * pnext - (n + 1) * s[q].ca * pcur) / s[q].sa * pref;
              const double tr = tmag * er[q], ti = -tmag * ei[q];
              const double rho = s[q].rho, sa = s[q].sa, ca = s[q].ca, cb = s[q].cb, sb = s[q].sb;
              const double brr = (double)n / rho * yr, bri = (double)n / rho * yi;
              const double ber = (double)m * yi, bei = -(double)m * yr;
              const double gxr = sa * cb * brr + ca * cb / rho * tr - sb / rho / sa * ber;
              const double gxi = sa * cb * bri + ca * cb / rho * ti - sb / rho / sa * bei;
              const double gyr = sa * sb * brr + ca * sb / rho * tr + cb / rho / sa * ber;
              const double gyi = sa * sb * bri + ca * sb / rho * ti + cb / rho / sa * bei;
              const double gzr = ca * brr - sa / rho * tr;
              const double gzi = ca * bri - sa / rho * ti;
              vr = n0 * gxr + n1 * gyr + n2 * gzr;
              vi = n0 * gxi + n1 * gyi + n2 * gzi;
              if constexpr (NT == 3) {
                const double gr[3] = {gxr, gyr, gzr}, gi[3] = {gxi, gyi, gzi};
#pragma unroll
                for (int e = 0; e < 3; ++e) { acc[e].x += aw[q] * gr[e]; acc[e].y += aw[q] * gi[e]; }
              }
            }
            if constexpr (NT == 1) { acc[0].x += aw[q] * vr; acc[0].y += aw[q] * vi; }
            else if constexpr (NT != 3) {
              const double wt[4] = {aw[q], aw[q] * qx[q], aw[q] * qy[q], aw[q] * qz[q]};
#pragma unroll
              for (int e = 0; e < NT; ++e) { acc[e].x += wt[e] * vr; acc[e].y += wt[e] * vi; }
            }
            p1[q] = pcur; p[q] = pnext;
            rhon[q] *= s[q].rho;
          }
          const int idx = n * (n + 1) / 2 + m;
          if constexpr (NT == 1) {
            if (d.p2m_packed) {
              if (m) out[packed_cpos(n, m)] = acc[0];
              else reinterpret_cast<double*>(out + d.p2m_real_off)[n] = acc[0].x;
            } else out[idx] = acc[0];
          } else {
#pragma unroll
            for (int e = 0; e < (NT == 3 ? 3 : NT); ++e) out[(size_t)e * SM + idx] = acc[e];
          }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          pn[q] = -pn[q] * fact * s[q].sa;
          rhom[q] *= s[q].rho;
          const double nr = er[q] * s[q].cb - ei[q] * s[q].sb, ni = er[q] * s[q].sb + ei[q] * s[q].cb;
          er[q] = nr; ei[q] = ni;
        }
        fact += 2;
      }
    }
  }
}

#ifndef FMMBEM_P2M_INFLIGHT
#define FMMBEM_P2M_INFLIGHT 4
#endif
typedef double tvec2 __attribute__((ext_vector_type(2)));   // native 16-B vector (the nontemporal builtin needs it)

template <int NT>
__global__ __launch_bounds__(4 * kWave) void p2m_apply_kernel(DevicePlan d, const int P) {
  const int S = P * (P + 1) / 2, SM = d.s_max, TS = d.p2m_stride;      // strides of an expansion and of a table record
  const int lane = threadIdx.x & (kWave - 1);
  const double2* __restrict__ tab = d.p2m_tab - (size_t)d.p2m_tab_row0 * NT * TS;      // indexed by tree-order panel
  __shared__ double2 low_part[4][2 * kWave];
  for (int li = blockIdx.x * 4 + threadIdx.x / kWave; li < d.n_p2m; li += gridDim.x * 4) {
    const int leaf = d.p2m_leaf[li], box = d.leaf_box[leaf];
    const int row0 = d.leaf_row0[leaf], nrows = d.leaf_nrows[leaf];
    if (NT == 1 && S <= kWave / 2) {
      // low order (where the relaxed solver spends most iterations): S <= 32 coefficients would leave most lanes idle, so
      // G = 64 / S groups of lanes take every G-th panel and the groups are added in order through the LDS
      const int G = kWave / S, g = lane / S, idx = lane - g * S;
      int cn, cm;
      coef_nm(idx, cn, cm);
      const int tslot = !d.p2m_packed ? idx : cm ? packed_cpos(cn, cm) : 0;
      double2 m0 = {0, 0}, m1 = {0, 0};
      if (g < G)
        for (int r = g; r < nrows; r += G) {
          const int64_t i = row0 + r;
          const double2* rec = tab + (size_t)i * TS;
          const double2 t = (d.p2m_packed && !cm) ? double2{reinterpret_cast<const double*>(rec + d.p2m_real_off)[cn], 0.0} : rec[tslot];
          const double x = d.xt[i];
          if (d.bc[i]) { m1.x = fma(x, t.x, m1.x); m1.y = fma(x, t.y, m1.y); }
          else { m0.x = fma(x, t.x, m0.x); m0.y = fma(x, t.y, m0.y); }
        }
      double2* part = low_part[threadIdx.x / kWave];
      wave_lds_sync();
      part[lane] = m0; part[kWave + lane] = m1;
      wave_lds_sync();
      if (lane < S) {
        double2 s0 = {0, 0}, s1 = {0, 0};
        for (int q = 0; q < G; ++q) {
          const double2 a = part[q * S + lane], b = part[kWave + q * S + lane];
          s0.x += a.x; s0.y += a.y; s1.x += b.x; s1.y += b.y;
        }
        for (int a = 0; a < d.n_act; ++a) {
          const int slot = a == 0 ? d.act[0] : d.act[1];
          d.M[((size_t)box * d.nslots + slot) * SM + lane] = slot ? s1 : s0;
        }
      }
      continue;
    }
    for (int idx = lane; idx < S; idx += kWave) {
      if (NT == 1) {
        int cn, cm;
        coef_nm(idx, cn, cm);
        const bool real_only = d.p2m_packed && !cm;    // a packed record holds the m = 0 moments as bare reals
        const int tslot = !d.p2m_packed ? idx : cm ? packed_cpos(cn, cm) : 0;
        double2 m0 = {0, 0}, m1 = {0, 0};              // G moments (POTENTIAL panels) / dG/dn moments (NORMAL_DERIV panels)
        constexpr int U = FMMBEM_P2M_INFLIGHT;         // panels' records in flight, added in panel order; the last batch of a leaf is
                                                       // a masked one (one latency, not one per leftover panel).  The table is read
                                                       // once per matvec and is 0.9 GB at N = 1M, p = 10: nontemporal, like the near blocks
        for (int r = 0; r < nrows; r += U) {
          const int64_t i = row0 + r;
          tvec2 t[U];
          double x[U];
          bool dn[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const bool ok = r + u < nrows;
            const int64_t iu = ok ? i + u : i;
            if (real_only) t[u] = ok ? tvec2{__builtin_nontemporal_load(reinterpret_cast<const double*>(tab + (size_t)iu * TS + d.p2m_real_off) + cn), 0.0} : tvec2{0, 0};
            else t[u] = ok ? __builtin_nontemporal_load(reinterpret_cast<const tvec2*>(tab + (size_t)iu * TS + tslot)) : tvec2{0, 0};
            x[u] = d.xt[iu]; dn[u] = d.bc[iu] != 0;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (dn[u]) { m1.x = fma(x[u], t[u].x, m1.x); m1.y = fma(x[u], t[u].y, m1.y); }
            else { m0.x = fma(x[u], t[u].x, m0.x); m0.y = fma(x[u], t[u].y, m0.y); }
          }
        }
        for (int a = 0; a < d.n_act; ++a) {
          const int slot = a == 0 ? d.act[0] : d.act[1];
          d.M[((size_t)box * d.nslots + slot) * SM + idx] = slot ? m1 : m0;
        }
      } else if (NT == 3) {
        // Stokes double layer (stresslet), the far field of TRACTION targets.  With phi = 1/|x - y| and the source's density g
        // and normal n,   -3 (d.n) d_i (d.g) / r^5  =  x_k d_i Psi_k - d_i Psi_0 - Theta_i   (d = x - y, d_i = d/dx_i),
        //   Psi_k = sum w A n_k (g . grad_y) phi,   Psi_0 = sum w A (y.n) (g . grad_y) phi,   Theta_i = sum w A g_i (n . grad_y) phi
        // -- seven harmonic dipole potentials (from d_i d_j / r^5 = (d_i d_j phi + delta_ij / r^3) / 3).  Their multipoles are
        // combinations of the panel's three gradient records G = sum_q w_q A grad(rho^n Ynm): slots 4..6 n_k (g.G), slot 7
        // (c.n)(g.G) -- y.n is constant on a flat panel --, slots 8..10 g_i (n.G).
        const double2* gtab = d.p2m_tab_g - (size_t)d.p2m_tab_row0 * 3 * TS;
        double2 m[7] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll 2
        for (int r = 0; r < nrows; ++r) {
          const int64_t i = row0 + r;
          const double2* t = gtab + (size_t)i * 3 * TS + idx;
          const double f0 = d.xt[3 * i], f1 = d.xt[3 * i + 1], f2 = d.xt[3 * i + 2];
          const double n0 = d.nx[i], n1 = d.ny[i], n2 = d.nz[i];
          const double yn = d.cx[i] * n0 + d.cy[i] * n1 + d.cz[i] * n2;
          const double2 t0 = t[0], t1 = t[TS], t2 = t[2 * TS];
          const double2 gG = {fma(f0, t0.x, fma(f1, t1.x, f2 * t2.x)), fma(f0, t0.y, fma(f1, t1.y, f2 * t2.y))};
          const double2 nG = {fma(n0, t0.x, fma(n1, t1.x, n2 * t2.x)), fma(n0, t0.y, fma(n1, t1.y, n2 * t2.y))};
          m[0].x = fma(n0, gG.x, m[0].x); m[0].y = fma(n0, gG.y, m[0].y);
          m[1].x = fma(n1, gG.x, m[1].x); m[1].y = fma(n1, gG.y, m[1].y);
          m[2].x = fma(n2, gG.x, m[2].x); m[2].y = fma(n2, gG.y, m[2].y);
          m[3].x = fma(yn, gG.x, m[3].x); m[3].y = fma(yn, gG.y, m[3].y);
          m[4].x = fma(f0, nG.x, m[4].x); m[4].y = fma(f0, nG.y, m[4].y);
          m[5].x = fma(f1, nG.x, m[5].x); m[5].y = fma(f1, nG.y, m[5].y);
          m[6].x = fma(f2, nG.x, m[6].x); m[6].y = fma(f2, nG.y, m[6].y);
        }
#pragma unroll
        for (int e = 0; e < 7; ++e) d.M[((size_t)box * d.nslots + 4 + e) * SM + idx] = m[e];
      } else {
        double2 m[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll 2
        for (int r = 0; r < nrows; ++r) {
          const int64_t i = row0 + r;
          const double2* t = tab + (size_t)i * 4 * TS + idx;
          const double f0 = d.xt[3 * i], f1 = d.xt[3 * i + 1], f2 = d.xt[3 * i + 2];
          const double2 t0 = t[0], t1 = t[TS], t2 = t[2 * TS], t3 = t[3 * TS];
          m[0].x = fma(f0, t0.x, m[0].x); m[0].y = fma(f0, t0.y, m[0].y);
          m[1].x = fma(f1, t0.x, m[1].x); m[1].y = fma(f1, t0.y, m[1].y);
          m[2].x = fma(f2, t0.x, m[2].x); m[2].y = fma(f2, t0.y, m[2].y);
          m[3].x = fma(f0, t1.x, fma(f1, t2.x, fma(f2, t3.x, m[3].x)));
          m[3].y = fma(f0, t1.y, fma(f1, t2.y, fma(f2, t3.y, m[3].y)));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) d.M[((size_t)box * d.nslots + e) * SM + idx] = m[e];
      }
    }
  }
}

// p2m_apply, one expansion per box (every panel the same boundary condition: the reference's drivers) and more than 32
// coefficients: the streaming form.  The kernel above spends three vector loads per panel (record, charge, flag) where one
// carries data, waits for every batch of four records before asking for the next, and finds its leaf through a chain of four
// dependent vector loads.  Here everything that is the same for all lanes is scalar -- the leaf index is wave-uniform by
// construction (readfirstlane), leaf record, row range and the charges come through the scalar cache (constant address
// space: the M stores of this kernel would otherwise count as clobbers), the NEXT leaf's record is fetched while this leaf
// streams -- and the only vector loads are the 16-byte table entries, kP2MStream panels' worth in flight per lane.
// 0.219 -> 0.194 ms at N = 1M, p = 10 (0.94 GB: 4.8 TB/s).  What it does NOT respond to (profiles/r03f, r03g): 4, 12 or 16
// entries in flight (0.192 / 0.203 / 0.208), 4 to 16 workgroups per CU (0.197 ... 0.192), contiguous leaf ranges per
// workgroup instead of the strided deal (0.194): 57 MB are in flight at 8 192 wavefronts x 7 KB, the memory system returns
// 4.8 TB/s for this pattern whatever the kernel asks.
constexpr int kP2MStream = 8;
typedef __attribute__((address_space(4))) int ConstInt;
typedef __attribute__((address_space(4))) double ConstDouble;
template <class T, class U>
__device__ __forceinline__ const T* as_const_space(const U* p) { return reinterpret_cast<const T*>(reinterpret_cast<uintptr_t>(p)); }

__global__ __launch_bounds__(4 * kWave) void p2m_stream_kernel(DevicePlan d, const int P) {
  constexpr int U = kP2MStream;
  const int S = P * (P + 1) / 2, SM = d.s_max, TS = d.p2m_stride;
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const double2* __restrict__ tab = d.p2m_tab - (size_t)d.p2m_tab_row0 * TS;      // indexed by tree-order panel
  const ConstInt* leaf_of = as_const_space<ConstInt>(d.p2m_leaf);
  const ConstInt* box_of = as_const_space<ConstInt>(d.leaf_box);
  const ConstInt* row0_of = as_const_space<ConstInt>(d.leaf_row0);
  const ConstInt* nrows_of = as_const_space<ConstInt>(d.leaf_nrows);
  const ConstDouble* xt = as_const_space<ConstDouble>(d.xt);
  const int slot = d.act[0], stride = gridDim.x * 4, n = d.n_p2m;
  // what this lane streams (at most two items: 128 >= the 120 + 8 of p = 16): the table slot it loads and where the sums go.
  // Classic records: item = coefficient.  Packed records: the complex (n, m >= 1) in order, then the m = 0 reals two to a slot.
  int tslot[2], st0[2], st1[2];
  {
    const int nb = d.p2m_packed ? P * (P - 1) / 2 : S, items = d.p2m_packed ? nb + (P + 1) / 2 : S;
    for (int it = 0; it < 2; ++it) {
      const int item = lane + it * kWave;
      tslot[it] = -1; st0[it] = 0; st1[it] = -2;
      if (item >= items) continue;
      if (!d.p2m_packed) { tslot[it] = item; st0[it] = item; }
      else if (item < nb) {
        int cn = 1;
        while (cn * (cn + 1) / 2 <= item) ++cn;          // degree of packed position `item`
        tslot[it] = item; st0[it] = item + cn + 1;       // n (n + 1) / 2 + m  =  cpos + n + 1
      } else {
        const int k = item - nb;
        tslot[it] = d.p2m_real_off + k;
        st0[it] = (2 * k) * (2 * k + 1) / 2;
        st1[it] = 2 * k + 1 < P ? (2 * k + 1) * (2 * k + 2) / 2 : -1;
      }
    }
  }
  int li = blockIdx.x * 4 + wv;
  if (li >= n) return;
  int leaf = leaf_of[li];
  int box = box_of[leaf], row0 = row0_of[leaf], nrows = nrows_of[leaf];
  for (; li < n; li += stride) {
    // the next leaf's record, on its way while this one streams
    const int nl = li + stride < n ? li + stride : li;
    const int nleaf = leaf_of[nl];
    const int nbox = box_of[nleaf], nrow0 = row0_of[nleaf], nnrows = nrows_of[nleaf];
    for (int it = 0; it < 2; ++it) {
      if (tslot[it] < 0) break;
      double2 m = {0, 0};
      for (int r = 0; r < nrows; r += U) {
        tvec2 t[U];
        double x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                    // a leaf's last batch is a short one: the row test is scalar, a branch
          t[u] = tvec2{0, 0}; x[u] = 0.0;                //  around the load (re-reading row r instead costs L2 requests: 24 for 19 rows)
          if (r + u < nrows) {
            const int64_t iu = (int64_t)row0 + r + u;
            t[u] = __builtin_nontemporal_load(reinterpret_cast<const tvec2*>(tab + (size_t)iu * TS + tslot[it]));
            x[u] = xt[iu];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { m.x = fma(x[u], t[u].x, m.x); m.y = fma(x[u], t[u].y, m.y); }   // panel order, as p2m_apply_kernel
      }
      double2* Mb = d.M + ((size_t)box * d.nslots + slot) * SM;
      if (st1[it] == -2) Mb[st0[it]] = m;                // a complex coefficient (classic layout, or packed m >= 1)
      else {                                             // a packed pair of m = 0 moments: two real coefficients
        Mb[st0[it]] = double2{m.x, 0.0};
        if (st1[it] >= 0) Mb[st1[it]] = double2{m.y, 0.0};
      }
    }
    leaf = nleaf; box = nbox; row0 = nrow0; nrows = nnrows;
  }
}

// ---------------------------------------------------------------------------------------------
// M2M, one tree level per launch.  A workgroup = kShiftWaves wavefronts takes one parent box; wavefront w
// translates child w (children are the <= 8 occupied octants) with the precomputed sparse operator
// (shift_ops.hpp): lanes = output rows, ELL term lists read coalesced, the child's M and the class's
// regular harmonics in LDS.  The per-child results are combined in child order (the reference's order,
// EvalInteractionLazySparse.hpp:185-190) and stored -- a parent's M has no other contribution.
// Workgroups stride over the level's parents.
// ---------------------------------------------------------------------------------------------
constexpr int kShiftWaves = 8;
#ifndef FMMBEM_L2L_WAVES
#define FMMBEM_L2L_WAVES 4
#endif

// One wavefront applies a shift operator: in[] = source expansion (LDS), Y[] = the class's harmonics (LDS);
// every lane sums whole pieces (<= T terms, ELL over pieces: coalesced) into pv[] (LDS), then lane = output row adds
// its pieces in order.  The caller wraps it in wavefront barriers.
__device__ __forceinline__ double2 shift_apply_row(const ShiftOpDev& op, const double2* pv, int S, int idx) {
  double2 sum = {0, 0};
  const int np = op.npiece[idx];
  for (int k = 0; k < np; ++k) {
    const double2 t = pv[op.piece[(size_t)k * S + idx]];
    sum.x += t.x; sum.y += t.y;
  }
  return sum;
}
__device__ __forceinline__ void shift_pieces(const ShiftOpDev& op, const double2* in, const double2* Y, double2* pv, int lane) {
  for (int v = lane; v < op.V; v += kWave) {            // V is a multiple of 64; padding pieces are zeros
    double2 acc = {0, 0};
    for (int i = 0; i < op.T; i += 4) {                 // four terms' operands in flight
      unsigned sc[4], yi[4];
      double r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t e = (size_t)(i + u) * op.V + v;
        sc[u] = op.src[e]; yi[u] = op.y[e]; r[u] = op.real[e];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        double2 x = in[sc[u] & 0x7fff];
        if (sc[u] & 0x8000) x.y = -x.y;
        const double2 t = cmul(x, Y[yi[u]]);
        acc.x = fma(t.x, r[u], acc.x); acc.y = fma(t.y, r[u], acc.y);
      }
    }
    pv[v] = acc;
  }
}
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (Round 3 tried the levels near the root as ONE launch with a device-wide barrier between the levels: a barrier through L2
// atomics costs ~10 us, a dependent launch 4.5 us including the kernel -- profiles/r03e_fused_levels_and_p2m_stream.txt.)
__global__ __launch_bounds__(kShiftWaves * kWave) void m2m_kernel(DevicePlan d, ShiftOpDev op, const int P, int first, int count) {
  const int slot_i = blockIdx.y, wg = blockIdx.x, nwg = gridDim.x;
  extern __shared__ double2 lds2[];
  const int S = P * (P + 1) / 2, P2 = P * P, SM = d.s_max, W = S + P2 + op.V;
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  double2* Ms = lds2 + (size_t)w * W;                 // this wavefront's child multipole
  double2* Y = Ms + S;                                // ... its translation harmonics
  double2* pv = Y + P2;                               // ... and its per-piece partial sums
  double2* part = lds2 + (size_t)kShiftWaves * W + (size_t)w * S;
  const int slot = d.act[slot_i];
  for (int it = wg; it < count; it += nwg) {
    const int parent = d.m2m_parent[first + it];
    const int cb = d.box_child_begin[parent], nchild = d.box_child_end[parent] - cb;
    if (w < nchild) {
      const int c = cb + w;
      const double2* src = d.M + ((size_t)c * d.nslots + slot) * SM;
      const double2* tab = d.up_tab + (size_t)d.up_cls[c] * d.p2_max;
      for (int i = lane; i < S; i += kWave) Ms[i] = src[i];
      for (int i = lane; i < P2; i += kWave) Y[i] = tab[i];
      wave_lds_sync();
      shift_pieces(op, Ms, Y, pv, lane);
      wave_lds_sync();
      for (int idx = lane; idx < S; idx += kWave) part[idx] = shift_apply_row(op, pv, S, idx);
    }
    __syncthreads();
    double2* dst = d.M + ((size_t)parent * d.nslots + slot) * SM;
    const double2* parts = lds2 + (size_t)kShiftWaves * W;
    for (int idx = threadIdx.x; idx < S; idx += blockDim.x) {
      double2 sum = {0, 0};
      for (int c = 0; c < nchild; ++c) { sum.x += parts[(size_t)c * S + idx].x; sum.y += parts[(size_t)c * S + idx].y; }
      dst[idx] = sum;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// mh_prep: Mh[n,m] = i^{-m} A[n,m] M[n,m] for the stored orders m >= 0 of every M2L source box
// (negative orders follow from Mh[n,-m] = (-1)^m conj(Mh[n,m]) inside the M2L kernel).  Stored ORDER-major,
// Mh[m*P - m(m-1)/2 + (n-m)]: the M2L kernel sums over n for a fixed m and reads the run with wide scalar loads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void mh_prep_kernel(DevicePlan d, const int P) {
  const int S = P * (P + 1) / 2;
  const int box = d.mh_box[blockIdx.x];
  const int slot = d.act[blockIdx.y];
  const double2* M = d.M + ((size_t)box * d.nslots + slot) * d.s_max;
  double2* Mh = d.Mh + ((size_t)box * d.nslots + slot) * d.s_max;
  for (int idx = threadIdx.x; idx < S; idx += kWave) {
    const int n = kJK.j[idx], m = kJK.k[idx];
    const double2 v = M[idx];
    const double a = d.tabA[n * n + n + m];
    Mh[m * P - m * (m - 1) / 2 + (n - m)] = mul_i_pow(double2{v.x * a, v.y * a}, -m);
  }
}

// ---------------------------------------------------------------------------------------------
// L2L, one tree level per launch: L[child] += shift(L[parent]).  One wavefront per child box (kShiftWaves
// independent children per workgroup), same sparse-operator scheme as M2M; the terms usable at order p are
// a prefix of each row's list (down_cnt).  Workgroups stride over the level's children.
// ---------------------------------------------------------------------------------------------
constexpr int kL2LWaves = FMMBEM_L2L_WAVES;            // independent children per workgroup (LDS: 10.7 KB each at p = 10)
__global__ __launch_bounds__(kL2LWaves * kWave) void l2l_kernel(DevicePlan d, ShiftOpDev op, const int P, int first, int count) {
  const int slot_i = blockIdx.y, wg = blockIdx.x, nwg = gridDim.x;
  extern __shared__ double2 lds2[];
  const int S = P * (P + 1) / 2, P2 = P * P, SM = d.s_max, W = S + P2 + op.V;
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  double2* Ls = lds2 + (size_t)w * W;
  double2* Y = Ls + S;
  double2* pv = Y + P2;
  const int slot = d.act[slot_i];
  for (int it = wg * kL2LWaves + w; it < count; it += nwg * kL2LWaves) {
    const int child = d.l2l_child[first + it];
    const int parent = d.box_parent[child];
    const double2* src = d.L + ((size_t)parent * d.nslots + slot) * SM;
    const double2* tab = d.down_tab + (size_t)d.down_cls[child] * d.p2_max;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < S; i += kWave) Ls[i] = src[i];
    for (int i = lane; i < P2; i += kWave) Y[i] = tab[i];
    wave_lds_sync();
    shift_pieces(op, Ls, Y, pv, lane);
    wave_lds_sync();
    double2* dst = d.L + ((size_t)child * d.nslots + slot) * SM;
    for (int idx = lane; idx < S; idx += kWave) {
      const double2 add = shift_apply_row(op, pv, S, idx);
      double2 acc = dst[idx];
      acc.x += add.x; acc.y += add.y;
      dst[idx] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// L2P: lane = panel centroid.  A leaf of the bench tree holds ~19 panels, so a wavefront takes a GROUP of consecutive
// leaves (<= 64 rows, <= 8 leaves, DevicePlan::l2p_grp; a leaf with more rows is a group of its own and is walked in
// chunks); the L of the group's leaves (active slots) is staged in LDS and every lane reads its own leaf's.
// y_tree[i] += r0 (POTENTIAL target) or -= r1 (NORMAL_DERIV target): the far field joins the near field in TREE order
// (coalesced; the one scatter to the caller's order comes after, plan.hip).
// ---------------------------------------------------------------------------------------------
// The L of a group's leaves into this wavefront's LDS slice, `slots` expansions per leaf ([leaf][slot][S], the flat index of
// an element IS its LDS index).  Lane k looks up leaf k -- two dependent loads for all leaves at once -- and the copy then has
// every load of a batch in flight before the first LDS write.  (Leaf by leaf in a loop it was three dependent round trips per
// leaf, 24 per group, most of the kernel's time: 91 -> see DESIGN.md section 4.)  Returns through the references this lane's leaf
// within the group, that leaf's first lane, the rows of the group; my_* = what lane k found for leaf k.
template <class SlotOf>
__device__ __forceinline__ void l2p_stage_group(const DevicePlan& d, int l0, int nl, int slots, int S, int lane, double2* Lw, SlotOf slot_of,
                                                int& my_leaf, int& my_box, int& my_nr, int& g, int& first, int& total) {
  my_leaf = 0; my_box = 0; my_nr = 0;
  if (lane < nl) { my_leaf = d.l2p_leaf[l0 + lane]; my_box = d.leaf_box[my_leaf]; my_nr = d.leaf_nrows[my_leaf]; }
  const int per = slots * S, T = nl * per;
  constexpr int U = 8;
  for (int e0 = 0; e0 < T; e0 += U * kWave) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * kWave + lane;
      const int ee = e < T ? e : T - 1;
      const int k = ee / per, rem = ee - k * per, a = rem / S, i = rem - a * S;
      const int box = __shfl(my_box, k, kWave);
      v[u] = d.L[((size_t)box * d.nslots + slot_of(a)) * d.s_max + i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * kWave + lane;
      if (e < T) Lw[e] = v[u];
    }
  }
  g = -1; first = 0; total = 0;
  for (int k = 0; k < nl; ++k) {
    const int nr = __shfl(my_nr, k, kWave);
    if (lane >= total && lane < total + nr) { g = k; first = total; }
    total += nr;
  }
}

constexpr int kL2PLeaves = 8, kL2PWaves = 4;
// Stokes stages 4 or 7 expansions per leaf: with room for 8 leaves a wavefront's LDS slice (18 KB at p = 8) leaves two wavefronts per
// SIMD; groups of at most 4 leaves (plan.hip asks l2p_group_leaves) halve it
constexpr int kL2PLeavesStokes = 4;
constexpr int kL2PUnrollMax = 12;                     // orders with an L2P kernel of their own (the orders of the rotation kernels)
static bool l2p_generic() { const char* e = std::getenv("FMMBEM_L2P_GENERIC"); return e && std::atoi(e) != 0; }   // A/B runs
// store: y[i] = the far field (the near field runs beside this on another stream and the two meet in the delivery kernel)
// instead of y[i] += (the near field is already there).
// PT > 0: the order at compile time -- both loops unrolled, every table entry and every coefficient at a constant LDS offset,
// no loop counters or index arithmetic left (the runtime-order loop spends more instructions on those than on the recurrences:
// L2P at N = 1M, p = 10: 0.089 ms generic, see DESIGN.md section 4 for the unrolled figure).  PT = 0: any order (p > kL2PUnrollMax).
template <int PT>
__global__ __launch_bounds__(kL2PWaves * kWave) __attribute__((amdgpu_waves_per_eu(4))) void l2p_kernel(DevicePlan d, const int Prt, double* __restrict__ y, const int store) {
  extern __shared__ double2 l2p_lds[];                  // step tables (3 S doubles), then [wave][leaf][active slot][S]
  const int P = PT > 0 ? PT : Prt;
  const int S = P * (P + 1) / 2, na = d.n_act;
  double* sPref = reinterpret_cast<double*>(l2p_lds);
  double* sC1 = sPref + S;
  double* sC2 = sC1 + S;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  double2* Lw = l2p_lds + (3 * S + 1) / 2 + (size_t)wave * kL2PLeaves * na * S;
  if (wave == 0) fill_step_tables(d, P, lane, sPref, sC1, sC2);
  __syncthreads();
  for (int gi = blockIdx.x * nwaves + wave; gi < d.n_l2p_grp; gi += gridDim.x * nwaves) {
    const int l0 = d.l2p_grp[gi], nl = d.l2p_grp[gi + 1] - l0;
    wave_lds_sync();                                    // the previous group's reads are done
    int g, first, total, my_leaf, my_box, my_nr;        // this lane's leaf within the group, its first lane, rows of the group
    const int act0 = d.act[0], act1 = d.act[1];
    l2p_stage_group(d, l0, nl, na, S, lane, Lw, [&](int a) { return a == 0 ? act0 : act1; }, my_leaf, my_box, my_nr, g, first, total);
    wave_lds_sync();
    if (nl == 1) { g = 0; first = 0; }                  // single leaf: every lane, chunk by chunk
    const int leaf = __shfl(my_leaf, g < 0 ? 0 : g, kWave), box = __shfl(my_box, g < 0 ? 0 : g, kWave);
    const int row0 = d.leaf_row0[leaf], nrows = __shfl(my_nr, g < 0 ? 0 : g, kWave);
    const double c0 = d.box_center[3 * box], c1 = d.box_center[3 * box + 1], c2 = d.box_center[3 * box + 2];
    for (int chunk = 0; chunk < total; chunk += kWave) {
      const int r_in_leaf = chunk + lane - first;
      if (g < 0 || r_in_leaf >= nrows) continue;
      const int64_t i = row0 + r_in_leaf;
      const int tb = d.bc[i] ? 1 : 0;
      const double2* Lt = Lw + (size_t)(g * na + (na == 2 ? tb : 0)) * S;
      const Sph s = cart2sph(d.cx[i] - c0, d.cy[i] - c1, d.cz[i] - c2);
      double r = 0;
      double pn = 1, rhom = 1, er = 1, ei = 0, fact = 1;
      auto order = [&](int m, int& step, auto unrolled) FMMBEM_INLINE {
        // unrolled: an offset the compiler cannot see through, tied to the sum so far -- at constant addresses all 4 S LDS reads of
        // a row are otherwise issued at its head (or lifted out of the chunk loop): 512 VGPRs and scratch at p = 10
        int tz = 0;
        if constexpr (decltype(unrolled)::value) asm volatile("" : "+v"(tz), "+v"(r));
        const double *tPref = sPref + tz, *tC1 = sC1 + tz, *tC2 = sC2 + tz;
        const double2* Lm = Lt + tz;
        double p = pn, p1 = p, rhon = rhom;
        const double w = m == 0 ? 1.0 : 2.0;
        auto degree = [&](int n) FMMBEM_INLINE {
          const double mag = rhon * p * tPref[step];
          const double2 Lc = Lm[n * (n + 1) / 2 + m];
          r += w * (Lc.x * (mag * er) - Lc.y * (mag * ei));        // Re(L * Ynm), Ynm = mag e^{+i m beta}
          const double pcur = p;
          p = tC1[step] * s.ca * pcur - tC2[step] * p1;
          p1 = pcur;
          rhon *= s.rho;
          ++step;
        };
        if constexpr (decltype(unrolled)::value) {
#pragma unroll
          for (int n = m; n < PT; ++n) degree(n);
        } else {
#pragma nounroll
          for (int n = m; n < P; ++n) degree(n);
        }
        pn = -pn * fact * s.sa;
        fact += 2;
        rhom *= s.rho;
        const double nr = er * s.cb - ei * s.sb, ni = er * s.sb + ei * s.cb;
        er = nr; ei = ni;
      };
      int step = 0;
      if constexpr (PT > 0) {
#pragma unroll
        for (int m = 0; m < PT; ++m) order(m, step, std::true_type{});
      } else {
#pragma nounroll
        for (int m = 0; m < P; ++m) order(m, step, std::false_type{});
      }
      y[i] = (store ? 0.0 : y[i]) + (tb ? -r : r);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Stokes L2P: StokesSpherical::L2P (kernel/StokesSpherical.hpp:318-401) then the 1/(2 mu) of
// StokesSphericalBEM::L2P (kernel/StokesSphericalBEM.hpp:512-522).  Lane = target panel; the leaf's four
// harmonic potentials L[0..3] in LDS; per lane the potentials' values and spherical gradients are
// accumulated along the same recurrence as the Laplace L2P (Ynm and its theta derivative), then
//   u_k = phi_k - x_0 d_k phi_0 - x_1 d_k phi_1 - x_2 d_k phi_2 + d_k phi_3   (Tornberg-Greengard)
// ---------------------------------------------------------------------------------------------
// TRAC: the rows of TRACTION targets, from the seven potentials of the double layer (p2m_apply_kernel<3>):
//   t_i = x_k d_i Psi_k - d_i Psi_0 - Theta_i     (slots 4..6 Psi_k, 7 Psi_0: gradients; 8..10 Theta_i: values; no 1/(2 mu))
// PT as in l2p_kernel: the order at compile time (both loops unrolled) or 0 for any order.
template <bool TRAC, int PT>
__global__ __launch_bounds__(kL2PWaves * kWave) __attribute__((amdgpu_waves_per_eu(PT > 0 ? 2 : 1))) void l2p_stokes_kernel(DevicePlan d, const int Prt, double* __restrict__ y, const int store) {
  constexpr int NE = TRAC ? 7 : 4, SB = TRAC ? 4 : 0;   // potentials staged per leaf, first slot
  extern __shared__ double2 l2p_lds[];                  // as l2p_kernel: step tables, then [wave][leaf][4 potentials][S]
  const int P = PT > 0 ? PT : Prt;
  const int S = P * (P + 1) / 2;
  double* sPref = reinterpret_cast<double*>(l2p_lds);
  double* sC1 = sPref + S;
  double* sC2 = sC1 + S;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  double2* Lw = l2p_lds + (3 * S + 1) / 2 + (size_t)wave * kL2PLeavesStokes * NE * S;
  if (wave == 0) fill_step_tables(d, P, lane, sPref, sC1, sC2);
  __syncthreads();
  for (int gi = blockIdx.x * nwaves + wave; gi < d.n_l2p_grp; gi += gridDim.x * nwaves) {
    const int l0 = d.l2p_grp[gi], nl = d.l2p_grp[gi + 1] - l0;
    wave_lds_sync();
    int g, first, total, my_leaf, my_box, my_nr;
    l2p_stage_group(d, l0, nl, NE, S, lane, Lw, [&](int e) { return SB + e; }, my_leaf, my_box, my_nr, g, first, total);
    wave_lds_sync();
    if (nl == 1) { g = 0; first = 0; }
    const int leaf = __shfl(my_leaf, g < 0 ? 0 : g, kWave), box = __shfl(my_box, g < 0 ? 0 : g, kWave);
    const int row0 = d.leaf_row0[leaf], nrows = __shfl(my_nr, g < 0 ? 0 : g, kWave);
    const double2* Ls = Lw + (size_t)(g < 0 ? 0 : g) * NE * S;
    const double c0 = d.box_center[3 * box], c1 = d.box_center[3 * box + 1], c2 = d.box_center[3 * box + 2];
    for (int chunk = 0; chunk < total; chunk += kWave) {
      const int r_in_leaf = chunk + lane - first;
      if (g < 0 || r_in_leaf >= nrows) continue;
      const int64_t i = row0 + r_in_leaf;
      if ((d.bc[i] != 0) != TRAC) continue;              // the target's flag picks the operator (StokesSphericalBEM.hpp:377-389)
      const double tx = d.cx[i], ty = d.cy[i], tz = d.cz[i];
      const Sph s = cart2sph(tx - c0, ty - c1, tz - c2);
      double val[3] = {0, 0, 0};               // potentials phi_0..2 at the target
      double g[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // (d/dr, d/dtheta, d/dphi-ish) of phi_0..3
      double pn = 1, rhom = 1, er = 1, ei = 0, fact = 1;
      const double inv_sa = 1.0 / s.sa, inv_rho = 1.0 / s.rho;     // one division each per row instead of one per coefficient
      auto order = [&](int m, int& step, auto unrolled) FMMBEM_INLINE {
        double p = pn, p1 = p, rhon = rhom;
        const double w = m == 0 ? 1.0 : 2.0;
        auto degree = [&](int n) FMMBEM_INLINE {
          // unrolled: the LDS reads of a coefficient (tables + 4 or 7 expansions) stay with that coefficient (see l2p_kernel; an
          // order at a time is already 280 VGPRs here)
          int tz = 0;
          if constexpr (decltype(unrolled)::value)         // every running sum: tied to one alone, the compiler runs that one's chain ahead through all coefficients
            asm volatile("" : "+v"(tz), "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(g[0][0]), "+v"(g[0][1]), "+v"(g[0][2]), "+v"(g[1][0]), "+v"(g[1][1]),
                         "+v"(g[1][2]), "+v"(g[2][0]), "+v"(g[2][1]), "+v"(g[2][2]), "+v"(g[3][0]), "+v"(g[3][1]), "+v"(g[3][2]));
          const double *tPref = sPref + tz, *tC1 = sC1 + tz, *tC2 = sC2 + tz;
          const double2* Lm = Ls + tz;
          const double pref = tPref[step];
          const double mag = rhon * p * pref;
          const double yr = mag * er, yi = mag * ei;                 // Ynm = mag e^{+i m beta}
          const double pcur = p;
          const double pnext = tC1[step] * s.ca * pcur - tC2[step] * p1;
          double tmag;                                               // YnmTheta (LaplaceSpherical.hpp:471,482)
          if (n == m) tmag = rhon * (pnext - (m + 1) * s.ca * pcur) * inv_sa * pref;
          else tmag = rhon * ((n - m + 1) * pnext - (n + 1) * s.ca * pcur) * inv_sa * pref;
          const double tr = tmag * er, ti = tmag * ei;
          const double factor = inv_rho * n;
          const int idx = n * (n + 1) / 2 + m;
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const double2 L = Lm[e * S + idx];
            const double re = L.x * yr - L.y * yi;                   // Re(L Ynm)
            if (TRAC ? e >= 4 : e < 3) val[TRAC ? e - 4 : e] += w * re;
            if (e < 4) {
              g[e][0] += w * re * factor;
              g[e][1] += w * (L.x * tr - L.y * ti);                  // Re(L YnmTheta)
              if (m) g[e][2] += 2 * (-(L.x * yi + L.y * yr)) * m;    // Re(L Ynm i) m
            }
          }
          p1 = pcur; p = pnext;
          rhon *= s.rho;
          ++step;
        };
        if constexpr (decltype(unrolled)::value) {
#pragma unroll
          for (int n = m; n < PT; ++n) degree(n);
        } else {
#pragma nounroll
          for (int n = m; n < P; ++n) degree(n);
        }
        pn = -pn * fact * s.sa;
        fact += 2;
        rhom *= s.rho;
        const double nr = er * s.cb - ei * s.sb, ni = er * s.sb + ei * s.cb;
        er = nr; ei = ni;
      };
      int step = 0;
      if constexpr (PT > 0) {
#pragma unroll
        for (int m = 0; m < PT; ++m) order(m, step, std::true_type{});
      } else {
#pragma nounroll
        for (int m = 0; m < P; ++m) order(m, step, std::false_type{});
      }
      // sph2cart (kernel/LaplaceSpherical.hpp:546-561) of each gradient, then the recombination
      const double r = s.rho, st = s.sa, ct = s.ca, cp = s.cb, sp = s.sb;
      double res[3] = {TRAC ? -val[0] : val[0], TRAC ? -val[1] : val[1], TRAC ? -val[2] : val[2]};
      const double tgt[3] = {tx, ty, tz};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const double cx_ = st * cp * g[e][0] + ct * cp / r * g[e][1] - sp / r / st * g[e][2];
        const double cy_ = st * sp * g[e][0] + ct * sp / r * g[e][1] + cp / r / st * g[e][2];
        const double cz_ = ct * g[e][0] - st / r * g[e][1];
        const double f = (TRAC ? -1.0 : 1.0) * (e < 3 ? -tgt[e] : 1.0);
        res[0] += f * cx_; res[1] += f * cy_; res[2] += f * cz_;
      }
      const double sc = TRAC ? 1.0 : 1. / 2 / d.mu;
      y[3 * (size_t)i] = (store ? 0.0 : y[3 * (size_t)i]) + sc * res[0];
      y[3 * (size_t)i + 1] = (store ? 0.0 : y[3 * (size_t)i + 1]) + sc * res[1];
      y[3 * (size_t)i + 2] = (store ? 0.0 : y[3 * (size_t)i + 2]) + sc * res[2];
    }
  }
}

// __constant__ symbols and function attributes belong to a device: once per device the process uses (one process per GPU is
// the deployment model, but the C ABI takes a device ordinal and a process may create plans on several)
hipError_t upload_constants_once() {
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (done[dev]) return hipSuccess;
  const hipError_t st = [] {
    const JK t = make_jk();
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(kJK), &t, sizeof(t));
    double recip[40];
    recip[0] = 0;
    for (int k = 1; k < 40; ++k) recip[k] = 1.0 / k;
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(kRecip), recip, sizeof(recip));
    // M2M at p = 16 needs a little over 64 KiB of dynamic LDS
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(m2m_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(l2l_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    // L2P with one wavefront per workgroup at p = 16: 8 leaves x 4 potentials x 136 coefficients = 69.6 KB (Stokes)
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(l2p_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(l2p_stokes_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(l2p_stokes_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    return e;
  }();
  if (st == hipSuccess) done[dev] = true;
  return st;
}

}  // namespace

hipError_t launch_p2m_table(const DevicePlan& d, double2* tab, hipStream_t s) {
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (d.n_p2m <= 0) return hipSuccess;
  const dim3 g(d.n_p2m < 256 * 32 ? d.n_p2m : 256 * 32), b(kWave);
  // FMMBEM_P2M_TABLE_POINTS=0: the point-by-point kernel at every rule (A/B: the same records to <= 4e-16)
  const bool pts = !(std::getenv("FMMBEM_P2M_TABLE_POINTS") && std::atoi(std::getenv("FMMBEM_P2M_TABLE_POINTS")) == 0);
#define P2M_TABLE_CASE(NTV)                                                                             \
  if (pts && d.nq == 1) hipLaunchKernelGGL((p2m_table_points_kernel<NTV, 1>), g, b, 0, s, d, tab);       \
  else if (pts && d.nq == 3) hipLaunchKernelGGL((p2m_table_points_kernel<NTV, 3>), g, b, 0, s, d, tab);  \
  else if (pts && d.nq == 4) hipLaunchKernelGGL((p2m_table_points_kernel<NTV, 4>), g, b, 0, s, d, tab);  \
  else hipLaunchKernelGGL((p2m_table_kernel<NTV>), g, b, 0, s, d, tab)
  if (d.kernel == 1) { P2M_TABLE_CASE(4); } else { P2M_TABLE_CASE(1); }
  return hipGetLastError();
}

hipError_t launch_p2m_table_grad(const DevicePlan& d, double2* tab, hipStream_t s) {      // Stokes double layer: three gradient records per panel
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (d.n_p2m <= 0) return hipSuccess;
  const dim3 g(d.n_p2m < 256 * 32 ? d.n_p2m : 256 * 32), b(kWave);
  const bool pts = !(std::getenv("FMMBEM_P2M_TABLE_POINTS") && std::atoi(std::getenv("FMMBEM_P2M_TABLE_POINTS")) == 0);
  P2M_TABLE_CASE(3);
#undef P2M_TABLE_CASE
  return hipGetLastError();
}

hipError_t launch_p2m(const DevicePlan& d, int p, hipStream_t s) {
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (d.n_p2m <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  if (d.p2m_tab) {
    const int nb = (d.n_p2m + 3) / 4;
    const char* se = std::getenv("FMMBEM_P2M_STREAM");               // read per launch: tests switch it per plan
    const bool stream_off = se && std::atoi(se) == 0;
    if (d.n_act == 1 && p * (p + 1) / 2 > kWave / 2 && !stream_off)      // one expansion per box, more than 32 coefficients
      hipLaunchKernelGGL(p2m_stream_kernel, dim3(nb < 256 * 8 ? nb : 256 * 8), dim3(4 * kWave), 0, s, d, p);
    else
      hipLaunchKernelGGL((p2m_apply_kernel<1>), dim3(nb < 256 * 16 ? nb : 256 * 16), dim3(4 * kWave), 0, s, d, p);
    return hipGetLastError();
  }
  const int nblk = (d.n_p2m + kP2MWaves - 1) / kP2MWaves;
  const dim3 g(nblk < 256 * 8 ? nblk : 256 * 8), b(kP2MWaves * kWave);
  for (int a = 0; a < d.n_act; ++a) {
    if (d.act[a] == 0) hipLaunchKernelGGL((p2m_kernel<0>), g, b, 0, s, d, p, 0, 0);
    else hipLaunchKernelGGL((p2m_kernel<1>), g, b, 0, s, d, p, 0, 1);
  }
  return hipGetLastError();
}

hipError_t launch_m2m_level(const DevicePlan& d, const ShiftOpDev& op, int p, int first, int count, hipStream_t s) {
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (count <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  {
    const size_t S = (size_t)p * (p + 1) / 2, P2 = (size_t)p * p;
    const size_t lds = (kShiftWaves * (S + P2 + op.V) + kShiftWaves * S) * sizeof(double2);
    hipLaunchKernelGGL(m2m_kernel, dim3(count < 1024 ? count : 1024, d.n_act), dim3(kShiftWaves * kWave), lds, s, d, op, p, first, count);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Sharded upward pass (multi-GPU): every shard computes P2M/M2M only for the boxes whose bodies are all its own,
// the multipoles travel in ONE all-gather (the caller's collective, equal chunks of xch_max boxes), and the
// few boxes spanning shards are then translated by everybody.  pack: M of my boxes -> send; unpack: every other
// shard's chunk -> M.  Layout of a chunk: [box index within the shard][active slot][S(p)] complex.
// ---------------------------------------------------------------------------------------------
__global__ void xch_pack_kernel(DevicePlan d, const int P, double2* __restrict__ send) {
  const int S = P * (P + 1) / 2, per = d.n_act * S;
  const int first = d.xch_ptr[d.xch_rank], count = d.xch_ptr[d.xch_rank + 1] - first;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)count * per) return;
  const int idx = (int)(t / per), rem = (int)(t - (int64_t)idx * per), a = rem / S, i = rem - a * S;
  send[t] = d.M[((size_t)d.xch_box[first + idx] * d.nslots + d.act[a]) * d.s_max + i];
}

__global__ void xch_unpack_kernel(DevicePlan d, const int P, const double2* __restrict__ recv) {
  const int S = P * (P + 1) / 2, per = d.n_act * S;
  const int total = d.xch_ptr[d.xch_world];
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)total * per) return;
  const int g = (int)(t / per), rem = (int)(t - (int64_t)g * per), a = rem / S, i = rem - a * S;
  int r = 0;
  while (g >= d.xch_ptr[r + 1]) ++r;
  if (r == d.xch_rank) return;                          // my own boxes are already in M
  const int idx = g - d.xch_ptr[r];
  d.M[((size_t)d.xch_box[g] * d.nslots + d.act[a]) * d.s_max + i] = recv[((size_t)r * d.xch_max + idx) * per + rem];
}

// selective exchange (shard_upward == 2): M of a list of boxes -> buffer, buffer -> M of a list of boxes; the buffer holds
// [position in the list][active slot][S(p)] complex, the lists the boxes of all peers one peer after the other
__global__ void xch_list_kernel(DevicePlan d, const int P, const int* __restrict__ boxes, int count, double2* __restrict__ buf, int to_buf) {
  const int S = P * (P + 1) / 2, per = d.n_act * S;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= (int64_t)count * per) return;
  const int idx = (int)(t / per), rem = (int)(t - (int64_t)idx * per), a = rem / S, i = rem - a * S;
  double2* m = d.M + ((size_t)boxes[idx] * d.nslots + d.act[a]) * d.s_max + i;
  if (to_buf) buf[t] = *m; else *m = buf[t];
}

hipError_t launch_xch_pack(const DevicePlan& d, int p, double2* send, hipStream_t s) {
  if (d.xsel_send_box) {
    const int64_t n = (int64_t)d.xsel_send_n * d.n_act * (p * (p + 1) / 2);
    if (n > 0) hipLaunchKernelGGL(xch_list_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, p, d.xsel_send_box, d.xsel_send_n, send, 1);
    return hipGetLastError();
  }
  const int64_t n = (int64_t)(d.xch_ptr[d.xch_rank + 1] - d.xch_ptr[d.xch_rank]) * d.n_act * (p * (p + 1) / 2);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(xch_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, p, send);
  return hipGetLastError();
}

hipError_t launch_xch_unpack(const DevicePlan& d, int p, const double2* recv, hipStream_t s) {
  if (d.xsel_recv_box) {
    const int64_t n = (int64_t)d.xsel_recv_n * d.n_act * (p * (p + 1) / 2);
    if (n > 0) hipLaunchKernelGGL(xch_list_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, p, d.xsel_recv_box, d.xsel_recv_n, const_cast<double2*>(recv), 0);
    return hipGetLastError();
  }
  const int64_t n = (int64_t)d.xch_ptr[d.xch_world] * d.n_act * (p * (p + 1) / 2);
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(xch_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d, p, recv);
  return hipGetLastError();
}

hipError_t launch_mh_prep(const DevicePlan& d, int p, hipStream_t s) {
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (d.n_mh <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mh_prep_kernel, dim3(d.n_mh, d.n_act), dim3(kWave), 0, s, d, p);
  return hipGetLastError();
}

hipError_t launch_l2l_level(const DevicePlan& d, const ShiftOpDev& op, int p, int first, int count, hipStream_t s) {
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (count <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  {
    const size_t S = (size_t)p * (p + 1) / 2, P2 = (size_t)p * p;
    const size_t lds = kL2LWaves * (S + P2 + op.V) * sizeof(double2);
    const int blocks = (count + kL2LWaves - 1) / kL2LWaves;
    hipLaunchKernelGGL(l2l_kernel, dim3(blocks < 1024 ? blocks : 1024, d.n_act), dim3(kL2LWaves * kWave), lds, s, d, op, p, first, count);
  }
  return hipGetLastError();
}

int l2p_group_leaves(int kernel) { return kernel == 1 ? kL2PLeavesStokes : kL2PLeaves; }   // 1 = FMMBEM_KERNEL_STOKES_BEM

hipError_t launch_l2p(const DevicePlan& d, int p, double* y, hipStream_t s) {
  constexpr bool store = false;                         // the kernels can also overwrite y (a far field on its own); no caller does
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (d.n_l2p <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  const int S = p * (p + 1) / 2;
  const size_t per_wave = sizeof(double2) * (size_t)kL2PLeaves * d.n_act * S;          // 7 KB at p = 10, 35 KB at p = 16 with both slots
  int nw = (int)((48 * 1024) / per_wave);
  nw = nw < 1 ? 1 : nw > kL2PWaves ? kL2PWaves : nw;
  const int nblk = (d.n_l2p_grp + nw - 1) / nw;
  const size_t lds = sizeof(double2) * (size_t)((3 * S + 1) / 2) + nw * per_wave;
  const dim3 grid(nblk < 256 * 8 ? nblk : 256 * 8), block(nw * kWave);
#define L2P_CASE(PP) case PP: hipLaunchKernelGGL(l2p_kernel<PP>, grid, block, lds, s, d, p, y, store ? 1 : 0); break;
  switch (p <= kL2PUnrollMax && !l2p_generic() ? p : 0) {
    L2P_CASE(1) L2P_CASE(2) L2P_CASE(3) L2P_CASE(4) L2P_CASE(5) L2P_CASE(6) L2P_CASE(7) L2P_CASE(8)
    L2P_CASE(9) L2P_CASE(10) L2P_CASE(11) L2P_CASE(12)
    default: hipLaunchKernelGGL(l2p_kernel<0>, grid, block, lds, s, d, p, y, store ? 1 : 0);
  }
#undef L2P_CASE
  return hipGetLastError();
}

}  // namespace fmmbem

namespace fmmbem {
hipError_t launch_p2m_stokes(const DevicePlan& d, int p, hipStream_t s) {
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (d.n_p2m <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  const int nb = (d.n_p2m + 3) / 4;
  const int nblk_r = (d.n_p2m + kP2MWaves - 1) / kP2MWaves;
  const dim3 gr(nblk_r < 256 * 8 ? nblk_r : 256 * 8), br(kP2MWaves * kWave);
  if (d.p2m_tab_g) hipLaunchKernelGGL((p2m_apply_kernel<3>), dim3(nb < 256 * 16 ? nb : 256 * 16), dim3(4 * kWave), 0, s, d, p);
  else if (d.stokes_traction_targets)                  // no gradient records (FMMBEM_P2M_TABLE=0, or more than 16 GB of them): the recurrences, slot by slot
    for (int e = 0; e < 7; ++e) hipLaunchKernelGGL((p2m_kernel<1>), gr, br, 0, s, d, p, 5 + e, 4 + e);
  if (!d.stokes_velocity_targets) return hipGetLastError();
  if (d.p2m_tab) {
    hipLaunchKernelGGL((p2m_apply_kernel<4>), dim3(nb < 256 * 16 ? nb : 256 * 16), dim3(4 * kWave), 0, s, d, p);
    return hipGetLastError();
  }
  const int nblk = (d.n_p2m + kP2MWaves - 1) / kP2MWaves;
  const dim3 g(nblk < 256 * 8 ? nblk : 256 * 8), b(kP2MWaves * kWave);
  for (int e = 0; e < 4; ++e) hipLaunchKernelGGL((p2m_kernel<0>), g, b, 0, s, d, p, e + 1, e);
  return hipGetLastError();
}

hipError_t launch_l2p_stokes(const DevicePlan& d, int p, double* y, hipStream_t s) {
  constexpr bool store = false;
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (d.n_l2p <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  const int S = p * (p + 1) / 2;
  for (int trac = 0; trac < 2; ++trac) {
    if (!(trac ? d.stokes_traction_targets : d.stokes_velocity_targets)) continue;
    const size_t per_wave = sizeof(double2) * (size_t)kL2PLeavesStokes * (trac ? 7 : 4) * S;
    int nw = (int)((48 * 1024) / per_wave);
    nw = nw < 1 ? 1 : nw > kL2PWaves ? kL2PWaves : nw;
    const int nblk = (d.n_l2p_grp + nw - 1) / nw;
    const size_t lds = sizeof(double2) * (size_t)((3 * S + 1) / 2) + nw * per_wave;
    const dim3 grid(nblk < 256 * 8 ? nblk : 256 * 8), block(nw * kWave);
#define L2PS_CASE(PP) case PP: if (trac) hipLaunchKernelGGL((l2p_stokes_kernel<true, PP>), grid, block, lds, s, d, p, y, store ? 1 : 0); \
                               else hipLaunchKernelGGL((l2p_stokes_kernel<false, PP>), grid, block, lds, s, d, p, y, store ? 1 : 0); break;
    switch (p <= kL2PUnrollMax && !l2p_generic() ? p : 0) {
      L2PS_CASE(1) L2PS_CASE(2) L2PS_CASE(3) L2PS_CASE(4) L2PS_CASE(5) L2PS_CASE(6) L2PS_CASE(7) L2PS_CASE(8)
      L2PS_CASE(9) L2PS_CASE(10) L2PS_CASE(11) L2PS_CASE(12)
      default: L2PS_CASE(0)
    }
#undef L2PS_CASE
  }
  return hipGetLastError();
}
}  // namespace fmmbem
