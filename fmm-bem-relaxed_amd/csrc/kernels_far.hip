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
#include "device_plan.hpp"

namespace fmmbem {

namespace {

constexpr int kWave = 64;
constexpr int kPmaxDev = 16, kSmax = kPmaxDev * (kPmaxDev + 1) / 2;
constexpr double kEps = 1e-12;                         // kernel/LaplaceSpherical.hpp:30

// stored index -> (j,k), up to p = 16
struct JK { unsigned char j[136], k[136]; };
__constant__ JK kJK;
JK make_jk() {
  JK t;
  int i = 0;
  for (int j = 0; j < 16; ++j)
    for (int k = 0; k <= j; ++k, ++i) { t.j[i] = (unsigned char)j; t.k[i] = (unsigned char)k; }
  return t;
}

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
__device__ inline double2 cmul(double2 a, double2 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ inline double2 cconj(double2 a) { return {a.x, -a.y}; }
__device__ inline void cfma(double2& acc, double2 a, double2 b) {     // acc += a*b
  acc.x = fma(a.x, b.x, acc.x); acc.x = fma(-a.y, b.y, acc.x);
  acc.y = fma(a.x, b.y, acc.y); acc.y = fma(a.y, b.x, acc.y);
}
__device__ inline double2 mul_i_pow(double2 a, int q) {              // a * i^q
  switch (q & 3) {
    case 0: return a;
    case 1: return {-a.y, a.x};
    case 2: return {-a.x, -a.y};
    default: return {a.y, -a.x};
  }
}
__device__ inline int oddeven(int n) { return (n & 1) ? -1 : 1; }

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

// ---------------------------------------------------------------------------------------------
// P2M: one wavefront per source leaf and expansion slot; lane = panel, NQ quadrature points advanced
// in lock-step through the Legendre / rho^n / e^{im beta} recurrences of evalMultipole(rho,alpha,-beta)
// (kernel/LaplaceSpherical.hpp:455-488); each coefficient is reduced over the wavefront by shuffles.
// ---------------------------------------------------------------------------------------------
template <int NQ, int slot>
__global__ __launch_bounds__(kWave) void p2m_kernel(DevicePlan d, const int P) {
  const int S = P * (P + 1) / 2;
  __shared__ double2 acc[kSmax];
  const int lane = threadIdx.x;
  const int leaf = d.p2m_leaf[blockIdx.x];
  const int box = d.leaf_box[leaf];
  const int row0 = d.leaf_row0[leaf], nrows = d.leaf_nrows[leaf];
  const double c0 = d.box_center[3 * box], c1 = d.box_center[3 * box + 1], c2 = d.box_center[3 * box + 2];
  for (int i = lane; i < S; i += kWave) acc[i] = {0, 0};
  const int64_t N = d.n;

  for (int chunk = 0; chunk < nrows; chunk += kWave) {
    const int64_t i = row0 + chunk + lane;
    const bool live = (chunk + lane < nrows) && (d.bc[i] == slot);
    const double wpanel = live ? d.xt[i] * d.area[i] : 0.0;
    const double n0 = live ? d.nx[i] : 0, n1 = live ? d.ny[i] : 0, n2 = live ? d.nz[i] : 0;
    for (int q0 = 0; q0 < d.nq; q0 += NQ) {
      Sph s[NQ];
      double wq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bool on = live && (q0 + q < d.nq);
        const int qq = on ? q0 + q : 0;
        const double dx = on ? d.quad[(qq * 3 + 0) * N + i] - c0 : 0.3;
        const double dy = on ? d.quad[(qq * 3 + 1) * N + i] - c1 : 0.4;
        const double dz = on ? d.quad[(qq * 3 + 2) * N + i] - c2 : 0.5;
        s[q] = cart2sph(dx, dy, dz);
        wq[q] = on ? wpanel * d.qw[qq] : 0.0;
      }
      double pn[NQ], rhom[NQ], er[NQ], ei[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) { pn[q] = 1; rhom[q] = 1; er[q] = 1; ei[q] = 0; }
      double fact = 1;
#pragma nounroll
      for (int m = 0; m < P; ++m) {
        double p[NQ], p1[NQ], rhon[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { p[q] = pn[q]; p1[q] = p[q]; rhon[q] = rhom[q]; }
#pragma nounroll
        for (int n = m; n < P; ++n) {
          const double pref = d.tabPref[n * n + n + m];
          double vr = 0, vi = 0;
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            // Ynm[n,m] at (rho, alpha, -beta): rho^n P_n^m(cos a) pref e^{-i m beta}
            const double mag = rhon[q] * p[q] * pref;
            const double yr = mag * er[q], yi = -mag * ei[q];
            // advance the Legendre recurrence (p -> P_{n+1}^m), keeping P_n^m in pcur
            const double pcur = p[q];
            double pnext;
            if (n == m) pnext = s[q].ca * (2 * m + 1) * pcur;
            else pnext = (s[q].ca * (2 * n + 1) * pcur - (n + m) * p1[q]) / (double)(n - m + 1);
            if (slot == 0) {                          // source BC POTENTIAL: G moments (:326)
              vr = fma(wq[q], yr, vr); vi = fma(wq[q], yi, vi);
            } else {                                  // NORMAL_DERIV: (n . grad)(rho^n Ynm) (:331-343)
              double tmag;                            // YnmTheta magnitude
              if (n == m) tmag = rhon[q] * (pnext - (m + 1) * s[q].ca * pcur) / s[q].sa * pref;
              else tmag = rhon[q] * ((n - m + 1) * pnext - (n + 1) * s[q].ca * pcur) / s[q].sa * pref;
              const double tr = tmag * er[q], ti = -tmag * ei[q];
              const double rho = s[q].rho, sa = s[q].sa, ca = s[q].ca, cb = s[q].cb, sb = s[q].sb;
              const double brr = (double)n / rho * yr, bri = (double)n / rho * yi;       // d/d rho
              const double ber = (double)m * yi, bei = -(double)m * yr;                  // -i m Ynm
              const double gxr = sa * cb * brr + ca * cb / rho * tr - sb / rho / sa * ber;
              const double gxi = sa * cb * bri + ca * cb / rho * ti - sb / rho / sa * bei;
              const double gyr = sa * sb * brr + ca * sb / rho * tr + cb / rho / sa * ber;
              const double gyi = sa * sb * bri + ca * sb / rho * ti + cb / rho / sa * bei;
              const double gzr = ca * brr - sa / rho * tr;
              const double gzi = ca * bri - sa / rho * ti;
              vr += wq[q] * (n0 * gxr + n1 * gyr + n2 * gzr);
              vi += wq[q] * (n0 * gxi + n1 * gyi + n2 * gzi);
            }
            p1[q] = pcur; p[q] = pnext;
            rhon[q] *= s[q].rho;
          }
          vr = wave_sum(vr);
          vi = wave_sum(vi);
          if (lane == 0) { double2& a = acc[n * (n + 1) / 2 + m]; a.x += vr; a.y += vi; }
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
  __syncthreads();
  double2* M = d.M + ((size_t)box * 2 + slot) * d.s_max;
  for (int i = lane; i < S; i += kWave) M[i] = acc[i];
}

// ---------------------------------------------------------------------------------------------
// M2M, one tree level per launch: one wavefront per parent box and slot, children in turn.
// The regular harmonics of the parent-child translation come from a per-class table.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void m2m_kernel(DevicePlan d, const int P, int first) {
  const int S = P * (P + 1) / 2, P2 = P * P;
  constexpr int SLOTS = (kSmax + kWave - 1) / kWave;
  __shared__ double2 Ms[kSmax];
  __shared__ double2 Y[kPmaxDev * kPmaxDev];
  const int lane = threadIdx.x;
  const int parent = d.m2m_parent[first + blockIdx.x];
  const int slot = d.act[blockIdx.y];
  const double* A = d.tabA;
  const double* iA = d.tabInvA;
  double2 acc[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) acc[s] = {0, 0};
  for (int c = d.box_child_begin[parent]; c < d.box_child_end[parent]; ++c) {
    __syncthreads();
    const double2* src = d.M + ((size_t)c * 2 + slot) * d.s_max;
    const double2* tab = d.up_tab + (size_t)d.up_cls[c] * d.p2_max;
    for (int i = lane; i < S; i += kWave) Ms[i] = src[i];
    for (int i = lane; i < P2; i += kWave) Y[i] = tab[i];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int idx = lane + s * kWave;
      if (idx >= S) continue;
      const int j = kJK.j[idx], k = kJK.k[idx];
      const int jk = j * j + j + k;
      double2 sum = {0, 0};
      for (int n = 0; n <= j; ++n) {
        const int mhi = (k - 1 < n) ? k - 1 : n;
        for (int m = -n; m <= mhi; ++m) {
          if (j - n >= k - m) {
            const int jnkm = (j - n) * (j - n) + j - n + k - m;
            const int jnkms = (j - n) * (j - n + 1) / 2 + k - m;
            const int nm = n * n + n + m;
            const double ph = (m < 0 && (m & 1)) ? -1.0 : 1.0;               // i^{m-|m|}
            const double coef = ph * oddeven(n) * A[nm] * A[jnkm] * iA[jk];
            const double2 t = cmul(Ms[jnkms], Y[nm]);
            sum.x = fma(t.x, coef, sum.x); sum.y = fma(t.y, coef, sum.y);
          }
        }
        for (int m = k; m <= n; ++m) {
          if (j - n >= m - k) {
            const int jnkm = (j - n) * (j - n) + j - n + k - m;
            const int jnkms = (j - n) * (j - n + 1) / 2 - k + m;
            const int nm = n * n + n + m;
            const double coef = oddeven(k + n + m) * A[nm] * A[jnkm] * iA[jk];
            const double2 t = cmul(cconj(Ms[jnkms]), Y[nm]);
            sum.x = fma(t.x, coef, sum.x); sum.y = fma(t.y, coef, sum.y);
          }
        }
      }
      acc[s].x = fma(sum.x, kEps, acc[s].x);
      acc[s].y = fma(sum.y, kEps, acc[s].y);
    }
  }
  double2* dst = d.M + ((size_t)parent * 2 + slot) * d.s_max;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int idx = lane + s * kWave;
    if (idx < S) dst[idx] = acc[s];
  }
}

// ---------------------------------------------------------------------------------------------
// mh_prep: Mh[n,m] = i^{-m} A[n,m] M[n,m] for the stored orders m >= 0 of every M2L source box
// (negative orders follow from Mh[n,-m] = (-1)^m conj(Mh[n,m]) inside the M2L kernel).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void mh_prep_kernel(DevicePlan d, const int P) {
  const int S = P * (P + 1) / 2;
  const int box = d.mh_box[blockIdx.x];
  const int slot = d.act[blockIdx.y];
  const double2* M = d.M + ((size_t)box * 2 + slot) * d.s_max;
  double2* Mh = d.Mh + ((size_t)box * 2 + slot) * d.s_max;
  for (int idx = threadIdx.x; idx < S; idx += kWave) {
    const int n = kJK.j[idx], m = kJK.k[idx];
    const double2 v = M[idx];
    const double a = d.tabA[n * n + n + m];
    Mh[idx] = mul_i_pow(double2{v.x * a, v.y * a}, -m);
  }
}

// ---------------------------------------------------------------------------------------------
// L2L, one tree level per launch: one wavefront per child box and slot: L[child] += shift(L[parent]).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void l2l_kernel(DevicePlan d, const int P, int first) {
  const int S = P * (P + 1) / 2, P2 = P * P;
  constexpr int SLOTS = (kSmax + kWave - 1) / kWave;
  __shared__ double2 Ls[kSmax];
  __shared__ double2 Y[kPmaxDev * kPmaxDev];
  const int lane = threadIdx.x;
  const int child = d.l2l_child[first + blockIdx.x];
  const int parent = d.box_parent[child];
  const int slot = d.act[blockIdx.y];
  const double* A = d.tabA;
  const double* iA = d.tabInvA;
  const double2* src = d.L + ((size_t)parent * 2 + slot) * d.s_max;
  const double2* tab = d.down_tab + (size_t)d.down_cls[child] * d.p2_max;
  for (int i = lane; i < S; i += kWave) Ls[i] = src[i];
  for (int i = lane; i < P2; i += kWave) Y[i] = tab[i];
  __syncthreads();
  double2* dst = d.L + ((size_t)child * 2 + slot) * d.s_max;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int idx = lane + s * kWave;
    if (idx >= S) continue;
    const int j = kJK.j[idx], k = kJK.k[idx];
    const int jk = j * j + j + k;
    double2 sum = {0, 0};
    for (int n = j; n < P; ++n) {
      for (int m = j + k - n; m < 0; ++m) {
        const int jnkm = (n - j) * (n - j) + n - j + m - k;
        const int nm = n * n + n - m;
        const int nms = n * (n + 1) / 2 - m;
        const double coef = oddeven(k) * A[jnkm] * A[jk] * iA[nm];
        const double2 t = cmul(cconj(Ls[nms]), Y[jnkm]);
        sum.x = fma(t.x, coef, sum.x); sum.y = fma(t.y, coef, sum.y);
      }
      for (int m = 0; m <= n; ++m) {
        const int dmk = m - k, admk = dmk < 0 ? -dmk : dmk;
        if (n - j >= admk) {
          const int jnkm = (n - j) * (n - j) + n - j + m - k;
          const int nm = n * n + n + m;
          const int nms = n * (n + 1) / 2 + m;
          const double ph = (dmk < 0 && (dmk & 1)) ? -1.0 : 1.0;             // i^{m-k-|m-k|}
          const double coef = ph * A[jnkm] * A[jk] * iA[nm];
          const double2 t = cmul(Ls[nms], Y[jnkm]);
          sum.x = fma(t.x, coef, sum.x); sum.y = fma(t.y, coef, sum.y);
        }
      }
    }
    double2 cur = dst[idx];
    cur.x = fma(sum.x, kEps, cur.x); cur.y = fma(sum.y, kEps, cur.y);
    dst[idx] = cur;
  }
}

// ---------------------------------------------------------------------------------------------
// L2P: one wavefront per target leaf; lane = panel centroid; the leaf's L (active slots) in LDS.
// y[perm[i]] += r0 (POTENTIAL target) or -= r1 (NORMAL_DERIV target).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void l2p_kernel(DevicePlan d, const int P, double* __restrict__ y) {
  const int S = P * (P + 1) / 2;
  __shared__ double2 Ls[2][kSmax];
  const int lane = threadIdx.x;
  const int leaf = d.l2p_leaf[blockIdx.x];
  const int box = d.leaf_box[leaf];
  const int row0 = d.leaf_row0[leaf], nrows = d.leaf_nrows[leaf];
  for (int s = 0; s < 2; ++s) {
    const double2* src = d.L + ((size_t)box * 2 + s) * d.s_max;
    for (int i = lane; i < S; i += kWave) Ls[s][i] = src[i];
  }
  __syncthreads();
  const double c0 = d.box_center[3 * box], c1 = d.box_center[3 * box + 1], c2 = d.box_center[3 * box + 2];
  for (int chunk = 0; chunk < nrows; chunk += kWave) {
    if (chunk + lane >= nrows) break;
    const int64_t i = row0 + chunk + lane;
    const int tb = d.bc[i] ? 1 : 0;
    const double2* Lt = Ls[tb];
    const Sph s = cart2sph(d.cx[i] - c0, d.cy[i] - c1, d.cz[i] - c2);
    double r = 0;
    double pn = 1, rhom = 1, er = 1, ei = 0, fact = 1;
#pragma nounroll
    for (int m = 0; m < P; ++m) {
      double p = pn, p1 = p, rhon = rhom;
      const double w = m == 0 ? 1.0 : 2.0;
#pragma nounroll
      for (int n = m; n < P; ++n) {
        const double mag = rhon * p * d.tabPref[n * n + n + m];
        const double2 Lc = Lt[n * (n + 1) / 2 + m];
        r += w * (Lc.x * (mag * er) - Lc.y * (mag * ei));        // Re(L * Ynm), Ynm = mag e^{+i m beta}
        const double pcur = p;
        if (n == m) p = s.ca * (2 * m + 1) * pcur;
        else p = (s.ca * (2 * n + 1) * pcur - (n + m) * p1) / (double)(n - m + 1);
        p1 = pcur;
        rhon *= s.rho;
      }
      pn = -pn * fact * s.sa;
      fact += 2;
      rhom *= s.rho;
      const double nr = er * s.cb - ei * s.sb, ni = er * s.sb + ei * s.cb;
      er = nr; ei = ni;
    }
    const uint32_t o = d.perm[i];
    y[o] += tb ? -r : r;
  }
}

hipError_t upload_constants_once() {
  static hipError_t st = [] {
    const JK t = make_jk();
    return hipMemcpyToSymbol(HIP_SYMBOL(kJK), &t, sizeof(t));
  }();
  return st;
}

}  // namespace

hipError_t launch_p2m(const DevicePlan& d, int p, hipStream_t s) {
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (d.n_p2m <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  const dim3 g(d.n_p2m), b(kWave);
  for (int a = 0; a < d.n_act; ++a) {
    const int slot = d.act[a];
    if (d.nq == 3) {
      if (slot == 0) hipLaunchKernelGGL((p2m_kernel<3, 0>), g, b, 0, s, d, p);
      else hipLaunchKernelGGL((p2m_kernel<3, 1>), g, b, 0, s, d, p);
    } else if (d.nq == 4) {
      if (slot == 0) hipLaunchKernelGGL((p2m_kernel<4, 0>), g, b, 0, s, d, p);
      else hipLaunchKernelGGL((p2m_kernel<4, 1>), g, b, 0, s, d, p);
    } else {
      if (slot == 0) hipLaunchKernelGGL((p2m_kernel<1, 0>), g, b, 0, s, d, p);
      else hipLaunchKernelGGL((p2m_kernel<1, 1>), g, b, 0, s, d, p);
    }
  }
  return hipGetLastError();
}

hipError_t launch_m2m_level(const DevicePlan& d, int p, int first, int count, hipStream_t s) {
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (count <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  hipLaunchKernelGGL(m2m_kernel, dim3(count, d.n_act), dim3(kWave), 0, s, d, p, first);
  return hipGetLastError();
}

hipError_t launch_mh_prep(const DevicePlan& d, int p, hipStream_t s) {
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (d.n_mh <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mh_prep_kernel, dim3(d.n_mh, d.n_act), dim3(kWave), 0, s, d, p);
  return hipGetLastError();
}

hipError_t launch_l2l_level(const DevicePlan& d, int p, int first, int count, hipStream_t s) {
  if (hipError_t e = upload_constants_once(); e != hipSuccess) return e;
  if (count <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  hipLaunchKernelGGL(l2l_kernel, dim3(count, d.n_act), dim3(kWave), 0, s, d, p, first);
  return hipGetLastError();
}

hipError_t launch_l2p(const DevicePlan& d, int p, double* y, hipStream_t s) {
  if (d.n_l2p <= 0) return hipSuccess;
  if (p < 1 || p > kPmaxDev) return hipErrorInvalidValue;
  hipLaunchKernelGGL(l2p_kernel, dim3(d.n_l2p), dim3(kWave), 0, s, d, p, y);
  return hipGetLastError();
}

}  // namespace fmmbem
