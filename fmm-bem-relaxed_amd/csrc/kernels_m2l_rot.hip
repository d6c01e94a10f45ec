// kernels_m2l_rot.hip -- M2L as rotation / axial translation / rotation back, one (target, source) PAIR per lane.
// Reference: LaplaceSpherical::M2L + evalLocal (kernel/LaplaceSpherical.hpp:296-329, 491-524), called once per LR_list pair
// and expansion (executor/EvalInteractionLazySparse.hpp:269-283).  The algebra and the constant tables are in m2l_rot.hpp.
//
// Mapping.  The M2L pairs are kept in CSR order by target (traversal order inside a target, as the oracle sums them).  The
// host cuts that list into ITEMS: runs of whole targets with at most 64 pairs, or one target with more.  A wavefront takes
// an item 64 pairs at a time: lane = pair.  The lane loads its source's multipole (S complex, 16-byte loads straight from
// M -- no rescaled copy, no mh_prep pass) and the five numbers of its translation class (1/rho, cos/sin alpha, cos/sin
// beta), and then runs the SAME straight-line instruction stream as every other lane on its own registers:
//     z-rotation by beta, fixed rotation, z-rotation by alpha, fixed rotation back, scale by rho^-n,
//     axial translation, scale by rho^-(j+1), fixed rotation, z-rotation by -alpha, fixed rotation back, z-rotation by -beta
// The rotation and translation constants are wave-uniform: they stream through the scalar cache into SGPR operands of
// v_fma_f64.  No LDS in the arithmetic, no barriers, no divergence; ~3 000 FMAs per pair at p = 10 (15 400 for the
// reference's double sum, 280 x 55 in kernels_m2l.hip).
// Reduction.  The lanes of one target are then added in pair order (four interleaved partial sums per coefficient,
// combined in a fixed order: deterministic, and shards of one operator produce the same bits): eight coefficients at a time
// go through a padded LDS tile [coefficient][lane]; a target spread over several passes accumulates in LDS.  Every L is
// written exactly once -- no atomics.
#include "device_plan.hpp"
#include "m2l_rot.hpp"

namespace fmmbem {

namespace {

constexpr int kWave = 64;
constexpr int kTile = 8;                              // coefficients per reduction tile
constexpr int kChains = 4;                            // partial sums per (target, coefficient)
#ifndef FMMBEM_ROT_BATCH
#define FMMBEM_ROT_BATCH 20
#endif
constexpr int kBatch = FMMBEM_ROT_BATCH;              // constants between two scheduling fences (2 SGPRs each)
typedef __attribute__((address_space(4))) const double ConstD;      // wave-uniform constants: scalar loads

#ifndef FMMBEM_ROT_XCD_CHUNK
#define FMMBEM_ROT_XCD_CHUNK 32
#endif

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int idx_of(int n, int m) { return n * (n + 1) / 2 + m; }

// The constants are loop-invariant: left alone, the compiler hoists all 700 scalar loads of a pass out of the pass loop and
// spills them to VGPR lanes (5 000 v_readlane per pass at p = 10).  Forgetting the pointer's provenance at the head of
// every block keeps each block's loads next to their uses.
__device__ __forceinline__ ConstD* opaque(ConstD* p) {
  uintptr_t v = reinterpret_cast<uintptr_t>(p);
  asm volatile("" : "+s"(v));
  return reinterpret_cast<ConstD*>(v);
}

// out = R in for the degree-n block, R = conj(X) (BACK = false) or X^T (BACK = true); coefficients at tab[rot_off(n) ...]
template <int P, bool BACK>
__device__ __forceinline__ void fixed_rotation(double (&a)[P * (P + 1) / 2], double (&b)[P * (P + 1) / 2], ConstD* tab_in) {
#pragma unroll
  for (int n = 1; n < P; ++n) {                       // degree 0 is the identity
    double na[P], nb[P];
    ConstD* tab = opaque(tab_in);
    int ci = rot_off(n), fence_at = ci;
#pragma unroll
    for (int m = 0; m <= n; ++m) {
      if (ci - fence_at >= kBatch) { __builtin_amdgcn_sched_barrier(0); fence_at = ci; }
      double sa = 0, sb = 0;
      bool ia = false, ib = false;                    // the first product initialises the sum
#pragma unroll
      for (int mp = 0; mp <= n; ++mp) {
        if (!rot_live(n, m, mp)) continue;
        const bool neg_back = BACK && ((m + mp) & 1);
        const bool even = ((n + m) & 1) == 0;
        const double src = (mp == 0 || even) ? a[idx_of(n, mp)] : b[idx_of(n, mp)];
        const int kk = ((mp == 0 ? m : m + 3 * mp) + ((mp != 0 && !even) ? 1 : 0)) & 3;   // i^kk: 0 +Re, 1 +Im, 2 -Re, 3 -Im
        const double c = ((kk >= 2) != neg_back) ? -tab[ci] : tab[ci];
        ++ci;
        if ((kk & 1) == 0) { sa = ia ? fma(c, src, sa) : c * src; ia = true; }
        else { sb = ib ? fma(c, src, sb) : c * src; ib = true; }
      }
      na[m] = sa; nb[m] = sb;
    }
#pragma unroll
    for (int m = 0; m <= n; ++m) { a[idx_of(n, m)] = na[m]; b[idx_of(n, m)] = nb[m]; }
  }
}

// v[n,m] *= e^{i m g} for every degree, (c1, s1) = (cos g, sin g)
template <int P>
__device__ __forceinline__ void z_rotation(double (&a)[P * (P + 1) / 2], double (&b)[P * (P + 1) / 2], double c1, double s1) {
  double cm = c1, sm = s1;
#pragma unroll
  for (int m = 1; m < P; ++m) {
#pragma unroll
    for (int n = m; n < P; ++n) {
      const double x = a[idx_of(n, m)], y = b[idx_of(n, m)];
      a[idx_of(n, m)] = fma(x, cm, -(y * sm));
      b[idx_of(n, m)] = fma(x, sm, y * cm);
    }
    const double c2 = fma(cm, c1, -(sm * s1)), s2 = fma(sm, c1, cm * s1);
    cm = c2; sm = s2;
  }
}

// Registers: a pair's data alone is 2 S doubles = 4 S VGPRs (112 at p = 7, 220 at p = 10), plus a degree block of
// temporaries.  Wavefronts per SIMD asked of the compiler, measured at N = 1M (M2L ms, Laplace): p = 7: two 0.41, one 0.50;
// p = 8: two 0.72, one 0.70; p = 9..12: two 1.00 / 1.51 / 3.28 / 5.59 (1-2 KB of scratch per lane), one 0.96 / 1.29 / 2.07 /
// 2.73 (the overflow goes to AGPRs: no scratch up to p = 10).
#ifndef FMMBEM_ROT_OCC
#define FMMBEM_ROT_OCC(P) ((P) <= 4 ? 4 : (P) <= 6 ? 3 : (P) <= 7 ? 2 : 1)
#endif
constexpr int rot_waves(int P) { return FMMBEM_ROT_OCC(P); }

template <int P>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(rot_waves(P), rot_waves(P)))) void m2l_rot_kernel(const DevicePlan d, const double* __restrict__ tab_g) {
  constexpr int S = P * (P + 1) / 2;
  constexpr int NT = (S + kTile - 1) / kTile;
  __shared__ double2 tile[kTile][kWave + 1];          // +1: the rows of one column sit in different banks
  __shared__ double2 carry[S];                        // a target spread over several passes
  __shared__ int seg_first[kWave + 1], seg_tgt[kWave];
  ConstD* tab = reinterpret_cast<ConstD*>(reinterpret_cast<uintptr_t>(tab_g));
  const int lane = threadIdx.x;
  // workgroups are dealt round-robin to the 8 XCDs: keep runs of consecutive items (neighbouring targets, which share
  // sources) on one XCD's L2.  gridDim.x is a multiple of 8 * CH.
  constexpr int CH = FMMBEM_ROT_XCD_CHUNK;
  const int rnd = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
  const int item = (rnd / CH) * 8 * CH + xcd * CH + rnd % CH;
  if (item >= d.n_rot_items) return;
  const int ib = d.rot_item_ptr[item], ie = d.rot_item_ptr[item + 1];
  const bool multi = ie - ib > kWave;                 // one target, several passes

  for (int q = 0; q < d.n_act; ++q) {
    const int slot = d.act[q];
    for (int pb = ib; pb < ie; pb += kWave) {
      const int cnt = ie - pb < kWave ? ie - pb : kWave;
      const bool live = lane < cnt;
      const int pi = live ? pb + lane : ie - 1;
      const int src = d.rot_src[pi], cls = d.rot_cls[pi], tgt = d.rot_tgt[pi];
      // ---- segments (targets) of this pass ----
      const int prev = __shfl_up(tgt, 1, kWave);
      const bool start = live && (lane == 0 || prev != tgt);
      const unsigned long long smask = __ballot(start);
      const int nseg = __popcll(smask);
      const int segid = __popcll(smask & ((2ull << lane) - 1)) - 1;
      if (start) { seg_first[segid] = lane; seg_tgt[segid] = tgt; }
      if (lane == 0) seg_first[nseg] = cnt;

      // ---- this lane's pair ----
      double a[S], b[S];
      {
        const double2* M = d.M + ((size_t)src * d.nslots + slot) * d.s_max;
#pragma unroll
        for (int i = 0; i < S; ++i) { const double2 v = M[i]; a[i] = v.x; b[i] = v.y; }
      }
      const double* cr = d.rot_cls_rec + (size_t)cls * 8;
      const double inv_rho = cr[0], ca = cr[1], sa = cr[2], cb = cr[3], sb = cr[4];
      z_rotation<P>(a, b, cb, sb);
      fixed_rotation<P, false>(a, b, tab);
      z_rotation<P>(a, b, ca, sa);
      fixed_rotation<P, true>(a, b, tab);
      {                                               // M''[n,m] = rho^-n M'[n,m]
        double r = inv_rho;
#pragma unroll
        for (int n = 1; n < P; ++n) {
#pragma unroll
          for (int m = 0; m <= n; ++m) { a[idx_of(n, m)] *= r; if (m) b[idx_of(n, m)] *= r; }
          r *= inv_rho;
        }
      }
      // axial translation, order by order: L'[j,k] = sum_{n>=k} Tz[j,n,k] M''[n,k]
#pragma unroll
      for (int k = 0; k < P; ++k) {
        double la[P], lb[P];
        ConstD* tabk = opaque(tab);
        int ti = tz_off(P, k), fence_at = ti;
#pragma unroll
        for (int j = k; j < P; ++j) {
          if (ti - fence_at >= kBatch) { __builtin_amdgcn_sched_barrier(0); fence_at = ti; }
          double s1 = 0, s2 = 0;
#pragma unroll
          for (int n = k; n < P; ++n) {
            const double t = tabk[ti++];
            s1 = n == k ? t * a[idx_of(n, k)] : fma(t, a[idx_of(n, k)], s1);
            if (k) s2 = n == k ? t * b[idx_of(n, k)] : fma(t, b[idx_of(n, k)], s2);
          }
          la[j] = s1; lb[j] = s2;
        }
#pragma unroll
        for (int j = k; j < P; ++j) { a[idx_of(j, k)] = la[j]; b[idx_of(j, k)] = lb[j]; }
      }
      {                                               // L'[j,k] *= rho^-(j+1)
        double r = inv_rho;
#pragma unroll
        for (int j = 0; j < P; ++j) {
#pragma unroll
          for (int k = 0; k <= j; ++k) { a[idx_of(j, k)] *= r; if (k) b[idx_of(j, k)] *= r; }
          r *= inv_rho;
        }
      }
      fixed_rotation<P, false>(a, b, tab);
      z_rotation<P>(a, b, ca, -sa);
      fixed_rotation<P, true>(a, b, tab);
      z_rotation<P>(a, b, cb, -sb);

      // ---- add the lanes of each target, pair order, kChains interleaved partial sums ----
      wave_sync();                                    // segment tables written
      const bool first_pass = pb == ib, last_pass = pb + kWave >= ie;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int c = 0; c < kTile; ++c) {
          const int i = t * kTile + c;
          if (i < S) tile[c][lane] = double2{a[i], b[i]};     // b[n,0] = 0 by construction
        }
        wave_sync();
        for (int task = lane; task < nseg * kTile * kChains; task += kWave) {
          const int h = task & (kChains - 1), c = (task / kChains) & (kTile - 1), s = task / (kChains * kTile);
          const int i = t * kTile + c;
          const int f = seg_first[s], e = seg_first[s + 1];
          double2 sum = {0, 0};
          // at most 64 / kChains terms per chain: all loads issued, then added in order (a lane past the end adds zero)
          double2 v[kWave / kChains];
#pragma unroll
          for (int u = 0; u < kWave / kChains; ++u) {
            const int l = f + h + u * kChains;
            v[u] = l < e ? tile[c][l] : double2{0, 0};
          }
#pragma unroll
          for (int u = 0; u < kWave / kChains; ++u) { sum.x += v[u].x; sum.y += v[u].y; }
          // chains 0..3 of one (target, coefficient) sit in four consecutive lanes: (0 + 1) + (2 + 3)
          sum.x += __shfl_xor(sum.x, 1, kWave); sum.y += __shfl_xor(sum.y, 1, kWave);
          sum.x += __shfl_xor(sum.x, 2, kWave); sum.y += __shfl_xor(sum.y, 2, kWave);
          if (h == 0 && i < S) {
            double2* L = d.L + ((size_t)seg_tgt[s] * d.nslots + slot) * d.s_max;
            if (!multi) L[i] = sum;
            else {
              if (!first_pass) { const double2 old = carry[i]; sum.x += old.x; sum.y += old.y; }
              if (last_pass) L[i] = sum; else carry[i] = sum;
            }
          }
        }
        wave_sync();
      }
    }
  }
}

// boxes that hold a local expansion but have no M2L source of their own (they only inherit from the parent): L = 0
__global__ void m2l_rot_zero_kernel(const DevicePlan d, int S) {
  const int box = d.rot_empty[blockIdx.x];
  for (int q = 0; q < d.n_act; ++q) {
    double2* L = d.L + ((size_t)box * d.nslots + d.act[q]) * d.s_max;
    for (int i = threadIdx.x; i < S; i += blockDim.x) L[i] = double2{0, 0};
  }
}

}  // namespace

bool m2l_rot_supported(int p) { return p >= 1 && p <= kRotPmax; }

hipError_t launch_m2l_rot(const DevicePlan& d, const DevicePlan* d_dev, int p, hipStream_t s) {
  (void)d_dev;
  if (d.n_rot_empty > 0) hipLaunchKernelGGL(m2l_rot_zero_kernel, dim3(d.n_rot_empty), dim3(kWave), 0, s, d, p * (p + 1) / 2);
  if (d.n_rot_items <= 0) return hipGetLastError();
  constexpr int CH = FMMBEM_ROT_XCD_CHUNK;
  const int grid = (d.n_rot_items + 8 * CH - 1) / (8 * CH) * (8 * CH);
  const double* tab = d.rot_tab + d.rot_tab_off[p - 1];
#define ROT_CASE(PP) case PP: hipLaunchKernelGGL((m2l_rot_kernel<PP>), dim3(grid), dim3(kWave), 0, s, d, tab); break;
  switch (p) {
    ROT_CASE(1) ROT_CASE(2) ROT_CASE(3) ROT_CASE(4) ROT_CASE(5) ROT_CASE(6)
    ROT_CASE(7) ROT_CASE(8) ROT_CASE(9) ROT_CASE(10) ROT_CASE(11) ROT_CASE(12)
    default: return hipErrorInvalidValue;
  }
#undef ROT_CASE
  return hipGetLastError();
}

}  // namespace fmmbem
