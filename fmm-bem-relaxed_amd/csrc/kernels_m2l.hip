// kernels_m2l.hip -- the M2L translation, the FP64-FMA-bound hot kernel of the far field.
// Reference: LaplaceSpherical::M2L + evalLocal (kernel/LaplaceSpherical.hpp:296-329, 491-524), called
// once per LR_list pair and expansion (executor/EvalInteractionLazySparse.hpp:269-283).
//
// Algebra (kernels_far.hip header): L[j,k] += i^{-k} (-1)^j A[j,k] * sum_{n<P,|m|<=n} Mh[n,m] * Yh[j+n, m-k]
// with Mh the rescaled source multipole (mh_prep) and Yh the rescaled singular harmonics of the translation
// vector, tabulated per translation class at plan build.
//
// Mapping: one wavefront per target box (a TEAM of 2-3 wavefronts when P(P+1)/2 > 64), one output (j,k) per
// lane, L accumulated in registers over the target's whole source list (CSR by target) and written once:
// no atomics, fixed summation order.  Per source:
//   * the class table ((2P)^2 complex, L2-resident) is PREFETCHED into registers while the previous source is
//     being computed, then scattered into the wavefront's LDS image with a constant row stride chosen so the
//     per-lane ds_read_b128 of the inner loop are bank-conflict free (m2l_layout.hpp);
//   * Mh[n,m] is wave-uniform and comes through the scalar cache (s_load_dwordx16) straight into the SGPR
//     operand of v_fma_f64, so the LDS pipe only carries the Yh gather: 1 ds_read_b128 per 4 FP64 FMAs.
// One instantiation per p = 1..16: the solver's per-iteration relaxation of p only picks among them.
#include "device_plan.hpp"
#include "m2l_layout.hpp"

#include <type_traits>

namespace fmmbem {

namespace {

constexpr int kWave = 64;
struct C2 { double x, y; };
typedef __attribute__((address_space(4))) C2 ConstC2;      // complex value in the constant address space
constexpr int kM2LTargets = 4;         // independent single-wavefront targets per workgroup when TEAM == 1

constexpr int isqrt_cut(int P, int i, int NS) {          // round(P * sqrt(i / NS)) without <cmath>
  int best = 0;
  for (int c = 0; c <= P; ++c) {
    const long a = (long)c * c * NS - (long)P * P * i, b = (long)best * best * NS - (long)P * P * i;
    if ((a < 0 ? -a : a) < (b < 0 ? -b : b)) best = c;
  }
  return best;
}

template <int P, int NS_> struct Shape {
  static constexpr int S = P * (P + 1) / 2, Y2 = 4 * P * P;
  static constexpr int TEAM = m2l_team(P);               // wavefronts needed to give every output a lane
  // The (n,m) terms of every output can additionally be split between NS wavefronts (contiguous n ranges
  // of about P^2/NS terms each) that share one LDS image: NS times the wavefronts per CU for the same
  // LDS, which is what hides the scalar-load latency of Mh.
  static constexpr int NS = NS_;
  static constexpr int cut(int i) { return i <= 0 ? 0 : (i >= NS ? P : isqrt_cut(P, i, NS)); }
  static constexpr int WAVES = TEAM * NS;
  static constexpr int TARGETS = WAVES == 1 ? kM2LTargets : 1;
  static constexpr int THREADS = WAVES * TARGETS * kWave;
  static constexpr int R = m2l_stride(P), C0 = m2l_col0(P), SLOTS = m2l_lds_slots(P);
  static constexpr int NLOAD = (Y2 + WAVES * kWave - 1) / (WAVES * kWave);   // table entries copied per lane
};

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

template <int P, int NS_>
__global__ __launch_bounds__((Shape<P, NS_>::THREADS)) void m2l_kernel(DevicePlan d) {
  using Sh = Shape<P, NS_>;
  constexpr int Y2 = Sh::Y2, TEAM = Sh::TEAM, TARGETS = Sh::TARGETS, R = Sh::R, NLOAD = Sh::NLOAD;
  constexpr int NS = Sh::NS, WAVES = Sh::WAVES;
  __shared__ double2 Yall[TARGETS][Sh::SLOTS];
  __shared__ double2 Comb[NS == 1 ? 1 : (NS - 1) * TEAM * kWave];   // partial sums of the other n-ranges
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int ti = blockIdx.x * TARGETS + (WAVES == 1 ? wave : 0);
  if (ti >= d.n_m2l_tgt) return;                       // WAVES==1: whole wavefront; else: whole workgroup
  const int tgt = d.m2l_tgt[ti];
  const int slot = d.act[blockIdx.y];
  double2* Yt = Yall[WAVES == 1 ? wave : 0];
  const int tid = WAVES == 1 ? lane : (int)threadIdx.x;  // index within the team (table copy)
  const int npart = WAVES == 1 ? 0 : wave % NS;          // which n-range this wavefront sums
  const int otid = WAVES == 1 ? lane : (wave / NS) * kWave + lane;   // index in the lane -> output map

  // this lane's output (conflict-free dealing, m2l_layout.hpp)
  const int packed = d.m2l_lane[(P - 1) * kM2LMaxThreads + otid];
  const bool valid = packed >= 0;
  const int j = valid ? (packed & 0xff) : 0, k = valid ? ((packed >> 8) & 0xff) : 0, idx = valid ? (packed >> 16) : 0;
  const double2* ybase = Yt + (j * R - k + Sh::C0);
  double2 acc = {0, 0};

  // This lane's share of the table copy: linear entries tid, tid + TEAM*64, ... -> LDS slots.  The staging
  // registers are NAMED scalars (macro-expanded, at most 8 per lane), not an array: hipcc keeps a 7 x 16-B
  // array that is live across the loop in scratch memory, which serialises the prefetch.
  static_assert(NLOAD <= 8, "table copy needs more staging registers");
  const int* scat = d.m2l_scat + d.m2l_scat_off[P - 1];
#define FMMBEM_REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define DECL(u)                                                                   \
  const int i##u = tid + u * WAVES * kWave;                                       \
  const int src##u = (u < NLOAD && i##u < Y2) ? i##u : 0;                         \
  const int dst##u = (u < NLOAD && i##u < Y2) ? scat[i##u] : -1;                  \
  double2 pre##u = {0, 0};
  FMMBEM_REP8(DECL)
#undef DECL
#define LOAD(u) if (u < NLOAD) pre##u = tab[src##u];
#define STORE(u) if (u < NLOAD && dst##u >= 0) Yt[dst##u] = pre##u;

  const int pb = d.m2l_ptr[tgt], pe = d.m2l_ptr[tgt + 1];
  if (pb == pe) {                                      // a box that only inherits from its parent: L = 0
    if (valid) d.L[((size_t)tgt * d.nslots + slot) * d.s_max + idx] = {0, 0};
    return;
  }
  {
    const int cls = __builtin_amdgcn_readfirstlane(d.m2l_cls[pb]);
    const double2* tab = d.m2l_tab + (size_t)cls * d.y2_max;
    FMMBEM_REP8(LOAD)
  }
  for (int pi = pb; pi < pe; ++pi) {
    const int src = __builtin_amdgcn_readfirstlane(d.m2l_src[pi]);
    const int pn = pi + 1 < pe ? pi + 1 : pi;          // last iteration re-reads its own table (harmless)
    const int cls_next = __builtin_amdgcn_readfirstlane(d.m2l_cls[pn]);
    if (WAVES == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();    // previous source's reads are done
    FMMBEM_REP8(STORE)
    if (WAVES == 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();
    }
    {                                                  // next table in flight during this source's FMAs
      const double2* tab = d.m2l_tab + (size_t)cls_next * d.y2_max;
      FMMBEM_REP8(LOAD)
    }
    // Mh (orders m >= 0 only; Mh[n,-m] = (-1)^m conj(Mh[n,m]) costs only sign modifiers on the FMAs) was
    // written by the preceding kernel and is immutable here: address it through the CONSTANT address space
    // so that the wave-uniform loads are always selected as scalar (SMEM) loads feeding SGPR operands.
    const ConstC2* mh = reinterpret_cast<const ConstC2*>(
        reinterpret_cast<uintptr_t>(d.Mh + ((size_t)src * d.nslots + slot) * d.s_max));
    // Warm the L2 for the NEXT source's Mh (a random 880-B record of a >100 MB array): one 16-B vector load
    // per lane now turns next iteration's dependent scalar loads from HBM/MALL misses into L2 hits.
    {
      const int src_next = __builtin_amdgcn_readfirstlane(d.m2l_src[pn]);
      const double2* nxt = d.Mh + ((size_t)src_next * d.nslots + slot) * d.s_max;
      if (tid < Sh::S) {
        const double2 touch = nxt[tid];
        asm volatile("" ::"v"(touch.x), "v"(touch.y));
      }
    }
    auto mac_range = [&](auto lo_c, auto hi_c) {
      constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#pragma unroll
      for (int n = LO; n < HI; ++n) {
#pragma unroll
        for (int m = -n; m <= n; ++m) {
          const int am = m < 0 ? -m : m;
          const double sr = (m < 0 && (am & 1)) ? -1.0 : 1.0;          // (-1)^m for m < 0
          const double si = (m < 0) ? -sr : 1.0;                       // conj
          const double ar = sr * mh[n * (n + 1) / 2 + am].x, ai = si * mh[n * (n + 1) / 2 + am].y;
          cfma(acc, double2{ar, ai}, ybase[n * R + m]);
        }
      }
    };
    if (valid) {
      if (npart == 0) mac_range(std::integral_constant<int, 0>{}, std::integral_constant<int, Sh::cut(1)>{});
      if (NS > 1 && npart == 1) mac_range(std::integral_constant<int, Sh::cut(1)>{}, std::integral_constant<int, Sh::cut(2)>{});
      if (NS > 2 && npart == 2) mac_range(std::integral_constant<int, Sh::cut(2)>{}, std::integral_constant<int, Sh::cut(3)>{});
      if (NS > 3 && npart == 3) mac_range(std::integral_constant<int, Sh::cut(3)>{}, std::integral_constant<int, Sh::cut(4)>{});
    }
  }
#undef LOAD
#undef STORE
#undef FMMBEM_REP8
  if (NS > 1) {                                        // fold the other n-ranges into the first, fixed order
    if (npart > 0) Comb[(npart - 1) * TEAM * kWave + otid] = acc;
    __syncthreads();
    if (npart == 0) {
#pragma unroll
      for (int q = 0; q < NS - 1; ++q) { acc.x += Comb[q * TEAM * kWave + otid].x; acc.y += Comb[q * TEAM * kWave + otid].y; }
    }
  }
  if (valid && npart == 0) {
    double2* L = d.L + ((size_t)tgt * d.nslots + slot) * d.s_max;
    const double f = ((j & 1) ? -1.0 : 1.0) * d.tabA[j * j + j + k];
    L[idx] = mul_i_pow(double2{acc.x * f, acc.y * f}, -k);
  }
}

#define FMMBEM_DISPATCH_P(p, ...)                                                                     \
  switch (p) {                                                                                         \
    case 1: { constexpr int PP = 1; __VA_ARGS__; } break;   case 2: { constexpr int PP = 2; __VA_ARGS__; } break;    \
    case 3: { constexpr int PP = 3; __VA_ARGS__; } break;   case 4: { constexpr int PP = 4; __VA_ARGS__; } break;    \
    case 5: { constexpr int PP = 5; __VA_ARGS__; } break;   case 6: { constexpr int PP = 6; __VA_ARGS__; } break;    \
    case 7: { constexpr int PP = 7; __VA_ARGS__; } break;   case 8: { constexpr int PP = 8; __VA_ARGS__; } break;    \
    case 9: { constexpr int PP = 9; __VA_ARGS__; } break;   case 10: { constexpr int PP = 10; __VA_ARGS__; } break;  \
    case 11: { constexpr int PP = 11; __VA_ARGS__; } break; case 12: { constexpr int PP = 12; __VA_ARGS__; } break;  \
    case 13: { constexpr int PP = 13; __VA_ARGS__; } break; case 14: { constexpr int PP = 14; __VA_ARGS__; } break;  \
    case 15: { constexpr int PP = 15; __VA_ARGS__; } break; case 16: { constexpr int PP = 16; __VA_ARGS__; } break;  \
    default: return hipErrorInvalidValue;                                                              \
  }

}  // namespace

hipError_t launch_m2l(const DevicePlan& d, int p, hipStream_t s) {
  if (d.n_m2l_tgt <= 0) return hipSuccess;
#define LAUNCH(NSV)                                                                                        \
  hipLaunchKernelGGL((m2l_kernel<PP, NSV>),                                                                \
                     dim3((d.n_m2l_tgt + Shape<PP, NSV>::TARGETS - 1) / Shape<PP, NSV>::TARGETS, d.n_act),  \
                     dim3(Shape<PP, NSV>::THREADS), 0, s, d)
  // NS = 2 measured best at p = 10 on MI355X (N = 1M: NS 1/2/3/4 -> 3.24 / 2.37 / 2.47 / 2.57 ms)
  FMMBEM_DISPATCH_P(p, if (PP < 6) { LAUNCH(1); } else { LAUNCH(2); })
#undef LAUNCH
  return hipGetLastError();
}

}  // namespace fmmbem
