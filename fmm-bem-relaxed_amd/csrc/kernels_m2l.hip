// kernels_m2l.hip -- the M2L translation, the FP64-FMA-bound hot kernel of the far field.
// Reference: LaplaceSpherical::M2L + evalLocal (kernel/LaplaceSpherical.hpp:296-329, 491-524), called
// once per LR_list pair and expansion (executor/EvalInteractionLazySparse.hpp:269-283).
//
// Algebra (kernels_far.hip header): L[j,k] += i^{-k} (-1)^j A[j,k] * sum_{n<P,|m|<=n} Mh[n,m] * Yh[j+n, m-k]
// with Mh the rescaled source multipole (mh_prep) and Yh[r,c] = i^{|c|} EPS Y[r,c] / A[r,c] the rescaled singular
// harmonics of the translation vector (rho, alpha, beta).  Yh factors into a phase and a REAL radial/polar part:
//     Yh[r,c] = Z^c * gh[r,c],   Z = i e^{i beta},   gh[r,c] = G[r,|c|] * (c < 0 ? (-1)^c : 1),
//     G[r,a] = EPS rho^{-r-1} P_r^a(cos alpha) pref[r,a] / A[r,a]
// so that the double sum becomes
//     Z^{-k} * sum_m Z^m * T_m,      T_m = sum_{n >= |m|} Mh[n,m] * gh[j+n, m-k]      (complex x real: 2 FMAs)
// i.e. 2 FMAs per (n,m) term + 4 per m + 4 per source = 280 instead of 400 at p = 10, and -- what matters more,
// the complex-table form of this kernel ran at the LDS bandwidth limit (one ds_read_b128 per 4 FMAs, 128 B/clk
// per CU) -- 8 bytes of LDS per term instead of 16: one ds_read_b128 serves n and n+1 (m2l_layout.hpp).
//
// Mapping: one wavefront per target box (a TEAM of 2-3 wavefronts when P(P+1)/2 > 64), one output (j,k) per
// lane, L accumulated in registers over the target's whole source list (CSR by target) and written once:
// no atomics, fixed summation order.  Per source:
//   * the class table G (P(2P+1) doubles, L2-resident) is PREFETCHED into registers while the previous source is
//     being computed, then scattered (both signs of c, both alignment copies) into the LDS image;
//   * Mh[n,m] and Z^m are wave-uniform and come through the scalar cache straight into the SGPR operands of
//     v_fma_f64; Z^k is one 16-byte vector load per lane, prefetched with the table.
// One instantiation per p = 1..16: the solver's per-iteration relaxation of p only picks among them.
#include "device_launch.hpp"
#include "m2l_layout.hpp"

#include <type_traits>

namespace fmmbem {

namespace {

constexpr int kWave = 64;
struct C2 { double x, y; };
typedef __attribute__((address_space(4))) C2 ConstC2;      // complex value in the constant address space
typedef __attribute__((address_space(4))) DevicePlan ConstPlan;
#ifndef FMMBEM_M2L_TARGETS
#define FMMBEM_M2L_TARGETS 1
#endif
// independent single-wavefront targets per workgroup when TEAM == 1.  Same box, N = 1M, M2L ms at p = 8 / 6 / 3 (small kernel):
// 1: 1.23 / 0.76 / 0.121   2: 1.27 / 0.76 / 0.126   4: 1.31 / 0.80 / 0.128   8: 1.67 / 0.88 / 0.135
constexpr int kM2LTargets = FMMBEM_M2L_TARGETS;
#ifndef FMMBEM_M2L_XCD
#define FMMBEM_M2L_XCD 1
#endif
constexpr bool kM2LXcdRemap = FMMBEM_M2L_XCD != 0;
#ifndef FMMBEM_M2L_XCD_CHUNK
#define FMMBEM_M2L_XCD_CHUNK 64
#endif
constexpr int kM2LXcdChunk = FMMBEM_M2L_XCD_CHUNK;

template <int P, int NS_> struct Shape {
  static constexpr int S = P * (P + 1) / 2, NE = m2l_entries(P);
  static constexpr int TEAM = m2l_team(P);               // wavefronts needed to give every output a lane
  // The m values of every output can additionally be split between NS wavefronts that share one LDS image:
  // NS times the wavefronts per CU for the same LDS, which is what hides the scalar-load latency of Mh.
  static constexpr int NS = NS_;
  static constexpr int WAVES = TEAM * NS;
  static constexpr int TARGETS = WAVES == 1 ? kM2LTargets : 1;
  static constexpr int THREADS = WAVES * TARGETS * kWave;
  static constexpr int COPYT = WAVES * kWave;             // threads sharing one target's table copy
  static constexpr int RR = m2l_rr(P), C0 = m2l_c0(P), O0 = m2l_odd_base(P), LDSD = m2l_lds_doubles(P);
  static constexpr int NLOAD = (NE + WAVES * kWave - 1) / (WAVES * kWave);   // table entries copied per lane
  // ds_read_b128 per m: pairs (n0, n0+1), n0 even, covering n = |m| .. P-1
  static constexpr int reads(int m) { const int am = m < 0 ? -m : m; return (P - (am & ~1) + 1) / 2; }
  // which wavefront sums the orders +-am (they share the Mh run): greedy over am = 1..P-1, 0 -> balanced reads
  struct Parts { int v[P]; };
  static constexpr Parts make_parts() {
    Parts t{};
    int load[4] = {0, 0, 0, 0};
    for (int i = 0; i < P; ++i) {
      const int am = i + 1 < P ? i + 1 : 0;
      int best = 0;
      for (int q = 1; q < NS; ++q) if (load[q] < load[best]) best = q;
      load[best] += reads(am) * (am ? 2 : 1);
      t.v[am] = best;
    }
    return t;
  }
  static constexpr Parts PT = make_parts();
};

__device__ inline double2 mul_i_pow(double2 a, int q) {              // a * i^q
  switch (q & 3) {
    case 0: return a;
    case 1: return {-a.y, a.x};
    case 2: return {-a.x, -a.y};
    default: return {a.y, -a.x};
  }
}

// NSLOT expansion slots (Stokes: the four harmonic potentials; Laplace with mixed BC: G and dG/dn) share one pass over
// a target's sources: same class table, same LDS reads, same barriers -- only Mh and the accumulators differ.
#ifndef FMMBEM_M2L_OCC4
#define FMMBEM_M2L_OCC4 4                              // four slots per pass (Stokes): 129 VGPRs without the hint, one over four wavefronts per
                                                       // SIMD; with it 128 and 12 B of scratch: 1.69 -> 1.52 ms at p = 8.  The same hint (5, 6) on the
                                                       // single-slot kernels changes nothing
#endif
template <int P, int NS_, int NSLOT>
__global__ __launch_bounds__((Shape<P, NS_>::THREADS), (NSLOT == 4 && P <= 10 ? FMMBEM_M2L_OCC4 : 1)) void m2l_kernel(const DevicePlan* __restrict__ d_in) {
  // The plan is read through a pointer: inside the source loop the pointer is made opaque once per iteration, so the
  // few fields the loop needs are re-read by scalar loads (scalar cache hits) instead of being kept alive in SGPRs --
  // the FMA region wants nearly all of them for Mh, and what else is live across it gets spilled to VGPR lanes and
  // restored with VALU instructions every source.
  const DevicePlan& d = *d_in;
  using Sh = Shape<P, NS_>;
  constexpr int NE = Sh::NE, TEAM = Sh::TEAM, TARGETS = Sh::TARGETS, RR = Sh::RR, NLOAD = Sh::NLOAD;
  constexpr int NS = Sh::NS, WAVES = Sh::WAVES;
  // + one spare double per copying thread: table entries without a place are stored there, so that the scatter
  // needs no per-destination branch (whose loop-invariant exec masks would sit in SGPRs across the FMA region)
  __shared__ double2 Gall[TARGETS][(Sh::LDSD + Sh::COPYT + 1) / 2];
  __shared__ double2 Comb[NS == 1 ? 1 : NSLOT * (NS - 1) * TEAM * kWave];   // partial sums of the other m sets
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: give XCD x the x-th CONTIGUOUS eighth of the
  // target list (level order, Morton order within a level) instead of every eighth target, so that the Mh records and
  // class tables neighbouring targets share are fetched into one L2 instead of eight.  gridDim.x is a multiple of 8.
  constexpr int CH = kM2LXcdChunk;                     // consecutive workgroups-worth of targets that stay on one XCD
  const int rnd = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
  const int tb = kM2LXcdRemap ? (rnd / CH) * 8 * CH + xcd * CH + rnd % CH : (int)blockIdx.x;
  const int ti = tb * TARGETS + (WAVES == 1 ? wave : 0);
  if (ti >= d.n_m2l_tgt) return;                       // WAVES==1: whole wavefront; else: whole workgroup
  const int tgt = d.m2l_tgt[ti];
  int slot[NSLOT];
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) slot[q] = d.act[blockIdx.y * NSLOT + q];
  double* Gt = reinterpret_cast<double*>(Gall[WAVES == 1 ? wave : 0]);
  const int tid = WAVES == 1 ? lane : (int)threadIdx.x;  // index within the team (table copy)
  const int npart = WAVES == 1 ? 0 : wave % NS;          // which m set this wavefront sums
  const int otid = WAVES == 1 ? lane : (wave / NS) * kWave + lane;   // index in the lane -> output map

  // this lane's output (conflict-free dealing, m2l_layout.hpp)
  const int packed = d.m2l_lane[(P - 1) * kM2LMaxThreads + otid];
  const bool valid = packed >= 0;
  const int j = valid ? (packed & 0xff) : 0, k = valid ? ((packed >> 8) & 0xff) : 0, idx = valid ? (packed >> 16) : 0;
  // 16-byte slot of gh[j + 0, 0 - k] in this lane's alignment copy; (m + C0) * RR + n0 is added per term
  const double2* gbase = reinterpret_cast<const double2*>(Gt) + (((j & 1) ? Sh::O0 + 1 : 0) + j - k * RR) / 2;
  double2 acc[NSLOT];
#pragma unroll
  for (int q = 0; q < NSLOT; ++q) acc[q] = {0, 0};
  // Where only one half of a 16-byte LDS read feeds an FMA (first pair of an odd order), hipcc narrows the read to
  // 8 bytes and pairs such reads into ds_read2_b64, which the conflict-free lane dealing does not cover: the unused
  // half is multiplied by a zero the compiler cannot see and starts the sum instead.
  const double zero = (double)(d.n_m2l_tgt >> 31);
  // ... and the half it multiplies may be a place the table scatter never writes (|c| > r): clear the image once,
  // stale LDS contents can be NaN or Inf.
  for (int i = tid; i < Sh::LDSD; i += WAVES * kWave) Gt[i] = 0.0;

  // This lane's share of the table copy: entries tid, tid + WAVES*64, ...; every entry has up to four LDS places.
  // The staging registers are NAMED scalars (macro-expanded), not arrays: hipcc keeps an array that is live across
  // the loop in scratch memory, which serialises the prefetch.
  static_assert(NLOAD <= 4, "table copy needs more staging registers");
  const int spare = (Sh::LDSD + tid) << 1;             // this thread's spare place, same encoding as the scatter map
  auto fix_dst = [](int dst, int sp) { return dst >= 0 ? dst : sp; };
  auto flip_sign = [](double v, int dst) {             // -v where bit 0 of the encoded place is set (integer xor, no mask)
    return __hiloint2double(__double2hiint(v) ^ (dst << 31), __double2loint(v));
  };
  const int* scat = d.m2l_scat + d.m2l_scat_off[P - 1];
#define FMMBEM_REP4(X) X(0) X(1) X(2) X(3)
  // Everything loop-invariant that the scatter needs lives in VGPRs (place, sign bit, clamped table index): as
  // conditions they would be exec masks parked in SGPR pairs across the FMA region, i.e. spilled and restored.
#define DECL(u)                                                                   \
  const int e##u = tid + u * Sh::COPYT;                                           \
  const bool on##u = u < NLOAD && e##u < NE;                                      \
  const int le##u = on##u ? e##u : 0;                                             \
  const int da##u = fix_dst(on##u ? scat[4 * e##u + 0] : -1, spare), db##u = fix_dst(on##u ? scat[4 * e##u + 1] : -1, spare); \
  const int dc##u = fix_dst(on##u ? scat[4 * e##u + 2] : -1, spare), dd##u = fix_dst(on##u ? scat[4 * e##u + 3] : -1, spare); \
  double pre##u = 0;
  FMMBEM_REP4(DECL)
#undef DECL
#define LOAD(u) if (u < NLOAD) pre##u = tab[le##u];
#define PUT(dst, v) Gt[dst >> 1] = flip_sign(v, dst);
#define STORE(u) if (u < NLOAD) { PUT(da##u, pre##u) PUT(db##u, pre##u) PUT(dc##u, pre##u) PUT(dd##u, pre##u) }

  const int pb = d.m2l_ptr[tgt], pe = d.m2l_ptr[tgt + 1];
  if (pb == pe) {                                      // a box that only inherits from its parent: L = 0
    if (valid)
      for (int q = 0; q < NSLOT; ++q) d.L[((size_t)tgt * d.nslots + slot[q]) * d.s_max + idx] = {0, 0};
    return;
  }
  double2 zk;                                          // Z^k of the class in flight
  {
    const int cls = __builtin_amdgcn_readfirstlane(d.m2l_cls[pb]);
    const double* tab = d.m2l_g + (size_t)cls * d.g_max;
    FMMBEM_REP4(LOAD)
    zk = d.m2l_z[(size_t)cls * d.p_max + k];
  }
  for (int pi = pb; pi < pe; ++pi) {
    uintptr_t dl = reinterpret_cast<uintptr_t>(d_in);
    asm volatile("" : "+s"(dl));                       // forget what was loaded through it
    const ConstPlan& d = *reinterpret_cast<const ConstPlan*>(dl);   // constant address space: scalar loads
    const int src = __builtin_amdgcn_readfirstlane(d.m2l_src[pi]);
    const int cls = __builtin_amdgcn_readfirstlane(d.m2l_cls[pi]);
    const int pn = pi + 1 < pe ? pi + 1 : pi;          // last iteration re-reads its own table (harmless)
    const int cls_next = __builtin_amdgcn_readfirstlane(d.m2l_cls[pn]);
    if (WAVES == 1) __builtin_amdgcn_wave_barrier(); else __syncthreads();    // previous source's reads are done
    FMMBEM_REP4(STORE)
    const double2 zk_now = zk;
    if (WAVES == 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();
    }
    {                                                  // next table in flight during this source's FMAs
      const double* tab = d.m2l_g + (size_t)cls_next * d.g_max;
      FMMBEM_REP4(LOAD)
      zk = d.m2l_z[(size_t)cls_next * d.p_max + k];
    }
    // Mh (orders m >= 0 only; Mh[n,-m] = (-1)^m conj(Mh[n,m]) costs only sign modifiers on the FMAs) was
    // written by the preceding kernel and is immutable here, like the class phases Z^m: address both through the
    // CONSTANT address space so that the wave-uniform loads are always selected as scalar (SMEM) loads feeding
    // SGPR operands.
    const ConstC2* mh[NSLOT];
#pragma unroll
    for (int q = 0; q < NSLOT; ++q)
      mh[q] = reinterpret_cast<const ConstC2*>(reinterpret_cast<uintptr_t>(d.Mh + ((size_t)src * d.nslots + slot[q]) * d.s_max));
    const ConstC2* zm = reinterpret_cast<const ConstC2*>(reinterpret_cast<uintptr_t>(d.m2l_z + (size_t)cls * d.p_max));
    // Warm the L2 for the NEXT source's Mh (a random 880-B record of a >100 MB array): one 16-B vector load
    // per lane now turns next iteration's dependent scalar loads from HBM/MALL misses into L2 hits.
    {
      const int src_next = __builtin_amdgcn_readfirstlane(d.m2l_src[pn]);
#pragma unroll
      for (int q = 0; q < NSLOT; ++q) {
        const double2* nxt = d.Mh + ((size_t)src_next * d.nslots + slot[q]) * d.s_max;
        if (tid < Sh::S) {
          const double2 touch = nxt[tid];
          asm volatile("" ::"v"(touch.x), "v"(touch.y));
        }
      }
    }
    double2 as[NSLOT];                                 // sum_m Z^m T_m of this source (this wavefront's m set)
#pragma unroll
    for (int q = 0; q < NSLOT; ++q) as[q] = {0, 0};
    auto mac_part = [&](auto part_c) {
      constexpr int PART = decltype(part_c)::value;
      // orders +am and -am together: they share the Mh run (Mh[n,-m] = (-1)^m conj(Mh[n,m]): sign modifiers) and
      // give four independent FMA chains per pass instead of two
#pragma unroll
      for (int am = 0; am < P; ++am) {
        if (Sh::PT.v[am] != PART) continue;
        const double sr = (am & 1) ? -1.0 : 1.0;                       // (-1)^m for the negative order; conj: -sr on im
        double trp[NSLOT], tqp[NSLOT], trn[NSLOT], tqn[NSLOT];
        bool started = false, started_r = false;   // the first product initialises the sums (no v_mov 0 + fma)
#pragma unroll
        for (int n0 = am & ~1; n0 < P; n0 += 2) {
          const double2 gp = gbase[((am + Sh::C0) * RR + n0) / 2];
          const double2 gn = am ? gbase[((-am + Sh::C0) * RR + n0) / 2] : gp;
          if (n0 < am) {                               // keeps the reads 16 bytes wide (see `zero`)
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) { trp[q] = zero * gp.x; if (am) trn[q] = zero * gn.x; }
            started_r = true;
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int n = n0 + h;
            if (n < am || n >= P) continue;
            const double gpv = h ? gp.y : gp.x, gnv = h ? gn.y : gn.x;
#pragma unroll
            for (int q = 0; q < NSLOT; ++q) {
              const double ar = mh[q][am * P - am * (am - 1) / 2 + n - am].x, ai = mh[q][am * P - am * (am - 1) / 2 + n - am].y;
              trp[q] = started_r ? fma(ar, gpv, trp[q]) : ar * gpv;
              tqp[q] = started ? fma(ai, gpv, tqp[q]) : ai * gpv;
              if (am) {
                trn[q] = started_r ? fma(sr * ar, gnv, trn[q]) : (sr * ar) * gnv;
                tqn[q] = started ? fma(-sr * ai, gnv, tqn[q]) : (-sr * ai) * gnv;
              }
            }
            started = started_r = true;
          }
        }
        const double er = zm[am].x, ei = zm[am].y;                     // Z^m; Z^{-m} = conj
#pragma unroll
        for (int q = 0; q < NSLOT; ++q) {
          as[q].x = fma(er, trp[q], as[q].x); as[q].x = fma(-ei, tqp[q], as[q].x);
          as[q].y = fma(er, tqp[q], as[q].y); as[q].y = fma(ei, trp[q], as[q].y);
          if (am) {
            as[q].x = fma(er, trn[q], as[q].x); as[q].x = fma(ei, tqn[q], as[q].x);
            as[q].y = fma(er, tqn[q], as[q].y); as[q].y = fma(-ei, trn[q], as[q].y);
          }
        }
      }
    };
    if ((TEAM == 1 && Sh::S > 48) || valid) {          // a full wavefront of outputs: the few lanes without one run along
      // (no exec mask to keep alive); otherwise idle 16-lane groups are masked off the LDS
      if (npart == 0) mac_part(std::integral_constant<int, 0>{});
      if (NS > 1 && npart == 1) mac_part(std::integral_constant<int, 1>{});
      if (NS > 2 && npart == 2) mac_part(std::integral_constant<int, 2>{});
      if (NS > 3 && npart == 3) mac_part(std::integral_constant<int, 3>{});
      // acc += Z^{-k} * as = conj(Z^k) * as
#pragma unroll
      for (int q = 0; q < NSLOT; ++q) {
        acc[q].x = fma(zk_now.x, as[q].x, acc[q].x); acc[q].x = fma(zk_now.y, as[q].y, acc[q].x);
        acc[q].y = fma(zk_now.x, as[q].y, acc[q].y); acc[q].y = fma(-zk_now.y, as[q].x, acc[q].y);
      }
    }
  }
#undef LOAD
#undef PUT
#undef STORE
#undef FMMBEM_REP4
  if (NS > 1) {                                        // fold the other m sets into the first, fixed order
    if (npart > 0)
#pragma unroll
      for (int e = 0; e < NSLOT; ++e) Comb[(e * (NS - 1) + npart - 1) * TEAM * kWave + otid] = acc[e];
    __syncthreads();
    if (npart == 0) {
#pragma unroll
      for (int e = 0; e < NSLOT; ++e)
#pragma unroll
        for (int q = 0; q < NS - 1; ++q) {
          acc[e].x += Comb[(e * (NS - 1) + q) * TEAM * kWave + otid].x;
          acc[e].y += Comb[(e * (NS - 1) + q) * TEAM * kWave + otid].y;
        }
    }
  }
  if (valid && npart == 0) {
    const double f = ((j & 1) ? -1.0 : 1.0) * d.tabA[j * j + j + k];
#pragma unroll
    for (int e = 0; e < NSLOT; ++e) {
      double2* L = d.L + ((size_t)tgt * d.nslots + slot[e]) * d.s_max;
      L[idx] = mul_i_pow(double2{acc[e].x * f, acc[e].y * f}, -k);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Low orders (p <= 4, where the relaxed solver spends most of its iterations): with 1-10 outputs per box the kernel
// above is a chain of per-source latencies (scalar Mh loads, table scatter, two barriers: 0.32 ms at p = 1 for
// 1.96 M pairs).  Here a wavefront takes one target and its LANES take the sources: each lane gathers its source's
// Mh, class table and phases straight from L2, forms all S outputs in registers, and the lanes are summed by a
// fixed butterfly.  Same algebra: sum_m Z^m sum_n Mh[n,m] gh[j+n, m-k], times Z^{-k}.
// ---------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(kM2LTargets * kWave) void m2l_small_kernel(DevicePlan d) {
  constexpr int S = P * (P + 1) / 2;
  const int lane = threadIdx.x & (kWave - 1);
  const int ti = blockIdx.x * kM2LTargets + threadIdx.x / kWave;
  if (ti >= d.n_m2l_tgt) return;                       // whole wavefront
  const int tgt = d.m2l_tgt[ti];
  const int slot = d.act[blockIdx.y];
  const int pb = d.m2l_ptr[tgt], pe = d.m2l_ptr[tgt + 1];
  double2 acc[S];
#pragma unroll
  for (int i = 0; i < S; ++i) acc[i] = {0, 0};
  for (int pi = pb + lane; pi < pe; pi += kWave) {
    const int src = d.m2l_src[pi], cls = d.m2l_cls[pi];
    const double2* mh = d.Mh + ((size_t)src * d.nslots + slot) * d.s_max;     // order-major: [am*P - am(am-1)/2 + n - am]
    const double* G = d.m2l_g + (size_t)cls * d.g_max;                         // [r(r+1)/2 + a]
    const double2* Z = d.m2l_z + (size_t)cls * d.p_max;                        // Z^m
    double2 mv[S], z[P];
    double g[(2 * P - 1) * P];                         // r <= 2P-2
#pragma unroll
    for (int i = 0; i < S; ++i) mv[i] = mh[i];
#pragma unroll
    for (int i = 0; i < P; ++i) z[i] = Z[i];
#pragma unroll
    for (int i = 0; i < (2 * P - 1) * P; ++i) g[i] = G[i];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int k = 0; k <= j; ++k) {
        double2 as = {0, 0};
#pragma unroll
        for (int m = -(P - 1); m <= P - 1; ++m) {
          const int am = m < 0 ? -m : m;
          const double sr = (m < 0 && (am & 1)) ? -1.0 : 1.0, si = (m < 0) ? -sr : 1.0;   // Mh[n,-m] = (-1)^m conj
          double tr = 0, tq = 0;
#pragma unroll
          for (int n = am; n < P; ++n) {
            const int c = m - k, a = c < 0 ? -c : c, r = j + n;
            const double gh = ((c < 0 && (a & 1)) ? -1.0 : 1.0) * g[r * (r + 1) / 2 + a];
            const double2 v = mv[am * P - am * (am - 1) / 2 + n - am];
            tr = fma(sr * v.x, gh, tr);
            tq = fma(si * v.y, gh, tq);
          }
          const double er = z[am].x, ei = (m < 0 ? -1.0 : 1.0) * z[am].y;       // Z^m, Z^{-m} = conj
          as.x = fma(er, tr, as.x); as.x = fma(-ei, tq, as.x);
          as.y = fma(er, tq, as.y); as.y = fma(ei, tr, as.y);
        }
        double2& t = acc[j * (j + 1) / 2 + k];          // += Z^{-k} * as = conj(Z^k) * as
        t.x = fma(z[k].x, as.x, t.x); t.x = fma(z[k].y, as.y, t.x);
        t.y = fma(z[k].x, as.y, t.y); t.y = fma(-z[k].y, as.x, t.y);
      }
  }
#pragma unroll
  for (int i = 0; i < S; ++i) {
#pragma unroll
    for (int off = 32; off; off >>= 1) {
      acc[i].x += __shfl_xor(acc[i].x, off, kWave);
      acc[i].y += __shfl_xor(acc[i].y, off, kWave);
    }
  }
  double2* L = d.L + ((size_t)tgt * d.nslots + slot) * d.s_max;
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int k = 0; k <= j; ++k)
      if (lane == j * (j + 1) / 2 + k) {
        const double f = ((j & 1) ? -1.0 : 1.0) * d.tabA[j * j + j + k];
        const double2 a = acc[j * (j + 1) / 2 + k];
        L[j * (j + 1) / 2 + k] = mul_i_pow(double2{a.x * f, a.y * f}, -k);
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

hipError_t launch_m2l(const DevicePlan& d, const DevicePlan* d_dev, int p, hipStream_t s) {
  if (d.n_m2l_tgt <= 0) return hipSuccess;
#define LAUNCH_Q(NSV, NQ)                                                                                     \
  hipLaunchKernelGGL((m2l_kernel<PP, NSV, NQ>),                                                               \
                     dim3((((d.n_m2l_tgt + Shape<PP, NSV>::TARGETS - 1) / Shape<PP, NSV>::TARGETS) + 8 * kM2LXcdChunk - 1) / (8 * kM2LXcdChunk) * (8 * kM2LXcdChunk), d.n_act / NQ), \
                     dim3(Shape<PP, NSV>::THREADS), 0, s, d_dev)
  // several active expansion slots per pass (Stokes: 4; Laplace with mixed BC: 2)
  // (Stokes config 4, p = 8, ms: 2.64 with one slot per pass, 2.32 with two; with the +-m pairing 1.96 with two, 1.84 with four)
#define LAUNCH(NSV) do { if (d.n_act % 4 == 0 && PP <= 10) { LAUNCH_Q(NSV, 4); } else if (d.n_act % 2 == 0 && PP <= 12) { LAUNCH_Q(NSV, 2); } else { LAUNCH_Q(NSV, 1); } } while (0)
#define LAUNCH_SMALL()                                                                                     \
  hipLaunchKernelGGL((m2l_small_kernel<(PP <= 4 ? PP : 1)>), dim3((d.n_m2l_tgt + kM2LTargets - 1) / kM2LTargets, d.n_act), \
                     dim3(kM2LTargets * kWave), 0, s, d)
  // wavefronts sharing a target's m sets, NS (N = 1M, ms with NS = 1 / 2 / 3 / 4): p = 6: 0.85 / 0.90 / 1.25 / 1.34;
  // p = 8: 1.46 / 1.50 / 1.54 / 1.91; p = 10: 2.57 / 2.03 / 2.45 / 2.17; p = 12 (two wavefronts of outputs): 4.80 / 5.01 / 5.87 / 5.59
  // with several slots per pass the work per source grows and two wavefronts pay off from p = 7 (Stokes p = 8: 2.05 / 1.84)
  // re-measured on one box with one target per workgroup and the XCD rounds (ms, NS = 1 / 2): p = 5 0.58 / 0.77, 6 0.75 / 0.83,
  // 7 0.94 / 1.02, 8 1.21 / 1.35, 9 1.64 / 1.80, 10 2.18 / 1.93, 11 4.00 / 4.01, 12 3.94 / 4.11; Stokes (four slots per pass):
  // p = 6 1.12 / 1.15, 7 1.23 / 1.36, 8 1.71 / 1.70, 9 2.02 / 2.32, 10 1.69 / 1.68  =>  two wavefronts at p = 10 only
#ifndef FMMBEM_M2L_NS
#define FMMBEM_M2L_NS 0                                // tuning: 1 or 2 forces the wavefronts per target for p >= 5
#endif
  FMMBEM_DISPATCH_P(p, if (PP <= 4) { LAUNCH_SMALL(); } else if (FMMBEM_M2L_NS == 2 || (FMMBEM_M2L_NS == 0 && PP == 10)) { LAUNCH(2); } else { LAUNCH(1); })
#undef LAUNCH_SMALL
#undef LAUNCH
#undef LAUNCH_Q
  return hipGetLastError();
}

}  // namespace fmmbem
