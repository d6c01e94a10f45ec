// kernels_m2l_rot2.hip -- M2L by rotation / axial translation / rotation, SPLIT form: one (target, source) pair on TWO lanes.
// Reference: LaplaceSpherical::M2L + evalLocal (kernel/LaplaceSpherical.hpp:296-329, 491-524); the algebra is that of
// kernels_m2l_rot.hip, the split and its constant streams are in m2l_rot.hpp ("Split form").
//
// STATUS (round 3): built, bit-level parity green, NOT the default -- 0.61-0.66 ms against 0.56 ms at p = 10, 1.08 against 1.03 at
// p = 12 (N = 1M).  The arithmetic is as short as planned (2 350 instructions per 32-pair pass: 4 700 per 64 pairs, no AGPR
// moves) and two wavefronts share a SIMD, but the gather of the source multipoles (30 x 16 bytes per lane: 1.7 GB per launch out
// of L2, ~0.3 ms) stands in front of every pass: at 256 registers per wavefront there is no room to fetch the next pass's
// operands ahead, which is how the one-pair-per-lane form hides it.  FMMBEM_M2L_ROT2=1 selects it; profiles/r03p_m2l_split_form.txt.
//
// Why.  At p >= 9 a pair's coefficients alone are 4 S VGPRs (220 at p = 10): one wavefront per SIMD, the overflow shuttling
// through AGPRs, and nothing to cover a stall -- VALU busy half the time.  Here a lane holds the even degrees of its pair (E
// lanes: rows 0 and 2 of the wavefront) or the odd ones (O lanes: rows 1 and 3; the partner is lane ^ 16): half the registers,
// two wavefronts per SIMD.  Both lanes run the instruction stream of the ODD degrees -- the even degree 2q embeds into the
// program of degree 2q + 1 with the order shifted by one -- and take their own constants from the two interleaved streams through
// the DPP row broadcast.  The z rotations and the scalings are local; the axial translation sums over the lane's own degrees for
// the outputs of both parities and swaps the foreign halves with the partner, once per pass.  A wavefront takes 32 pairs per pass;
// the reduction over the pairs of a target is the one of kernels_m2l_rot.hip in pair space (same chains, same order of additions
// per target whatever the cut into passes and shards).
#include "device_launch.hpp"
#include "m2l_rot.hpp"

#include <type_traits>

namespace fmmbem {

namespace {

constexpr int kWave = 64;
constexpr int kPairs = 32;                            // pairs per pass: two lanes each
constexpr int kChains = 4;                            // partial sums per (target, coefficient)
constexpr int kRow2 = kPairs + 4;                     // tile row in double2: rows 16 banks apart (see kernels_m2l_rot.hip)

#ifndef FMMBEM_ROT_XCD_CHUNK
#define FMMBEM_ROT_XCD_CHUNK 32
#endif

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int idx_of(int n, int m) { return n * (n + 1) / 2 + m; }

template <int CTRL>
__device__ __forceinline__ double quad_swap(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// the partner's value (lane ^ 16)
__device__ __forceinline__ double partner(double v) { return __shfl_xor(v, 16, kWave); }

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(static_cast<F&&>(f));
  }
}
#define FMMBEM_INLINE __attribute__((always_inline))

// The constant stream of the split form: groups of sixteen positions, per group 16 E constants then 16 O constants
// (m2l_rot.hpp build_rot2_stream).  A lane loads the constant of its parity at lane & 15; everything else as ConstFeed of
// kernels_m2l_rot.hip: loads issued by hand kRotAhead groups ahead, in-order vmcnt waits, the FMA takes lane k of its row.
// the DPP hazard remedy of kernels_m2l_rot.hip: orders listed in FMMBEM_ROT_NOP_ORDERS (bit p - 1; set by csrc/Makefile when
// tools/check_rot_isa.py finds the hazard in the generated code) get "s_nop 1" in front of every DPP FMA
#ifndef FMMBEM_ROT_NOP_ORDERS
#define FMMBEM_ROT_NOP_ORDERS 0u
#endif
constexpr bool rot2_needs_nop(int P) { return ((FMMBEM_ROT_NOP_ORDERS) >> (P - 1)) & 1u; }

template <int NG, bool kNop = false>
struct ConstFeed2 {
  static constexpr int NB = kRotAhead + 1;
  static constexpr int kGroupBytes = 2 * kRotGroup * 8;
  static constexpr int kWindow = 4096 / kGroupBytes;  // groups per 12-bit immediate window
  const double* base;
  double cv[NB];
  template <int G>
  __device__ __forceinline__ void issue() {
    if constexpr (G < NG) {
      if constexpr (G % kWindow == 0 && G != 0) {
        base += kWindow * 2 * kRotGroup;
        asm volatile("" : "+v"(base));
      }
      asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=&v"(cv[G % NB]) : "v"(base), "n"((G % kWindow) * kGroupBytes));
    }
  }
  __device__ __forceinline__ void start(const double* stream, int lane) {
    base = stream + ((lane >> 4) & 1) * kRotGroup + (lane & 15);
    static_for<0, kRotAhead>([&](auto g) FMMBEM_INLINE { issue<decltype(g)::value>(); });
  }
  template <int K>
  static __device__ __forceinline__ void dpp_fma(double& acc, double c, double src) {
    if constexpr (kNop) asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(src), "n"(K));
    else asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(src), "n"(K));
  }
  template <int E>
  __device__ __forceinline__ void fma1(double& acc, double src) {
    constexpr int g = E / kRotGroup, k = E % kRotGroup;
    if constexpr (k == 0) {
      issue<g + kRotAhead>();
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(cv[g % NB]) : "n"(NG - 1 - g < kRotAhead ? NG - 1 - g : kRotAhead));
    }
    dpp_fma<k>(acc, cv[g % NB], src);
  }
  template <int E>
  __device__ __forceinline__ void fma2(double& acc1, double src1, double& acc2, double src2) {
    fma1<E>(acc1, src1);
    constexpr int g = E / kRotGroup, k = E % kRotGroup;
    dpp_fma<k>(acc2, cv[g % NB], src2);
  }
};

// out = R in, degree pair by degree pair: the program of the odd degree 2q + 1 on the slots (q, 0 .. 2q+1) of both kinds of lane
template <int P, int STAGE, class Feed>
__device__ __forceinline__ void fixed_rotation2(double (&a)[rot2_nslots(P)], double (&b)[rot2_nslots(P)], Feed& cf) {
  static_for<0, rot2_pairs(P)>([&](auto Q_) FMMBEM_INLINE {
    constexpr int q = decltype(Q_)::value, n = 2 * q + 1;
    double na[n + 1], nb[n + 1];
    static_for<0, n + 1>([&](auto M_) FMMBEM_INLINE {
      constexpr int m = decltype(M_)::value;
      double sa = 0, sb = 0;
      static_for<0, n + 1>([&](auto MP) FMMBEM_INLINE {
        constexpr int mp = decltype(MP)::value;
        if constexpr (rot_live(n, m, mp)) {
          constexpr int e = rot2_stage_base(P, STAGE) + rot2_rot_index(q, m, mp);
          constexpr bool even = ((n + m) & 1) == 0;
          const double src = (mp == 0 || even) ? a[rot2_sidx(q, mp)] : b[rot2_sidx(q, mp)];
          if constexpr ((rot_kk(n, m, mp) & 1) == 0) cf.template fma1<e>(sa, src); else cf.template fma1<e>(sb, src);
        }
      });
      na[m] = sa; nb[m] = sb;
    });
#pragma unroll
    for (int m = 0; m <= n; ++m) { a[rot2_sidx(q, m)] = na[m]; b[rot2_sidx(q, m)] = nb[m]; }
  });
}

// slot (q, t) turns by e^{i t g} on an O lane, by e^{i (t - 1) g} on an E lane: (c0, s0) = the lane's phase of slot order 1
template <int P>
__device__ __forceinline__ void z_rotation2(double (&a)[rot2_nslots(P)], double (&b)[rot2_nslots(P)], double c1, double s1, double c0, double s0) {
  double cm = c0, sm = s0;
#pragma unroll
  for (int t = 1; t < 2 * rot2_pairs(P); ++t) {
#pragma unroll
    for (int q = rot2_qmin(t); q < rot2_pairs(P); ++q) {
      const double x = a[rot2_sidx(q, t)], y = b[rot2_sidx(q, t)];
      a[rot2_sidx(q, t)] = fma(x, cm, -(y * sm));
      b[rot2_sidx(q, t)] = fma(x, sm, y * cm);
    }
    const double c2 = fma(cm, c1, -(sm * s1)), s2 = fma(sm, c1, cm * s1);
    cm = c2; sm = s2;
  }
}

// every slot of pair q times r0 * step^q
template <int P>
__device__ __forceinline__ void scale2(double (&a)[rot2_nslots(P)], double (&b)[rot2_nslots(P)], double r0, double step) {
  double r = r0;
#pragma unroll
  for (int q = 0; q < rot2_pairs(P); ++q) {
#pragma unroll
    for (int t = 0; t <= 2 * q + 1; ++t) { a[rot2_sidx(q, t)] *= r; if (t) b[rot2_sidx(q, t)] *= r; }
    r *= step;
  }
}

constexpr int rot2_tile(int P) {                      // coefficients per reduction round: the share of 160 KB of one of 8 wavefronts per CU
  const int S = P * (P + 1) / 2, budget = 160 * 1024 / 8 - 1024;
  for (int R = 1; R <= S; ++R) {
    const int kt = (S + R - 1) / R;
    if (kt * kRow2 * 16 + S * kChains * 16 + 512 <= budget) return kt;
  }
  return 1;
}

// first lane (the E lane) of pair j of a pass
__device__ __forceinline__ int lane_of_pair(int j) { return (j & 15) | ((j >> 4) << 5); }

// OP: kRotM2L, or one of the shifts of the tree passes (kRotM2M: lane pair = (parent, child), the children of a parent added
// like the pairs of an M2L target; kRotL2L: lane pair = (child, parent), added into the child's L).  The shifts run items of ONE
// pass of 32 pairs, one expansion slot per wavefront (blockIdx.y): a level of the tree is a single pass on a chip that is not
// full, and this form's pass is half as long as that of kernels_m2l_rot.hip.
template <int P, int OP>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void m2l_rot2_kernel(const DevicePlan d, const RotWork w) {
  constexpr int S = P * (P + 1) / 2, Q = rot2_pairs(P), NS = rot2_nslots(P);
  constexpr int KT = rot2_tile(P);
  constexpr int NT = (S + KT - 1) / KT;
  __shared__ double2 tile[KT][kRow2];
  __shared__ double2 carry[S][kChains];
  const int lane = threadIdx.x;
  int par = (lane >> 4) & 1;                           // 0: E lane (even degrees), 1: O lane (odd degrees)
  const int pj = (lane & 15) | ((lane >> 5) << 4);     // the pair of the pass this lane works on
  constexpr int CH = FMMBEM_ROT_XCD_CHUNK;
  const int rnd = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
  const int item = (rnd / CH) * 8 * CH + xcd * CH + rnd % CH;
  if (item >= w.n_items) return;
  const int ib = w.item_ptr[item], ie = w.item_ptr[item + 1];
#ifdef FMMBEM_ROT2_EXP_OFFSET
  // experiment: the two wavefronts of a SIMD start together and would load together, compute together: hold every second one back
  if ((blockIdx.x / FMMBEM_ROT2_EXP_OFFSET) & 1) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
#endif
  const double mask_o = par ? 1.0 : 0.0, mask_e = par ? 0.0 : 1.0;

  const int qs_begin = OP == kRotM2L ? 0 : (int)blockIdx.y, qs_end = OP == kRotM2L ? d.n_act : qs_begin + 1;
  for (int qs = qs_begin; qs < qs_end; ++qs) {
    const int slot = d.act[qs];
    const double2* Mslot = (OP == kRotL2L ? d.L : d.M) + (size_t)slot * d.s_max;       // the operand: M, or the parent's L
    const size_t box_stride = (size_t)d.nslots * d.s_max;
    bool cont_in = false;
    int cont_q = 0;
    int pi = ib + pj < ie ? ib + pj : ie - 1;          // pairs past the end repeat the item's last pair
    int src = w.src[pi], cls = w.cls[pi], tgt = w.tgt[pi];
    for (int pb = ib; pb < ie; pb += kPairs) {
      const int cnt = ie - pb < kPairs ? ie - pb : kPairs;
      const bool live = pj < cnt;
      // the lane's parity, opaque per pass: everything selected by it (30 load offsets, 30 tile rows) is then computed where it is
      // used -- one v_cndmask each -- instead of being hoisted out of the pass loop, kept in registers and spilled to scratch
      int par_ = par;
      asm volatile("" : "+v"(par_));
      const bool odd = par_ != 0;
      // ---- segments (targets) of this pass, in pair space: bit j of smask = pair j begins a target ----
      const int prev = __shfl(tgt, lane_of_pair(pj > 0 ? pj - 1 : 0), kWave);
      const unsigned long long bal = __ballot(live && !odd && (pj == 0 || prev != tgt));
      const unsigned smask = (unsigned)(bal & 0xffffull) | ((unsigned)((bal >> 32) & 0xffffull) << 16);
      const int nseg = __popc(smask);
      const int flast = 31 - __clz((int)smask);
      const bool more = pb + kPairs < ie;
      const int npi = more ? (pb + kPairs + pj < ie ? pb + kPairs + pj : ie - 1) : pi;
      const int nsrc = w.src[npi], ncls = w.cls[npi], ntgt = w.tgt[npi];

      // ---- this lane's half of its pair: O lane slot (q, t) = M[2q+1, t], E lane slot (q, t) = M[2q, t-1] ----
      // (round 3 also tried the multipoles re-laid-out per parity by a preparation kernel -- one base address and immediates per
      // lane, four cache lines of its own: 0.74 ms against 0.66, the preparation pass costs more than the loads gain)
      double a[NS], b[NS];
      {
        const double2* M = Mslot + (size_t)src * box_stride;
        static_for<0, Q>([&](auto Q_) FMMBEM_INLINE {
          constexpr int q = decltype(Q_)::value;
          static_for<0, 2 * q + 2>([&](auto T_) FMMBEM_INLINE {
            constexpr int t = decltype(T_)::value;
            constexpr bool has_o = 2 * q + 1 < P, has_e = t >= 1;
            constexpr int io = has_o ? idx_of(2 * q + 1, t) : 0, ie_ = has_e ? idx_of(2 * q, t - 1) : 0;
            const bool has = odd ? has_o : has_e;
            const double2 v = M[odd ? io : ie_];
            a[rot2_sidx(q, t)] = has ? v.x : 0.0;
            b[rot2_sidx(q, t)] = has ? v.y : 0.0;
          });
        });
      }
      const double* cr = w.rec + (size_t)cls * 8;
      const double inv_rho = cr[0], ca = cr[1], sa = cr[2], cb = cr[3], sb = cr[4];
      // powers in front of and behind the axial operator (kernels_m2l_rot.hip): M2L rho^-n, rho^-(j+1); M2M rho^-n, rho^j; L2L rho^n, rho^-j
      const double pre = OP == kRotL2L ? cr[5] : inv_rho, post = OP == kRotM2M ? cr[5] : inv_rho;
      const double pre2 = pre * pre, post2 = post * post;
      ConstFeed2<(rot2_stream_len(P) + kRotGroup - 1) / kRotGroup, rot2_needs_nop(P)> cf;
#ifndef FMMBEM_ROT2_EXP_NOARITH                          // experiment builds: where does a pass spend its time
      cf.start(w.stream, lane);
      z_rotation2<P>(a, b, cb, sb, odd ? cb : 1.0, odd ? sb : 0.0);
      fixed_rotation2<P, 0>(a, b, cf);
      z_rotation2<P>(a, b, ca, sa, odd ? ca : 1.0, odd ? sa : 0.0);
      fixed_rotation2<P, 1>(a, b, cf);
      scale2<P>(a, b, odd ? pre : 1.0, pre2);          // pre^n: E degree 2q, O degree 2q + 1
      {
        // axial translation, slot order by slot order (m2l_rot.hpp): own rows in place, the other parity's rows to the partner
        double pa[Q], pb_[Q];                            // E lanes: what the partner sent at the previous step, for slot t
#pragma unroll
        for (int i = 0; i < Q; ++i) { pa[i] = 0; pb_[i] = 0; }
        static_for<0, 2 * Q>([&](auto T_) FMMBEM_INLINE {
          constexpr int t = decltype(T_)::value;
          constexpr int q0 = rot2_qmin(t), q0o = rot2_qmin_other(t);
          double oa[Q], ob[Q], xa[Q], xb[Q];
          static_for<q0o, Q>([&](auto QO) FMMBEM_INLINE {
            constexpr int qo = decltype(QO)::value;
            static_for<(qo >= q0 ? 0 : 1), 2>([&](auto OT) FMMBEM_INLINE {
              constexpr int other = decltype(OT)::value;
              double s1 = 0, s2 = 0;
              static_for<q0, Q>([&](auto QI) FMMBEM_INLINE {
                constexpr int qi = decltype(QI)::value;
                constexpr int e = rot2_stage_base(P, 2) + rot2_axial_index(P, t, qo, other, qi);
                if constexpr (t != 0) cf.template fma2<e>(s1, a[rot2_sidx(qi, t)], s2, b[rot2_sidx(qi, t)]);
                else cf.template fma1<e>(s1, a[rot2_sidx(qi, t)]);
              });
              if constexpr (other) { xa[qo] = s1; xb[qo] = s2; } else { oa[qo] = s1; ob[qo] = s2; }
            });
          });
#pragma unroll
          for (int qo = q0; qo < Q; ++qo) { a[rot2_sidx(qo, t)] = oa[qo]; b[rot2_sidx(qo, t)] = ob[qo]; }
          // E lanes: the partner's sums of the previous step belong to this slot order
#pragma unroll
          for (int qo = q0; qo < Q; ++qo) {
            a[rot2_sidx(qo, t)] = fma(pa[qo], mask_e, a[rot2_sidx(qo, t)]);
            if (t) b[rot2_sidx(qo, t)] = fma(pb_[qo], mask_e, b[rot2_sidx(qo, t)]);
          }
          // the exchange; O lanes add what they receive into slot order t - 1 now, E lanes into t + 1 one step later
#pragma unroll
          for (int qo = q0o; qo < Q; ++qo) {
            const double ra = partner(xa[qo]), rb = t ? partner(xb[qo]) : 0.0;
            if (t >= 1) {
              a[rot2_sidx(qo, t - 1)] = fma(ra, mask_o, a[rot2_sidx(qo, t - 1)]);
              if (t >= 2) b[rot2_sidx(qo, t - 1)] = fma(rb, mask_o, b[rot2_sidx(qo, t - 1)]);
            }
            const bool exists = t + 1 <= 2 * qo + 1;     // compile-time after unrolling
            pa[qo] = exists ? ra : 0.0; pb_[qo] = exists ? rb : 0.0;
          }
#pragma unroll
          for (int qo = 0; qo < q0o; ++qo) { pa[qo] = 0; pb_[qo] = 0; }
        });
      }
      if constexpr (OP == kRotM2L) scale2<P>(a, b, odd ? post2 : post, post2);     // L'[j, k] *= rho^-(j+1)
      else scale2<P>(a, b, odd ? post : 1.0, post2);                                // rho^j (M2M), rho^-j (L2L)
      fixed_rotation2<P, 3>(a, b, cf);
      z_rotation2<P>(a, b, ca, -sa, odd ? ca : 1.0, odd ? -sa : 0.0);
      fixed_rotation2<P, 4>(a, b, cf);
      z_rotation2<P>(a, b, cb, -sb, odd ? cb : 1.0, odd ? -sb : 0.0);
#else
      a[0] += inv_rho + ca + sa + cb + sb + pre2 + post2 + mask_o + mask_e;
#endif

      // does the last target go on in the next pass?  (pairs past cnt repeat the item's last pair: pair 31 is the last pair)
      const bool cont_out = more && __builtin_amdgcn_readfirstlane(ntgt) == __shfl(tgt, lane_of_pair(kPairs - 1), kWave);

      // ---- add the pairs of each target: chain h of a target = its pairs h, h + 4, ... in order (pair space) ----
#ifdef FMMBEM_ROT2_EXP_NOREDUCE
      {
        double sx = 0, sy = 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) { sx += a[i]; sy += b[i]; }
        if (sx == 1.2345 && sy == 5.4321) d.L[(size_t)tgt * d.nslots * d.s_max] = double2{sx, sy};
        (void)smask; (void)nseg; (void)live;
      }
#else
      if constexpr (OP == kRotL2L) {
        // every pair is a child of its own: L[child] += the shifted parent, each lane its half, straight from the registers
        if (live) {
          double2* own = d.L + ((size_t)tgt * d.nslots + slot) * d.s_max;
          static_for<0, Q>([&](auto Q_) FMMBEM_INLINE {
            constexpr int q = decltype(Q_)::value;
            static_for<0, 2 * q + 2>([&](auto TT) FMMBEM_INLINE {
              constexpr int t = decltype(TT)::value;
              constexpr bool has_o = 2 * q + 1 < P, has_e = t >= 1;
              constexpr int io = has_o ? idx_of(2 * q + 1, t) : 0, ie_ = has_e ? idx_of(2 * q, t - 1) : 0;
              if (odd ? has_o : has_e) {
                double2* at = own + (odd ? io : ie_);
                double2 v = *at;
                v.x += a[rot2_sidx(q, t)]; v.y += b[rot2_sidx(q, t)];
                *at = v;
              }
            });
          });
        }
      } else if constexpr (OP == kRotM2M) {
        // a parent has at most eight children and an item holds whole parents in its one pass: lane = (parent, coefficient) adds the
        // parent's children in the chain order of the general case -- (c0 + c4) + (c1 + c5) and so on (kernels_m2l_rot.hip)
        __shared__ int seg_first[kPairs + 1], seg_tgt[kPairs];
        const int segid = __popc(smask & ((2u << pj) - 1u)) - 1;
        if (live && !odd && (pj == 0 || prev != tgt)) { seg_first[segid] = pj; seg_tgt[segid] = tgt; }
        if (lane == 0) seg_first[nseg] = cnt;
        static_for<0, NT>([&](auto T_) FMMBEM_INLINE {
          constexpr int tr = decltype(T_)::value;
          constexpr int kt = S - tr * KT < KT ? S - tr * KT : KT;
          static_for<0, Q>([&](auto Q_) FMMBEM_INLINE {
            constexpr int q = decltype(Q_)::value;
            static_for<0, 2 * q + 2>([&](auto TT) FMMBEM_INLINE {
              constexpr int t = decltype(TT)::value;
              constexpr bool has_o = 2 * q + 1 < P, has_e = t >= 1;
              constexpr int io = has_o ? idx_of(2 * q + 1, t) : -1, ie_ = has_e ? idx_of(2 * q, t - 1) : -1;
              constexpr bool in_o = io >= tr * KT && io < tr * KT + kt, in_e = ie_ >= tr * KT && ie_ < tr * KT + kt;
              if constexpr (in_o || in_e) {
                const bool mine = odd ? in_o : in_e;
                const int row = (odd ? io : ie_) - tr * KT;
                if (mine) tile[row][pj] = double2{a[rot2_sidx(q, t)], b[rot2_sidx(q, t)]};
              }
            });
          });
          wave_sync();
          for (int task = lane; task < nseg * kt; task += kWave) {
            const int sg = task / kt, c = task - sg * kt;
            const int f = seg_first[sg], e = seg_first[sg + 1];
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = f + u < e ? tile[c][f + u] : double2{0, 0};
            double2 sum;
            sum.x = ((v[0].x + v[4].x) + (v[1].x + v[5].x)) + ((v[2].x + v[6].x) + (v[3].x + v[7].x));
            sum.y = ((v[0].y + v[4].y) + (v[1].y + v[5].y)) + ((v[2].y + v[6].y) + (v[3].y + v[7].y));
            d.M[((size_t)seg_tgt[sg] * d.nslots + slot) * d.s_max + tr * KT + c] = sum;
          }
          wave_sync();
        });
      } else
      static_for<0, NT>([&](auto T_) FMMBEM_INLINE {
        constexpr int tr = decltype(T_)::value;
        constexpr int kt = S - tr * KT < KT ? S - tr * KT : KT;          // coefficients of this round
        constexpr int NTASK = (kt * kChains + kWave - 1) / kWave;
        // every lane puts the coefficients of its half that fall into this round into the tile: [coefficient][pair]
        static_for<0, Q>([&](auto Q_) FMMBEM_INLINE {
          constexpr int q = decltype(Q_)::value;
          static_for<0, 2 * q + 2>([&](auto TT) FMMBEM_INLINE {
            constexpr int t = decltype(TT)::value;
            constexpr bool has_o = 2 * q + 1 < P, has_e = t >= 1;
            constexpr int io = has_o ? idx_of(2 * q + 1, t) : -1, ie_ = has_e ? idx_of(2 * q, t - 1) : -1;
            constexpr bool in_o = io >= tr * KT && io < tr * KT + kt, in_e = ie_ >= tr * KT && ie_ < tr * KT + kt;
            if constexpr (in_o || in_e) {
              const bool mine = odd ? in_o : in_e;
              const int row = (odd ? io : ie_) - tr * KT;
              if (mine) tile[row][pj] = double2{a[rot2_sidx(q, t)], b[rot2_sidx(q, t)]};
            }
          });
        });
        wave_sync();
        unsigned rest = smask;
        for (int s = 0; s < nseg; ++s) {
          const int f = __ffs((int)rest) - 1;
          rest &= rest - 1;
          const int e = rest ? __ffs((int)rest) - 1 : cnt;
          const int stgt = __builtin_amdgcn_readlane(tgt, lane_of_pair(f));
          const bool head = s == 0 && cont_in;
          const bool tail = s == nseg - 1 && cont_out;
          const int q0 = head ? cont_q : 0;
          double2* Ls = (OP == kRotM2M ? d.M : d.L) + ((size_t)stgt * d.nslots + slot) * d.s_max + tr * KT;
          auto segment = [&](auto NU_) FMMBEM_INLINE {
            constexpr int NU = decltype(NU_)::value;                    // terms per chain
            double2 sum[NTASK], v[NTASK][NU];
            bool valid[NTASK];
            int cc[NTASK];
#pragma unroll
            for (int k = 0; k < NTASK; ++k) {
              const int task = lane + k * kWave, h = task & (kChains - 1);
              valid[k] = task < kt * kChains;
              cc[k] = valid[k] ? task / kChains : kt - 1;
              const int first = f + ((h - q0) & (kChains - 1));
              sum[k] = head ? carry[tr * KT + cc[k]][h] : double2{0, 0};
#pragma unroll
              for (int u = 0; u < NU; ++u) {
                const int l = first + u * kChains;
                v[k][u] = l < e ? tile[cc[k]][l] : double2{0, 0};
              }
            }
#pragma unroll
            for (int k = 0; k < NTASK; ++k) {
#pragma unroll
              for (int u = 0; u < NU; ++u) { sum[k].x += v[k][u].x; sum[k].y += v[k][u].y; }
            }
#pragma unroll
            for (int k = 0; k < NTASK; ++k) {
              const int h = lane & (kChains - 1);
              if (tail) { if (valid[k]) carry[tr * KT + cc[k]][h] = sum[k]; }
              else {
                sum[k].x += quad_swap<0xB1>(sum[k].x); sum[k].y += quad_swap<0xB1>(sum[k].y);
                sum[k].x += quad_swap<0x4E>(sum[k].x); sum[k].y += quad_swap<0x4E>(sum[k].y);
                if (h == 0 && valid[k]) {
                  if constexpr (OP == kRotL2L) { const double2 own = Ls[cc[k]]; sum[k].x += own.x; sum[k].y += own.y; }   // L[child] += ...
                  Ls[cc[k]] = sum[k];
                }
              }
            }
          };
          const int len = e - f;
          if (len <= 16) segment(std::integral_constant<int, 4>{});
          else segment(std::integral_constant<int, 8>{});
        }
        wave_sync();
      });
#endif
      cont_q = cont_out ? (nseg == 1 && cont_in ? cont_q : 0) + (cnt - flast) : 0;
      cont_in = cont_out;
      pi = npi; src = nsrc; cls = ncls; tgt = ntgt;
    }
  }
}

}  // namespace

unsigned rot_nop_orders_rot2() { return (unsigned)(FMMBEM_ROT_NOP_ORDERS); }      // fmmbem_stats.rot_nop_orders

bool shift_rot2_supported(int p) { return p >= 8 && p <= 12; }

// the shifts of the tree passes in the same form: op = kRotM2M or kRotL2L, items of ONE pass (at most 32 pairs of whole targets)
hipError_t launch_shift_rot2(const DevicePlan& d, const RotWork& w, int p, int op, hipStream_t s) {
  if (w.n_items <= 0) return hipSuccess;
  constexpr int CH = FMMBEM_ROT_XCD_CHUNK;
  const dim3 grid((w.n_items + 8 * CH - 1) / (8 * CH) * (8 * CH), d.n_act);
#define SHIFT2_CASE(PP) case PP: if (op == kRotM2M) hipLaunchKernelGGL((m2l_rot2_kernel<PP, kRotM2M>), grid, dim3(kWave), 0, s, d, w); \
                                 else hipLaunchKernelGGL((m2l_rot2_kernel<PP, kRotL2L>), grid, dim3(kWave), 0, s, d, w); break;
  switch (p) {
    SHIFT2_CASE(8) SHIFT2_CASE(9) SHIFT2_CASE(10) SHIFT2_CASE(11) SHIFT2_CASE(12)
    default: return hipErrorInvalidValue;
  }
#undef SHIFT2_CASE
  return hipGetLastError();
}

}  // namespace fmmbem
