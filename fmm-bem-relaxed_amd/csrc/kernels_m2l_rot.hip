// kernels_m2l_rot.hip -- M2L as rotation / axial translation / rotation back, one (target, source) PAIR per lane.
// Reference: LaplaceSpherical::M2L + evalLocal (kernel/LaplaceSpherical.hpp:296-329, 491-524), called once per LR_list pair
// and expansion (executor/EvalInteractionLazySparse.hpp:269-283).  The algebra and the constant tables are in m2l_rot.hpp.
//
// Mapping.  The M2L pairs are kept in box order of their targets (traversal order inside a target, as the oracle sums them).
// The host cuts that list into ITEMS: runs of whole targets of about 256 pairs (HostPlan::build_rot_items).  A wavefront takes
// an item 64 pairs at a time: lane = pair.  The lane has its source's multipole (S complex, straight from M -- no rescaled
// copy, no mh_prep pass) and the numbers of its translation class (1/rho, cos/sin alpha, cos/sin beta, rho), and runs the SAME
// straight-line instruction stream as every other lane on its own registers:
//     z-rotation by beta, fixed rotation, z-rotation by alpha, fixed rotation back, scale by rho^-n,
//     axial translation, scale by rho^-(j+1), fixed rotation, z-rotation by -alpha, fixed rotation back, z-rotation by -beta
// The rotation and translation constants are wave-uniform: one stream, read sixteen at a time into a VGPR pair and applied by
// v_fmac_f64_dpp row_newbcast (ConstFeed below).  No LDS in the arithmetic, no barriers, no divergence; ~3 000 FMAs per pair at
// p = 10 (15 400 for the reference's double sum, 280 x 55 in kernels_m2l.hip).
// Reduction.  The lanes of one target are then added in pair order as four interleaved chains -- chain h takes the target's
// pairs h, h + 4, h + 8, ... one after the other -- combined (0 + 1) + (2 + 3) at the end.  An item is a run of whole
// targets cut into passes of 64 pairs wherever they fall: a target that straddles a pass boundary carries its four chain sums
// through LDS into the next pass and goes on adding, so the sequence of additions of a target does not depend on where in an
// item it sits -- any cut of the list, and any shard of the operator, produces the same bits.  The coefficients go through
// a padded LDS tile [coefficient][lane], as many at a time as the LDS share of a wavefront holds (two rounds at p = 10).
// Every L is written exactly once -- no atomics.
#include "device_plan.hpp"
#include "m2l_rot.hpp"

#include <type_traits>

namespace fmmbem {

namespace {

// This source is compiled three times (csrc/Makefile): FMMBEM_ROT_OP = 0 the M2L kernel, 1 the M2M and 2 the L2L of the tree
// passes -- the same rotations around a different axial operator (m2l_rot.hpp), other source and destination arrays.
#ifndef FMMBEM_ROT_OP
#define FMMBEM_ROT_OP 0
#endif
constexpr int OP = FMMBEM_ROT_OP;
// one kernel name per operator, so that traces tell them apart
#if FMMBEM_ROT_OP == 0
#define ROT_KERNEL m2l_rot_kernel
#elif FMMBEM_ROT_OP == 1
#define ROT_KERNEL m2m_rot_kernel
#else
#define ROT_KERNEL l2l_rot_kernel
#endif

constexpr int kWave = 64;
constexpr int kChains = 4;                            // partial sums per (target, coefficient)
// tile row: 64 lanes + 4.  A task group of four lanes (the chains of one coefficient) reads four consecutive double2; with
// rows 4 double2 = 16 banks apart, the sixteen lanes that share an LDS cycle (4 coefficients x 4 chains) hit all 64 banks.
constexpr int kRow = kWave + 4;

#ifndef FMMBEM_ROT_XCD_CHUNK
#define FMMBEM_ROT_XCD_CHUNK 32
#endif

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int idx_of(int n, int m) { return n * (n + 1) / 2 + m; }

// the value of another lane of the same quad (DPP quad_perm: 0xB1 = lanes 1 0 3 2, 0x4E = lanes 2 3 0 1): a VALU move per
// half instead of the round trip through the LDS crossbar that __shfl_xor compiles to
template <int CTRL>
__device__ __forceinline__ double quad_swap(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// ---- the constant stream ------------------------------------------------------------------------------------------------
// Every FMA of the rotations and of the axial translation multiplies per-lane data by a WAVE-UNIFORM constant that is read
// once per pass, in sequence (m2l_rot.hpp build_rot_stream).  As SGPR operands (the obvious path, measured first) hipcc
// cannot hold them: 2 300 v_writelane + 2 300 v_readlane of spilled constants per pass at p = 10 beside 3 100 FMAs, whatever
// the fencing.  So the constants travel in VGPRs, sixteen to a register pair -- one 8-byte load per lane, lane & 15, puts the
// same sixteen numbers into every 16-lane row -- and the FMA takes lane k of its row as the multiplier:
//     v_fmac_f64_dpp acc, cv, src row_newbcast:k
// which issues at the rate of v_fma_f64 with an SGPR operand (tools/microbench/dpp64: 56.8 against 57.8 TFLOP/s).  No SGPR
// pressure, no lane moves; kRotAhead groups are in flight, their loads issued by hand (ConstFeed).
// compile-time loop: f(std::integral_constant<int, i>) for i = B .. E-1 (the stream positions must be constant expressions:
// they end up in the DPP control field of the instruction)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(static_cast<F&&>(f));
  }
}
#define FMMBEM_INLINE __attribute__((always_inline))

// Hazard the compiler cannot see inside the asm: a DPP source needs two wait states after a VALU write of that register
// (wrong results from p = 8 when the constants were C++ loads that the register allocator parked in AGPRs and restored with
// v_accvgpr_read right in front of their use).  With the loads in asm the ring is written by VMEM only and the allocator has
// so far left it alone -- but nothing obliges it to, so the build CHECKS the generated code (tools/check_rot_isa.py, run
// by the Makefile on every compile) and an order that shows the hazard is listed here to get "s_nop 1" in front of each FMA
// (+30% M2L time at one wavefront per SIMD, where the s_nop takes an issue slot of its own).
#ifndef FMMBEM_ROT_NOP_ORDERS
#define FMMBEM_ROT_NOP_ORDERS 0u                      // bit p - 1
#endif
constexpr bool rot_needs_nop(int P) { return ((FMMBEM_ROT_NOP_ORDERS) >> (P - 1)) & 1u; }

template <int NG, bool kNop>                          // NG: groups of sixteen in this order's stream
struct ConstFeed {
  static constexpr int NB = kRotAhead + 1;
  const double* base;                                 // this order's stream + (lane & 15)
  double cv[NB];
  // The loads are issued by hand: as plain C++ loads the compiler sinks each one to just in front of its first use
  // ("global_load; s_waitcnt vmcnt(0); FMAs" -- a full L2 latency per sixteen FMAs, 1.2 ms of the 1.3 at p = 10).  As
  // volatile asm they stay where they are written, kRotAhead groups ahead of their use; VMEM returns in order, so the wait in
  // front of group g is vmcnt(number of groups issued after g) -- kRotAhead, fewer at the end of the stream: nothing is
  // fetched past it, so the queue is empty when the last group has arrived and the kernel can put the NEXT pass's multipole
  // loads behind it.  (The compiler's own waits stay correct: more loads in flight than it counts can only make an in-order
  // counter wait longer.)
  // the group's address is the running pointer (moved every 32 groups) plus an immediate: as  base + 128 g  in C++ the
  // compiler materialises, hoists and then spills a VGPR pair per group
  static constexpr int kWindow = 32;                  // 32 groups x 128 bytes = the 12-bit immediate
  template <int G>
  __device__ __forceinline__ void issue() {
    if constexpr (G < NG) {
      if constexpr (G % kWindow == 0 && G != 0) {
        base += kWindow * kRotGroup;
        asm volatile("" : "+v"(base));                // one live pointer, not one per window
      }
      asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=&v"(cv[G % NB]) : "v"(base), "n"((G % kWindow) * kRotGroup * 8));
    }
  }
  __device__ __forceinline__ void start(const double* stream, int lane) {
    base = stream + (lane & 15);
    static_for<0, kRotAhead>([&](auto g) FMMBEM_INLINE { issue<decltype(g)::value>(); });
  }
  template <int K>
  static __device__ __forceinline__ void dpp_fma(double& acc, double c, double src) {
    if constexpr (kNop) asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(src), "n"(K));
    else asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(src), "n"(K));
  }
  // acc += stream[E] * src; at the head of a group of sixteen: fetch the group kRotAhead further on, wait for this one
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
  __device__ __forceinline__ void fma2(double& acc1, double src1, double& acc2, double src2) {   // one constant, two products
    fma1<E>(acc1, src1);
    constexpr int g = E / kRotGroup, k = E % kRotGroup;
    dpp_fma<k>(acc2, cv[g % NB], src2);
  }
};

// out = R in for every degree-n block; STAGE picks the stream segment: conj(X) (0, 3) or X^T (1, 4), signs folded in
template <int P, int STAGE, class Feed>
__device__ __forceinline__ void fixed_rotation(double (&a)[P * (P + 1) / 2], double (&b)[P * (P + 1) / 2], Feed& cf) {
  static_for<1, P>([&](auto N) FMMBEM_INLINE {         // degree 0 is the identity
    constexpr int n = decltype(N)::value;
    double na[P], nb[P];
    // first the rows that read a[n, .] (n + m even), then the rows that read b[n, .] (m2l_rot.hpp rot_index: the a[] of the degree
    // are dead before the second class's outputs need registers); within a class two output rows at a time, their terms side by
    // side: consecutive FMAs feed four accumulators in turn
    static_for<0, 2>([&](auto C_) FMMBEM_INLINE {
      constexpr int cls = decltype(C_)::value, cnt = rot_class_count(n, cls);
      static_for<0, (cnt + 1) / 2>([&](auto I_) FMMBEM_INLINE {
        constexpr int i0 = 2 * decltype(I_)::value;
        double sa[2] = {0, 0}, sb[2] = {0, 0};
        static_for<0, n + 1>([&](auto Q) FMMBEM_INLINE {
          constexpr int mp = decltype(Q)::value;
          static_for<0, 2>([&](auto R_) FMMBEM_INLINE {
            constexpr int r = decltype(R_)::value;
            if constexpr (i0 + r < cnt) {
              constexpr int m = rot_class_row(n, cls, i0 + r);
              if constexpr (rot_live(n, m, mp)) {
                constexpr int e = rot_stage_base(P, STAGE, OP) + rot_index(n, m, mp);
                const double src = cls == 0 ? a[idx_of(n, mp)] : b[idx_of(n, mp)];       // class 1 has no live mp = 0
                if constexpr ((rot_kk(n, m, mp) & 1) == 0) cf.template fma1<e>(sa[r], src); else cf.template fma1<e>(sb[r], src);
              }
            }
          });
        });
        na[rot_class_row(n, cls, i0)] = sa[0]; nb[rot_class_row(n, cls, i0)] = sb[0];
        if constexpr (i0 + 1 < cnt) { na[rot_class_row(n, cls, i0 + 1)] = sa[1]; nb[rot_class_row(n, cls, i0 + 1)] = sb[1]; }
      });
    });
#pragma unroll
    for (int m = 0; m <= n; ++m) { a[idx_of(n, m)] = na[m]; b[idx_of(n, m)] = nb[m]; }
  });
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
// temporaries.  Wavefronts per SIMD asked of the compiler: four up to p = 4, two at 5-8, ONE from p = 9, where the overflow goes
// to AGPRs (no scratch up to p = 11).  Same box, present kernel, M2L ms at N = 1M: p = 4 four 0.091, three 0.093, five 0.180;
// p = 5 two 0.111, three 0.120, four 0.161; p = 6 two 0.169, three 0.180; p = 7 one 0.284, two 0.236, three 0.367; p = 8 one
// 0.399, two 0.337; p = 9 one 0.461, two 0.472.  Measured with the first (SGPR-constant) kernel at N = 1M, M2L ms: p = 7:
// two 0.41, one 0.50; p = 8: two 0.72, one 0.70; p = 9..12: two 1.00 / 1.51 / 3.28 / 5.59, one 0.96 / 1.29 / 2.07 / 2.73; with the
// present one at p = 10: two 1.27, one 1.07 (before the later steps that took it to 0.56).
#ifndef FMMBEM_ROT_OCC
#define FMMBEM_ROT_OCC(P) ((P) <= 4 ? 4 : (P) <= 8 ? 2 : 1)
#endif
constexpr int rot_waves(int P) { return FMMBEM_ROT_OCC(P); }

// The NEXT pass's operands, fetched while this pass is still busy.  At one wavefront per SIMD nothing hides a load: the chain
// pair index -> source box -> 55 gathered 16-byte loads (every lane its own cache lines) + the class record cost a quarter of
// the pass at p = 10.  The AGPRs are free once the last fixed rotation is through (the allocator parks values there during
// the arithmetic only), the VMEM queue is empty behind the last group of constants, and the z rotation and the reduction that
// follow use neither: the loads go out there, as asm so that they stay there, into AGPRs, and the next pass begins by waiting
// for them and moving them over.  PF double2 of the multipole are fetched ahead -- all of them at p = 8, 9, 32 of 55 at p = 10
// (36 until the rotation rows were taken two at a time: four accumulators instead of two, and 36 then spills; same box, M2L ms:
// row by row with 36: 0.547-0.558; two rows with 36: 0.606, 35: 0.522-0.534, 34: 0.561, 32: 0.533-0.536 and no scratch),
// 16 at p = 11, none at p = 12: what the AGPRs hold beside the allocator's own use of them (more: the prefetched values are
// spilled to scratch at once, which tools/check_rot_isa.py reports) -- and the rest at the head of the pass.  The tree-pass
// operators (OP != M2L) take items of one pass and fetch nothing ahead.
typedef double v2d __attribute__((ext_vector_type(2)));
#ifndef FMMBEM_ROT_PF
#define FMMBEM_ROT_PF(P) ((P) <= 9 ? 64 : (P) == 10 ? 32 : (P) == 11 ? 16 : 0)
#endif
constexpr int rot_prefetch(int P) { return (rot_waves(P) > 1 || OP != kRotM2L) ? 0 : (P * (P + 1) / 2 < FMMBEM_ROT_PF(P) ? P * (P + 1) / 2 : FMMBEM_ROT_PF(P)); }

template <int PF>
struct NextPass {
  v2d m[PF > 0 ? PF : 1];                             // multipole of the lane's next pair
  v2d r01, r23, r45;                                  // its class record: 1/rho, cos a | sin a, cos b | sin b, rho
  const double2* from;
  __device__ __forceinline__ void issue(const double2* M, const double* rec) {
    from = M;                                         // (a member: asm operands inside a generic lambda do not capture locals)
    static_for<0, PF>([&](auto I) FMMBEM_INLINE {
      constexpr int i = decltype(I)::value;
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(m[i]) : "v"(from), "n"(i * 16));
    });
    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(r01) : "v"(rec));
    asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=a"(r23) : "v"(rec));
    asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=a"(r45) : "v"(rec));
  }
  __device__ __forceinline__ void wait() {            // everything after this in program order sees the loaded values
    asm volatile("s_waitcnt vmcnt(0)" : "+a"(r01), "+a"(r23), "+a"(r45));
    static_for<0, PF>([&](auto I) FMMBEM_INLINE { asm volatile("" : "+a"(m[decltype(I)::value])); });
  }
};

// coefficients per reduction round: as many as fit the wavefront's share of the 160 KB of LDS at the occupancy asked for
constexpr int rot_tile(int P) {
  const int S = P * (P + 1) / 2, budget = 160 * 1024 / (4 * rot_waves(P)) - 2048;
  for (int R = 1; R <= S; ++R) {
    const int kt = (S + R - 1) / R;
    if (kt * kRow * 16 + S * kChains * 16 + 1024 <= budget) return kt;
  }
  return 1;
}

template <int P>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(rot_waves(P), rot_waves(P)))) void ROT_KERNEL(const DevicePlan d, const RotWork w) {
  constexpr int S = P * (P + 1) / 2;
  constexpr int KT = rot_tile(P);
  constexpr int NT = (S + KT - 1) / KT;
  __shared__ double2 tile[KT][kRow];
  __shared__ double2 carry[S][kChains];               // chain sums of a target that goes on in the next pass
  const int lane = threadIdx.x;
  if constexpr (OP == kRotM2L) {                      // column kWave of every tile row stays zero: what a chain reads past its segment's end
    for (int c = lane; c < KT; c += kWave) tile[c][kWave] = double2{0, 0};
    wave_sync();
  }
  // workgroups are dealt round-robin to the 8 XCDs: keep runs of consecutive items (neighbouring targets, which share
  // sources) on one XCD's L2.  gridDim.x is a multiple of 8 * CH.
  constexpr int CH = FMMBEM_ROT_XCD_CHUNK;
  const int rnd = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
  const int item = (rnd / CH) * 8 * CH + xcd * CH + rnd % CH;
  if (item >= w.n_items) return;
  const int ib = w.item_ptr[item], ie = w.item_ptr[item + 1];

  constexpr int PF = rot_prefetch(P);
  constexpr bool kAhead = rot_waves(P) == 1 && OP == kRotM2L;   // fetch the next pass's operands ahead (NextPass); with several
                                                      // wavefronts per SIMD the others cover the loads, and registers are scarce
  // M2L: a wavefront takes the item's pairs through every active expansion slot in turn (tens of passes per SIMD either way).
  // The shifts: one slot per wavefront (blockIdx.y) -- a level is a single pass on a chip that is not full, and what one waits for
  // is the length of one wavefront's program
  const int q_begin = OP == kRotM2L ? 0 : (int)blockIdx.y, q_end = OP == kRotM2L ? d.n_act : q_begin + 1;
  for (int q = q_begin; q < q_end; ++q) {
    const int slot = d.act[q];
    const double2* Mslot = (OP == kRotL2L ? d.L : d.M) + (size_t)slot * d.s_max;      // the operand: M, or the parent's L
    const size_t box_stride = (size_t)d.nslots * d.s_max;
    bool cont_in = false;                             // the first lane's target began in an earlier pass ...
    int cont_q = 0;                                   // ... which took this many of its pairs
    int pi = ib + lane < ie ? ib + lane : ie - 1;     // lanes past the end repeat the item's last pair
    int src = w.src[pi], cls = w.cls[pi], tgt = w.tgt[pi];
    NextPass<PF> nx;
    if constexpr (kAhead) nx.issue(Mslot + (size_t)src * box_stride, w.rec + (size_t)cls * 8);
    for (int pb = ib; pb < ie; pb += kWave) {
      const int cnt = ie - pb < kWave ? ie - pb : kWave;
      const bool live = lane < cnt;
      // ---- segments (targets) of this pass: bit l of smask = lane l begins a target ----
      const int prev = __shfl_up(tgt, 1, kWave);
      const unsigned long long smask = __ballot(live && (lane == 0 || prev != tgt));
      const int nseg = __popcll(smask);
      const int flast = 63 - __clzll(smask);          // first lane of the last segment
      // the pairs of the next pass (the last pass reads its own again: nothing waits for that)
      const bool more = pb + kWave < ie;
      const int npi = more ? (pb + kWave + lane < ie ? pb + kWave + lane : ie - 1) : pi;
      // asm: as C++ loads they would sink to their use, behind the arithmetic.  Into AGPRs: a VGPR destination the compiler,
      // for which the asm has delivered when it is over, parks in an AGPR at once -- before the data is there
      // (tools/check_rot_isa.py looks for exactly that in every build)
      int nsrc, ncls, ntgt;
      if constexpr (kAhead) {
        asm volatile("global_load_dword %0, %1, off" : "=a"(nsrc) : "v"(w.src + npi));
        asm volatile("global_load_dword %0, %1, off" : "=a"(ncls) : "v"(w.cls + npi));
        asm volatile("global_load_dword %0, %1, off" : "=a"(ntgt) : "v"(w.tgt + npi));
      } else { nsrc = w.src[npi]; ncls = w.cls[npi]; ntgt = w.tgt[npi]; }

      // ---- this lane's pair ----
      double a[S], b[S];
      if constexpr (PF < S) {                          // what was not fetched ahead
        const double2* M = Mslot + (size_t)src * box_stride;
#pragma unroll
        for (int i = PF; i < S; ++i) { const double2 v = M[i]; a[i] = v.x; b[i] = v.y; }
      }
      double inv_rho, ca, sa, cb, sb, rho;
      if constexpr (kAhead) {
        nx.wait();
#pragma unroll
        for (int i = 0; i < PF; ++i) { a[i] = nx.m[i].x; b[i] = nx.m[i].y; }
        inv_rho = nx.r01.x; ca = nx.r01.y; sa = nx.r23.x; cb = nx.r23.y; sb = nx.r45.x; rho = nx.r45.y;
      } else {
        const double* cr = w.rec + (size_t)cls * 8;
        inv_rho = cr[0]; ca = cr[1]; sa = cr[2]; cb = cr[3]; sb = cr[4]; rho = cr[5];
      }
      ConstFeed<(rot_stream_len(P, OP) + kRotGroup - 1) / kRotGroup, rot_needs_nop(P)> cf;
      cf.start(w.stream, lane);
      z_rotation<P>(a, b, cb, sb);
      fixed_rotation<P, 0>(a, b, cf);
      z_rotation<P>(a, b, ca, sa);
      fixed_rotation<P, 1>(a, b, cf);
      {                                               // M2L, M2M: M''[n,m] = rho^-n M'[n,m];  L2L: rho^n
        const double base = OP == kRotL2L ? rho : inv_rho;
        double r = base;
#pragma unroll
        for (int n = 1; n < P; ++n) {
#pragma unroll
          for (int m = 0; m <= n; ++m) { a[idx_of(n, m)] *= r; if (m) b[idx_of(n, m)] *= r; }
          r *= base;
        }
      }
      // axial translation, order by order: out[j,k] = sum_n T[j,n,k] in[n,k], n over the operator's row (m2l_rot.hpp)
      static_for<0, P>([&](auto K_) FMMBEM_INLINE {
        constexpr int k = decltype(K_)::value;
        double la[P], lb[P];
        // two output rows j at a time, n ascending with the two rows' terms side by side (m2l_rot.hpp axial_index)
        static_for<0, (P - k + 1) / 2>([&](auto J) FMMBEM_INLINE {
          constexpr int j0 = k + 2 * decltype(J)::value;
          double s1[2] = {0, 0}, s2[2] = {0, 0};
          static_for<k, P>([&](auto N) FMMBEM_INLINE {
            constexpr int n = decltype(N)::value;
            static_for<0, 2>([&](auto R_) FMMBEM_INLINE {
              constexpr int r = decltype(R_)::value, j = j0 + r;
              if constexpr (j < P) {
                if constexpr (n >= axial_row_begin(P, OP, k, j) && n < axial_row_end(P, OP, k, j)) {
                  constexpr int e = rot_stage_base(P, 2, OP) + axial_index(P, OP, k, j, n);
                  if constexpr (k != 0) cf.template fma2<e>(s1[r], a[idx_of(n, k)], s2[r], b[idx_of(n, k)]);
                  else cf.template fma1<e>(s1[r], a[idx_of(n, k)]);
                }
              }
            });
          });
          la[j0] = s1[0]; lb[j0] = s2[0];
          if constexpr (j0 + 1 < P) { la[j0 + 1] = s1[1]; lb[j0 + 1] = s2[1]; }
        });
#pragma unroll
        for (int j = k; j < P; ++j) { a[idx_of(j, k)] = la[j]; b[idx_of(j, k)] = lb[j]; }
      });
      {                                               // M2L: L'[j,k] *= rho^-(j+1);  M2M: rho^j;  L2L: rho^-j
        const double base = OP == kRotM2M ? rho : inv_rho;
        double r = OP == kRotM2L ? base : 1.0;
#pragma unroll
        for (int j = OP == kRotM2L ? 0 : 1; j < P; ++j) {
          if (OP != kRotM2L) r *= base;
#pragma unroll
          for (int k = 0; k <= j; ++k) { a[idx_of(j, k)] *= r; if (k) b[idx_of(j, k)] *= r; }
          if (OP == kRotM2L) r *= base;
        }
      }
      fixed_rotation<P, 3>(a, b, cf);
      z_rotation<P>(a, b, ca, -sa);
      fixed_rotation<P, 4>(a, b, cf);
      if constexpr (kAhead) {
        asm volatile("s_waitcnt vmcnt(0)" : "+a"(nsrc), "+a"(ncls), "+a"(ntgt));    // long since there
        if (more) nx.issue(Mslot + (size_t)nsrc * box_stride, w.rec + (size_t)ncls * 8);
      }
      z_rotation<P>(a, b, cb, -sb);
      // does the last target go on in the next pass?  (lanes past cnt repeat the item's last pair: lane 63 is the last pair)
      const bool cont_out = more && __builtin_amdgcn_readfirstlane(ntgt) == __shfl(tgt, kWave - 1, kWave);

      // ---- add the lanes of each target: chain h of a target = its pairs h, h + 4, ... in order ----
      if constexpr (OP == kRotL2L) {
        // every lane is a child of its own: L[child] += the shifted parent, straight from the registers (an item of the
        // shifts is ONE pass -- plan.hip cuts them so -- and nothing is carried)
        if (live) {
          double2* own = d.L + ((size_t)tgt * d.nslots + slot) * d.s_max;
#pragma unroll
          for (int i = 0; i < S; ++i) { double2 v = own[i]; v.x += a[i]; v.y += b[i]; own[i] = v; }
        }
      } else if constexpr (OP == kRotM2M) {
        // a parent has at most eight children, an item holds whole parents in one pass: lane = (parent, coefficient) adds the
        // parent's children in the chain order of the general case -- (c0 + c4) + (c1 + c5) and so on -- eight LDS reads, no loop
        __shared__ int seg_first[kWave + 1], seg_tgt[kWave];
        const int segid = __popcll(smask & ((2ull << lane) - 1)) - 1;
        if (live && (lane == 0 || prev != tgt)) { seg_first[segid] = lane; seg_tgt[segid] = tgt; }
        if (lane == 0) seg_first[nseg] = cnt;
        static_for<0, NT>([&](auto T_) FMMBEM_INLINE {
          constexpr int t = decltype(T_)::value;
          constexpr int kt = S - t * KT < KT ? S - t * KT : KT;
#pragma unroll
          for (int c = 0; c < kt; ++c) tile[c][lane] = double2{a[t * KT + c], b[t * KT + c]};
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
            d.M[((size_t)seg_tgt[sg] * d.nslots + slot) * d.s_max + t * KT + c] = sum;
          }
          wave_sync();
        });
      } else
      static_for<0, NT>([&](auto T_) FMMBEM_INLINE {
        constexpr int t = decltype(T_)::value;
        constexpr int kt = S - t * KT < KT ? S - t * KT : KT;          // coefficients of this round
        constexpr int NTASK = (kt * kChains + kWave - 1) / kWave;       // (coefficient, chain) tasks per lane and segment
#pragma unroll
        for (int c = 0; c < kt; ++c) tile[c][lane] = double2{a[t * KT + c], b[t * KT + c]};     // b[n,0] = 0 by construction
        wave_sync();
        // segment by segment (everything about a segment is wave-uniform: scalar registers, no tables); a lane takes NTASK
        // tasks, all their loads are issued before the first addition
        unsigned long long rest = smask;
        for (int s = 0; s < nseg; ++s) {
          const int f = __ffsll((long long)rest) - 1;
          rest &= rest - 1;
          const int e = rest ? __ffsll((long long)rest) - 1 : cnt;
          const int stgt = __builtin_amdgcn_readlane(tgt, f);
          const bool head = s == 0 && cont_in;                          // goes on from the previous pass
          const bool tail = s == nseg - 1 && cont_out;                  // goes on in the next pass
          const int q0 = head ? cont_q : 0;
          double2* Ls = (OP == kRotM2M ? d.M : d.L) + ((size_t)stgt * d.nslots + slot) * d.s_max + t * KT;
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
              sum[k] = head ? carry[t * KT + cc[k]][h] : double2{0, 0};
#pragma unroll
              for (int u = 0; u < NU; ++u) {
                const int l = first + u * kChains;
                v[k][u] = tile[cc[k]][l < e ? l : kWave];            // past the segment's end: the row's zero column (one select on the index, not four on the value)
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
              if (tail) { if (valid[k]) carry[t * KT + cc[k]][h] = sum[k]; }      // keep the chains apart
              else {
                // chains 0..3 of one (target, coefficient) sit in four consecutive lanes: (0 + 1) + (2 + 3)
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
          else if (len <= 32) segment(std::integral_constant<int, 8>{});
          else if (len <= 48) segment(std::integral_constant<int, 12>{});
          else segment(std::integral_constant<int, 16>{});
        }
        wave_sync();
      });
      cont_q = cont_out ? (nseg == 1 && cont_in ? cont_q : 0) + (cnt - flast) : 0;
      cont_in = cont_out;
      pi = npi; src = nsrc; cls = ncls; tgt = ntgt;
    }
  }
}

#if FMMBEM_ROT_OP == 0
// boxes that hold a local expansion but have no M2L source of their own (they only inherit from the parent): L = 0
__global__ void m2l_rot_zero_kernel(const DevicePlan d, int S) {
  const int box = d.rot_empty[blockIdx.x];
  for (int q = 0; q < d.n_act; ++q) {
    double2* L = d.L + ((size_t)box * d.nslots + d.act[q]) * d.s_max;
    for (int i = threadIdx.x; i < S; i += blockDim.x) L[i] = double2{0, 0};
  }
}
#endif

hipError_t launch_rot(const DevicePlan& d, const RotWork& w, int p, hipStream_t s) {
  if (w.n_items <= 0) return hipSuccess;
  constexpr int CH = FMMBEM_ROT_XCD_CHUNK;
  const int grid = (w.n_items + 8 * CH - 1) / (8 * CH) * (8 * CH);
#define ROT_CASE(PP) case PP: hipLaunchKernelGGL((ROT_KERNEL<PP>), dim3(grid, OP == kRotM2L ? 1 : d.n_act), dim3(kWave), 0, s, d, w); break;
  switch (p) {
#ifdef FMMBEM_ROT_ONLY                                 // experiment builds (tools/rot_variant.sh): one order, 20 s instead of 3.5 min
    ROT_CASE(FMMBEM_ROT_ONLY)
#elif FMMBEM_ROT_OP == 0
    ROT_CASE(1) ROT_CASE(2) ROT_CASE(3) ROT_CASE(4) ROT_CASE(5) ROT_CASE(6)
    ROT_CASE(7) ROT_CASE(8) ROT_CASE(9) ROT_CASE(10) ROT_CASE(11) ROT_CASE(12)
#else
    ROT_CASE(1) ROT_CASE(2) ROT_CASE(3) ROT_CASE(4) ROT_CASE(5) ROT_CASE(6)
    ROT_CASE(7) ROT_CASE(8) ROT_CASE(9) ROT_CASE(10) ROT_CASE(11) ROT_CASE(12)
#endif
    default: return hipErrorInvalidValue;
  }
#undef ROT_CASE
  return hipGetLastError();
}

}  // namespace

// the orders of this object that were built with "s_nop 1" in front of their DPP FMAs (fmmbem_stats.rot_nop_orders)
#if FMMBEM_ROT_OP == 0
unsigned rot_nop_orders_m2l() { return (unsigned)(FMMBEM_ROT_NOP_ORDERS); }
#elif FMMBEM_ROT_OP == 1
unsigned rot_nop_orders_m2m() { return (unsigned)(FMMBEM_ROT_NOP_ORDERS); }
#else
unsigned rot_nop_orders_l2l() { return (unsigned)(FMMBEM_ROT_NOP_ORDERS); }
#endif

#if FMMBEM_ROT_OP == 0
bool m2l_rot_supported(int p) { return p >= 1 && p <= kRotPmax; }
// the orders that run one wavefront per SIMD take the long items (host_plan.cpp build_rot_items)
bool m2l_rot_long_items(int p) { return p >= 1 && p <= kRotPmax && rot_waves(p) == 1; }

// L = 0 for the boxes that hold a local expansion but have no M2L source
hipError_t launch_m2l_rot_zero(const DevicePlan& d, int p, hipStream_t s) {
  if (d.n_rot_empty > 0) hipLaunchKernelGGL(m2l_rot_zero_kernel, dim3(d.n_rot_empty), dim3(kWave), 0, s, d, p * (p + 1) / 2);
  return hipGetLastError();
}

hipError_t launch_m2l_rot(const DevicePlan& d, const DevicePlan* d_dev, int p, hipStream_t s) {
  (void)d_dev;
  if (hipError_t e = launch_m2l_rot_zero(d, p, s); e != hipSuccess) return e;
  RotWork w;
  w.src = d.rot_src; w.cls = d.rot_cls; w.tgt = d.rot_tgt;
  if (m2l_rot_long_items(p)) { w.item_ptr = d.rot_item_ptr_long; w.n_items = d.n_rot_items_long; }
  else { w.item_ptr = d.rot_item_ptr; w.n_items = d.n_rot_items; }
  w.rec = d.rot_cls_rec; w.stream = d.rot_tab + d.rot_tab_off[p - 1];
  if (hipError_t e = launch_rot(d, w, p, s); e != hipSuccess) return e;
  return hipGetLastError();
}
#elif FMMBEM_ROT_OP == 1
bool shift_rot_supported(int p) { return p >= kShiftRotPmin && p <= kRotPmax; }
hipError_t launch_m2m_rot(const DevicePlan& d, const RotWork& w, int p, hipStream_t s) { return launch_rot(d, w, p, s); }
#else
hipError_t launch_l2l_rot(const DevicePlan& d, const RotWork& w, int p, hipStream_t s) { return launch_rot(d, w, p, s); }
#endif

}  // namespace fmmbem
