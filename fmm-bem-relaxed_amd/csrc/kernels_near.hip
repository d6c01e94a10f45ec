// kernels_near.hip -- gfx950 kernels of the near field (P2P):
//   near_assemble   plan-build: A[i,j] = K(target_i, source_j) for every leaf-leaf block
//                   (reference: P2P_Lazy::to_matrix, executor/EvalP2P.hpp:47-98, calling
//                    LaplaceSphericalBEM::operator(), kernel/LaplaceSphericalBEM.hpp:159-297)
//   gather_x        x_tree[i] = x[perm[i]]        (EvalInteractionLazySparse.hpp:137-138)
//   near_spmv       y_tree = A_near * x_tree      (Matvec<>, include/Matvec.hpp:14-33) -- HBM-bound hot kernel
//   scatter_y       y[perm[i]] = y_tree[i]        (EvalInteractionLazySparse.hpp:146-148)
#include "device_launch.hpp"

#include <algorithm>
#include <cstdlib>

namespace fmmbem {

namespace {

constexpr int kWave = 64;
typedef double dvec2 __attribute__((ext_vector_type(2)));   // native 16-B vector (nontemporal builtin needs it)

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

__device__ inline void wave_lds_fence() {             // LDS writes of this wavefront visible to its other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct V3 { double x, y, z; };
__device__ inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline V3 cross(V3 u, V3 v) { return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; }
__device__ inline double norm(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ inline V3 mul3(const double* M, V3 v) {       // row-major 3x3 times vector (include/Mat3.hpp:76-82)
  return {M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z};
}

// 5-point Gauss-Legendre in the polar angle along one triangle edge
// (examples/BEM/SemiAnalytical.hpp:13-71, LAPLACE branch; only G is needed by the Laplace near field)
__device__ inline double edge_angle_integral(double z, double x, double v1, double v2) {
  const double t1 = atan2(v1, x), t2 = atan2(v2, x);
  const double dt = t2 - t1, tm = (t2 + t1) / 2;
  const double az = fabs(z);
  const double xk[5] = {-9.06179846e-01, -5.38469310e-01, 1.78162900e-17, 9.06179846e-01, 5.38469310e-01};
  const double wk[5] = {0.23692689, 0.47862867, 0.56888889, 0.23692689, 0.47862867};
  double G = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const double tk = dt / 2 * xk[i] + tm;
    const double Rt = x / cos(tk);
    const double R = sqrt(Rt * Rt + z * z);
    G += wk[i] * (R - az) * dt / 2;
  }
  return G;
}

// contribution of one edge v1->v2, in the panel plane with the collocation point at the origin and
// height p above the plane (examples/BEM/SemiAnalytical.hpp:81-145)
__device__ inline double edge_term(V3 v1, V3 v2, double p) {
  const V3 e = sub(v2, v1);
  const double len = norm(e);
  const V3 u = {e.x / len, e.y / len, e.z / len};
  const V3 o = cross(V3{0, 0, 1}, u);
  double R[9] = {o.x, u.x, 0, o.y, u.y, 0, o.z, u.z, 1};
  V3 a = mul3(R, v1);
  if (a.x < 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = -R[i];
    R[8] = 1.;
    a = mul3(R, v1);
  }
  const V3 b = mul3(R, v2);
  if ((a.y > 0 && b.y < 0) || (a.y < 0 && b.y > 0))
    return edge_angle_integral(p, a.x, 0, a.y) + edge_angle_integral(p, a.x, b.y, 0);
  return -edge_angle_integral(p, a.x, a.y, b.y);
}

// int_panel 1/|x - y| dS(y), semi-analytically (examples/BEM/SemiAnalytical.hpp:148-203)
__device__ inline double semi_analytic_G(V3 y0, V3 y1, V3 y2, V3 x) {
  const V3 xp = sub(x, y0), e1 = sub(y1, y0), e2 = sub(y2, y0);
  V3 X = e1, Z = cross(e1, e2);
  const double xn = norm(X), zn = norm(Z);
  X = {X.x / xn, X.y / xn, X.z / xn};
  Z = {Z.x / zn, Z.y / zn, Z.z / zn};
  const V3 Y = cross(Z, X);
  const double rot[9] = {X.x, X.y, X.z, Y.x, Y.y, Y.z, Z.x, Z.y, Z.z};
  const V3 q0 = mul3(rot, V3{0, 0, 0}), q1 = mul3(rot, e1), q2 = mul3(rot, e2), xq = mul3(rot, xp);
  const V3 f0 = {q0.x - xq.x, q0.y - xq.y, q0.z}, f1 = {q1.x - xq.x, q1.y - xq.y, q1.z}, f2 = {q2.x - xq.x, q2.y - xq.y, q2.z};
  return edge_term(f0, f1, xq.z) + edge_term(f1, f2, xq.z) + edge_term(f2, f0, xq.z);
}

// the 16-point "K_fine" rule keyed 17 (examples/BEM/GaussQuadrature.hpp:86-116), barycentric
__constant__ double kFine[16][4] = {
    {1. / 3, 1. / 3, 1. / 3, 0.144315607677787},
    {0.081414823414554, 0.459292588292723, 0.459292588292723, 0.095091634267285},
    {0.459292588292723, 0.081414823414554, 0.459292588292723, 0.095091634267285},
    {0.459292588292723, 0.459292588292723, 0.081414823414554, 0.095091634267285},
    {0.658861384496480, 0.170569307751760, 0.170569307751760, 0.103217370534718},
    {0.170569307751760, 0.658861384496480, 0.170569307751760, 0.103217370534718},
    {0.170569307751760, 0.170569307751760, 0.658861384496480, 0.103217370534718},
    {0.898905543365938, 0.050547228317031, 0.050547228317031, 0.032458497623198},
    {0.050547228317031, 0.898905543365938, 0.050547228317031, 0.032458497623198},
    {0.050547228317031, 0.050547228317031, 0.898905543365938, 0.032458497623198},
    {0.008394777409958, 0.263112829634638, 0.728492392955404, 0.027230314174435},
    {0.008394777409958, 0.728492392955404, 0.263112829634638, 0.027230314174435},
    {0.263112829634638, 0.008394777409958, 0.728492392955404, 0.027230314174435},
    {0.263112829634638, 0.728492392955404, 0.008394777409958, 0.027230314174435},
    {0.728492392955404, 0.008394777409958, 0.263112829634638, 0.027230314174435},
    {0.728492392955404, 0.263112829634638, 0.008394777409958, 0.027230314174435}};

// One near-matrix entry: target centroid t with BC flag, source panel j (tree index).
// kernel/LaplaceSphericalBEM.hpp:273-297 -> eval_G (:159-205) / eval_dGdn (:208-264)
// In two parts, so that the assembly can run the expensive regime with full wavefronts: laplace_entry_far gives the entry of a
// pair in the far regime (the K stored Gauss points; also the 2 pi of a NORMAL_DERIV self pair) or says `deferred`;
// laplace_entry_near gives the near regime (semi-analytic G, :166-178; the 16-point rule for dG/dn, :222-243).
// the far regime's arithmetic on a source panel held in registers (one text for both callers: the bits must not depend on who asks)
template <class Quad>
__device__ __forceinline__ double laplace_far_from(V3 t, int tbc, V3 c, double A, V3 nrm, int nq, const double* qw, Quad&& quad, bool& deferred) {
  const double dist = norm(sub(t, c));
  const bool nearby = sqrt(2 * A) / dist >= 0.5;
  deferred = false;
  if (tbc == 0) {                                   // POTENTIAL target: int G
    if (nearby) { deferred = true; return 0; }
    double r = 0;
    for (int q = 0; q < nq; ++q) {
      const V3 qp = quad(q);
      r += qw[q] * A / norm(sub(t, qp));
    }
    return r;
  }
  // NORMAL_DERIV target: int dG/dn
  if (dist < 1e-8) return 2 * M_PI;
  if (nearby) { deferred = true; return 0; }
  double r = 0;
  for (int q = 0; q < nq; ++q) {
    const V3 qp = quad(q);
    const V3 dx = sub(qp, t);
    const double r2 = dx.x * dx.x + dx.y * dx.y + dx.z * dx.z;
    r += qw[q] * A * (dx.x * nrm.x + dx.y * nrm.y + dx.z * nrm.z) / (r2 * sqrt(r2));
  }
  return r;
}
__device__ inline double laplace_entry_far(const DevicePlan& d, V3 t, int tbc, int64_t j, bool& deferred) {
  const int64_t N = d.n;
  const V3 c = {d.cx[j], d.cy[j], d.cz[j]};
  const V3 nrm = tbc ? V3{d.nx[j], d.ny[j], d.nz[j]} : V3{0, 0, 0};
  return laplace_far_from(t, tbc, c, d.area[j], nrm, d.nq, d.qw,
                          [&](int q) { return V3{d.quad[(q * 3 + 0) * N + j], d.quad[(q * 3 + 1) * N + j], d.quad[(q * 3 + 2) * N + j]}; }, deferred);
}
__device__ inline double laplace_entry_near(const DevicePlan& d, V3 t, int tbc, int64_t j) {
  const int64_t N = d.n;
  const V3 v0 = {d.vert[0 * N + j], d.vert[1 * N + j], d.vert[2 * N + j]};
  const V3 v1 = {d.vert[3 * N + j], d.vert[4 * N + j], d.vert[5 * N + j]};
  const V3 v2 = {d.vert[6 * N + j], d.vert[7 * N + j], d.vert[8 * N + j]};
  if (tbc == 0) return semi_analytic_G(v0, v1, v2, t);
  const double A = d.area[j];
  const V3 nrm = {d.nx[j], d.ny[j], d.nz[j]};
  double r = 0;
  for (int q = 0; q < 16; ++q) {
    const V3 pt = {v0.x * kFine[q][0] + v1.x * kFine[q][1] + v2.x * kFine[q][2],
                   v0.y * kFine[q][0] + v1.y * kFine[q][1] + v2.y * kFine[q][2],
                   v0.z * kFine[q][0] + v1.z * kFine[q][1] + v2.z * kFine[q][2]};
    const V3 dx = sub(pt, t);
    const double r2 = dx.x * dx.x + dx.y * dx.y + dx.z * dx.z;
    r += kFine[q][3] * A * (dx.x * nrm.x + dx.y * nrm.y + dx.z * nrm.z) / (r2 * sqrt(r2));
  }
  return r;
}
__device__ inline double laplace_entry(const DevicePlan& d, V3 t, int tbc, int64_t j) {
  bool deferred;
  const double v = laplace_entry_far(d, t, tbc, j, deferred);
  return deferred ? laplace_entry_near(d, t, tbc, j) : v;
}

// Column staging shared by near_assemble and near_spmv.  The columns of a target leaf's row block are
// the rows of its source leaves in ascending order; adjacent source leaves were merged on the host into
// runs (first row, first column).  All run descriptors are fetched with ONE parallel load, then every
// thread locates its column's run by a binary search in LDS -- two dependent global-memory latencies per
// workgroup instead of two per source leaf.
constexpr int kAsmChunk = 4096;                      // columns mapped in LDS at a time
struct Runs { int* row0; int* off; int n; };
__device__ inline Runs load_runs(const DevicePlan& d, int t, int* lds_row0, int* lds_off) {
  const int64_t rb = d.near_ptr[t];
  const int nruns = (int)(d.near_ptr[t + 1] - rb);
  for (int i = threadIdx.x; i < nruns; i += blockDim.x) {
    lds_row0[i] = d.near_run_row0[rb + i];
    lds_off[i] = d.near_run_off[rb + i];
  }
  __syncthreads();
  return {lds_row0, lds_off, nruns};
}
__device__ inline int column_to_row(const Runs& r, int c) {
  int lo = 0, hi = r.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (r.off[mid] <= c) lo = mid; else hi = mid - 1;
  }
  return r.row0[lo] + (c - r.off[lo]);
}

// ---------------------------------------------------------------------------------------------
// near_assemble: workgroups stride over the owned target leaves, one thread per matrix entry.
// The near regime is 4.5 % of the entries and a hundred times the work of the others (semi_analytic_G: three edges x a 5-point
// polar rule of logarithms and arc tangents); taken where it is met, nearly every wavefront holds a few such lanes and all 64
// wait for them (46 ms at N = 1M, the far regime alone is ~4).  So a pass over kAsmBatch entries per thread evaluates the far
// regime in place and QUEUES the near-regime entries in LDS; the workgroup then takes the queue with full wavefronts.  Same
// functions per entry, same values.
// ---------------------------------------------------------------------------------------------
constexpr int kAsmBatch = 8;                          // entries per thread in one pass
__global__ __launch_bounds__(256) void near_assemble_kernel(DevicePlan d) {
  extern __shared__ int lds_i[];
  // the queue is drained when it holds a round of 256 or more (and at the end of a chunk): ~90 entries join per pass, taken pass by
  // pass they would fill 36 % of the lanes.  Room for what is left below a round plus one pass in which every entry joins
  __shared__ int queue[256 * (kAsmBatch + 1)];
  __shared__ int queued;
  int* colmap = lds_i;                               // [kAsmChunk]
  int* run_row0 = lds_i + kAsmChunk;                 // [max_runs]
  int* run_off = run_row0 + d.max_runs;              // [max_runs]
  for (int t = d.leaf_begin + blockIdx.x; t < d.leaf_end; t += gridDim.x) {
    if (d.near_rec && d.near_rec[t]) continue;        // hybrid plan: this leaf keeps no matrix (its far regime is recomputed)
    const int ncols = d.near_ncols[t], stride = d.near_stride[t], nrows = d.leaf_nrows[t];
    const int row0 = d.leaf_row0[t];
    const Runs runs = load_runs(d, t, run_row0, run_off);
    double* blk = d.near_val + d.near_off[t];
    for (int c0 = 0; c0 < stride; c0 += kAsmChunk) {
      const int cw = stride - c0 < kAsmChunk ? stride - c0 : kAsmChunk;
      if (c0) __syncthreads();
      for (int c = threadIdx.x; c < cw; c += blockDim.x) colmap[c] = c0 + c < ncols ? column_to_row(runs, c0 + c) : -1;
      if (threadIdx.x == 0) queued = 0;
      __syncthreads();
      const int total = nrows * cw;
      for (int e0 = 0; e0 < total; e0 += 256 * kAsmBatch) {
        for (int u = 0; u < kAsmBatch; ++u) {
          const int e = e0 + u * 256 + (int)threadIdx.x;
          if (e >= total) break;
          const int r = e / cw, c = e - r * cw;
          double v = 0;                               // padding column (odd ncols) stays zero
          bool deferred = false;
          if (colmap[c] >= 0) {
            const int64_t i = row0 + r;
            v = laplace_entry_far(d, V3{d.cx[i], d.cy[i], d.cz[i]}, d.bc[i], colmap[c], deferred);
          }
          if (deferred) queue[atomicAdd(&queued, 1)] = e;
          else blk[(int64_t)r * stride + c0 + c] = v;
        }
        __syncthreads();
        const int nq = queued;
        __syncthreads();                              // everybody has read the count: the next pass may add to it
        if (nq < 256 && e0 + 256 * kAsmBatch < total) continue;
        for (int k = threadIdx.x; k < nq; k += 256) {
          const int e = queue[k];
          const int r = e / cw, c = e - r * cw;
          const int64_t i = row0 + r;
          blk[(int64_t)r * stride + c0 + c] = laplace_entry_near(d, V3{d.cx[i], d.cy[i], d.cz[i]}, d.bc[i], colmap[c]);
        }
        if (threadIdx.x == 0) queued = 0;              // (read by all before the barrier above; written again only after the one below)
        __syncthreads();
      }
    }
    __syncthreads();
  }
}

// The same assembly with a thread per COLUMN: the source panel (centroid, area, normal, the K <= 4 Gauss points: 20 doubles) is read
// once and kept in registers down the rows of the block; the rows' centroids come from LDS.  Same entry functions, same values;
// one load per entry instead of fourteen.  Rules of more than four points take the entry-per-thread kernel above.
template <int NQ>
__global__ __launch_bounds__(256) void near_assemble_cols_kernel(DevicePlan d) {
  extern __shared__ int lds_i[];
  __shared__ int queue[256 * (kAsmBatch + 1)];
  __shared__ int queued;
  __shared__ double trow[64][4];                     // centroid and flag of the rows of a row block
  int* colmap = lds_i;                               // [kAsmChunk]
  int* run_row0 = lds_i + kAsmChunk;                 // [max_runs]
  int* run_off = run_row0 + d.max_runs;              // [max_runs]
  const int64_t N = d.n;
  for (int t = d.leaf_begin + blockIdx.x; t < d.leaf_end; t += gridDim.x) {
    if (d.near_rec && d.near_rec[t]) continue;
    const int ncols = d.near_ncols[t], stride = d.near_stride[t], nrows = d.leaf_nrows[t];
    const int row0 = d.leaf_row0[t];
    const Runs runs = load_runs(d, t, run_row0, run_off);
    double* blk = d.near_val + d.near_off[t];
    for (int c0 = 0; c0 < stride; c0 += kAsmChunk) {
      const int cw = stride - c0 < kAsmChunk ? stride - c0 : kAsmChunk;
      if (c0) __syncthreads();
      for (int c = threadIdx.x; c < cw; c += blockDim.x) colmap[c] = c0 + c < ncols ? column_to_row(runs, c0 + c) : -1;
      if (threadIdx.x == 0) queued = 0;
      __syncthreads();
      // the queue holds entries as row * cw + column of this chunk; drained when a round of 256 has gathered and at the chunk's end
      auto drain = [&](bool last) {
        __syncthreads();
        const int nq = queued;
        __syncthreads();
        if (nq < 256 && !last) return;
        for (int k = threadIdx.x; k < nq; k += 256) {
          const int e = queue[k];
          const int r = e / cw, c = e - r * cw;
          const int64_t i = row0 + r;
          blk[(int64_t)r * stride + c0 + c] = laplace_entry_near(d, V3{d.cx[i], d.cy[i], d.cz[i]}, d.bc[i], colmap[c]);
        }
        if (threadIdx.x == 0) queued = 0;
        __syncthreads();
      };
      for (int rb = 0; rb < nrows; rb += 64) {
        const int nr = nrows - rb < 64 ? nrows - rb : 64;
        if ((int)threadIdx.x < nr) {
          const int64_t i = row0 + rb + threadIdx.x;
          trow[threadIdx.x][0] = d.cx[i]; trow[threadIdx.x][1] = d.cy[i]; trow[threadIdx.x][2] = d.cz[i]; trow[threadIdx.x][3] = (double)d.bc[i];
        }
        __syncthreads();
        for (int cb = 0; cb < cw; cb += 256) {
          const int c = cb + (int)threadIdx.x;
          const int j = c < cw ? colmap[c] : -1;
          V3 sc = {0, 0, 0}, sn = {0, 0, 0};
          double A = 0, qx[NQ], qy[NQ], qz[NQ];
          if (j >= 0) {
            sc = {d.cx[j], d.cy[j], d.cz[j]}; sn = {d.nx[j], d.ny[j], d.nz[j]}; A = d.area[j];
#pragma unroll
            for (int q = 0; q < NQ; ++q) { qx[q] = d.quad[(q * 3 + 0) * N + j]; qy[q] = d.quad[(q * 3 + 1) * N + j]; qz[q] = d.quad[(q * 3 + 2) * N + j]; }
          }
          for (int r8 = 0; r8 < nr; r8 += kAsmBatch) {
            const int rend = r8 + kAsmBatch < nr ? r8 + kAsmBatch : nr;
            if (c < cw)
              for (int r = r8; r < rend; ++r) {
                double v = 0;                           // padding column (odd ncols) stays zero
                bool deferred = false;
                if (j >= 0)
                  v = laplace_far_from(V3{trow[r][0], trow[r][1], trow[r][2]}, (int)trow[r][3], sc, A, sn, NQ, d.qw,
                                       [&](int q) { return V3{qx[q], qy[q], qz[q]}; }, deferred);
                if (deferred) queue[atomicAdd(&queued, 1)] = (rb + r) * cw + c;
                else blk[(int64_t)(rb + r) * stride + c0 + c] = v;
              }
            drain(false);
          }
        }
        drain(rb + 64 >= nrows);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Stokes (velocity boundary condition): one near-matrix entry is the 3x3 block
//   (1/2mu) int_source ( I/r + d d^T / r^3 ) dS,  d = target centroid - y
// (kernel/StokesSphericalBEM.hpp:260-375).  Regimes: far -> the K stored Gauss points (:352-369);
// near (sqrt(2A)/dist >= 0.5) -> the K_fine rule on the vertices (:302-321); self -> Fata's closed form.
// ---------------------------------------------------------------------------------------------
__device__ inline void stokeslet_point(double* res, double wA, V3 t, V3 pnt) {
  const V3 dd = sub(t, pnt);
  const double r2 = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z;
  double invR2 = 1. / r2;
  if (r2 < 1e-8) invR2 = 0;
  const double f = wA * invR2 * sqrt(invR2);
  const double dv[3] = {dd.x, dd.y, dd.z};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) res[3 * i + j] += f * ((i == j ? r2 : 0.0) + dv[i] * dv[j]);
}

// Self term: AnalyticalIntegral::FataAnalytical<STOKES>(y1,y2,y3,.,x = centroid, self = true, G)
// (examples/BEM/FataAnalytical.hpp:414-690, self branch :535-539) + Integration<STOKES>::integrate (:273-341).
// With the collocation point in the panel plane (et = 0) and chi left at {0,0,0} in the self branch, only
// omega (three logarithms) and the rho-difference terms survive.
__device__ inline void stokes_self(V3 y1, V3 y2, V3 y3, V3 x, double* IU) {
  const double pi = M_PI;
  const V3 v1 = sub(y2, y1), v3 = sub(y3, y1);
  const double snrm = v1.x * v1.x + v1.y * v1.y + v1.z * v1.z, nrm = sqrt(snrm);
  const double al = (v1.x * v3.x + v1.y * v3.y + v1.z * v3.z) / snrm;
  V3 e2 = {v3.x - al * v1.x, v3.y - al * v1.y, v3.z - al * v1.z};
  const double nrx = norm(e2);
  const V3 e1 = {v1.x / nrm, v1.y / nrm, v1.z / nrm};
  e2 = {e2.x / nrx, e2.y / nrx, e2.z / nrx};
  const V3 e3 = {e1.y * e2.z - e2.y * e1.z, e1.z * e2.x - e2.z * e1.x, e1.x * e2.y - e2.x * e1.y};
  const double bQ = v1.x * e1.x + v1.y * e1.y + v1.z * e1.z;
  const double aQ = v3.x * e2.x + v3.y * e2.y + v3.z * e2.z;
  const double cQ = v3.x * e1.x + v3.y * e1.y + v3.z * e1.z;
  const double bmc = bQ - cQ, aQs = aQ * aQ;
  const double th0 = acos(cQ / sqrt(cQ * cQ + aQs)), th1 = acos(bmc / sqrt(bmc * bmc + aQs));
  const double alpha2 = pi - th1, alpha3 = pi + th0;
  const double cs2 = cos(alpha2), sn2 = sin(alpha2), cs3 = cos(alpha3), sn3 = sin(alpha3);
  const V3 r1 = sub(x, y1);
  const double xi = r1.x * e1.x + r1.y * e1.y + r1.z * e1.z;
  const double zt = r1.x * e2.x + r1.y * e2.y + r1.z * e2.z;
  double q[3];
  const double p11 = -xi, p12 = bQ - xi;
  q[0] = -zt;
  const double x3 = cQ + p11, z3 = aQ + q[0];
  const double p22 = p12 * cs2 + q[0] * sn2, p23 = x3 * cs2 + z3 * sn2;
  q[1] = q[0] * cs2 - p12 * sn2;
  const double p31 = p11 * cs3 + q[0] * sn3, p33 = x3 * cs3 + z3 * sn3;
  q[2] = q[0] * cs3 - p11 * sn3;
  const double rho[3] = {sqrt(p11 * p11 + q[0] * q[0]), sqrt(p12 * p12 + q[0] * q[0]), sqrt(p33 * p33 + q[2] * q[2])};
  const double omega = q[0] * log((p11 + rho[0]) / (p12 + rho[1])) + q[1] * log((p22 + rho[1]) / (p23 + rho[2])) +
                       q[2] * log((p33 + rho[2]) / (p31 + rho[0]));
  const double alpha[3] = {0., alpha2, alpha3};
  const double rb[3] = {rho[0] - rho[1], rho[1] - rho[2], rho[2] - rho[0]};
  double Ixx = 0, Izz = 0, Izx = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Ixx += (rb[i] * sin(alpha[i])) * cos(alpha[i]);
    Izz += (-rb[i] * cos(alpha[i])) * sin(alpha[i]);
    Izx += (rb[i] * sin(alpha[i])) * sin(alpha[i]);
  }
  const double E[3][3] = {{e1.x, e1.y, e1.z}, {e2.x, e2.y, e2.z}, {e3.x, e3.y, e3.z}};
  const double coef[3][3] = {{omega + Ixx, Izx, 0.0}, {Izx, omega + Izz, 0.0}, {0.0, 0.0, omega}};
#pragma unroll
  for (int i = 0; i < 9; ++i) IU[i] = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) IU[3 * i + j] += coef[a][b] * E[a][i] * E[b][j];
}

// one Gauss point of the traction (double-layer) integrand: res += w A (d . n) d d^T / r^5, d = target - point
// (kernel/StokesSphericalBEM.hpp:205-225, 236-252)
__device__ inline void stresslet_point(double* res, double wA, V3 t, V3 pnt, V3 nrm) {
  const V3 dd = sub(t, pnt);
  const double r2 = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z;
  double invR2 = 1. / r2;
  if (r2 < 1e-8) invR2 = 0;
  const double invR5 = invR2 * invR2 * sqrt(invR2);
  const double dn = dd.x * nrm.x + dd.y * nrm.y + dd.z * nrm.z;
  const double f = wA * dn * invR5;
  const double dv[3] = {dd.x, dd.y, dd.z};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) res[3 * i + j] += f * dv[i] * dv[j];
}

// tbc = the TARGET's flag (kernel/StokesSphericalBEM.hpp:377-389): 0 VELOCITY -> eval_velocity_integral (:260-375),
// 1 TRACTION -> eval_traction_integral (:160-258): self 2 pi I, near K_fine, far K points, times -3, no 1/(2 mu)
// In two parts like laplace_entry: stokes_far_from gives the block of a pair in the far regime (the K stored Gauss points; the
// 2 pi I of a TRACTION self pair) from a source panel held in registers, or says `deferred`; stokes_entry_near the near regime
// (the K_fine rule on the vertices, the closed form of the self pair).  One text for the arithmetic, whoever asks.
template <class Quad>
__device__ __forceinline__ bool stokes_far_from(const DevicePlan& d, V3 t, int tbc, V3 c, double A, V3 nrm, Quad&& quad, double* out) {
  const double dist = norm(sub(t, c));
#pragma unroll
  for (int i = 0; i < 9; ++i) out[i] = 0;
  if (tbc) {
    if (fabs(dist) < 1e-8) { out[0] = out[4] = out[8] = 2 * M_PI; return false; }
    if (sqrt(2 * A) / dist >= 0.5) return true;
    for (int q = 0; q < d.nq; ++q) stresslet_point(out, d.qw[q] * A, t, quad(q), nrm);
#pragma unroll
    for (int i = 0; i < 9; ++i) out[i] *= -3.;
    return false;
  }
  if (sqrt(2 * A) / dist >= 0.5) return true;
  for (int q = 0; q < d.nq; ++q) stokeslet_point(out, d.qw[q] * A, t, quad(q));
  const double sc = 1. / 2 / d.mu;
#pragma unroll
  for (int i = 0; i < 9; ++i) out[i] *= sc;
  return false;
}
__device__ inline void stokes_entry_near(const DevicePlan& d, V3 t, int tbc, int64_t j, double* out) {
  const int64_t N = d.n;
  const V3 c = {d.cx[j], d.cy[j], d.cz[j]};
  const double A = d.area[j];
  const double dist = norm(sub(t, c));
#pragma unroll
  for (int i = 0; i < 9; ++i) out[i] = 0;
  const V3 v0 = {d.vert[0 * N + j], d.vert[1 * N + j], d.vert[2 * N + j]};
  const V3 v1 = {d.vert[3 * N + j], d.vert[4 * N + j], d.vert[5 * N + j]};
  const V3 v2 = {d.vert[6 * N + j], d.vert[7 * N + j], d.vert[8 * N + j]};
  if (tbc) {
    const V3 nrm = {d.nx[j], d.ny[j], d.nz[j]};
    for (int q = 0; q < d.nqf; ++q) {
      const V3 pt = {v0.x * d.qf[4 * q + 0] + v1.x * d.qf[4 * q + 1] + v2.x * d.qf[4 * q + 2],
                     v0.y * d.qf[4 * q + 0] + v1.y * d.qf[4 * q + 1] + v2.y * d.qf[4 * q + 2],
                     v0.z * d.qf[4 * q + 0] + v1.z * d.qf[4 * q + 1] + v2.z * d.qf[4 * q + 2]};
      stresslet_point(out, d.qf[4 * q + 3] * A, t, pt, nrm);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) out[i] *= -3.;
    return;
  }
  if (dist < 1e-8) {
    stokes_self(v0, v1, v2, t, out);
  } else {
    for (int q = 0; q < d.nqf; ++q) {
      const V3 pt = {v0.x * d.qf[4 * q + 0] + v1.x * d.qf[4 * q + 1] + v2.x * d.qf[4 * q + 2],
                     v0.y * d.qf[4 * q + 0] + v1.y * d.qf[4 * q + 1] + v2.y * d.qf[4 * q + 2],
                     v0.z * d.qf[4 * q + 0] + v1.z * d.qf[4 * q + 1] + v2.z * d.qf[4 * q + 2]};
      stokeslet_point(out, d.qf[4 * q + 3] * A, t, pt);
    }
  }
  const double sc = 1. / 2 / d.mu;
#pragma unroll
  for (int i = 0; i < 9; ++i) out[i] *= sc;
}
__device__ inline void stokes_entry(const DevicePlan& d, V3 t, int tbc, int64_t j, double* out) {
  const int64_t N = d.n;
  const V3 c = {d.cx[j], d.cy[j], d.cz[j]};
  const V3 nrm = tbc ? V3{d.nx[j], d.ny[j], d.nz[j]} : V3{0, 0, 0};
  if (stokes_far_from(d, t, tbc, c, d.area[j], nrm,
                      [&](int q) { return V3{d.quad[(q * 3 + 0) * N + j], d.quad[(q * 3 + 1) * N + j], d.quad[(q * 3 + 2) * N + j]}; }, out))
    stokes_entry_near(d, t, tbc, j, out);
}

// one thread per (target panel, source panel) pair of the leaf block; writes its 3x3 block into the
// 3 rows x 3 columns it occupies in the row-major block of unknowns
__global__ __launch_bounds__(256) void near_assemble_stokes_kernel(DevicePlan d) {
  extern __shared__ int lds_i[];
  int* colmap = lds_i;                               // [kAsmChunk] panel columns
  int* run_row0 = lds_i + kAsmChunk;
  int* run_off = run_row0 + d.max_runs;
  for (int t = d.leaf_begin + blockIdx.x; t < d.leaf_end; t += gridDim.x) {
    if (d.near_rec && d.near_rec[t]) continue;        // hybrid plan: this leaf keeps no matrix (its far regime is recomputed)
    const int ncols = d.near_ncols[t], stride = d.near_stride[t], nrows = d.leaf_nrows[t];
    const int row0 = d.leaf_row0[t];
    const Runs runs = load_runs(d, t, run_row0, run_off);
    double* blk = d.near_sym ? nullptr : d.near_val + d.near_off[t];
    dvec2* sym = d.near_sym ? reinterpret_cast<dvec2*>(d.near_sym + d.near_sym_off[t]) : nullptr;
    if (blk && stride > 3 * ncols)                    // padding column (odd number of unknowns per row)
      for (int r = threadIdx.x; r < 3 * nrows; r += blockDim.x) blk[(int64_t)r * stride + 3 * ncols] = 0.0;
    for (int c0 = 0; c0 < ncols; c0 += kAsmChunk) {
      const int cw = ncols - c0 < kAsmChunk ? ncols - c0 : kAsmChunk;
      if (c0) __syncthreads();
      for (int c = threadIdx.x; c < cw; c += blockDim.x) colmap[c] = column_to_row(runs, c0 + c);
      __syncthreads();
      const int total = nrows * cw;
      for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int r = e / cw, c = e - r * cw;
        const int64_t i = row0 + r;
        double m[9];
        stokes_entry(d, V3{d.cx[i], d.cy[i], d.cz[i]}, d.bc[i], colmap[c], m);
        if (sym) {                                     // the six entries (a <= b) of the symmetric block, three planes per panel row
          dvec2* row = sym + (int64_t)r * 3 * ncols + c0 + c;
          row[0] = dvec2{m[0], m[1]};
          row[ncols] = dvec2{m[2], m[4]};
          row[2 * ncols] = dvec2{m[5], m[8]};
        } else {
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) blk[(int64_t)(3 * r + a) * stride + 3 * (c0 + c) + b] = m[3 * a + b];
        }
      }
    }
    __syncthreads();
  }
}

// The Stokes assembly with a thread per column (cf. near_assemble_cols_kernel): the source panel's centroid, area, normal and K <= 4
// Gauss points in registers down the rows of the block, near-regime pairs (the K_fine rule: five times the work; the self pair)
// queued in LDS and taken by full wavefronts.  Same entry functions, same values.
__device__ __forceinline__ void stokes_store_block(const DevicePlan& d, double* blk, dvec2* sym, int r, int cg, int ncols, int stride, const double* m) {
  if (sym) {                                           // the six entries (a <= b) of the symmetric block, three planes per panel row
    dvec2* row = sym + (int64_t)r * 3 * ncols + cg;
    row[0] = dvec2{m[0], m[1]};
    row[ncols] = dvec2{m[2], m[4]};
    row[2 * ncols] = dvec2{m[5], m[8]};
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) blk[(int64_t)(3 * r + a) * stride + 3 * cg + b] = m[3 * a + b];
  }
}
template <int NQ>
__global__ __launch_bounds__(256) void near_assemble_stokes_cols_kernel(DevicePlan d) {
  extern __shared__ int lds_i[];
  __shared__ int queue[256 * (kAsmBatch + 1)];
  __shared__ int queued;
  __shared__ double trow[64][4];
  int* colmap = lds_i;                               // [kAsmChunk] panel columns
  int* run_row0 = lds_i + kAsmChunk;
  int* run_off = run_row0 + d.max_runs;
  const int64_t N = d.n;
  for (int t = d.leaf_begin + blockIdx.x; t < d.leaf_end; t += gridDim.x) {
    if (d.near_rec && d.near_rec[t]) continue;
    const int ncols = d.near_ncols[t], stride = d.near_stride[t], nrows = d.leaf_nrows[t];
    const int row0 = d.leaf_row0[t];
    const Runs runs = load_runs(d, t, run_row0, run_off);
    double* blk = d.near_sym ? nullptr : d.near_val + d.near_off[t];
    dvec2* sym = d.near_sym ? reinterpret_cast<dvec2*>(d.near_sym + d.near_sym_off[t]) : nullptr;
    if (blk && stride > 3 * ncols)                    // padding column (odd number of unknowns per row)
      for (int r = threadIdx.x; r < 3 * nrows; r += blockDim.x) blk[(int64_t)r * stride + 3 * ncols] = 0.0;
    for (int c0 = 0; c0 < ncols; c0 += kAsmChunk) {
      const int cw = ncols - c0 < kAsmChunk ? ncols - c0 : kAsmChunk;
      if (c0) __syncthreads();
      for (int c = threadIdx.x; c < cw; c += blockDim.x) colmap[c] = column_to_row(runs, c0 + c);
      if (threadIdx.x == 0) queued = 0;
      __syncthreads();
      auto drain = [&](bool last) {
        __syncthreads();
        const int nq = queued;
        __syncthreads();
        if (nq < 256 && !last) return;
        for (int k = threadIdx.x; k < nq; k += 256) {
          const int e = queue[k];
          const int r = e / cw, c = e - r * cw;
          const int64_t i = row0 + r;
          double m[9];
          stokes_entry_near(d, V3{d.cx[i], d.cy[i], d.cz[i]}, d.bc[i], colmap[c], m);
          stokes_store_block(d, blk, sym, r, c0 + c, ncols, stride, m);
        }
        if (threadIdx.x == 0) queued = 0;
        __syncthreads();
      };
      for (int rb = 0; rb < nrows; rb += 64) {
        const int nr = nrows - rb < 64 ? nrows - rb : 64;
        if ((int)threadIdx.x < nr) {
          const int64_t i = row0 + rb + threadIdx.x;
          trow[threadIdx.x][0] = d.cx[i]; trow[threadIdx.x][1] = d.cy[i]; trow[threadIdx.x][2] = d.cz[i]; trow[threadIdx.x][3] = (double)d.bc[i];
        }
        __syncthreads();
        for (int cb = 0; cb < cw; cb += 256) {
          const int c = cb + (int)threadIdx.x;
          const bool live = c < cw;
          V3 sc = {0, 0, 0}, sn = {0, 0, 0};
          double A = 0, qx[NQ], qy[NQ], qz[NQ];
          if (live) {
            const int j = colmap[c];
            sc = {d.cx[j], d.cy[j], d.cz[j]}; sn = {d.nx[j], d.ny[j], d.nz[j]}; A = d.area[j];
#pragma unroll
            for (int q = 0; q < NQ; ++q) { qx[q] = d.quad[(q * 3 + 0) * N + j]; qy[q] = d.quad[(q * 3 + 1) * N + j]; qz[q] = d.quad[(q * 3 + 2) * N + j]; }
          }
          for (int r8 = 0; r8 < nr; r8 += kAsmBatch) {
            const int rend = r8 + kAsmBatch < nr ? r8 + kAsmBatch : nr;
            if (live)
              for (int r = r8; r < rend; ++r) {
                double m[9];
                const int tbc = (int)trow[r][3];
                const bool deferred = stokes_far_from(d, V3{trow[r][0], trow[r][1], trow[r][2]}, tbc, sc, A, tbc ? sn : V3{0, 0, 0},
                                                      [&](int q) { return V3{qx[q], qy[q], qz[q]}; }, m);
                if (deferred) queue[atomicAdd(&queued, 1)] = (rb + r) * cw + c;
                else stokes_store_block(d, blk, sym, rb + r, c0 + c, ncols, stride, m);
              }
            drain(false);
          }
        }
        drain(rb + 64 >= nrows);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// near_matfree -- the matrix-free near field of EvalInteractionLazy (sparse_local = false): r_i += sum_j K(t_i, s_j) c_j
// recomputed every matvec (executor/EvalInteractionLazy.hpp:239-252 -> executor/P2P.hpp:20-36 -> Direct::eval asymmetric,
// include/Direct.hpp:99-125).  Third form (one unknown per panel).  The literal first form (round 1; git history) repeated the
// reference's arithmetic entry by entry: every entry re-reads its source panel (14 doubles from L2), takes a square root and a division per
// quadrature point, and a wavefront that meets ONE near-regime pair walks the whole semi-analytic integral (three edges,
// atan2, cos, five-point rules) with the other lanes idle: 45 ms at N = 1M against 0.7 ms for the assembled matrix.  The
// second form (round 2: lane = source panel applied to 64 rows out of registers, near-regime pairs queued in LDS and worked
// off with all lanes busy) took 15.5 ms, 9 of them the semi-analytic integrals of the 4.5 % of pairs in the near regimes --
// the same numbers every matvec.  Here the point of sparse_local = 0 is kept (no 4.1 GB matrix) and those are not recomputed:
//   * plan creation: the pairs that are, or might be, in a near regime (the reference's test sqrt(2A)/d >= 0.5, taken as
//     d^2 <= 8A with a guard band) are listed per target row (mf_sweep<COUNT>, <FILL>: columns ascending) and evaluated ONCE by
//     the laplace_entry the assembled path uses, one thread per pair (mf_side_eval_kernel): DevicePlan::side_ptr/col/val,
//     ~23 M entries = 280 MB at N = 1M;
//   * every matvec (mf_sweep<APPLY>): a wavefront takes a work item of near_spmv (a row range of a target leaf, kMfRows rows at a
//     time) and walks its columns 64 at a time: lane = SOURCE panel, read once into registers and applied to every row of the
//     block (the row's centroid is an LDS broadcast); the far regime is K reciprocal square roots per pair, no division; a
//     listed pair contributes nothing here (same predicate as at creation: same code).  The lane keeps its share of every
//     row's sum in registers, the lanes are added once at the end, then lane = row adds the row's listed entries in order.
//   * a row's sum is formed in a fixed order: column groups in order, then the lanes' tree sum, then the list.
// ---------------------------------------------------------------------------------------------
#ifndef FMMBEM_MF_ROWS
#define FMMBEM_MF_ROWS 24                             /* N = 1M, K = 3 (near ms): 12 rows 2.04, 16 1.96, 20 1.90, 24 1.85 (122 VGPRs), 28 2.21, 32 2.11, 40 2.19 */
#endif
constexpr int kMfRows = FMMBEM_MF_ROWS;               // rows per block: their partial sums live in registers (2 VGPRs each)

// 1 / sqrt(x) for a positive, normal x (a squared distance between distinct points): the hardware estimate (2^-26) and one
// Newton step carried to second order -- five FMAs instead of the library's two steps plus special-case handling
__device__ __forceinline__ double rsqrt_pos(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);                 // 1 - x y^2
  return fma(y * e, fma(0.375, e, 0.5), y);
}

// far regime with more than three quadrature points (K = 4 .. 79): the points are read again for every row
__device__ __noinline__ double mf_far_general(const DevicePlan& d, int64_t j, double tx, double ty, double tz, int dn, double A,
                                              double nx, double ny, double nz) {
  const int64_t N = d.n;
  double v = 0;
  for (int q = 0; q < d.nq; ++q) {
    const double ex = d.quad[(q * 3 + 0) * N + j] - tx, ey = d.quad[(q * 3 + 1) * N + j] - ty, ez = d.quad[(q * 3 + 2) * N + j] - tz;
    const double ir = rsqrt(ex * ex + ey * ey + ez * ez);
    v += dn ? d.qw[q] * A * (ex * nx + ey * ny + ez * nz) * (ir * ir * ir) : d.qw[q] * A * ir;
  }
  return v;
}

enum MfMode { kMfCount = 0, kMfFill = 1, kMfApply = 2 };

// "this pair is listed": squared centroid distance against 8 A (1 + 1e-9).  ONE function with the contractions spelled out, used by
// the kernels that list the pairs at plan creation and by every kernel that masks them in a matvec (mf_sweep, mf_sweep3_apply,
// the hybrid kernels): a pair whose distance sits within rounding of the band must get the same verdict in all of them
__device__ __forceinline__ bool mf_listed(double dx, double dy, double dz, double near2) {
  return fma(dz, dz, fma(dy, dy, dx * dx)) <= near2;
}

// side_cnt (COUNT): listed pairs per tree-order row; side_ptr / side_col (FILL): the CSR being filled; APPLY: d.side_*
// GEN: rules of more than three points (the points re-read per row by a called function, whose frame costs the kernel a third
// of its registers: 209 instead of 139 VGPRs) -- the reference's K = 1, 3 take the kernel without it
template <int MODE, bool GEN>
__global__ __launch_bounds__(kWave) void mf_sweep_kernel(DevicePlan d, int* __restrict__ side_cnt, const int64_t* __restrict__ side_ptr,
                                                         int* __restrict__ side_col) {
  extern __shared__ int mf_lds[];
  __shared__ double tcx[kMfRows], tcy[kMfRows], tcz[kMfRows];
  __shared__ int tbc[kMfRows];
  int* run_row0 = mf_lds;
  int* run_off = run_row0 + d.max_runs;
  const int lane = threadIdx.x;
  const int64_t N = d.n;
  const int nq = d.nq;
  // work units: the row ranges near_spmv streams (plan.hip: a leaf's rows cut so that no unit exceeds 256 KB of matrix, largest
  // first) -- one coarse leaf of the bench tree is 58 rows x 16 031 columns, a wavefront alone on it would finish milliseconds
  // after everyone else
  for (int item = blockIdx.x; item < d.near_nitems; item += gridDim.x) {
    const int4 it = d.near_items[item];
    const int t = it.x;
    const int ncols = d.near_ncols[t], nrows_all = it.z, row0 = d.leaf_row0[t] + it.y;
    __builtin_amdgcn_wave_barrier();
    const Runs runs = load_runs(d, t, run_row0, run_off);
    wave_lds_fence();
    for (int rb = 0; rb < nrows_all; rb += kMfRows) {
      const int nr = nrows_all - rb < kMfRows ? nrows_all - rb : kMfRows;
      if (lane < nr) {
        const int64_t i = row0 + rb + lane;
        tcx[lane] = d.cx[i]; tcy[lane] = d.cy[i]; tcz[lane] = d.cz[i]; tbc[lane] = d.bc[i];
      }
      double acc[kMfRows];                               // APPLY, lane = column: this lane's columns' share of every row (registers: the
#pragma unroll                                           // row loop below is unrolled; a wave_sum per row and column group, the
      for (int r = 0; r < kMfRows; ++r) acc[r] = 0;      // first attempt, cost three times the arithmetic)
      int listed = 0;                                    // COUNT / FILL, lane = row: listed pairs of the row so far
      wave_lds_fence();
      for (int c0 = 0; c0 < ncols; c0 += kWave) {
        const int c = c0 + lane;
        const bool valid = c < ncols;
        const int j = valid ? column_to_row(runs, c) : 0;
        // this lane's source panel, once for all rows
        const double sx = d.cx[j], sy = d.cy[j], sz = d.cz[j], A = d.area[j];
        const double near2 = 8.0 * A * (1.0 + 1e-9);      // d^2 <= 8 A  <=>  sqrt(2 A) / d >= 0.5; the band goes to the exact test
        double xj = 0, nx = 0, ny = 0, nz = 0;
        double qx[3], qy[3], qz[3], wA[3];               // up to three far-regime points in registers (K = 1, 3); more: reloaded
        const int nqr = nq <= 3 ? nq : 0;
        if constexpr (MODE == kMfApply) {
          xj = valid ? d.xt[j] : 0.0;
          nx = d.nx[j]; ny = d.ny[j]; nz = d.nz[j];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const bool have = q < nqr;
            qx[q] = have ? d.quad[(q * 3 + 0) * N + j] : 0.0; qy[q] = have ? d.quad[(q * 3 + 1) * N + j] : 0.0;
            qz[q] = have ? d.quad[(q * 3 + 2) * N + j] : 1.0; wA[q] = have ? d.qw[q] * A : 0.0;
          }
        }
#pragma unroll
        for (int r = 0; r < kMfRows; ++r) {
          if (r < nr) {                                    // wave-uniform
            const double tx = tcx[r], ty = tcy[r], tz = tcz[r];
            const double dx = tx - sx, dy = ty - sy, dz = tz - sz;
            const bool slow = valid && mf_listed(dx, dy, dz, near2);       // a listed pair: the SAME function at creation and in every matvec
            if constexpr (MODE == kMfApply) {
              const int dn = __builtin_amdgcn_readfirstlane(tbc[r]);      // the row's operator: a scalar branch, not a select per point
              double v = 0;
              if (nqr) {                                   // K <= 3: straight-line, the listed lanes' values are dropped below
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                  const double ex = qx[q] - tx, ey = qy[q] - ty, ez = qz[q] - tz;
                  const double ir = rsqrt_pos(fma(ex, ex, fma(ey, ey, ez * ez)));
                  if (dn) v = fma(wA[q] * fma(ex, nx, fma(ey, ny, ez * nz)), ir * ir * ir, v); else v = fma(wA[q], ir, v);
                }
              } else if constexpr (GEN) {
                if (valid && !slow) v = mf_far_general(d, j, tx, ty, tz, dn, A, nx, ny, nz);
              }
              // lanes past the last column ran the arithmetic against panel 0: its value may be inf/NaN (K = 1: the only point of
              // panel 0 IS the centroid of row 0) and NaN * 0 is NaN -- masked by validity, not by xj = 0
              acc[r] = fma((slow || !valid) ? 0.0 : v, xj, acc[r]);
              // one row after the other: left to itself the scheduler interleaves all rows of the unrolled loop (every LDS
              // broadcast hoisted, 512 registers and 2.4 KB of scratch per lane)
              __builtin_amdgcn_sched_barrier(0);
            } else {
              const unsigned long long m = __ballot(slow);
              if (m) {
                if constexpr (MODE == kMfFill)
                  if (slow) side_col[side_ptr[row0 + rb + r] + __builtin_amdgcn_readlane(listed, r) + __popcll(m & ((1ull << lane) - 1))] = j;
                if (lane == r) listed += __popcll(m);
              }
            }
          }
        }
      }
      if constexpr (MODE == kMfApply) {
        double racc = 0;                                   // lane = row: the row's sum
#pragma unroll
        for (int r = 0; r < kMfRows; ++r)
          if (r < nr) {
            const double srow = wave_sum(acc[r]);
            if (lane == r) racc = srow;
          }
        if (lane < nr) {
          const int64_t i = row0 + rb + lane;
          for (int64_t k = d.side_ptr[i]; k < d.side_ptr[i + 1]; ++k) racc = fma(d.side_val[k], d.xt[d.side_col[k]], racc);
          d.yt[i] = racc;
        }
      } else if constexpr (MODE == kMfCount) {
        if (lane < nr) side_cnt[row0 + rb + lane] = listed;
      }
      wave_lds_fence();
    }
  }
}

// The same sweep for StokesSphericalBEM (three unknowns per panel; StokesBEM -disable_sparse): the far regime of a pair applied
// to the source's Vec<3,double> charge WITHOUT forming the 3 x 3 block --
//   velocity target   (1/2mu) sum_q w_q A [ x / r + d (d.x) / r^3 ]          d = target - point q   (StokesSphericalBEM.hpp:352-369)
//   TRACTION target   -3 sum_q w_q A (d.n)(d.x) d / r^5                       (:236-252; n = the source's normal)
// one reciprocal square root per point; listed pairs (near regime: the K_fine rule; self: Fata's closed form / 2 pi I) come
// from the list with their 3 x 3 blocks.  Rows in blocks of kMfRows3 (three partial sums each in registers).
#ifndef FMMBEM_MF_ROWS3
#define FMMBEM_MF_ROWS3 24                            /* red blood cell N = 524 288 (near ms): 8 rows 2.60, 12 2.35, 16 2.21, 20 2.15, 24 2.10 (243 VGPRs) */
#endif
constexpr int kMfRows3 = FMMBEM_MF_ROWS3;
__global__ __launch_bounds__(kWave) void mf_sweep3_apply_kernel(DevicePlan d) {
  extern __shared__ int mf_lds[];
  __shared__ double tcx[kMfRows3], tcy[kMfRows3], tcz[kMfRows3];
  __shared__ int tbc[kMfRows3];
  int* run_row0 = mf_lds;
  int* run_off = run_row0 + d.max_runs;
  const int lane = threadIdx.x;
  const int64_t N = d.n;
  const int nq = d.nq;
  const double sc = 1. / 2 / d.mu;
  for (int item = blockIdx.x; item < d.near_nitems; item += gridDim.x) {
    const int4 it = d.near_items[item];
    const int t = it.x;
    const int ncols = d.near_ncols[t], nrows_all = it.z, row0 = d.leaf_row0[t] + it.y;
    __builtin_amdgcn_wave_barrier();
    const Runs runs = load_runs(d, t, run_row0, run_off);
    wave_lds_fence();
    for (int rb = 0; rb < nrows_all; rb += kMfRows3) {
      const int nr = nrows_all - rb < kMfRows3 ? nrows_all - rb : kMfRows3;
      if (lane < nr) {
        const int64_t i = row0 + rb + lane;
        tcx[lane] = d.cx[i]; tcy[lane] = d.cy[i]; tcz[lane] = d.cz[i]; tbc[lane] = d.bc[i];
      }
      double a0[kMfRows3], a1[kMfRows3], a2[kMfRows3];
#pragma unroll
      for (int r = 0; r < kMfRows3; ++r) a0[r] = a1[r] = a2[r] = 0;
      wave_lds_fence();
      for (int c0 = 0; c0 < ncols; c0 += kWave) {
        const int c = c0 + lane;
        const bool valid = c < ncols;
        const int j = valid ? column_to_row(runs, c) : 0;
        const double sx = d.cx[j], sy = d.cy[j], sz = d.cz[j], A = d.area[j];
        const double near2 = 8.0 * A * (1.0 + 1e-9);
        const double x0 = valid ? d.xt[3 * (int64_t)j] : 0.0, x1 = valid ? d.xt[3 * (int64_t)j + 1] : 0.0, x2 = valid ? d.xt[3 * (int64_t)j + 2] : 0.0;
        const double nx = d.nx[j], ny = d.ny[j], nz = d.nz[j];
        double qx[4], qy[4], qz[4], wA[4];               // the far-regime points in registers: K = 1, 3, 4 (the launcher sends larger
        const int nqr = nq <= 4 ? nq : 0;                // rules to the literal kernel: a call to stokes_entry in here costs the sweep its registers)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool have = q < nqr;
          qx[q] = have ? d.quad[(q * 3 + 0) * N + j] : 0.0; qy[q] = have ? d.quad[(q * 3 + 1) * N + j] : 0.0;
          qz[q] = have ? d.quad[(q * 3 + 2) * N + j] : 1.0; wA[q] = have ? d.qw[q] * A : 0.0;
        }
#pragma unroll
        for (int r = 0; r < kMfRows3; ++r) {
          if (r < nr) {
            const double tx = tcx[r], ty = tcy[r], tz = tcz[r];
            const double dx = tx - sx, dy = ty - sy, dz = tz - sz;
            const bool slow = valid && mf_listed(dx, dy, dz, near2);             // the function of mf_sweep_kernel: the same pairs
            const int trac = __builtin_amdgcn_readfirstlane(tbc[r]);
            double u0 = 0, u1 = 0, u2 = 0;
            if (nqr) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const double ex = tx - qx[q], ey = ty - qy[q], ez = tz - qz[q];
                const double ir = rsqrt_pos(fma(ex, ex, fma(ey, ey, ez * ez)));
                const double ir3 = ir * ir * ir, dxq = fma(ex, x0, fma(ey, x1, ez * x2));
                if (trac) {
                  const double g = wA[q] * fma(ex, nx, fma(ey, ny, ez * nz)) * dxq * (ir3 * ir * ir);
                  u0 = fma(g, ex, u0); u1 = fma(g, ey, u1); u2 = fma(g, ez, u2);
                } else {
                  const double f1 = wA[q] * ir, g = wA[q] * ir3 * dxq;
                  u0 = fma(f1, x0, fma(g, ex, u0)); u1 = fma(f1, x1, fma(g, ey, u1)); u2 = fma(f1, x2, fma(g, ez, u2));
                }
              }
              const double f = trac ? -3.0 : sc;
              u0 *= f; u1 *= f; u2 *= f;
            }
            const bool drop = slow || !valid;
            a0[r] += drop ? 0.0 : u0; a1[r] += drop ? 0.0 : u1; a2[r] += drop ? 0.0 : u2;
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      double r0 = 0, r1 = 0, r2 = 0;
#pragma unroll
      for (int r = 0; r < kMfRows3; ++r)
        if (r < nr) {
          const double s0 = wave_sum(a0[r]), s1 = wave_sum(a1[r]), s2 = wave_sum(a2[r]);
          if (lane == r) { r0 = s0; r1 = s1; r2 = s2; }
        }
      if (lane < nr) {
        const int64_t i = row0 + rb + lane;
        for (int64_t k = d.side_ptr[i]; k < d.side_ptr[i + 1]; ++k) {
          const double* m = d.side_val + 9 * k;
          const int64_t cj = d.side_col[k];
          const double y0 = d.xt[3 * cj], y1 = d.xt[3 * cj + 1], y2 = d.xt[3 * cj + 2];
          r0 = fma(m[0], y0, fma(m[1], y1, fma(m[2], y2, r0)));
          r1 = fma(m[3], y0, fma(m[4], y1, fma(m[5], y2, r1)));
          r2 = fma(m[6], y0, fma(m[7], y1, fma(m[8], y2, r2)));
        }
        d.yt[3 * i] = r0; d.yt[3 * i + 1] = r1; d.yt[3 * i + 2] = r2;
      }
      wave_lds_fence();
    }
  }
}

// one thread per listed pair: the entry, by the function the assembled path uses (which repeats the exact regime test)
__global__ void mf_side_eval_kernel(DevicePlan d, const int* __restrict__ side_row, const int* __restrict__ side_col, double* __restrict__ side_val, int64_t nside) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= nside) return;
  const int64_t i = side_row[k];
  if (d.dof == 3) stokes_entry(d, V3{d.cx[i], d.cy[i], d.cz[i]}, d.bc[i], side_col[k], side_val + 9 * k);
  else side_val[k] = laplace_entry(d, V3{d.cx[i], d.cy[i], d.cz[i]}, d.bc[i], side_col[k]);
}

// The same for StokesSphericalBEM (StokesBEM -disable_sparse): the 3x3 panel integral of every (target, source) pair of the
// near list is recomputed and applied to the source's Vec<3,double> charge; one wavefront per target panel row.
__global__ __launch_bounds__(256) void near_matfree_stokes_kernel(DevicePlan d) {
  extern __shared__ double lds_d[];
  double* xs = lds_d;                                          // [3][kAsmChunk / 2]
  constexpr int kChunk = kAsmChunk / 2;
  int* colmap = reinterpret_cast<int*>(xs + 3 * kChunk);       // [kChunk]
  int* run_row0 = colmap + kChunk;
  int* run_off = run_row0 + d.max_runs;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
  for (int t = d.leaf_begin + blockIdx.x; t < d.leaf_end; t += gridDim.x) {
    const int ncols = d.near_ncols[t], nrows = d.leaf_nrows[t];
    const int row0 = d.leaf_row0[t];
    const Runs runs = load_runs(d, t, run_row0, run_off);
    for (int c0 = 0; c0 < ncols; c0 += kChunk) {
      const int cw = ncols - c0 < kChunk ? ncols - c0 : kChunk;
      if (c0) __syncthreads();
      for (int c = threadIdx.x; c < cw; c += blockDim.x) {
        const int j = column_to_row(runs, c0 + c);
        colmap[c] = j;
        xs[c] = d.xt[3 * (int64_t)j]; xs[kChunk + c] = d.xt[3 * (int64_t)j + 1]; xs[2 * kChunk + c] = d.xt[3 * (int64_t)j + 2];
      }
      __syncthreads();
      for (int r = wave; r < nrows; r += nwaves) {
        const int64_t i = row0 + r;
        const V3 tc = {d.cx[i], d.cy[i], d.cz[i]};
        const int tbc = d.bc[i];
        double a0 = 0, a1 = 0, a2 = 0;
        for (int c = lane; c < cw; c += kWave) {
          double m[9];
          stokes_entry(d, tc, tbc, colmap[c], m);
          const double x0 = xs[c], x1 = xs[kChunk + c], x2 = xs[2 * kChunk + c];
          a0 = fma(m[0], x0, fma(m[1], x1, fma(m[2], x2, a0)));
          a1 = fma(m[3], x0, fma(m[4], x1, fma(m[5], x2, a1)));
          a2 = fma(m[6], x0, fma(m[7], x1, fma(m[8], x2, a2)));
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) {
          double* y = d.yt + 3 * i;
          y[0] = c0 ? y[0] + a0 : a0; y[1] = c0 ? y[1] + a1 : a1; y[2] = c0 ? y[2] + a2 : a2;
        }
      }
    }
    __syncthreads();
  }
}

// dof unknowns per panel, interleaved (Stokes: Vec<3,double> per panel); one thread per unknown
__global__ void gather_x_kernel(const uint32_t* __restrict__ perm, const double* __restrict__ x,
                                double* __restrict__ xt, int64_t n, int dof, double* __restrict__ xt4) {
  const int64_t u = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (u < n * dof) {
    const int64_t i = u / dof;
    const int a = (int)(u - i * dof);
    const double v = x[(int64_t)perm[i] * dof + a];
    xt[u] = v;
    if (xt4) xt4[4 * i + a] = v;                        // hybrid Stokes plans: the charge padded to 32 bytes per panel (near_recompute3g)
  }
}

__global__ void scatter_y_kernel(const uint32_t* __restrict__ perm, const double* __restrict__ yt, double* __restrict__ y,
                                 int64_t row_begin, int64_t row_end, int dof) {
  const int64_t u = row_begin * dof + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (u < row_end * dof) {
    const int64_t i = u / dof;
    const int a = (int)(u - i * dof);
    y[(int64_t)perm[i] * dof + a] = yt[u];
  }
}

// ---------------------------------------------------------------------------------------------
// near_spmv (P2P): y_t = A_near x_t over the owned rows -- include/Matvec.hpp:14-33 on the dense leaf blocks.
//   1. the x values of a work item's columns are staged in LDS through the leaf's run descriptors;
//   2. each wavefront takes kRows rows at a time (rows w, w+4, ...); a row is `stride` contiguous doubles read as
//      16-B vectors, lanes striding over the columns (coalesced 1-KiB wave loads, nontemporal); all kRows x kVecs
//      loads of a batch are issued before the first FMA;
//   3. per-row wave shuffle reduction, lane 0 stores y_tree[row].
// Algorithmic bytes: 8 B per near entry (+ 8 B x read + 8 B y write per panel); no column indices.
// This plain form serves Stokes (3 unknowns per panel, 2048-column chunks); one unknown per panel takes the pipelined
// form further down.
// ---------------------------------------------------------------------------------------------
constexpr int kSpmvWaves = 4;
constexpr int kSpmvChunk = 2048;                     // columns of x staged in LDS at a time (16 KiB)

constexpr int kSpmvOcc = 5;                           // wavefronts per SIMD = workgroups per CU (82 VGPRs)
constexpr int kColRows = 8;                           // work items with fewer rows split the COLUMNS over the wavefronts

template <int kRows, int kVecs>
__global__ __launch_bounds__(kSpmvWaves * kWave, kSpmvOcc) void near_spmv_kernel(DevicePlan d) {
  extern __shared__ double xs[];                      // [kSpmvChunk] doubles, then the run descriptors
  __shared__ double part[kSpmvWaves][kColRows];
  int* run_row0 = reinterpret_cast<int*>(xs + kSpmvChunk);
  int* run_off = run_row0 + d.max_runs;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const dvec2* xv = reinterpret_cast<const dvec2*>(xs);
  const int dof = d.dof;

  // Persistent workgroups (a few per CU) are dealt the work items -- row ranges of one target leaf, largest
  // first -- round-robin.  One short-lived workgroup per leaf leaves the CUs mostly empty (3.5 resident
  // wavefronts per CU, launch-rate bound); whole leaves as items strand the workgroup that owns a coarse leaf;
  // a dynamic queue costs more than it balances (one atomic per item: 65 k same-address atomics, +0.2 ms).
  for (int item = blockIdx.x; item < d.near_nitems; item += gridDim.x) {
    const int4 it = d.near_items[item];
    const int t = it.x, r0 = it.y, nrows = it.z;
    const bool colsplit = it.w != 0;
    // rows / columns counted in unknowns: dof per panel (Stokes: the 3x3 blocks are simply 3 rows x 3 columns)
    const int ncols = dof * d.near_ncols[t], stride = d.near_stride[t];
    const Runs runs = load_runs(d, t, run_row0, run_off);
    const double* blk = d.near_val + d.near_off[t] + (int64_t)r0 * stride;
    double* yt = d.yt + dof * d.leaf_row0[t] + r0;
    // the x slice is staged kSpmvChunk columns at a time: a few coarse leaves of an adaptive tree see
    // >10^4 columns, and sizing the LDS for them would leave one workgroup per CU
    for (int c0 = 0; c0 < stride; c0 += kSpmvChunk) {
      const int cw = stride - c0 < kSpmvChunk ? stride - c0 : kSpmvChunk;
      if (c0) __syncthreads();
      for (int c = threadIdx.x; c < cw; c += blockDim.x) {
        const int cc = c0 + c, pc = cc / dof;            // unknown -> panel column
        xs[c] = cc < ncols ? d.xt[(int64_t)dof * column_to_row(runs, pc) + (cc - pc * dof)] : 0.0;
      }
      __syncthreads();
      const int nvec = cw >> 1;                       // 16-B vectors of this chunk per row
      // row mode: wavefront w takes rows w, w+4, ... over all columns; column mode: every wavefront takes
      // a quarter of the columns (64-B aligned) of every row and the quarters are summed in fixed order
      const int seg = colsplit ? ((((nvec + kSpmvWaves - 1) / kSpmvWaves) + 3) & ~3) : nvec;
      const int v0 = colsplit ? wave * seg : 0, v1 = min(nvec, v0 + seg);
      const int rstep = colsplit ? 1 : kSpmvWaves;
      for (int r = colsplit ? 0 : wave; r < nrows; r += colsplit ? kRows : kRows * kSpmvWaves) {
        const dvec2* row[kRows];
        double acc[kRows];
#pragma unroll
        for (int i = 0; i < kRows; ++i) {
          const int ri = r + i * rstep;
          row[i] = reinterpret_cast<const dvec2*>(blk + (int64_t)(ri < nrows ? ri : r) * stride + c0);
          acc[i] = 0;
        }
        for (int c = v0 + lane; c < v1; c += kVecs * kWave) {
          dvec2 v[kRows][kVecs];
#pragma unroll
          for (int u = 0; u < kVecs; ++u) {
            const int cc = c + u * kWave;
            const bool ok = cc < v1;
#pragma unroll
            for (int i = 0; i < kRows; ++i) v[i][u] = ok ? __builtin_nontemporal_load(&row[i][cc]) : dvec2{0, 0};
          }
#pragma unroll
          for (int u = 0; u < kVecs; ++u) {
            const int cc = c + u * kWave;
            if (cc < v1) {
              const dvec2 x2 = xv[cc];
#pragma unroll
              for (int i = 0; i < kRows; ++i) acc[i] = fma(v[i][u].x, x2.x, fma(v[i][u].y, x2.y, acc[i]));
            }
          }
        }
#pragma unroll
        for (int i = 0; i < kRows; ++i) acc[i] = wave_sum(acc[i]);
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < kRows; ++i) {
            const int ri = r + i * rstep;
            if (ri < nrows) {
              if (colsplit) part[wave][ri] = acc[i];
              else yt[ri] = c0 ? yt[ri] + acc[i] : acc[i];
            }
          }
        }
      }
      if (colsplit) {
        __syncthreads();
        if ((int)threadIdx.x < nrows) {
          const int ri = threadIdx.x;
          const double sum = ((part[0][ri] + part[1][ri]) + part[2][ri]) + part[3][ri];
          yt[ri] = c0 ? yt[ri] + sum : sum;
        }
      }
    }
    __syncthreads();                                  // xs / run descriptors / part are rewritten for the next item
  }
}

// ---------------------------------------------------------------------------------------------
// near_spmv, pipelined form (one unknown per panel): the same items, the same row/column-split arithmetic as
// near_spmv_kernel above, but the set-up of item i+1 runs while item i streams.  A workgroup's time per item was
// [x gather: one L2 round trip] [2-3 round trips of matrix rows] with two barriers, i.e. a streaming duty of ~80 %
// (5.5 of the 7.0 TB/s a bare read of the same blocks reaches).  Here
//   * item records are self-contained (NearItem) and fetched two items ahead by scalar loads;
//   * the run descriptors of item i+2 and the x values of item i+1 (first kSpmvPipeChunk columns) are loaded into
//     registers BEFORE item i's rows are streamed and written to the other halves of double-buffered LDS arrays after;
//   * one barrier per item.
// Wider leaves stage their further chunks in place as before.
// ---------------------------------------------------------------------------------------------
constexpr int kSpmvPipeChunk = 1024;                  // columns per x buffer (2 x 8 KiB)
constexpr int kSpmvPre = kSpmvPipeChunk / (kSpmvWaves * kWave);       // x values a thread prefetches

// the item records through the constant address space: scalar loads whatever else the kernel stores (6 VGPRs fewer than
// through the generic pointer)
typedef __attribute__((address_space(4))) NearItem ConstNearItem;
__device__ __forceinline__ NearItem load_item(const ConstNearItem* r) {
  NearItem o;
  o.val_off = r->val_off; o.run_begin = r->run_begin; o.nruns = r->nruns; o.yrow = r->yrow; o.nrows = r->nrows;
  o.ncols = r->ncols; o.stride = r->stride; o.colsplit = r->colsplit; o.pad[0] = o.pad[1] = 0;
  return o;
}

// the pipelined loop over items first, first + step, ... < end of one workgroup; LDS: xs_all [2][kSpmvPipeChunk] doubles then
// runbuf [2][2][max_runs] ints, part [kSpmvWaves][kColRows]
template <int kRows, int kVecs>
__device__ __forceinline__ void spmv_pipe_run(const DevicePlan& d, double* xs_all, double (*part)[kColRows], int first, int step, int end) {
  int* const runbuf = reinterpret_cast<int*>(xs_all + 2 * kSpmvPipeChunk);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int mr = d.max_runs, nitems = end, tid = threadIdx.x;
  const ConstNearItem* recs = reinterpret_cast<const ConstNearItem*>(reinterpret_cast<uintptr_t>(d.near_recs));

  int item = first;
  if (item >= nitems) return;
  NearItem it = load_item(recs + item);
  NearItem nx = load_item(recs + (item + step < nitems ? item + step : nitems - 1));
  // prologue: runs of the first two items, x of the first
  for (int i = tid; i < it.nruns; i += blockDim.x) { runbuf[i] = d.near_run_row0[it.run_begin + i]; runbuf[mr + i] = d.near_run_off[it.run_begin + i]; }
  for (int i = tid; i < nx.nruns; i += blockDim.x) { runbuf[2 * mr + i] = d.near_run_row0[nx.run_begin + i]; runbuf[3 * mr + i] = d.near_run_off[nx.run_begin + i]; }
  __syncthreads();
  {
    const Runs runs{runbuf, runbuf + mr, it.nruns};
    const int cw = it.stride < kSpmvPipeChunk ? it.stride : kSpmvPipeChunk;
    for (int c = tid; c < cw; c += blockDim.x) xs_all[c] = c < it.ncols ? d.xt[column_to_row(runs, c)] : 0.0;
  }
  __syncthreads();
  int xb = 0, rb = 0;                                 // which halves hold the current item's x chunk 0 / runs
  for (;; item += step) {
    const bool more = item + step < nitems;
    const int i2 = item + 2 * step;
    const NearItem nn = load_item(recs + (i2 < nitems ? i2 : nitems - 1));
    // ---- set-up of the following items, in flight while this item's rows stream ----
    // (issued behind the first matrix loads instead: the staging array lands in scratch, 0.91 ms)
    double px[kSpmvPre];
    int pr0 = 0, pr1 = 0;
    const bool prun = i2 < nitems && tid < nn.nruns;     // max_runs <= blockDim.x is checked by the launcher
    if (more) {
      const Runs nruns{runbuf + (rb ^ 1) * 2 * mr, runbuf + (rb ^ 1) * 2 * mr + mr, nx.nruns};
      const int ncw = nx.stride < kSpmvPipeChunk ? nx.stride : kSpmvPipeChunk;
#pragma unroll
      for (int u = 0; u < kSpmvPre; ++u) {
        const int c = tid + u * (kSpmvWaves * kWave);
        px[u] = (c < ncw && c < nx.ncols) ? d.xt[column_to_row(nruns, c)] : 0.0;
      }
      if (prun) { pr0 = d.near_run_row0[nn.run_begin + tid]; pr1 = d.near_run_off[nn.run_begin + tid]; }
    }
    // ---- this item ----
    const int nrows = it.nrows, ncols = it.ncols, stride = it.stride;
    const bool colsplit = it.colsplit != 0;
    const double* blk = d.near_val + it.val_off;
    double* yt = d.yt + it.yrow;
    double* xs = xs_all + xb * kSpmvPipeChunk;
    const dvec2* xv = reinterpret_cast<const dvec2*>(xs);
    const Runs runs{runbuf + rb * 2 * mr, runbuf + rb * 2 * mr + mr, it.nruns};
    for (int c0 = 0; c0 < stride; c0 += kSpmvPipeChunk) {
      const int cw = stride - c0 < kSpmvPipeChunk ? stride - c0 : kSpmvPipeChunk;
      if (c0) {                                       // further chunks of a wide leaf: staged in place
        __syncthreads();
        for (int c = tid; c < cw; c += blockDim.x) xs[c] = c0 + c < ncols ? d.xt[column_to_row(runs, c0 + c)] : 0.0;
        __syncthreads();
      }
      const int nvec = cw >> 1;                       // 16-B vectors of this chunk per row
      const int seg = colsplit ? ((((nvec + kSpmvWaves - 1) / kSpmvWaves) + 3) & ~3) : nvec;
      const int v0 = colsplit ? wave * seg : 0, v1 = min(nvec, v0 + seg);
      const int rstep = colsplit ? 1 : kSpmvWaves;
      for (int r = colsplit ? 0 : wave; r < nrows; r += colsplit ? kRows : kRows * kSpmvWaves) {
        const dvec2* row[kRows];
        double acc[kRows];
#pragma unroll
        for (int i = 0; i < kRows; ++i) {
          const int ri = r + i * rstep;
          row[i] = reinterpret_cast<const dvec2*>(blk + (int64_t)(ri < nrows ? ri : r) * stride + c0);
          acc[i] = 0;
        }
        for (int c = v0 + lane; c < v1; c += kVecs * kWave) {
          dvec2 v[kRows][kVecs];
#pragma unroll
          for (int u = 0; u < kVecs; ++u) {
            const int cc = c + u * kWave;
            const bool ok = cc < v1;
#pragma unroll
            for (int i = 0; i < kRows; ++i) v[i][u] = ok ? __builtin_nontemporal_load(&row[i][cc]) : dvec2{0, 0};
          }
#pragma unroll
          for (int u = 0; u < kVecs; ++u) {
            const int cc = c + u * kWave;
            if (cc < v1) {
              const dvec2 x2 = xv[cc];
#pragma unroll
              for (int i = 0; i < kRows; ++i) acc[i] = fma(v[i][u].x, x2.x, fma(v[i][u].y, x2.y, acc[i]));
            }
          }
        }
#pragma unroll
        for (int i = 0; i < kRows; ++i) acc[i] = wave_sum(acc[i]);
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < kRows; ++i) {
            const int ri = r + i * rstep;
            if (ri < nrows) {
              if (colsplit) part[wave][ri] = acc[i];
              else yt[ri] = c0 ? yt[ri] + acc[i] : acc[i];
            }
          }
        }
      }
      if (colsplit) {
        __syncthreads();
        if (tid < nrows) {
          const double sum = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
          yt[tid] = c0 ? yt[tid] + sum : sum;
        }
      }
    }
    if (!more) break;
    // ---- hand over: next item's x and the item after's runs into the halves nobody reads now ----
    {
      double* xn = xs_all + (xb ^ 1) * kSpmvPipeChunk;
      const int ncw = nx.stride < kSpmvPipeChunk ? nx.stride : kSpmvPipeChunk;
#pragma unroll
      for (int u = 0; u < kSpmvPre; ++u) {
        const int c = tid + u * (kSpmvWaves * kWave);
        if (c < ncw) xn[c] = px[u];
      }
    }
    // runs[rb] belonged to this item; a wide leaf's later chunks were the last to read it, and every thread has passed
    // them: the barriers inside the chunk loop order those reads before this write
    if (prun) { runbuf[rb * 2 * mr + tid] = pr0; runbuf[rb * 2 * mr + mr + tid] = pr1; }
    __syncthreads();
    it = nx; nx = nn; xb ^= 1; rb ^= 1;
  }
}

// Static deal: item blockIdx.x, + gridDim.x, ...  (A work queue -- batches of eight items drawn from an atomic counter, so
// that the grid could be any size and a workgroup that shares its CU simply takes fewer -- was built and measured in round 3:
// inside the pipelined loop it costs the loop its scalar registers (spills, 1.44 ms); with the pipeline restarted per batch
// every restart is an atomic and three dependent loads at the latency of a saturated memory system, ~60 us: 1.15 ms against
// 0.70.  profiles/r03d_overlap_near_far.txt.  One WAVEFRONT per item, free-running, no barrier at all -- 2.5 % instead of 17 % of
// the row slots empty on this tree's 19-row leaves -- takes 0.86-0.96 ms whatever it keeps in flight: the set-up of an item is a
// third of that wavefront's instruction stream.  profiles/r03m_near_spmv_free_running_wavefronts.txt.)
template <int kRows, int kVecs>
__global__ __launch_bounds__(kSpmvWaves * kWave, kSpmvOcc) void near_spmv_pipe_kernel(DevicePlan d) {
  extern __shared__ double xs_all[];
  __shared__ double part[kSpmvWaves][kColRows];
  spmv_pipe_run<kRows, kVecs>(d, xs_all, part, blockIdx.x, gridDim.x, d.near_rec ? d.near_nitems_stream : d.near_nitems);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// Stokes near field with the block symmetry folded (DevicePlan::near_sym): 48 instead of 72 bytes per panel pair.
// The assembly writes the six entries (a,b), a <= b, of every 3x3 block (the quadrature blocks are symmetric bit for bit, the
// analytic self blocks to rounding, 1e-16); the 9-value rows are not kept.
// near_spmv_sym3: the scheme of near_spmv_kernel with a PANEL row per wavefront step -- per source panel three 16-byte
// vectors and the three x components from LDS feed nine FMAs into (y_x, y_y, y_z).
// ---------------------------------------------------------------------------------------------
#ifndef FMMBEM_SYM_OCC
#define FMMBEM_SYM_OCC 5
#endif
#ifndef FMMBEM_SYM_CHUNK
#define FMMBEM_SYM_CHUNK 1024
#endif
constexpr int kSymOcc = FMMBEM_SYM_OCC;               // workgroups per CU
constexpr int kSymChunk = FMMBEM_SYM_CHUNK;           // source panels of x staged at a time (3 x 8 KiB)
template <int kRows, int kVecs>
__global__ __launch_bounds__(kSpmvWaves * kWave, kSymOcc) void near_spmv_sym3_kernel(DevicePlan d) {
  extern __shared__ double xs[];                      // [3][kSymChunk] doubles, then the run descriptors
  __shared__ double part[kSpmvWaves][kColRows][3];
  int* run_row0 = reinterpret_cast<int*>(xs + 3 * kSymChunk);
  int* run_off = run_row0 + d.max_runs;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  for (int item = blockIdx.x; item < d.sym_nitems; item += gridDim.x) {
    const int4 it = d.sym_items[item];
    const int t = it.x, r0 = it.y, nrows = it.z;       // panel rows
    const bool colsplit = it.w != 0;
    const int ncp = d.near_ncols[t];
    const Runs runs = load_runs(d, t, run_row0, run_off);
    const dvec2* blk = reinterpret_cast<const dvec2*>(d.near_sym + d.near_sym_off[t]) + (int64_t)r0 * 3 * ncp;
    double* yt = d.yt + 3 * (int64_t)(d.leaf_row0[t] + r0);
    for (int c0 = 0; c0 < ncp; c0 += kSymChunk) {
      const int cw = ncp - c0 < kSymChunk ? ncp - c0 : kSymChunk;
      if (c0) __syncthreads();
      for (int c = threadIdx.x; c < cw; c += blockDim.x) {
        const double* xp = d.xt + 3 * (int64_t)column_to_row(runs, c0 + c);
        xs[c] = xp[0]; xs[kSymChunk + c] = xp[1]; xs[2 * kSymChunk + c] = xp[2];
      }
      __syncthreads();
      const int seg = colsplit ? ((((cw + kSpmvWaves - 1) / kSpmvWaves) + 3) & ~3) : cw;
      const int v0 = colsplit ? wave * seg : 0, v1 = min(cw, v0 + seg);
      const int rstep = colsplit ? 1 : kSpmvWaves;
      for (int r = colsplit ? 0 : wave; r < nrows; r += colsplit ? kRows : kRows * kSpmvWaves) {
        const dvec2* row[kRows];
        double ax[kRows], ay[kRows], az[kRows];
#pragma unroll
        for (int i = 0; i < kRows; ++i) {
          const int ri = r + i * rstep;
          row[i] = blk + (int64_t)(ri < nrows ? ri : r) * 3 * ncp + c0;
          ax[i] = ay[i] = az[i] = 0;
        }
        for (int c = v0 + lane; c < v1; c += kVecs * kWave) {
          dvec2 v[kRows][kVecs][3];
#pragma unroll
          for (int u = 0; u < kVecs; ++u) {
            const int cc = c + u * kWave;
            const bool ok = cc < v1;
#pragma unroll
            for (int i = 0; i < kRows; ++i)
#pragma unroll
              for (int k = 0; k < 3; ++k) v[i][u][k] = ok ? __builtin_nontemporal_load(&row[i][k * ncp + cc]) : dvec2{0, 0};
          }
#pragma unroll
          for (int u = 0; u < kVecs; ++u) {
            const int cc = c + u * kWave;
            if (cc < v1) {
              const double x0 = xs[cc], x1 = xs[kSymChunk + cc], x2 = xs[2 * kSymChunk + cc];
#pragma unroll
              for (int i = 0; i < kRows; ++i) {
                const dvec2 a = v[i][u][0], b = v[i][u][1], e = v[i][u][2];   // (xx,xy) (xz,yy) (yz,zz)
                ax[i] = fma(a.x, x0, fma(a.y, x1, fma(b.x, x2, ax[i])));
                ay[i] = fma(a.y, x0, fma(b.y, x1, fma(e.x, x2, ay[i])));
                az[i] = fma(b.x, x0, fma(e.x, x1, fma(e.y, x2, az[i])));
              }
            }
          }
        }
#pragma unroll
        for (int i = 0; i < kRows; ++i) { ax[i] = wave_sum(ax[i]); ay[i] = wave_sum(ay[i]); az[i] = wave_sum(az[i]); }
        if (lane == 0) {
#pragma unroll
          for (int i = 0; i < kRows; ++i) {
            const int ri = r + i * rstep;
            if (ri < nrows) {
              if (colsplit) { part[wave][ri][0] = ax[i]; part[wave][ri][1] = ay[i]; part[wave][ri][2] = az[i]; }
              else {
                double* y = yt + 3 * ri;
                y[0] = c0 ? y[0] + ax[i] : ax[i]; y[1] = c0 ? y[1] + ay[i] : ay[i]; y[2] = c0 ? y[2] + az[i] : az[i];
              }
            }
          }
        }
      }
      if (colsplit) {
        __syncthreads();
        if ((int)threadIdx.x < 3 * nrows) {
          const int ri = threadIdx.x / 3, a = threadIdx.x % 3;
          const double sum = ((part[0][ri][a] + part[1][ri][a]) + part[2][ri][a]) + part[3][ri][a];
          yt[3 * ri + a] = c0 ? yt[3 * ri + a] + sum : sum;
        }
      }
    }
    __syncthreads();                                  // xs / run descriptors / part are rewritten for the next item
  }
}

// ---------------------------------------------------------------------------------------------
// Hybrid near field (fmmbem_options.near_stream_fraction < 1; round 5).  The target leaves of such a plan are of two kinds:
//   * STREAMED leaves keep their blocks (near_sym / near_val) and are served by near_spmv_sym3 / near_spmv_pipe as before:
//     HBM-bound, the VALU idle 95 % of the time;
//   * RECOMPUTED leaves keep NO matrix.  near_recompute3g (Stokes) / near_recompute1 (Laplace) evaluate their far-regime entries (K
//     quadrature points per pair) every matvec from the source panels' points; the near-regime pairs (semi-analytic / fine-rule /
//     self entries, 4.5 %) were listed and evaluated once at plan creation (side_ptr / side_col / side_val, as the matrix-free
//     plans do) and are applied by near_side_items, the tail of the recompute kernels.
// The kernels run SIDE BY SIDE on streams of their own, each with its own register budget and a grid sized so that all are resident
// on every CU from start to end (launch_near_hybrid): the arithmetic of the one runs in the issue slots the other leaves empty, and
// the matrix bytes of the recomputed leaves are neither read nor stored.  (Two grids that each fill the chip serialise:
// profiles/r05a_near_split_step_a.txt; one kernel that takes both kinds of items pays the larger register budget on its streaming
// wavefronts and loses what it gains; a register-prefetch form of the Stokes kernel at one workgroup per CU measured 2 % behind
// the shipped one and is gone: profiles/r05b_hybrid_near_stokes_sweeps.txt.)
//
// Common to the recompute kernels: persistent 256-thread workgroups; lane = source panel, the partial sums of a wavefront's rows
// stay in the lane's registers over the columns and cross the lanes through ONE 16- (8-) value halving butterfly (17 shuffles
// instead of 90); item records by scalar loads, run descriptors two items ahead; a row's sum is formed in a fixed order.
// ---------------------------------------------------------------------------------------------


// v[0..15] per lane -> every lane returns the sum over all 64 lanes of value hyb_value_of(lane); fixed tree.
// One halving step: the lanes whose `BIT` is set keep the upper W values, the others the lower W, and each adds what its
// partner (lane ^ BIT) held of them.  (Written as a template per step: as one loop over the steps the compiler does not unroll
// it, indexes v[] at run time and emits a 16-way select per access -- 1 400 instructions instead of 120.)
template <int W, int BIT, int NV>
__device__ __forceinline__ void hyb_halve(double (&v)[NV], int lane) {
  const bool hi = (lane & BIT) != 0;
#pragma unroll
  for (int k = 0; k < W; ++k) {
    const double send = hi ? v[k] : v[k + W];
    const double keep = hi ? v[k + W] : v[k];
    v[k] = keep + __shfl_xor(send, BIT, kWave);
  }
}
__device__ __forceinline__ double hyb_reduce16(double (&v)[16], int lane) {
  hyb_halve<8, 32>(v, lane);
  hyb_halve<4, 16>(v, lane);
  hyb_halve<2, 8>(v, lane);
  hyb_halve<1, 4>(v, lane);
  double t = v[0];
  t += __shfl_xor(t, 2, kWave);
  t += __shfl_xor(t, 1, kWave);
  return t;
}
// the same for 8 values: every lane returns the sum over all lanes of value (lane >> 3) & 7
__device__ __forceinline__ double hyb_reduce8(double (&v)[8], int lane) {
  hyb_halve<4, 32>(v, lane);
  hyb_halve<2, 16>(v, lane);
  hyb_halve<1, 8>(v, lane);
  double t = v[0];
  t += __shfl_xor(t, 4, kWave);
  t += __shfl_xor(t, 2, kWave);
  t += __shfl_xor(t, 1, kWave);
  return t;
}
__device__ __forceinline__ int hyb_value_of(int lane) { return (lane >> 2) & 15; }   // bits 5..2 of the lane, bit 5 = value bit 3

#ifndef FMMBEM_RC_OCC
#define FMMBEM_RC_OCC 2
#endif
#ifndef FMMBEM_RC_PRIO
#define FMMBEM_RC_PRIO 0
#endif
constexpr int kRcOcc = FMMBEM_RC_OCC;                 // register budget of the recompute kernel: 512 / kRcOcc VGPRs
constexpr int kRcChunk = kSpmvWaves * kWave;          // source panels per chunk = threads of the workgroup

// ys[rows of the recomputed leaves] = the rows' LISTED entries (near-regime pairs, evaluated once at plan creation) times x:
// a CSR product; a workgroup takes a run of rows with <= 256 entries, thread = entry for the products (LDS), then thread = row adds
// its products in entry order.  The recompute kernels run this as the TAIL of their own item loop -- a workgroup that is out of
// recompute items takes listed entries -- so that it needs no residency of its own: as a third kernel beside the two it either
// sat on the recompute stream's critical path (0.34 ms in front of that kernel) or, on a stream of its own, took the registers
// the recompute workgroups were sized for and starved them until the streaming kernel had finished (profiles/r05b: the f = 0.4
// trace).  near_side_add_kernel adds ys to y after the join.
template <int DOF>
__device__ __forceinline__ void near_side_items(const DevicePlan& d, double* prod /* [DOF][256] doubles of LDS */) {
  constexpr int kT = kSpmvWaves * kWave;
  const int tid = threadIdx.x;
  for (int item = blockIdx.x; item < d.side_nitems; item += gridDim.x) {
    const int4 it = d.side_items[item];
    const int e0 = it.x, e1 = it.y, r0 = it.z, nrows = it.w - it.z;
    double acc[DOF];
#pragma unroll
    for (int a = 0; a < DOF; ++a) acc[a] = 0.0;
    const int64_t mine_lo = tid < nrows ? d.side_ptr[r0 + tid] : 0, mine_hi = tid < nrows ? d.side_ptr[r0 + tid + 1] : 0;
    for (int base = e0; base < e1; base += kT) {         // more than one pass only for a single row of > 256 entries
      const int k = base + tid;
      if (k < e1) {                                      // thread = entry: the product of the entry with the source's charge
        const int64_t cj = d.side_col[k];
        if constexpr (DOF == 3) {
          const double* m = d.side_val + 9 * (int64_t)k;
          const double y0 = d.xt[3 * cj], y1 = d.xt[3 * cj + 1], y2 = d.xt[3 * cj + 2];
          prod[tid] = fma(m[0], y0, fma(m[1], y1, m[2] * y2));
          prod[kT + tid] = fma(m[3], y0, fma(m[4], y1, m[5] * y2));
          prod[2 * kT + tid] = fma(m[6], y0, fma(m[7], y1, m[8] * y2));
        } else {
          prod[tid] = d.side_val[k] * d.xt[cj];
        }
      }
      __syncthreads();
      if (tid < nrows) {                                 // thread = row: its products, in entry order
        const int lo = (int)(mine_lo > base ? mine_lo - base : 0);
        const int hi = (int)((mine_hi < (int64_t)base + kT ? mine_hi : (int64_t)base + kT) - base);
        for (int q = lo; q < hi; ++q)
#pragma unroll
          for (int a = 0; a < DOF; ++a) acc[a] += prod[a * kT + q];
      }
      __syncthreads();
    }
    if (tid < nrows)
#pragma unroll
      for (int a = 0; a < DOF; ++a) d.ys[(int64_t)DOF * (r0 + tid) + a] = acc[a];
  }
}

// y_tree[rows of the recomputed leaves] += their listed entries' sums (near_side_items), once both kernels are done
__global__ void near_side_add_kernel(DevicePlan d) {
  for (int item = blockIdx.x; item < d.rc_nitems; item += gridDim.x) {
    const RcItem it = d.rc_items[item];
    const int n = it.nrows * d.dof;
    const int64_t u0 = (int64_t)it.prow0 * d.dof;
    for (int k = threadIdx.x; k < n; k += blockDim.x) d.yt[u0 + k] += d.ys[u0 + k];
  }
}

typedef __attribute__((address_space(4))) RcItem ConstRcItem;
__device__ __forceinline__ RcItem load_rc_item(const ConstRcItem* r) {
  RcItem o;
  o.prow0 = r->prow0; o.nrows = r->nrows; o.ncp = r->ncp; o.nruns = r->nruns; o.run_begin = r->run_begin; o.leaf = r->leaf; o.pad = 0;
  return o;
}

// ---------------------------------------------------------------------------------------------
// near_recompute3g_kernel (StokesSphericalBEM): the source panels are loaded STRAIGHT INTO LDS (global_load_lds_dwordx4: no
// register holds a prefetch), so that the kernel fits 168 VGPRs and TWO of its workgroups are resident per CU beside two
// streaming ones (88 VGPRs: near_spmv_sym3<1, 2>) -- at one wavefront per SIMD every VALU instruction costs ~3 ns whatever it is
// (the lone-wavefront issue rate of the M2L study), at two the gaps of the one are the other's.
//   * packed per-panel records built once per plan (rc_src: 4 points, centroid, area = 16 doubles = 8 pieces of 16 bytes;
//     rc_nrm: the normal) and the charges padded to 4 doubles per panel by gather_x (xt4): a source panel is 10 (12) pieces;
//   * a chunk is 128 source panels: wavefront w loads pieces 2i + (w >> 1), i = 0.., of the 64 panels of half (w & 1) -- one
//     global_load_lds per piece, destination = wave-uniform LDS base + lane x 16, i.e. the image [piece][panel] the arithmetic
//     reads back with ds_read_b128; two buffers: the chunk after this one (of this item, or the first of the next) is in flight
//     while this one is worked on; ONE barrier per chunk (s_waitcnt vmcnt(0) + raw s_barrier: the loads of the next chunk are
//     issued behind it, so nothing is drained early);
//   * the rows' centroids (pieces 6 and 7 of the rows' own records) and the run descriptors of the item after the next arrive the
//     same way; item records by scalar loads.  No ordinary vector load is left in the loop (hipcc waits vmcnt(0) at the first
//     use of one while a global_load_lds is in flight).
// ---------------------------------------------------------------------------------------------
#ifndef FMMBEM_RCG_OCC
#define FMMBEM_RCG_OCC 3
#endif
#ifndef FMMBEM_RCG_QSCHED
#define FMMBEM_RCG_QSCHED 1
#endif
constexpr int kRcgChunk = 128;                        // source panels per chunk
#ifndef FMMBEM_RCG_ROWS
#define FMMBEM_RCG_ROWS 5                             /* S2 R2, f = 0.6 (near ms): 5 rows 1.50, 4 rows 1.56-1.58 (profiles/r05h) */
#endif
constexpr int kRcgRows = FMMBEM_RCG_ROWS;             // rows per block (3 sums each through the 16-value butterfly)
constexpr int kRcgBlocks = 16 / FMMBEM_RCG_ROWS;      // row blocks per wavefront: items of up to 4 x 15 = 60 (4 x 16 = 64) rows, i.e. whole leaves
constexpr int kRcgTgt = 64;                           // target slots per item
typedef __attribute__((address_space(1))) const void* GldsSrc;
typedef __attribute__((address_space(3))) void* GldsDst;
__device__ __forceinline__ void glds16(const double* g, double* l) { __builtin_amdgcn_global_load_lds((GldsSrc)g, (GldsDst)l, 16, 0, 0); }
__device__ __forceinline__ void glds4(const int* g, int* l) { __builtin_amdgcn_global_load_lds((GldsSrc)g, (GldsDst)l, 4, 0, 0); }
__device__ __forceinline__ void glds_wait_and_meet() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <bool TRAC>
__global__ __launch_bounds__(kSpmvWaves * kWave, FMMBEM_RCG_OCC) void near_recompute3g_kernel(DevicePlan d) {
  constexpr int P = TRAC ? 12 : 10;                   // 16-byte pieces per source panel
  constexpr int kBufD = P * kRcgChunk * 2;            // doubles per chunk buffer
  extern __shared__ double lds[];                     // buf [2][P][128][2], tgt [2][2][64][2] doubles, then runs [3][2][max_runs] ints
  double* const tgt_all = lds + 2 * kBufD;
  int* const runbuf = reinterpret_cast<int*>(tgt_all + 2 * 2 * kRcgTgt * 2);
  __shared__ int tflag_all[TRAC ? 2 * kRcgTgt : 1];   // TRACTION plans: the rows' flags (two items in flight)
  __shared__ double tot_all[kSpmvWaves * kRcgBlocks * 16];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, tid = threadIdx.x;
  const int mr = d.max_runs, nitems = d.rc_nitems, step = gridDim.x;
  const double sc = 1. / 2 / d.mu;
  const double w0 = d.qw[0], w1 = d.nq > 1 ? d.qw[1] : 0.0;        // the rules the host admits: weights of points 1.. all equal
  const int nq = d.nq;
  const ConstRcItem* recs = reinterpret_cast<const ConstRcItem*>(reinterpret_cast<uintptr_t>(d.rc_items));
  const int half = wave & 1, grp = wave >> 1;

  int item = blockIdx.x;
  if (item >= nitems) return;
  RcItem it = load_rc_item(recs + item);
  RcItem nx = load_rc_item(recs + (item + step < nitems ? item + step : nitems - 1));
  // run descriptors: slot of an item = (its ordinal in this workgroup's sequence) % 3
  auto runs_of = [&](int slot, int nruns) { return Runs{runbuf + slot * 2 * mr, runbuf + slot * 2 * mr + mr, nruns}; };
  for (int i = tid; i < it.nruns; i += blockDim.x) { runbuf[i] = d.near_run_row0[it.run_begin + i]; runbuf[mr + i] = d.near_run_off[it.run_begin + i]; }
  for (int i = tid; i < nx.nruns; i += blockDim.x) { runbuf[2 * mr + i] = d.near_run_row0[nx.run_begin + i]; runbuf[3 * mr + i] = d.near_run_off[nx.run_begin + i]; }
  __syncthreads();
  // one chunk of an item's source panels, and an item's row centroids, straight into LDS
  auto issue_chunk = [&](const Runs& runs, int c0, int ncp, double* buf) {
    const int c = c0 + kWave * half + lane;
    const size_t j = (size_t)column_to_row(runs, c < ncp ? c : 0);           // columns past the end repeat column 0; masked where they are used
    const double* rs = d.rc_src + 16 * j;
    const double* rx = d.xt4 + 4 * j;
    double* dst = buf + (size_t)(kWave * half) * 2;
#pragma unroll
    for (int i = 0; i < P / 2; ++i) {
      const int k = 2 * i + grp;                        // this wavefront's piece of the step
      const double* g = k < 8 ? rs + 2 * k : k < 10 ? rx + 2 * (k - 8) : d.rc_nrm + 4 * j + 2 * (k - 10);
      glds16(g, dst + (size_t)k * kRcgChunk * 2);
    }
  };
  auto issue_targets = [&](const RcItem& r, double* tg, int slot) {
    if (wave < 2 && lane < r.nrows) glds16(d.rc_src + 16 * (size_t)(r.prow0 + lane) + 12 + 2 * wave, tg + (size_t)wave * kRcgTgt * 2);
    if constexpr (TRAC) { if (wave == 2 && lane < r.nrows) tflag_all[slot * kRcgTgt + lane] = d.bc[r.prow0 + lane]; }
  };
  int seq = 0;                                           // ordinal of `it` in this workgroup's sequence
  int gk = 0;                                            // chunks issued so far: the buffer parity
  issue_chunk(runs_of(0, it.nruns), 0, it.ncp, lds);
  issue_targets(it, tgt_all, 0);
  for (;; item += step, ++seq) {
    const bool more = item + step < nitems;
    const int i2 = item + 2 * step;
    const RcItem nn = load_rc_item(recs + (i2 < nitems ? i2 : nitems - 1));
    const int nrows = it.nrows, ncp = it.ncp, prow0 = it.prow0;
    const Runs runs = runs_of(seq % 3, it.nruns), nruns = runs_of((seq + 1) % 3, nx.nruns);
    if (i2 < nitems && wave >= 2) {                      // the run descriptors of the item after the next: wavefront 2 the rows, 3 the offsets
      const int* g = wave == 2 ? d.near_run_row0 : d.near_run_off;
      int* l = runbuf + ((seq + 2) % 3) * 2 * mr + (wave == 2 ? 0 : mr);
      for (int b = 0; b < nn.nruns; b += kWave)
        if (b + lane < nn.nruns) glds4(g + nn.run_begin + b + lane, l + b);
    }
    const int rq = nrows / kSpmvWaves, rrem = nrows % kSpmvWaves;
    const int rw = wave * rq + (wave < rrem ? wave : rrem);      // this wavefront's first row of the item
    const int nrw = rq + (wave < rrem ? 1 : 0);                  // ... and how many it has (0: none)
    const double* const tg = tgt_all + (size_t)(seq & 1) * 2 * kRcgTgt * 2;
    const int* const tflag = tflag_all + (TRAC ? (seq & 1) * kRcgTgt : 0);
    // a wavefront's rows in blocks of kRcgRows: the block's sums cross the lanes once per CHUNK into the wavefront's LDS slots -- a
    // whole leaf (up to 64 rows) is ONE item, its source panels are loaded once, and a chunk is worked on for as many row blocks as
    // the wavefront has (the time the next chunk's loads have to arrive in)
    const int nblk = (nrw + kRcgRows - 1) / kRcgRows;
    double* const totw = tot_all + wave * (kRcgBlocks * 16);       // this wavefront's row sums: [block][value]
    if (lane < kRcgBlocks * 16) totw[lane] = 0.0;
    const int nchunks = (ncp + kRcgChunk - 1) / kRcgChunk;
    for (int ck = 0; ck < nchunks; ++ck, ++gk) {
      const double* const buf = lds + (size_t)(gk & 1) * kBufD;
      glds_wait_and_meet();                              // this chunk has landed for everybody; nobody reads the other buffer any more
      {                                                  // the chunk after this one -- of this item, or the first of the next with its rows
        double* const nb = lds + (size_t)((gk + 1) & 1) * kBufD;
        if (ck + 1 < nchunks) issue_chunk(runs, (ck + 1) * kRcgChunk, ncp, nb);
        else if (more) { issue_chunk(nruns, 0, nx.ncp, nb); issue_targets(nx, tgt_all + (size_t)((seq + 1) & 1) * 2 * kRcgTgt * 2, (seq + 1) & 1); }
      }
      const int cw = ncp - ck * kRcgChunk < kRcgChunk ? ncp - ck * kRcgChunk : kRcgChunk;
      const dvec2* const bp = reinterpret_cast<const dvec2*>(buf);
      const dvec2* const tp = reinterpret_cast<const dvec2*>(tg);
#pragma unroll 1
      for (int blk = 0; blk < nblk; ++blk) {             // (not unrolled: one copy of the block's code, 153 VGPRs)
        const int rb = rw + blk * kRcgRows, nrb = nrw - blk * kRcgRows < kRcgRows ? nrw - blk * kRcgRows : kRcgRows;
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.0;
        for (int cg = 0; cg * kWave < cw; ++cg) {
          const int c = cg * kWave + lane;
          const bool valid = c < cw;
          dvec2 pc[P];
#pragma unroll
          for (int k = 0; k < P; ++k) pc[k] = bp[k * kRcgChunk + c];
          const double qx[4] = {pc[0].x, pc[1].y, pc[3].x, pc[4].y}, qy[4] = {pc[0].y, pc[2].x, pc[3].y, pc[5].x}, qz[4] = {pc[1].x, pc[2].y, pc[4].x, pc[5].y};
          const double sx = pc[6].x, sy = pc[6].y, sz = pc[7].x, A = pc[7].y;
          const double x0 = pc[8].x, x1 = pc[8].y, x2 = pc[9].x;
          double nx_ = 0, ny_ = 0, nz_ = 0;
          if constexpr (TRAC) { nx_ = pc[10].x; ny_ = pc[10].y; nz_ = pc[11].x; }
          const double near2 = valid ? 8.0 * A * (1.0 + 1e-9) : 1e300;      // lanes past the last column: every pair "listed", i.e. dropped
          const double wA0 = w0 * A, wA1 = w1 * A;
#pragma unroll
          for (int r = 0; r < kRcgRows; ++r) {
            if (r < nrb) {                                 // wave-uniform
              const dvec2 ta = tp[rb + r], tb = tp[kRcgTgt + rb + r];     // the row's centroid: LDS broadcasts
              const double txr = ta.x, tyr = ta.y, tzr = tb.x;
              const bool slow = mf_listed(txr - sx, tyr - sy, tzr - sz, near2);
              bool trac = false;
              if constexpr (TRAC) trac = tflag[rb + r] != 0;
              double u0 = 0, u1 = 0, u2 = 0;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const double wA = q ? (q < nq ? wA1 : 0.0) : wA0;
                const double ex = txr - qx[q], ey = tyr - qy[q], ez = tzr - qz[q];
                const double ir = rsqrt_pos(fma(ex, ex, fma(ey, ey, ez * ez)));
                const double ir3 = ir * ir * ir, dxq = fma(ex, x0, fma(ey, x1, ez * x2));
                if (TRAC && trac) {
                  const double g = wA * fma(ex, nx_, fma(ey, ny_, ez * nz_)) * dxq * (ir3 * ir * ir);
                  u0 = fma(g, ex, u0); u1 = fma(g, ey, u1); u2 = fma(g, ez, u2);
                } else {
                  const double f1 = wA * ir, g = wA * ir3 * dxq;
                  u0 = fma(f1, x0, fma(g, ex, u0)); u1 = fma(f1, x1, fma(g, ey, u1)); u2 = fma(f1, x2, fma(g, ez, u2));
                }
                if ((q + 1) % FMMBEM_RCG_QSCHED == 0) __builtin_amdgcn_sched_barrier(0);
              }
              const double f = (TRAC && trac) ? -3.0 : sc;
              v[3 * r] += slow ? 0.0 : u0 * f; v[3 * r + 1] += slow ? 0.0 : u1 * f; v[3 * r + 2] += slow ? 0.0 : u2 * f;
            }
            __builtin_amdgcn_sched_barrier(0);             // one row after the other
          }
        }
        const double t = hyb_reduce16(v, lane);           // every lane: the total of value (lane >> 2) & 15 of this block and chunk
        if ((lane & 3) == 0) totw[blk * 16 + hyb_value_of(lane)] += t;      // (the wavefront's own LDS slots: 153 VGPRs instead of 167)
      }
    }
    {
      const int blk = lane >> 4, val = lane & 15;          // lane = (block, value)
      const int nrb = nrw - blk * kRcgRows < kRcgRows ? nrw - blk * kRcgRows : kRcgRows;
      if (val < 3 * nrb) d.yt[3 * (int64_t)(prow0 + rw + blk * kRcgRows) + val] = totw[lane];
    }
    if (!more) break;
    it = nx; nx = nn;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // nothing of this wavefront in flight from here on
  __syncthreads();
  near_side_items<3>(d, lds);                            // the tail: the recomputed rows' listed entries (the chunk buffers are free)
}

// the packed records of near_recompute3g_kernel, once per plan: rc_src[panel][16] = 4 points (x, y, z each), centroid, area;
// rc_nrm[panel][4] = normal, 0
__global__ void rc_pack_kernel(DevicePlan d, double* __restrict__ rc_src, double* __restrict__ rc_nrm) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= d.n) return;
  const int64_t N = d.n;
  double* o = rc_src + 16 * i;
  for (int q = 0; q < 4; ++q) {
    const bool have = q < d.nq;
    o[3 * q] = have ? d.quad[(q * 3 + 0) * N + i] : 0.0; o[3 * q + 1] = have ? d.quad[(q * 3 + 1) * N + i] : 0.0; o[3 * q + 2] = have ? d.quad[(q * 3 + 2) * N + i] : 1.0;
  }
  o[12] = d.cx[i]; o[13] = d.cy[i]; o[14] = d.cz[i]; o[15] = d.area[i];
  rc_nrm[4 * i] = d.nx[i]; rc_nrm[4 * i + 1] = d.ny[i]; rc_nrm[4 * i + 2] = d.nz[i]; rc_nrm[4 * i + 3] = 0.0;
}
int rcg_item_rows() { return kSpmvWaves * kRcgBlocks * kRcgRows; }      // rows an item of near_recompute3g_kernel may hold (host: plan.hip)
hipError_t launch_rc_pack(const DevicePlan& d, double* rc_src, double* rc_nrm, hipStream_t s) {
  hipLaunchKernelGGL(rc_pack_kernel, dim3((unsigned)((d.n + 255) / 256)), dim3(256), 0, s, d, rc_src, rc_nrm);
  return hipGetLastError();
}

// The same for LaplaceSphericalBEM (one unknown per panel): items of <= 32 rows, wavefront w owns up to 8 of them, one sum per row
// (8 values through one butterfly); a source panel is 14 doubles (3 points, centroid, area, charge; 17 with the normal when the
// plan has NORMAL_DERIV targets).  far regime: G = sum_q w_q A / |x - q| (kernel/LaplaceSphericalBEM.hpp:198-203), dG/dn =
// sum_q w_q A (q - x).n / |q - x|^3 (:251-262) -- the arithmetic of mf_sweep_kernel.
constexpr int kRc1Rows = 8;                           // rows per wavefront

template <bool DN>
__global__ __launch_bounds__(kSpmvWaves * kWave, kRcOcc) void near_recompute1_kernel(DevicePlan d) {
#if FMMBEM_RC_PRIO
  __builtin_amdgcn_s_setprio(FMMBEM_RC_PRIO);          // this kernel's wavefronts issue ahead of the streaming kernel's on the SIMD they share
#endif
  constexpr int F = DN ? 17 : 14;
  extern __shared__ double src[];                     // [F][kRcChunk] doubles, then runbuf [2][2][max_runs] ints
  int* const runbuf = reinterpret_cast<int*>(src + F * kRcChunk);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, tid = threadIdx.x;
  const int mr = d.max_runs, nitems = d.rc_nitems, step = gridDim.x;
  const int64_t N = d.n;
  const double w0 = d.qw[0], w1 = d.nq > 1 ? d.qw[1] : 0.0;        // the rules the host admits: weights of points 1.. all equal
  const int nq = d.nq;
  const ConstRcItem* recs = reinterpret_cast<const ConstRcItem*>(reinterpret_cast<uintptr_t>(d.rc_items));

  int item = blockIdx.x;
  if (item >= nitems) return;
  RcItem it = load_rc_item(recs + item);
  RcItem nx = load_rc_item(recs + (item + step < nitems ? item + step : nitems - 1));
  for (int i = tid; i < it.nruns; i += blockDim.x) { runbuf[i] = d.near_run_row0[it.run_begin + i]; runbuf[mr + i] = d.near_run_off[it.run_begin + i]; }
  for (int i = tid; i < nx.nruns; i += blockDim.x) { runbuf[2 * mr + i] = d.near_run_row0[nx.run_begin + i]; runbuf[3 * mr + i] = d.near_run_off[nx.run_begin + i]; }
  __syncthreads();
  double pre[F];
  auto fetch = [&](const Runs& runs, int c, int ncp) {
    const unsigned j = (unsigned)column_to_row(runs, c < ncp ? c : 0);       // columns past the end repeat column 0; masked where they are used
    const double* qp = d.quad + j;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bool have = q < nq;
      pre[3 * q] = have ? qp[(q * 3 + 0) * N] : 0.0; pre[3 * q + 1] = have ? qp[(q * 3 + 1) * N] : 0.0; pre[3 * q + 2] = have ? qp[(q * 3 + 2) * N] : 1.0;
    }
    pre[9] = d.cx[j]; pre[10] = d.cy[j]; pre[11] = d.cz[j]; pre[12] = d.area[j];
    pre[13] = d.xt[j];
    if constexpr (DN) { pre[14] = d.nx[j]; pre[15] = d.ny[j]; pre[16] = d.nz[j]; }
  };
  int rb = 0;
  fetch(Runs{runbuf, runbuf + mr, it.nruns}, tid, it.ncp);
  for (;; item += step) {
    const bool more = item + step < nitems;
    const int i2 = item + 2 * step;
    const RcItem nn = load_rc_item(recs + (i2 < nitems ? i2 : nitems - 1));
    const int nrows = it.nrows, ncp = it.ncp, prow0 = it.prow0;
    const Runs runs{runbuf + rb * 2 * mr, runbuf + rb * 2 * mr + mr, it.nruns};
    const Runs nruns{runbuf + (rb ^ 1) * 2 * mr, runbuf + (rb ^ 1) * 2 * mr + mr, nx.nruns};
    int pr0 = 0, pr1 = 0;
    const bool prun = i2 < nitems && tid < nn.nruns;
    if (prun) { pr0 = d.near_run_row0[nn.run_begin + tid]; pr1 = d.near_run_off[nn.run_begin + tid]; }
    const int rq = nrows / kSpmvWaves, rrem = nrows % kSpmvWaves;
    const int rw = wave * rq + (wave < rrem ? wave : rrem);      // this wavefront's first row of the item
    const int nrw = rq + (wave < rrem ? 1 : 0);                  // ... and how many it has (0: none)
    double tx[kRc1Rows], ty[kRc1Rows], tz[kRc1Rows];
    int tbw[kRc1Rows];
#pragma unroll
    for (int r = 0; r < kRc1Rows; ++r) {
      const int64_t i = prow0 + rw + (r < nrw ? r : 0);
      tx[r] = d.cx[i]; ty[r] = d.cy[i]; tz[r] = d.cz[i];
      tbw[r] = DN ? d.bc[i] : 0;
    }
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = 0.0;
    const int vidx = (lane >> 3) & 7;
    const bool mine = nrw > 0 && (lane & 7) == 0 && vidx < nrw;
    const int nchunks = (ncp + kRcChunk - 1) / kRcChunk;
    for (int ck = 0; ck < nchunks; ++ck) {
      __syncthreads();
#pragma unroll
      for (int f = 0; f < F; ++f) src[f * kRcChunk + tid] = pre[f];
      __syncthreads();
      if (ck + 1 < nchunks) fetch(runs, (ck + 1) * kRcChunk + tid, ncp);
      else if (more) fetch(nruns, tid, nx.ncp);
      if (nrw <= 0) continue;
      const int cw = ncp - ck * kRcChunk < kRcChunk ? ncp - ck * kRcChunk : kRcChunk;
      double cur[F], nxt[F];
#pragma unroll
      for (int f = 0; f < F; ++f) cur[f] = src[f * kRcChunk + lane];
      for (int cg = 0; cg * kWave < cw; ++cg) {
        const int c = cg * kWave + lane;
        const bool valid = c < cw;
        if ((cg + 1) * kWave < cw) {
#pragma unroll
          for (int f = 0; f < F; ++f) nxt[f] = src[f * kRcChunk + c + kWave];
        }
        const double sx = cur[9], sy = cur[10], sz = cur[11], A = cur[12], xj = cur[13];
        double nx_ = 0, ny_ = 0, nz_ = 0;
        if constexpr (DN) { nx_ = cur[14]; ny_ = cur[15]; nz_ = cur[16]; }
        const double near2 = valid ? 8.0 * A * (1.0 + 1e-9) : 1e300;        // lanes past the last column: every pair "listed", i.e. dropped
        const double wA0 = w0 * A, wA1 = w1 * A;
#pragma unroll
        for (int r = 0; r < kRc1Rows; ++r) {
          if (r < nrw) {                                   // wave-uniform
            const bool slow = mf_listed(tx[r] - sx, ty[r] - sy, tz[r] - sz, near2);
            bool dn = false;
            if constexpr (DN) dn = __builtin_amdgcn_readfirstlane(tbw[r]) != 0;
            double u = 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const double wA = q ? (q < nq ? wA1 : 0.0) : wA0;
              const double ex = cur[3 * q] - tx[r], ey = cur[3 * q + 1] - ty[r], ez = cur[3 * q + 2] - tz[r];
              const double ir = rsqrt_pos(fma(ex, ex, fma(ey, ey, ez * ez)));
              if (DN && dn) u = fma(wA * fma(ex, nx_, fma(ey, ny_, ez * nz_)), ir * ir * ir, u);
              else u = fma(wA, ir, u);
            }
            // a listed pair contributes through the list, not here (a select: K = 1 puts the self pair's point ON the centroid)
            v[r] = fma(slow ? 0.0 : u, xj, v[r]);
          }
          __builtin_amdgcn_sched_barrier(0);               // one row after the other
        }
#pragma unroll
        for (int f = 0; f < F; ++f) cur[f] = nxt[f];
      }
    }
    if (nrw > 0) {
      const double tot = hyb_reduce8(v, lane);
      if (mine) d.yt[prow0 + rw + vidx] = tot;
    }
    if (!more) break;
    if (prun) { runbuf[rb * 2 * mr + tid] = pr0; runbuf[rb * 2 * mr + mr + tid] = pr1; }
    it = nx; nx = nn; rb ^= 1;
  }
  __syncthreads();
  near_side_items<1>(d, src);                            // the tail: the recomputed rows' listed entries
}

// Panel(p0, p1, p2) of kernel/LaplaceSphericalBEM.hpp:64-97 for every panel, in TREE order, from the caller's vertices (original
// order) and the permutation: centroid, normal (p2-p0) x (p1-p0) / 2A, area, the rule's points, the vertices transposed.  The
// arithmetic of host_plan.cpp fill_panel operation for operation, contraction OFF: the same bits as the host form (which a
// host-only plan and fmmbem_kernel_entries still use).
__global__ void panel_setup_kernel(int64_t n, const uint32_t* __restrict__ perm, const double* __restrict__ v_orig, int nq,
                                   const double* __restrict__ pts, double* __restrict__ cx, double* __restrict__ cy,
                                   double* __restrict__ cz, double* __restrict__ nx, double* __restrict__ ny, double* __restrict__ nz,
                                   double* __restrict__ area, double* __restrict__ quad, double* __restrict__ vert) {
#pragma clang fp contract(off)
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* v = v_orig + 9 * (size_t)perm[i];
  double p0[3], p1[3], p2[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { p0[k] = v[k]; p1[k] = v[3 + k]; p2[k] = v[6 + k]; }
  cx[i] = (p0[0] + p1[0] + p2[0]) / 3;
  cy[i] = (p0[1] + p1[1] + p2[1]) / 3;
  cz[i] = (p0[2] + p1[2] + p2[2]) / 3;
  const double a0[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  const double a1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
  const double c[3] = {a0[1] * a1[2] - a0[2] * a1[1], -(a0[0] * a1[2] - a0[2] * a1[0]), a0[0] * a1[1] - a0[1] * a1[0]};
  const double A = 0.5 * sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  area[i] = A;
  nx[i] = c[0] / 2 / A; ny[i] = c[1] / 2 / A; nz[i] = c[2] / 2 / A;
  for (int q = 0; q < nq; ++q)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      quad[((size_t)q * 3 + k) * n + i] = p0[k] * pts[3 * q] + p1[k] * pts[3 * q + 1] + p2[k] * pts[3 * q + 2];
#pragma unroll
  for (int k = 0; k < 9; ++k) vert[(size_t)k * n + i] = v[k];
}
hipError_t launch_panel_setup(int64_t n, const uint32_t* perm, const double* v_orig, int nq, const double* pts, double* cx, double* cy,
                              double* cz, double* nx, double* ny, double* nz, double* area, double* quad, double* vert, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(panel_setup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, perm, v_orig, nq, pts, cx, cy, cz, nx, ny,
                     nz, area, quad, vert);
  return hipGetLastError();
}

// diag[original unknown] = A_near[u,u]: the self-interaction K(s,s) that Preconditioners::Diagonal divides by
// (examples/BEM/Preconditioner.hpp:19-42).  selfcol[leaf] = first column of the leaf's own panels in its block.
__global__ void near_diag_kernel(DevicePlan d, const int* __restrict__ selfcol, double* __restrict__ out) {
  const int t = d.leaf_begin + blockIdx.x;
  const int dof = d.dof, nrows = dof * d.leaf_nrows[t], stride = d.near_stride[t], row0 = d.leaf_row0[t];
  if (d.near_rec && d.near_rec[t]) {                  // hybrid plan, recomputed leaf: the self entry is a listed (near-regime) pair
    for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
      const int64_t i = row0 + r / dof;
      const int a = r % dof;
      double v = 0;
      for (int64_t k = d.side_ptr[i]; k < d.side_ptr[i + 1]; ++k)
        if (d.side_col[k] == i) v = dof == 3 ? d.side_val[9 * k + 4 * a] : d.side_val[k];
      out[(int64_t)d.perm[i] * dof + a] = v;
    }
    return;
  }
  if (d.near_sym) {                                   // Stokes, symmetric blocks: (a,a) of the self block of panel row tr
    const int ncp = d.near_ncols[t];
    const dvec2* sym = reinterpret_cast<const dvec2*>(d.near_sym + d.near_sym_off[t]);
    for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
      const int tr = r / 3, a = r % 3, cp = selfcol[t] / 3 + tr;
      const dvec2* row = sym + (int64_t)tr * 3 * ncp;
      out[(int64_t)d.perm[row0 + tr] * 3 + a] = a == 0 ? row[cp].x : a == 1 ? row[ncp + cp].y : row[2 * ncp + cp].y;
    }
    return;
  }
  const double* blk = d.near_val + d.near_off[t];
  for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
    const int panel = row0 + r / dof;
    out[(int64_t)d.perm[panel] * dof + r % dof] = blk[(int64_t)r * stride + selfcol[t] + r];
  }
}

// Kernel::operator()(target, source) for m independent panel pairs (fmmbem_kernel_entries): panels [0, m) are the targets,
// [m, 2m) the sources; the same entry functions as the near-matrix assembly.
__global__ void kernel_entries_kernel(DevicePlan d, int m, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const V3 t = {d.cx[i], d.cy[i], d.cz[i]};
  if (d.kernel == 1) {
    double b[9];
    stokes_entry(d, t, d.bc[i], (int64_t)m + i, b);
#pragma unroll
    for (int k = 0; k < 9; ++k) out[9 * (size_t)i + k] = b[k];
  } else {
    out[i] = laplace_entry(d, t, d.bc[i], (int64_t)m + i);
  }
}

// one near-matrix row of a plan's own panels, evaluated now (fmmbem_plan_get_near_row on a recomputed leaf of a hybrid plan):
// out[c] (Laplace) or out[9 c ..] (Stokes) = K(panel prow, panel cols[c]) by the entry functions of the assembly
__global__ void near_row_eval_kernel(DevicePlan d, int64_t prow, const int* __restrict__ cols, int n, double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const V3 t = {d.cx[prow], d.cy[prow], d.cz[prow]};
  if (d.dof == 3) stokes_entry(d, t, d.bc[prow], cols[c], out + 9 * (size_t)c);
  else out[c] = laplace_entry(d, t, d.bc[prow], cols[c]);
}
hipError_t launch_near_row_eval(const DevicePlan& d, int64_t prow, const int* cols, int n, double* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(near_row_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, s, d, prow, cols, n, out);
  return hipGetLastError();
}

hipError_t launch_kernel_entries(const DevicePlan& d, int m, double* out, hipStream_t s) {
  if (m <= 0) return hipSuccess;
  hipLaunchKernelGGL(kernel_entries_kernel, dim3((m + 63) / 64), dim3(64), 0, s, d, m, out);
  return hipGetLastError();
}

hipError_t launch_near_diag(const DevicePlan& d, const int* selfcol, double* out, hipStream_t s) {
  const int nb = d.leaf_end - d.leaf_begin;
  if (nb <= 0) return hipSuccess;
  hipLaunchKernelGGL(near_diag_kernel, dim3(nb), dim3(64), 0, s, d, selfcol, out);
  return hipGetLastError();
}

hipError_t launch_near_assemble(const DevicePlan& d, hipStream_t s) {
  const int nb = d.leaf_end - d.leaf_begin;
  if (nb <= 0) return hipSuccess;
  const size_t lds = ((size_t)kAsmChunk + 2 * (size_t)d.max_runs) * sizeof(int);
  const dim3 g(nb < 256 * 8 ? nb : 256 * 8), b(256);
  // FMMBEM_ASM_COLS=0: the entry-per-thread kernel at every rule (A/B: the two give the same bits)
  const bool cols = !(std::getenv("FMMBEM_ASM_COLS") && std::atoi(std::getenv("FMMBEM_ASM_COLS")) == 0);
  if (cols && d.nq == 1) hipLaunchKernelGGL(near_assemble_cols_kernel<1>, g, b, lds, s, d);
  else if (cols && d.nq == 3) hipLaunchKernelGGL(near_assemble_cols_kernel<3>, g, b, lds, s, d);
  else if (cols && d.nq == 4) hipLaunchKernelGGL(near_assemble_cols_kernel<4>, g, b, lds, s, d);
  else hipLaunchKernelGGL(near_assemble_kernel, g, b, lds, s, d);
  return hipGetLastError();
}

hipError_t launch_near_assemble_stokes(const DevicePlan& d, hipStream_t s) {
  const int nb = d.leaf_end - d.leaf_begin;
  if (nb <= 0) return hipSuccess;
  const size_t lds = ((size_t)kAsmChunk + 2 * (size_t)d.max_runs) * sizeof(int);
  const dim3 g(nb < 256 * 8 ? nb : 256 * 8), b(256);
  const bool cols = !(std::getenv("FMMBEM_ASM_COLS") && std::atoi(std::getenv("FMMBEM_ASM_COLS")) == 0);
  if (cols && d.nq == 1) hipLaunchKernelGGL(near_assemble_stokes_cols_kernel<1>, g, b, lds, s, d);
  else if (cols && d.nq == 3) hipLaunchKernelGGL(near_assemble_stokes_cols_kernel<3>, g, b, lds, s, d);
  else if (cols && d.nq == 4) hipLaunchKernelGGL(near_assemble_stokes_cols_kernel<4>, g, b, lds, s, d);
  else hipLaunchKernelGGL(near_assemble_stokes_kernel, g, b, lds, s, d);
  return hipGetLastError();
}

hipError_t launch_near_matfree(const DevicePlan& d, hipStream_t s) {
  const int nb = d.leaf_end - d.leaf_begin;
  if (nb <= 0) return hipSuccess;
  if (d.dof == 3) {
    if (d.side_ptr && d.nq <= 4) {
      hipLaunchKernelGGL(mf_sweep3_apply_kernel, dim3(d.near_nitems < 256 * 16 ? d.near_nitems : 256 * 16), dim3(kWave), 2 * (size_t)d.max_runs * sizeof(int), s, d);
      return hipGetLastError();
    }
    const size_t lds3 = (size_t)(kAsmChunk / 2) * (3 * sizeof(double) + sizeof(int)) + 2 * (size_t)d.max_runs * sizeof(int);
    hipLaunchKernelGGL(near_matfree_stokes_kernel, dim3(nb < 256 * 8 ? nb : 256 * 8), dim3(256), lds3, s, d);
    return hipGetLastError();
  }
  if (!d.side_ptr) return hipErrorInvalidValue;                       // a matrix-free plan always lists its near-regime pairs (plan.hip)
  const size_t lds2 = 2 * (size_t)d.max_runs * sizeof(int);
  const dim3 g(d.near_nitems < 256 * 16 ? d.near_nitems : 256 * 16);
  if (d.nq <= 3) hipLaunchKernelGGL((mf_sweep_kernel<kMfApply, false>), g, dim3(kWave), lds2, s, d, nullptr, nullptr, nullptr);
  else hipLaunchKernelGGL((mf_sweep_kernel<kMfApply, true>), g, dim3(kWave), lds2, s, d, nullptr, nullptr, nullptr);
  return hipGetLastError();
}

// matrix-free plans, at creation: the list of near-regime pairs.  phase 0: side_cnt[row] = listed pairs of every owned row;
// phase 1: side_col filled through side_ptr (the exclusive scan of the counts); phase 2: side_val from (side_row, side_col)
hipError_t launch_mf_side(const DevicePlan& d, int phase, int* side_cnt, const int64_t* side_ptr, int* side_col, const int* side_row,
                          double* side_val, int64_t nside, hipStream_t s) {
  if (d.near_nitems <= 0) return hipSuccess;
  const size_t lds2 = 2 * (size_t)d.max_runs * sizeof(int);
  const dim3 g(d.near_nitems < 256 * 16 ? d.near_nitems : 256 * 16), b(kWave);
  if (phase == 0) hipLaunchKernelGGL((mf_sweep_kernel<kMfCount, false>), g, b, lds2, s, d, side_cnt, nullptr, nullptr);
  else if (phase == 1) hipLaunchKernelGGL((mf_sweep_kernel<kMfFill, false>), g, b, lds2, s, d, nullptr, side_ptr, side_col);
  else if (nside > 0) hipLaunchKernelGGL(mf_side_eval_kernel, dim3((unsigned)((nside + 255) / 256)), dim3(256), 0, s, d, side_row, side_col, side_val, nside);
  return hipGetLastError();
}

hipError_t launch_gather_x(const DevicePlan& d, const double* x, hipStream_t s) {
  const int bs = 256;
  hipLaunchKernelGGL(gather_x_kernel, dim3((unsigned)((d.n * d.dof + bs - 1) / bs)), dim3(bs), 0, s, d.perm, x, d.xt, d.n, d.dof, d.xt4);
  return hipGetLastError();
}

// Hybrid plans (near_stream_fraction < 1): the streaming kernel over the stored leaves on `s`; the recompute kernel over the others (and
// the listed entries) on a stream forked from and joined to `s`: kS workgroups per CU of the one (84-88 VGPRs), kR of the other
// (167-187), sized so that BOTH are resident on every CU from start to end.
hipError_t launch_near_hybrid(const DevicePlan& d, hipStream_t s, const HybridStreams& hs) {
  if (!d.near_rec) return hipErrorInvalidValue;
  const bool stokes = d.dof == 3;
  static const int kS3 = [] { const char* e = std::getenv("FMMBEM_HYB_WG_STREAM"); return e ? std::atoi(e) : 2; }();
  static const int kS1 = [] { const char* e = std::getenv("FMMBEM_HYB_WG_STREAM"); return e ? std::atoi(e) : 3; }();
  static const int kR = [] { const char* e = std::getenv("FMMBEM_HYB_WG_RECOMPUTE"); return e ? std::atoi(e) : 1; }();
  const dim3 b(kSpmvWaves * kWave);
  const bool flag1 = stokes ? d.stokes_traction_targets != 0 : (d.n_act > 1 || d.act[0] == 1);   // TRACTION / NORMAL_DERIV targets present
  const int n_stream = stokes ? d.sym_nitems : d.near_nitems_stream;
  hipError_t e = hipSuccess;
  if (d.rc_nitems > 0) {
    // fork: the recompute kernel (its few, large workgroups must find room on every CU: launched first) on a stream of its own
    // beside the streaming kernel on `s`; it takes the listed entries' product as the tail of its item loop
    if ((e = hipEventRecord(hs.fork, s)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(hs.recompute, hs.fork, 0)) != hipSuccess) return e;
    const dim3 g(std::min(d.rc_nitems, 256 * kR));
    if (stokes) {                                        // sources straight into LDS: two workgroups per CU
      static const int kRg = [] { const char* e = std::getenv("FMMBEM_HYB_WG_RECOMPUTE"); return e ? std::atoi(e) : 2; }();
      const dim3 gg(std::min(d.rc_nitems, 256 * kRg));
      const size_t ldsg = (size_t)2 * (flag1 ? 12 : 10) * kRcgChunk * 2 * sizeof(double) + (size_t)2 * 2 * kRcgTgt * 2 * sizeof(double) + 6 * (size_t)d.max_runs * sizeof(int);
      if (flag1) hipLaunchKernelGGL((near_recompute3g_kernel<true>), gg, b, ldsg, hs.recompute, d);
      else hipLaunchKernelGGL((near_recompute3g_kernel<false>), gg, b, ldsg, hs.recompute, d);
    } else {
      const size_t ldsr = (size_t)(flag1 ? 17 : 14) * kRcChunk * sizeof(double) + 4 * (size_t)d.max_runs * sizeof(int);
      if (flag1) hipLaunchKernelGGL((near_recompute1_kernel<true>), g, b, ldsr, hs.recompute, d);
      else hipLaunchKernelGGL((near_recompute1_kernel<false>), g, b, ldsr, hs.recompute, d);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  if (n_stream > 0) {
    if (stokes) {
      const size_t lds3 = 3 * (size_t)kSymChunk * sizeof(double) + 2 * (size_t)d.max_runs * sizeof(int);
      // beside the recompute kernel: two source-panel vectors in flight per lane instead of three (2.11 against 2.10 ms alone) --
      // 88 VGPRs or fewer, so that 2 x 168 (two recompute workgroups) + 2 x 88 fit the 512 registers of a SIMD
      if (d.rc_nitems > 0) hipLaunchKernelGGL((near_spmv_sym3_kernel<1, 2>), dim3(std::min(n_stream, 256 * kS3)), b, lds3, s, d);
      else hipLaunchKernelGGL((near_spmv_sym3_kernel<1, 3>), dim3(std::min(n_stream, 256 * kSymOcc)), b, lds3, s, d);
    } else {
      const size_t lds2 = 2 * (size_t)kSpmvPipeChunk * sizeof(double) + 4 * (size_t)d.max_runs * sizeof(int);
      hipLaunchKernelGGL((near_spmv_pipe_kernel<2, 4>), dim3(std::min(n_stream, 256 * (d.rc_nitems > 0 ? kS1 : kSpmvOcc))), b, lds2, s, d);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  if (d.rc_nitems > 0) {
    if ((e = hipEventRecord(hs.join_recompute, hs.recompute)) != hipSuccess) return e;
    if ((e = hipStreamWaitEvent(s, hs.join_recompute, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL(near_side_add_kernel, dim3(std::min(d.rc_nitems, 256 * 8)), dim3(64), 0, s, d);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t launch_near_spmv(const DevicePlan& d, hipStream_t s) {
  if (d.near_rec) return hipErrorInvalidValue;         // hybrid plans go through launch_near_hybrid
  if (d.near_nitems <= 0) return hipSuccess;
  if (d.dof == 3 && d.near_sym) {
    const size_t lds3 = 3 * (size_t)kSymChunk * sizeof(double) + 2 * (size_t)d.max_runs * sizeof(int);
    // panel rows x source-panel vectors in flight per wavefront, red blood cell N = 524 288 (ms): 1x3 2.10, 1x2 2.11, 1x1 2.15,
    // 1x4 2.3-2.6, 2x2 3.39, 1x5 3.36, 4x1 4.14; 2x4, 4x2, 1x6 spill (7-9); the 9-value rows: 3.32
    hipLaunchKernelGGL((near_spmv_sym3_kernel<1, 3>), dim3(std::min(d.sym_nitems, 256 * kSymOcc)), dim3(kSpmvWaves * kWave), lds3, s, d);
    return hipGetLastError();
  }
  const size_t lds = (size_t)kSpmvChunk * sizeof(double) + 2 * (size_t)d.max_runs * sizeof(int);
  // rows x vectors in flight per wavefront measured at N = 1M (ms): 4x4 1.05,
  // 2x4 0.99, 4x2 1.14, 8x2 1.21, 2x2 1.09, 4x1 1.09; plain (non-nontemporal) loads +25 %
  // The grid is exactly what is resident at once (kSpmvOcc workgroups per CU): with one more per CU the stragglers
  // start when the others finish (0.98 ms instead of 0.79 at N = 1M); occupancy 6 (80 VGPRs) and 2x2 loads at
  // occupancy 8 measure within 3 % of this.
  const dim3 g(std::min(d.near_nitems, 256 * kSpmvOcc)), b(kSpmvWaves * kWave);
  if (d.dof == 1 && d.max_runs <= kSpmvWaves * kWave) {               // the pipelined form; a leaf with more runs than threads (never seen) takes the plain one
    const size_t lds2 = 2 * (size_t)kSpmvPipeChunk * sizeof(double) + 4 * (size_t)d.max_runs * sizeof(int);
    hipLaunchKernelGGL((near_spmv_pipe_kernel<2, 4>), g, b, lds2, s, d);
  } else {
    hipLaunchKernelGGL((near_spmv_kernel<2, 4>), g, b, lds, s, d);
  }
  return hipGetLastError();
}

// Result vectors assembled from per-shard slices (multi-GPU, include/fmmbem.h fmmbem_plan_assemble_slices_device): shard r
// computed the tree-order rows [cut[r], cut[r+1]); its slice starts at slices + r * chunk.  One thread per unknown.
__global__ void assemble_slices_kernel(const uint32_t* __restrict__ perm, const double* __restrict__ slices, double* __restrict__ y,
                                       int64_t n, int dof, int world, const int64_t* __restrict__ cut, int64_t chunk) {
  const int64_t u = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (u >= n * dof) return;
  const int64_t i = u / dof;
  const int a = (int)(u - i * dof);
  int r = 0;
  while (r + 1 < world && i >= cut[r + 1]) ++r;
  y[(int64_t)perm[i] * dof + a] = slices[(int64_t)r * chunk + (i - cut[r]) * dof + a];
}

hipError_t launch_assemble_slices(const DevicePlan& d, const double* slices, double* y, int world, const int64_t* d_cut, int64_t chunk,
                                  hipStream_t s) {
  const int bs = 256;
  hipLaunchKernelGGL(assemble_slices_kernel, dim3((unsigned)((d.n * d.dof + bs - 1) / bs)), dim3(bs), 0, s, d.perm, slices, y, d.n,
                     d.dof, world, d_cut, chunk);
  return hipGetLastError();
}

hipError_t launch_scatter_y(const DevicePlan& d, double* y, hipStream_t s) {
  const int64_t rows = (d.row_end - d.row_begin) * d.dof;
  if (rows <= 0) return hipSuccess;
  const int bs = 256;
  hipLaunchKernelGGL(scatter_y_kernel, dim3((unsigned)((rows + bs - 1) / bs)), dim3(bs), 0, s, d.perm, d.yt, y, d.row_begin, d.row_end, d.dof);
  return hipGetLastError();
}

}  // namespace fmmbem
