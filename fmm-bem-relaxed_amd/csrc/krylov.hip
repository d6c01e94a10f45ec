// krylov.hip -- the modified Gram-Schmidt column of the callers above the matvec (examples/BEM/GMRES.hpp:203-212,
// GMRES_Stokes.hpp the same loop on Vec<3,double> values): for k = 0..i:  h_k = <w, V_k>;  w -= h_k V_k;  then
// h_{i+1} = |w|,  V_{i+1} = w / h_{i+1}.  The reference runs 2(i + 1) + 2 passes over the vectors; so did solver.py with
// one torch call each, and at N = 1M the 27 iterations of the config-5 solve spent 5-6 ms of 37 there, most of it per-call
// overhead.  Here a column is ONE call and i + 3 launches: launch k subtracts h_{k-1} V_{k-1} and, on the updated w,
// accumulates <w, V_k> in the same sweep (the last launch accumulates <w, w>), then one scales and one collects the column.
// A dot product is 1 024 per-workgroup sums that whoever needs the total adds in index order: same bits every run.
// Same operations in the same order as the reference's loop; only the association inside a dot product differs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/fmmbem.h"

namespace fmmbem {
int fail(int code, const std::string& msg);        // plan.hip: records the message for fmmbem_last_error
struct SolverWs;
// plan.hip: where a plan lives, how many unknowns it has, and the slot in which it keeps the workspace of the solver below
// (freed with the plan through solver_ws_destroy)
int plan_solver_info(fmmbem_plan* plan, int* device, int64_t* unknowns, int* p_max, SolverWs*** slot);
void solver_ws_destroy(SolverWs* ws);
}

namespace {

constexpr int kBlocks = 1024, kThreads = 256;       // workgroups of a sweep = partial sums per dot product

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum of the kBlocks partial sums of one dot product, the same order in every workgroup that asks (unused slots hold zeros)
__device__ __forceinline__ double block_total(const double* __restrict__ partial, double* wsum) {
  double v[kBlocks / kThreads];
#pragma unroll
  for (int u = 0; u < kBlocks / kThreads; ++u) v[u] = partial[threadIdx.x + u * kThreads];
  double s = 0;
#pragma unroll
  for (int u = 0; u < kBlocks / kThreads; ++u) s += v[u];
  s = wave_sum64(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  double t = 0;
#pragma unroll
  for (int k = 0; k < kThreads / 64; ++k) t += wsum[k];
  return t;
}

// w -= h_prev * v_prev (when v_prev; h_prev = the total of part_prev, formed by every workgroup for itself -- no atomics, no
// device-wide fences: on this part a device-scope fence per workgroup writes the L2 back and cost 30 us per sweep), then this
// workgroup's share of <w, v_dot> (v_dot == nullptr: <w, w>) into part_out[blockIdx.x]
// vec: every vector of the call starts on a 16-byte boundary (the host checks the pointers and the stride); otherwise -- an odd
// ldv with Laplace's one unknown per panel puts every other row of V on an 8-byte boundary -- the whole sweep is scalar
__global__ __launch_bounds__(kThreads) void mgs_step_kernel(int64_t n, double* __restrict__ w, const double* __restrict__ v_prev,
                                                             const double* __restrict__ part_prev, const double* __restrict__ v_dot,
                                                             double* __restrict__ part_out, int vec) {
  __shared__ double wsum[kThreads / 64];
  const double hp = v_prev ? block_total(part_prev, wsum) : 0.0;
  double acc = 0;
  // the slots of this row no workgroup of this grid writes are part of every total: zero them here, every call (a scratch
  // reused after a call with a larger n would otherwise add stale partial sums)
  if (blockIdx.x == 0)
    for (int k = gridDim.x + threadIdx.x; k < kBlocks; k += kThreads) part_out[k] = 0.0;
  // four consecutive elements per thread and step, as two 16-byte vectors per array: all loads of a step are in flight before
  // the first FMA
  const int64_t n4 = vec ? n >> 2 : 0;
  typedef double dv2 __attribute__((ext_vector_type(2)));
  dv2* w2 = reinterpret_cast<dv2*>(w);
  const dv2* p2 = reinterpret_cast<const dv2*>(v_prev);
  const dv2* d2 = reinterpret_cast<const dv2*>(v_dot);
  for (int64_t q = blockIdx.x * (int64_t)kThreads + threadIdx.x; q < n4; q += (int64_t)gridDim.x * kThreads) {
    dv2 wa = w2[2 * q], wb = w2[2 * q + 1];
    dv2 pa = {0, 0}, pb = {0, 0}, da, db;
    if (v_prev) { pa = p2[2 * q]; pb = p2[2 * q + 1]; }
    if (v_dot) { da = d2[2 * q]; db = d2[2 * q + 1]; }
    if (v_prev) {
      wa.x = fma(-hp, pa.x, wa.x); wa.y = fma(-hp, pa.y, wa.y); wb.x = fma(-hp, pb.x, wb.x); wb.y = fma(-hp, pb.y, wb.y);
      w2[2 * q] = wa; w2[2 * q + 1] = wb;
    }
    if (!v_dot) { da = wa; db = wb; }
    acc = fma(wa.x, da.x, acc); acc = fma(wa.y, da.y, acc); acc = fma(wb.x, db.x, acc); acc = fma(wb.y, db.y, acc);
  }
  for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {   // n mod 4 leftovers
    double wi = w[i];
    if (v_prev) { wi = fma(-hp, v_prev[i], wi); w[i] = wi; }
    acc = fma(wi, v_dot ? v_dot[i] : wi, acc);
  }
  acc = wave_sum64(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int k = 0; k < kThreads / 64; ++k) s += wsum[k];
    part_out[blockIdx.x] = s;
  }
}

// x += a z
__global__ __launch_bounds__(kThreads) void mgs_axpy_kernel(int64_t n, double* __restrict__ x, double a, const double* __restrict__ z) {
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) x[i] = fma(a, z[i], x[i]);
}

// v_next = w / |w|, |w|^2 = the total of part_norm
__global__ __launch_bounds__(kThreads) void mgs_scale_kernel(int64_t n, const double* __restrict__ w, const double* __restrict__ part_norm,
                                                              double* __restrict__ v_next, int vec) {
  __shared__ double wsum[kThreads / 64];
  const double inv = 1.0 / sqrt(block_total(part_norm, wsum));
  typedef double dv2 __attribute__((ext_vector_type(2)));
  const int64_t n2 = vec ? n >> 1 : 0;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n2; q += (int64_t)gridDim.x * blockDim.x) {
    dv2 v = reinterpret_cast<const dv2*>(w)[q];
    v.x *= inv; v.y *= inv;
    reinterpret_cast<dv2*>(v_next)[q] = v;
  }
  for (int64_t i = (n2 << 1) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v_next[i] = w[i] * inv;
}

// h[k] = total of the k-th row of partial sums, k <= ncols; the last one is |w|^2 -> |w|
__global__ __launch_bounds__(kThreads) void mgs_finish_kernel(const double* __restrict__ partial, int ncols, double* __restrict__ h) {
  __shared__ double wsum[kThreads / 64];
  const double t = block_total(partial + (size_t)blockIdx.x * kBlocks, wsum);
  if (threadIdx.x == 0) h[blockIdx.x] = (int)blockIdx.x == ncols ? sqrt(t) : t;
}

}  // namespace

extern "C" int fmmbem_mgs_column_device(int64_t n, double* d_w, const double* d_V, int64_t ldv, int ncols, double* d_h,
                                        double* d_vnext, double* d_scratch, void* stream) {
  if (n <= 0 || !d_w || !d_V || ncols < 1 || !d_h || !d_vnext || !d_scratch || ldv < n)
    return fmmbem::fail(FMMBEM_ERR_INVALID, "fmmbem_mgs_column_device: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t want = (n / 4 + kThreads - 1) / kThreads + 1;
  const int grid = (int)(want < kBlocks ? want : kBlocks);          // slots grid .. kBlocks-1 of a row are zeroed by the sweep itself
  const auto a16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int vec = a16(d_w) && a16(d_V) && a16(d_vnext) && (ldv & 1) == 0;
  for (int k = 0; k <= ncols; ++k) {
    const double* v_prev = k ? d_V + (int64_t)(k - 1) * ldv : nullptr;
    const double* v_dot = k < ncols ? d_V + (int64_t)k * ldv : nullptr;
    hipLaunchKernelGGL(mgs_step_kernel, dim3(grid), dim3(kThreads), 0, s, n, d_w, v_prev, k ? d_scratch + (size_t)(k - 1) * kBlocks : nullptr,
                       v_dot, d_scratch + (size_t)k * kBlocks, vec);
  }
  hipLaunchKernelGGL(mgs_scale_kernel, dim3(grid), dim3(kThreads), 0, s, n, d_w, d_scratch + (size_t)ncols * kBlocks, d_vnext, vec);
  hipLaunchKernelGGL(mgs_finish_kernel, dim3(ncols + 1), dim3(kThreads), 0, s, d_scratch, ncols, d_h);
  return hipGetLastError() == hipSuccess ? FMMBEM_OK : fmmbem::fail(FMMBEM_ERR_HIP, "fmmbem_mgs_column_device: launch failed");
}

extern "C" int fmmbem_mgs_scratch_doubles(int max_cols) { return (max_cols + 1) * kBlocks; }


// ================================================================================================================
// Relaxed GMRES / FGMRES resident on the device (include/fmmbem.h; examples/BEM/GMRES.hpp:143-252, :276-380,
// GMRES_Stokes.hpp:173-320, SolverOptions.hpp:25-38).  Host side: the (R+1) x R Hessenberg matrix, the Givens rotations, the
// residual estimate, predict_p, restart, back substitution.  Device side: everything of length n.
// ================================================================================================================
namespace {

// w += a v, and this workgroup's share of <w, w> afterwards into part_out (r0 = A x0 - b and its norm in one sweep;
// v == nullptr: only the norm)
__global__ __launch_bounds__(kThreads) void axpy_norm_kernel(int64_t n, double* __restrict__ w, double a, const double* __restrict__ v,
                                                              double* __restrict__ part_out) {
  __shared__ double wsum[kThreads / 64];
  if (blockIdx.x == 0)
    for (int k = gridDim.x + threadIdx.x; k < kBlocks; k += kThreads) part_out[k] = 0.0;
  double acc = 0;
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    double wi = w[i];
    if (v) { wi = fma(a, v[i], wi); w[i] = wi; }
    acc = fma(wi, wi, acc);
  }
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int k = 0; k < kThreads / 64; ++k) s += wsum[k];
    part_out[blockIdx.x] = s;
  }
}

// out = a * in  (V_0 = -w / beta), or out = r .* in (the diagonal preconditioner) when r
__global__ __launch_bounds__(kThreads) void scale_kernel(int64_t n, const double* __restrict__ in, double a, const double* __restrict__ r,
                                                          double* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
    out[i] = r ? r[i] * in[i] : a * in[i];
}

// x += sum_j y[j] * (r .* B_j), j ascending as the reference's loop (GMRES.hpp:237-241 with M = identity or diagonal;
// FGMRES :368-371 with B = Z): the basis is read once, x once
__global__ __launch_bounds__(kThreads) void update_x_kernel(int64_t n, double* __restrict__ x, const double* __restrict__ B, int64_t ldb, int ncols,
                                                             const double* __restrict__ y, const double* __restrict__ r) {
  for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    double acc = x[i];
    const double ri = r ? r[i] : 1.0;
    for (int j = 0; j < ncols; ++j) {
      const double b = B[(int64_t)j * ldb + i];
      acc = fma(y[j], r ? ri * b : b, acc);
    }
    x[i] = acc;
  }
}

}  // namespace

namespace fmmbem {

struct SolverWs {
  int device = 0;
  int64_t n = 0, ld = 0;
  int vcols = 0, zcols = 0, hcap = 0;
  double *V = nullptr, *Z = nullptr, *w = nullptr, *z = nullptr, *d_h = nullptr, *d_scratch = nullptr, *d_y = nullptr;
  double *d_xb = nullptr, *d_recip = nullptr;       // staging of the host-pointer entry point
  double* h_pin = nullptr;                          // pinned: the Hessenberg column / the y coefficients cross here
};

void solver_ws_destroy(SolverWs* ws) {
  if (!ws) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(ws->device);
  for (double* p : {ws->V, ws->Z, ws->w, ws->z, ws->d_h, ws->d_scratch, ws->d_y, ws->d_xb, ws->d_recip})
    if (p) (void)hipFree(p);
  if (ws->h_pin) (void)hipHostFree(ws->h_pin);
  (void)hipSetDevice(prev);
  delete ws;
}

}  // namespace fmmbem

namespace {

using fmmbem::SolverWs;
using fmmbem::fail;

#define KRY_HIP(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return fail(e_ == hipErrorOutOfMemory ? FMMBEM_ERR_ALLOC : FMMBEM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define KRY_TRY(expr) do { int rc_ = (expr); if (rc_ != FMMBEM_OK) return rc_; } while (0)

struct DevGuard {
  int prev = 0;
  bool on = false;
  explicit DevGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) on = true; }
  ~DevGuard() { if (on) (void)hipSetDevice(prev); }
};

// FMMBEM_KRYLOV_FAIL_GROW=k (tests only): the k-th allocation of the solver workspace in this process fails once, as an
// out-of-memory would -- the retry path of ensure_ws cannot be exercised otherwise without filling 288 GB
int grow(double** p, size_t doubles) {
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  static int calls = 0;
  static const int fail_at = [] { const char* e = std::getenv("FMMBEM_KRYLOV_FAIL_GROW"); return e ? std::atoi(e) : 0; }();
  if (fail_at > 0 && ++calls == fail_at) return fail(FMMBEM_ERR_ALLOC, "solver workspace: allocation failure injected by FMMBEM_KRYLOV_FAIL_GROW");
  KRY_HIP(hipMalloc(reinterpret_cast<void**>(p), sizeof(double) * std::max<size_t>(doubles, 1)));
  return FMMBEM_OK;
}

// after a failed allocation: nothing of the workspace is trusted -- every buffer freed, every size zero, so that the next solve
// on this plan allocates from scratch instead of launching on the null pointers a half-grown workspace holds
void reset_ws(SolverWs* ws) {
  for (double** p : {&ws->w, &ws->z, &ws->V, &ws->Z, &ws->d_h, &ws->d_y, &ws->d_scratch}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  if (ws->h_pin) (void)hipHostFree(ws->h_pin);
  ws->h_pin = nullptr;
  ws->n = ws->ld = 0;
  ws->vcols = ws->zcols = ws->hcap = 0;
}

int grow_ws(SolverWs* ws, int64_t n, int vcols, int zcols) {
  const int64_t ld = (n + 1) & ~int64_t(1);                      // even stride: every basis vector on a 16-byte boundary
  if (ws->n != n) {
    reset_ws(ws);
    KRY_TRY(grow(&ws->w, (size_t)ld));
    KRY_TRY(grow(&ws->z, (size_t)ld));
    ws->n = n; ws->ld = ld;
  }
  if (ws->vcols < vcols) { ws->vcols = 0; KRY_TRY(grow(&ws->V, (size_t)ld * vcols)); ws->vcols = vcols; }
  if (ws->zcols < zcols) { ws->zcols = 0; KRY_TRY(grow(&ws->Z, (size_t)ld * zcols)); ws->zcols = zcols; }
  if (ws->hcap < vcols + 1) {
    const int hcap = vcols + 1;
    ws->hcap = 0;
    KRY_TRY(grow(&ws->d_h, (size_t)hcap));
    KRY_TRY(grow(&ws->d_y, (size_t)hcap));
    KRY_TRY(grow(&ws->d_scratch, (size_t)fmmbem_mgs_scratch_doubles(hcap)));
    if (ws->h_pin) (void)hipHostFree(ws->h_pin);
    ws->h_pin = nullptr;
    KRY_HIP(hipHostMalloc(reinterpret_cast<void**>(&ws->h_pin), sizeof(double) * (size_t)hcap, hipHostMallocDefault));
    ws->hcap = hcap;                                             // sizes are committed only once every buffer of the group exists
  }
  return FMMBEM_OK;
}

// the workspace a plan keeps between solves (an inner-plan preconditioner solves once per outer iteration: no allocation there)
int ensure_ws(SolverWs** slot, int device, int64_t n, int vcols, int zcols, SolverWs** out) {
  if (!*slot) { *slot = new SolverWs; (*slot)->device = device; }
  SolverWs* ws = *slot;
  const int rc = grow_ws(ws, n, vcols, zcols);
  if (rc != FMMBEM_OK) { reset_ws(ws); return rc; }
  *out = ws;
  return FMMBEM_OK;
}

int sweep_grid(int64_t n) {
  const int64_t want = (n + kThreads - 1) / kThreads;
  return (int)std::max<int64_t>(1, std::min<int64_t>(want, kBlocks));
}

// |w| (after w += a v when v) -> host; one synchronisation
int axpy_norm(SolverWs* ws, double* w, double a, const double* v, hipStream_t s, double* out) {
  hipLaunchKernelGGL(axpy_norm_kernel, dim3(sweep_grid(ws->n)), dim3(kThreads), 0, s, ws->n, w, a, v, ws->d_scratch);
  hipLaunchKernelGGL(mgs_finish_kernel, dim3(1), dim3(kThreads), 0, s, ws->d_scratch, 0, ws->d_h);      // ncols = 0: row 0 is a norm
  KRY_HIP(hipMemcpyAsync(ws->h_pin, ws->d_h, sizeof(double), hipMemcpyDeviceToHost, s));
  KRY_HIP(hipStreamSynchronize(s));
  *out = ws->h_pin[0];
  return FMMBEM_OK;
}

// SolverOptions::predict_p (SolverOptions.hpp:25-38), the (unsigned) cast of the reference kept as far as it matters: a
// non-positive residual estimate saturates nu at 1 and asks for order 0, which every call site then raises to its floor
int predict_p(const fmmbem_solver_options& o, double eps) {
  if (!o.variable_p) return o.max_p;
  if (!(eps > 0.0)) return o.relax_type == FMMBEM_RELAX_BOURAS ? 0 : o.max_p;
  double v;
  if (o.relax_type == FMMBEM_RELAX_BOURAS) {
    const double alpha = 1.0 / std::min(eps, 1.0);
    const double nu = std::min(alpha * o.residual, 1.0);
    v = std::ceil(-std::log2(nu));
  } else {
    v = std::ceil(-std::log2(eps));
  }
  if (v < 0) return o.max_p;                                     // (unsigned)negative is huge: min(., max_p)
  return v > (double)o.max_p ? o.max_p : (int)v;
}

int order_for(const fmmbem_solver_options& o, double resid, int plan_pmax) {
  const int pp = predict_p(o, std::fabs(resid));
  int p;
  switch (o.order_rule) {
    case FMMBEM_ORDER_GMRES_STOKES: p = std::max(o.p_min, pp - 1); break;      // GMRES_Stokes.hpp:229
    case FMMBEM_ORDER_FGMRES_STOKES: p = std::max(5, pp); break;                // GMRES_Stokes.hpp:373
    default: p = std::max(1, pp); break;                                        // GMRES.hpp:195 (and :324: no order 0 here)
  }
  return std::min(p, plan_pmax);
}

void plane_rotation(double dx, double dy, double* cs, double* sn) {             // GMRES.hpp:88-105 GeneratePlaneRotation
  if (dy == 0.0) { *cs = 1.0; *sn = 0.0; }
  else if (std::fabs(dy) > std::fabs(dx)) { const double t = dx / dy; *sn = 1.0 / std::sqrt(1.0 + t * t); *cs = t * *sn; }
  else { const double t = dy / dx; *cs = 1.0 / std::sqrt(1.0 + t * t); *sn = t * *cs; }
}

int solve(fmmbem_plan* plan, const fmmbem_solver_options& so, double* d_x, const double* d_b, const fmmbem_preconditioner* M,
          fmmbem_solver_log* log, hipStream_t s, int depth);

// z = M(v): returns in *z either v itself (identity) or ws->z / the given buffer
int apply_pc(SolverWs* ws, const fmmbem_preconditioner* M, const double* v, double* zbuf, const double** z, hipStream_t s, int depth) {
  const int kind = M ? M->kind : FMMBEM_PC_IDENTITY;
  if (kind == FMMBEM_PC_IDENTITY) { *z = v; return FMMBEM_OK; }
  if (kind == FMMBEM_PC_DIAGONAL) {
    hipLaunchKernelGGL(scale_kernel, dim3(sweep_grid(ws->n)), dim3(kThreads), 0, s, ws->n, v, 0.0, M->reciprocals, zbuf);
    *z = zbuf;
    return FMMBEM_OK;
  }
  // LocalPC.hpp:35-41: fill(y, 0); GMRES(plan, y, x, options)
  KRY_HIP(hipMemsetAsync(zbuf, 0, sizeof(double) * (size_t)ws->n, s));
  KRY_TRY(solve(M->inner_plan, M->inner, zbuf, v, nullptr, nullptr, s, depth + 1));
  *z = zbuf;
  return FMMBEM_OK;
}

int solve(fmmbem_plan* plan, const fmmbem_solver_options& so, double* d_x, const double* d_b, const fmmbem_preconditioner* M,
          fmmbem_solver_log* log, hipStream_t s, int depth) {
  int device = 0, plan_pmax = 0;
  int64_t n = 0;
  SolverWs** slot = nullptr;
  KRY_TRY(fmmbem::plan_solver_info(plan, &device, &n, &plan_pmax, &slot));
  if (depth > 1) return fail(FMMBEM_ERR_UNSUPPORTED, "fmmbem_gmres: a preconditioner's inner solve cannot itself be preconditioned by a plan");
  if (so.restart < 1 || so.max_iters < 0 || so.max_p < 1 || !(so.residual > 0.0))
    return fail(FMMBEM_ERR_INVALID, "fmmbem_gmres: restart >= 1, max_iters >= 0, max_p >= 1, residual > 0 required");
  const int kind = M ? M->kind : FMMBEM_PC_IDENTITY;
  if (kind == FMMBEM_PC_DIAGONAL && !M->reciprocals) return fail(FMMBEM_ERR_INVALID, "fmmbem_gmres: diagonal preconditioner without reciprocals");
  if (kind == FMMBEM_PC_INNER_PLAN) {
    if (!M->inner_plan || M->inner_plan == plan) return fail(FMMBEM_ERR_INVALID, "fmmbem_gmres: the preconditioner needs a plan of its own");
    int dv = 0, pm = 0; int64_t nn = 0; SolverWs** sl = nullptr;
    KRY_TRY(fmmbem::plan_solver_info(M->inner_plan, &dv, &nn, &pm, &sl));
    if (dv != device || nn != n) return fail(FMMBEM_ERR_INVALID, "fmmbem_gmres: the preconditioner's plan must hold the same panels on the same device");
  }
  if (kind < 0 || kind > FMMBEM_PC_INNER_PLAN) return fail(FMMBEM_ERR_INVALID, "fmmbem_gmres: unknown preconditioner kind");
  const int R = so.restart;
  // the inner loop runs while i + 1 < R and i + 1 <= max_iters (GMRES.hpp:221): at most min(R, max_iters + 1) columns
  const int most = (int)std::min<int64_t>(R, (int64_t)so.max_iters + 1);
  SolverWs* ws = nullptr;
  KRY_TRY(ensure_ws(slot, device, n, most + 1, so.flexible ? most : 0, &ws));
  const int64_t ld = ws->ld;
  const int grid = sweep_grid(n);
  std::vector<double> H((size_t)(R + 1) * R, 0.0), cs(R, 0.0), sn(R, 0.0), sv(R + 1, 0.0);
  auto Hm = [&](int r, int c) -> double& { return H[(size_t)r * R + c]; };
  if (log) { log->iterations = 0; log->residual = 0.0; }

  // scale residual by |b| (GMRES.hpp:162)
  double normb = 0;
  KRY_TRY(axpy_norm(ws, const_cast<double*>(d_b), 0.0, nullptr, s, &normb));
  if (normb == 0.0) return FMMBEM_OK;        // b = 0: the reference divides by zero and stops on NaN; x = x0 is returned
  int cur_p = std::min(so.initial_p > 0 ? so.initial_p : so.max_p, plan_pmax);
  int iter = 0, i = 0;
  double resid = 0;
  do {                                       // outer (restart) loop, :166
    KRY_TRY(fmmbem_plan_execute_device(plan, cur_p, d_x, ws->w, s));           // w = A x at the kernel's current order
    double beta = 0;
    KRY_TRY(axpy_norm(ws, ws->w, -1.0, d_b, s, &beta));                        // w -= b; beta = |w|
    if (beta == 0.0) { resid = 0.0; break; }                                   // x solves the system exactly
    hipLaunchKernelGGL(scale_kernel, dim3(grid), dim3(kThreads), 0, s, n, ws->w, -1.0 / beta, (const double*)nullptr, ws->V);   // V_0 = -w / beta
    sv[0] = beta;
    i = -1;
    resid = sv[0] / normb;
    do {                                     // inner loop, :186
      ++i;
      ++iter;
      cur_p = order_for(so, resid, plan_pmax);
      const double* z = nullptr;
      KRY_TRY(apply_pc(ws, M, ws->V + (int64_t)i * ld, so.flexible ? ws->Z + (int64_t)i * ld : ws->z, &z, s, depth));
      KRY_TRY(fmmbem_plan_execute_device(plan, cur_p, z, ws->w, s));
      // modified Gram-Schmidt against V_0..V_i, |w|, V_{i+1} = w / |w| (:203-212): one library call, then the column to the host
      KRY_TRY(fmmbem_mgs_column_device(n, ws->w, ws->V, ld, i + 1, ws->d_h, ws->V + (int64_t)(i + 1) * ld, ws->d_scratch, s));
      KRY_HIP(hipMemcpyAsync(ws->h_pin, ws->d_h, sizeof(double) * (size_t)(i + 2), hipMemcpyDeviceToHost, s));
      KRY_HIP(hipStreamSynchronize(s));
      for (int k = 0; k <= i + 1; ++k) Hm(k, i) = ws->h_pin[k];
      for (int k = 0; k < i; ++k) {          // PlaneRotation, :108-117
        const double t = cs[k] * Hm(k, i) + sn[k] * Hm(k + 1, i);
        Hm(k + 1, i) = -sn[k] * Hm(k, i) + cs[k] * Hm(k + 1, i);
        Hm(k, i) = t;
      }
      plane_rotation(Hm(i, i), Hm(i + 1, i), &cs[i], &sn[i]);
      {
        const double t = cs[i] * Hm(i, i) + sn[i] * Hm(i + 1, i);
        Hm(i + 1, i) = -sn[i] * Hm(i, i) + cs[i] * Hm(i + 1, i);
        Hm(i, i) = t;
      }
      sv[i + 1] = -sn[i] * sv[i];
      sv[i] = cs[i] * sv[i];
      resid = sv[i + 1] / normb;
      if (log && log->p && log->resid && iter <= log->capacity) { log->p[iter - 1] = cur_p; log->resid[iter - 1] = std::fabs(resid); }
      if (std::fabs(resid) < so.residual) break;
    } while (i + 1 < R && i + 1 <= so.max_iters && std::fabs(resid) > so.residual);
    // solve the upper triangular system in place (:228-234)
    for (int j = i; j >= 0; --j) {
      sv[j] /= Hm(j, j);
      for (int k = j - 1; k >= 0; --k) sv[k] -= Hm(k, j) * sv[j];
    }
    // update the solution (:237-241; FGMRES :368-371)
    if (kind == FMMBEM_PC_INNER_PLAN && !so.flexible) {
      for (int j = 0; j <= i; ++j) {         // x += y_j M(V_j): the inner solve again, column by column, as the reference
        const double* z = nullptr;
        KRY_TRY(apply_pc(ws, M, ws->V + (int64_t)j * ld, ws->z, &z, s, depth));
        hipLaunchKernelGGL(mgs_axpy_kernel, dim3(grid), dim3(kThreads), 0, s, n, d_x, sv[j], z);
      }
    } else {
      for (int j = 0; j <= i; ++j) ws->h_pin[j] = sv[j];
      KRY_HIP(hipMemcpyAsync(ws->d_y, ws->h_pin, sizeof(double) * (size_t)(i + 1), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(update_x_kernel, dim3(grid), dim3(kThreads), 0, s, n, d_x, so.flexible ? ws->Z : ws->V, ld, i + 1, ws->d_y,
                         (!so.flexible && kind == FMMBEM_PC_DIAGONAL) ? M->reciprocals : (const double*)nullptr);
      KRY_HIP(hipStreamSynchronize(s));      // h_pin is reused by the next cycle
    }
  } while (std::fabs(resid) > so.residual && iter < so.max_iters);
  KRY_HIP(hipStreamSynchronize(s));
  if (log) { log->iterations = iter; log->residual = std::fabs(resid); }
  return hipGetLastError() == hipSuccess ? FMMBEM_OK : fail(FMMBEM_ERR_HIP, "fmmbem_gmres: a launch failed");
}

}  // namespace

extern "C" void fmmbem_solver_options_default(fmmbem_solver_options* o) {     // SolverOptions(), SolverOptions.hpp:23
  if (!o) return;
  o->residual = 1e-5; o->max_iters = 500; o->restart = 500; o->max_p = 16; o->p_min = 5; o->variable_p = 1;
  o->relax_type = FMMBEM_RELAX_BOURAS; o->order_rule = FMMBEM_ORDER_GMRES; o->flexible = 0; o->initial_p = 0;
}

extern "C" int fmmbem_gmres_device(fmmbem_plan* plan, const fmmbem_solver_options* opts, double* d_x, const double* d_b,
                                   const fmmbem_preconditioner* M, fmmbem_solver_log* log, void* stream) {
  if (!plan || !opts || !d_x || !d_b) return fail(FMMBEM_ERR_INVALID, "fmmbem_gmres_device: null argument");
  int device = 0, pm = 0; int64_t n = 0; SolverWs** slot = nullptr;
  KRY_TRY(fmmbem::plan_solver_info(plan, &device, &n, &pm, &slot));
  DevGuard guard(device);
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = solve(plan, *opts, d_x, d_b, M, log, static_cast<hipStream_t>(stream), 0);
  if (log) log->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

extern "C" int fmmbem_gmres(fmmbem_plan* plan, const fmmbem_solver_options* opts, double* x, const double* b,
                            const fmmbem_preconditioner* M, fmmbem_solver_log* log) {
  if (!plan || !opts || !x || !b) return fail(FMMBEM_ERR_INVALID, "fmmbem_gmres: null argument");
  int device = 0, pm = 0; int64_t n = 0; SolverWs** slot = nullptr;
  KRY_TRY(fmmbem::plan_solver_info(plan, &device, &n, &pm, &slot));
  DevGuard guard(device);
  const auto t0 = std::chrono::steady_clock::now();
  if (!*slot) { *slot = new SolverWs; (*slot)->device = device; }
  SolverWs* ws = *slot;
  const size_t bytes = sizeof(double) * (size_t)n;
  if (!ws->d_xb || ws->n != n) {             // x and b staged once per solve: the only PCIe traffic of length n
    KRY_TRY(grow(&ws->d_xb, (size_t)2 * ((n + 1) & ~int64_t(1))));
    if (ws->d_recip) { (void)hipFree(ws->d_recip); ws->d_recip = nullptr; }
  }
  double* d_x = ws->d_xb;
  double* d_b = ws->d_xb + ((n + 1) & ~int64_t(1));
  fmmbem_preconditioner Md;
  const fmmbem_preconditioner* Mp = M;
  if (M && M->kind == FMMBEM_PC_DIAGONAL) {
    if (!M->reciprocals) return fail(FMMBEM_ERR_INVALID, "fmmbem_gmres: diagonal preconditioner without reciprocals");
    if (!ws->d_recip) KRY_TRY(grow(&ws->d_recip, (size_t)n));
    KRY_HIP(hipMemcpy(ws->d_recip, M->reciprocals, bytes, hipMemcpyHostToDevice));
    Md = *M;
    Md.reciprocals = ws->d_recip;
    Mp = &Md;
  }
  KRY_HIP(hipMemcpy(d_x, x, bytes, hipMemcpyHostToDevice));
  KRY_HIP(hipMemcpy(d_b, b, bytes, hipMemcpyHostToDevice));
  // ensure_ws below may see n change and must not free what was just staged: d_xb / d_recip are not touched by it
  const int rc = solve(plan, *opts, d_x, d_b, Mp, log, nullptr, 0);
  if (rc != FMMBEM_OK) return rc;
  KRY_HIP(hipMemcpy(x, d_x, bytes, hipMemcpyDeviceToHost));
  if (log) log->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return FMMBEM_OK;
}
