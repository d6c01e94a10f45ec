// krylov.hip -- the modified Gram-Schmidt column of the callers above the matvec (examples/BEM/GMRES.hpp:203-212,
// GMRES_Stokes.hpp the same loop on Vec<3,double> values): for k = 0..i:  h_k = <w, V_k>;  w -= h_k V_k;  then
// h_{i+1} = |w|,  V_{i+1} = w / h_{i+1}.  The reference runs 2(i + 1) + 2 passes over the vectors; so did solver.py with
// one torch call each, and at N = 1M the 27 iterations of the config-5 solve spent 5-6 ms of 37 there, most of it per-call
// overhead.  Here a column is ONE call and i + 3 launches: launch k subtracts h_{k-1} V_{k-1} and, on the updated w,
// accumulates <w, V_k> in the same sweep (the last launch accumulates <w, w>), then one scales and one collects the column.
// A dot product is 1 024 per-workgroup sums that whoever needs the total adds in index order: same bits every run.
// Same operations in the same order as the reference's loop; only the association inside a dot product differs.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/fmmbem.h"

namespace fmmbem {
int fail(int code, const std::string& msg);        // plan.hip: records the message for fmmbem_last_error
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
