// membw.hip -- read-bandwidth probes for the near_spmv access pattern (tuning aid, not part of the product).
//   A: every thread streams 16-B vectors grid-stride over the whole buffer (the ideal linear read)
//   B: persistent 256-thread workgroups take contiguous chunks of C bytes (a "leaf block"), each wavefront reads
//      ROWS of `rowb` bytes, kRows rows x kVecs 16-B vectors per lane in flight, exactly like near_spmv
// build: hipcc --offload-arch=gfx950 -O3 membw.hip -o membw ; run: ./membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <utility>
typedef double dvec2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void linear_read(const dvec2* __restrict__ a, size_t nvec, double* out, int nt) {
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    dvec2 v = nt ? __builtin_nontemporal_load(&a[i]) : a[i];
    acc += v.x + v.y;
  }
  if (acc == 1.2345e-300) out[0] = acc;
}

template <int kRows, int kVecs, int kWaves>
__global__ __launch_bounds__(kWaves * 64) void chunk_read(const dvec2* __restrict__ a, size_t chunk_vec, int nchunks, int row_vec, double* out, const int* __restrict__ order = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0;
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const dvec2* blk = a + (size_t)(order ? order[ch] : ch) * chunk_vec;
    const int nrows = (int)(chunk_vec / row_vec);
    for (int r = wave; r < nrows; r += kRows * kWaves) {
      const dvec2* row[kRows];
#pragma unroll
      for (int i = 0; i < kRows; ++i) { const int ri = r + i * kWaves; row[i] = blk + (size_t)(ri < nrows ? ri : r) * row_vec; }
      for (int c = lane; c < row_vec; c += kVecs * 64) {
        dvec2 v[kRows][kVecs];
#pragma unroll
        for (int u = 0; u < kVecs; ++u) {
          const int cc = c + u * 64;
#pragma unroll
          for (int i = 0; i < kRows; ++i) v[i][u] = cc < row_vec ? __builtin_nontemporal_load(&row[i][cc]) : dvec2{0, 0};
        }
#pragma unroll
        for (int u = 0; u < kVecs; ++u)
#pragma unroll
          for (int i = 0; i < kRows; ++i) acc += v[i][u].x + v[i][u].y;
      }
    }
    __syncthreads();
  }
  if (acc == 1.2345e-300) out[0] = acc;
}

// D: B + the x slice read from a packed global array (16 B per matrix vector, L1/L2 hits after the first row) and
//    the FMAs; no LDS, no barrier: wavefronts run free.  This is the shape near_spmv takes with a pre-packed x.
template <int kRows, int kVecs, int kWaves>
__global__ __launch_bounds__(kWaves * 64) void chunk_spmv(const dvec2* __restrict__ a, const dvec2* __restrict__ xp, size_t chunk_vec, int nchunks,
                                                          int row_vec, double* __restrict__ y, const int* __restrict__ order) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int chi = blockIdx.x; chi < nchunks; chi += gridDim.x) {
    const int ch = order ? order[chi] : chi;
    const dvec2* blk = a + (size_t)ch * chunk_vec;
    const dvec2* x = xp + (size_t)ch * row_vec;
    const int nrows = (int)(chunk_vec / row_vec);
    for (int r = wave; r < nrows; r += kRows * kWaves) {
      const dvec2* row[kRows];
      double acc[kRows];
#pragma unroll
      for (int i = 0; i < kRows; ++i) { const int ri = r + i * kWaves; row[i] = blk + (size_t)(ri < nrows ? ri : r) * row_vec; acc[i] = 0; }
      for (int c = lane; c < row_vec; c += kVecs * 64) {
        dvec2 v[kRows][kVecs], xv[kVecs];
#pragma unroll
        for (int u = 0; u < kVecs; ++u) {
          const int cc = c + u * 64;
#pragma unroll
          for (int i = 0; i < kRows; ++i) v[i][u] = cc < row_vec ? __builtin_nontemporal_load(&row[i][cc]) : dvec2{0, 0};
          xv[u] = cc < row_vec ? x[cc] : dvec2{0, 0};
        }
#pragma unroll
        for (int u = 0; u < kVecs; ++u)
#pragma unroll
          for (int i = 0; i < kRows; ++i) acc[i] = fma(v[i][u].x, xv[u].x, fma(v[i][u].y, xv[u].y, acc[i]));
      }
#pragma unroll
      for (int i = 0; i < kRows; ++i) {
        double t = acc[i];
        for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o);
        const int ri = r + i * kWaves;
        if (lane == 0 && ri < nrows) y[(size_t)ch * 64 + ri] = t;
      }
    }
  }
}

__global__ void fill_random(double* a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ULL; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 29;
    a[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}

int main(int argc, char** argv) {
  const bool randomize = argc > 1;
  const size_t bytes = (size_t)4 << 30, nvec = bytes / 16;
  dvec2* a; double* out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&out, 8));
  CK(hipMemset(a, 0, bytes));
  if (randomize) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (double*)a, bytes / 8); CK(hipDeviceSynchronize()); printf("buffer filled with random doubles\n"); }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char* name) {
    for (int i = 0; i < 2; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-58s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout);
  };
  char name[128];
  for (int nt = 0; nt < 2; ++nt)
    for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 64}) {
      snprintf(name, sizeof name, "A linear nt=%d grid=%d x256", nt, g);
      time([&] { hipLaunchKernelGGL(linear_read, dim3(g), dim3(256), 0, 0, a, nvec, out, nt); }, name);
    }
  for (int rowb : {3904}) {
    for (size_t chunk : {(size_t)32 << 10, (size_t)76 << 10, (size_t)256 << 10, (size_t)1 << 20, (size_t)4 << 20}) {
      const int row_vec = rowb / 16;
      const size_t chunk_vec = chunk / rowb * row_vec;      // whole rows
      const int nchunks = (int)(nvec / chunk_vec);
      for (int per_cu : {5, 8}) {
        snprintf(name, sizeof name, "B rows %d B, chunk %zu KB, <2,4,4w> %d/CU", rowb, chunk >> 10, per_cu);
        time([&] { hipLaunchKernelGGL((chunk_read<2, 4, 4>), dim3(256 * per_cu), dim3(256), 0, 0, a, chunk_vec, nchunks, row_vec, out); }, name);
      }
      snprintf(name, sizeof name, "B rows %d B, chunk %zu KB, <1,4,16w> 2/CU", rowb, chunk >> 10);
      time([&] { hipLaunchKernelGGL((chunk_read<1, 4, 16>), dim3(256 * 2), dim3(1024), 0, 0, a, chunk_vec, nchunks, row_vec, out); }, name);
    }
  }
  {
  for (int rowb : {3904, 3920}) {
    const int row_vec = rowb / 16;
    printf("-- D with rows of %d bytes\n", rowb);
    const size_t chunk = (size_t)76 << 10, chunk_vec = chunk / rowb * row_vec;
    const int nchunks = (int)(nvec / chunk_vec);
    dvec2* xp; double* y;
    CK(hipMalloc(&xp, (size_t)nchunks * rowb)); CK(hipMemset(xp, 0, (size_t)nchunks * rowb));
    CK(hipMalloc(&y, (size_t)nchunks * 64 * 8));
    std::vector<int> ord(nchunks);
    for (int i = 0; i < nchunks; ++i) ord[i] = i;
    unsigned long long sd = 999;
    for (int i = nchunks - 1; i > 0; --i) { sd = sd * 6364136223846793005ULL + 1442695040888963407ULL; int j = (int)((sd >> 33) % (unsigned)(i + 1)); std::swap(ord[i], ord[j]); }
    int* dord; CK(hipMalloc(&dord, nchunks * sizeof(int))); CK(hipMemcpy(dord, ord.data(), nchunks * sizeof(int), hipMemcpyHostToDevice));
    for (int per_cu : {4, 5, 6, 8}) {
      snprintf(name, sizeof name, "D spmv packed-x rows 3904 B chunk 76 KB <2,4,4w> %d/CU in order", per_cu);
      time([&] { hipLaunchKernelGGL((chunk_spmv<2, 4, 4>), dim3(256 * per_cu), dim3(256), 0, 0, a, xp, chunk_vec, nchunks, row_vec, y, (const int*)nullptr); }, name);
      snprintf(name, sizeof name, "D spmv packed-x rows 3904 B chunk 76 KB <2,4,4w> %d/CU RANDOM", per_cu);
      time([&] { hipLaunchKernelGGL((chunk_spmv<2, 4, 4>), dim3(256 * per_cu), dim3(256), 0, 0, a, xp, chunk_vec, nchunks, row_vec, y, dord); }, name);
    }
    for (int per_cu : {16, 20, 24, 32}) {
      snprintf(name, sizeof name, "D spmv packed-x <2,4,1w> wave-per-chunk %d waves/CU RANDOM", per_cu);
      time([&] { hipLaunchKernelGGL((chunk_spmv<2, 4, 1>), dim3(256 * per_cu), dim3(64), 0, 0, a, xp, chunk_vec, nchunks, row_vec, y, dord); }, name);
    }
    snprintf(name, sizeof name, "D spmv packed-x <2,2,4w> 8/CU RANDOM");
    time([&] { hipLaunchKernelGGL((chunk_spmv<2, 2, 4>), dim3(256 * 8), dim3(256), 0, 0, a, xp, chunk_vec, nchunks, row_vec, y, dord); }, name);
    snprintf(name, sizeof name, "D spmv packed-x <4,2,4w> 5/CU RANDOM");
    time([&] { hipLaunchKernelGGL((chunk_spmv<4, 2, 4>), dim3(256 * 5), dim3(256), 0, 0, a, xp, chunk_vec, nchunks, row_vec, y, dord); }, name);
    snprintf(name, sizeof name, "D spmv packed-x <4,4,4w> 4/CU RANDOM");
    time([&] { hipLaunchKernelGGL((chunk_spmv<4, 4, 4>), dim3(256 * 4), dim3(256), 0, 0, a, xp, chunk_vec, nchunks, row_vec, y, dord); }, name);
    hipFree(xp); hipFree(y); hipFree(dord);
  }
  }
  // C: the same chunks in RANDOM order (what a size-sorted work queue does to the address stream)
  for (size_t chunk : {(size_t)8 << 10, (size_t)32 << 10, (size_t)76 << 10, (size_t)256 << 10, (size_t)1 << 20}) {
    const int rowb = 4096 > chunk ? (int)chunk : 4096, row_vec = rowb / 16;
    const size_t chunk_vec = chunk / rowb * row_vec;
    const int nchunks = (int)(nvec / chunk_vec);
    std::vector<int> ord(nchunks);
    for (int i = 0; i < nchunks; ++i) ord[i] = i;
    unsigned long long sd = 12345;
    for (int i = nchunks - 1; i > 0; --i) { sd = sd * 6364136223846793005ULL + 1442695040888963407ULL; int j = (int)((sd >> 33) % (unsigned)(i + 1)); std::swap(ord[i], ord[j]); }
    int* dord; CK(hipMalloc(&dord, nchunks * sizeof(int))); CK(hipMemcpy(dord, ord.data(), nchunks * sizeof(int), hipMemcpyHostToDevice));
    for (int per_cu : {5, 8}) {
      snprintf(name, sizeof name, "C RANDOM order, chunk %zu KB, <2,4,4w> %d/CU", chunk >> 10, per_cu);
      time([&] { hipLaunchKernelGGL((chunk_read<2, 4, 4>), dim3(256 * per_cu), dim3(256), 0, 0, a, chunk_vec, nchunks, row_vec, out, dord); }, name);
      snprintf(name, sizeof name, "C in order,     chunk %zu KB, <2,4,4w> %d/CU", chunk >> 10, per_cu);
      time([&] { hipLaunchKernelGGL((chunk_read<2, 4, 4>), dim3(256 * per_cu), dim3(256), 0, 0, a, chunk_vec, nchunks, row_vec, out, (const int*)nullptr); }, name);
    }
    hipFree(dord);
  }
  return 0;
}
