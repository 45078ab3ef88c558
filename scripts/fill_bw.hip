// Microbenchmark (not product code): what does MI355X sustain for a write-only stream?  The predict grid writes
// 7.2 GB and reads next to nothing; this is its ceiling.
// hipcc -O3 --offload-arch=gfx950 scripts/fill_bw.hip -o scripts/fill_bw && scripts/fill_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
typedef float f4v __attribute__((ext_vector_type(4)));
// grid-stride, U stores per lane and iteration, plain or non-temporal
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_fill(float4 *p, size_t n4, float x) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const f4v v = {x, x + 1.f, x + 2.f, x + 3.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < n4) {
        if (NT) __builtin_nontemporal_store(v, (f4v *)&p[i + u * stride]);
        else *((f4v *)&p[i + u * stride]) = v;
      }
  }
}
// one workgroup per contiguous range (the shape of a tiled output: a workgroup streams its own rows)
template <bool NT>
__global__ __launch_bounds__(256) void k_fill_rows(float4 *p, size_t n4, float x) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  const f4v v = {x, x + 1.f, x + 2.f, x + 3.f};
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    if (NT) __builtin_nontemporal_store(v, (f4v *)&p[i]);
    else *((f4v *)&p[i]) = v;
  }
}
// the predict grid's store shape: a wave owns 32 rows of a row-major [rows][cols] fp32 matrix and walks along them;
// one store instruction writes R rows x S bytes (R * S = 1 KiB), a step covers SEG bytes of each of the wave's rows.
// Workgroup = 4 waves = 128 rows; grid.y cuts the columns into parts.
template <int S, int SEG, bool NT = true>
__global__ __launch_bounds__(256) void k_fill_tiles(float *out, int rows, int cols, int cols_per_part, float x) {
  constexpr int R = 1024 / S;                  // rows per store instruction
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r_in = lane / (S / 16), piece = lane % (S / 16);
  const int row0 = blockIdx.x * 128 + w * 32;
  const int c_lo = blockIdx.y * cols_per_part, c_hi = min(cols, c_lo + cols_per_part);
  const f4v v = {x, x + 1.f, x + 2.f, x + 3.f};
  for (int c = c_lo; c < c_hi; c += SEG / 4) {          // a step: SEG bytes of every row
#pragma unroll
    for (int sub = 0; sub < SEG / S; ++sub)             // SEG / S instructions per group of R rows
#pragma unroll 4
      for (int g = 0; g < 32 / R; ++g) {
        const int row = row0 + g * R + r_in, col = c + sub * (S / 4) + piece * 4;
        if (row < rows && col < c_hi) {
          if (NT) __builtin_nontemporal_store(v, (f4v *)&out[(size_t)row * cols + col]);
          else *((f4v *)&out[(size_t)row * cols + col]) = v;
        }
      }
  }
}
template <typename F>
static int timeit(const char *name, size_t bytes, F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  launch(); CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best; sum += ms;
  }
  printf("%-44s mean %.3f ms  best %.3f ms = %.2f TB/s\n", name, sum / 5, best, bytes / best / 1e9);
  return 0;
}
int main() {
  const size_t bytes = (size_t)100000 * 18000 * 4;  // the predict grid
  const size_t n4 = bytes / 16;
  float4 *p;
  CK(hipMalloc(&p, bytes));
  timeit("hipMemsetAsync", bytes, [&] { (void)hipMemsetAsync(p, 1, bytes, 0); });
  timeit("grid-stride x4 plain, 16384 wg", bytes, [&] { hipLaunchKernelGGL((k_fill<4, false>), dim3(16384), dim3(256), 0, 0, p, n4, 1.f); });
  timeit("grid-stride x4 nt, 16384 wg", bytes, [&] { hipLaunchKernelGGL((k_fill<4, true>), dim3(16384), dim3(256), 0, 0, p, n4, 1.f); });
  timeit("grid-stride x1 nt, 65536 wg", bytes, [&] { hipLaunchKernelGGL((k_fill<1, true>), dim3(65536), dim3(256), 0, 0, p, n4, 1.f); });
  timeit("grid-stride x8 nt, 2048 wg", bytes, [&] { hipLaunchKernelGGL((k_fill<8, true>), dim3(2048), dim3(256), 0, 0, p, n4, 1.f); });
  timeit("contiguous ranges plain, 8192 wg", bytes, [&] { hipLaunchKernelGGL((k_fill_rows<false>), dim3(8192), dim3(256), 0, 0, p, n4, 1.f); });
  timeit("contiguous ranges nt, 8192 wg", bytes, [&] { hipLaunchKernelGGL((k_fill_rows<true>), dim3(8192), dim3(256), 0, 0, p, n4, 1.f); });
  timeit("contiguous ranges nt, 1024 wg", bytes, [&] { hipLaunchKernelGGL((k_fill_rows<true>), dim3(1024), dim3(256), 0, 0, p, n4, 1.f); });
  float *o = reinterpret_cast<float *>(p);
  const int rows = 100000, cols = 18000, parts = 5, cpp = 3648;  // 57 tiles of 64 columns per part, like the kernel
  const dim3 g((rows + 127) / 128, parts);
  timeit("tiles: 4 rows x 256 B, 256 B per step", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<256, 256>), g, dim3(256), 0, 0, o, rows, cols, cpp, 1.f); });
  timeit("tiles: 4 rows x 256 B, 512 B per step", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<256, 512>), g, dim3(256), 0, 0, o, rows, cols, cpp, 1.f); });
  timeit("tiles: 2 rows x 512 B, 512 B per step", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<512, 512>), g, dim3(256), 0, 0, o, rows, cols, cpp, 1.f); });
  timeit("tiles: 2 rows x 512 B, 1 KiB per step", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<512, 1024>), g, dim3(256), 0, 0, o, rows, cols, cpp, 1.f); });
  timeit("tiles: 1 row x 1 KiB, 1 KiB per step", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<1024, 1024>), g, dim3(256), 0, 0, o, rows, cols, cpp, 1.f); });
  timeit("tiles: 1 row x 1 KiB, 2 KiB per step", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<1024, 2048>), g, dim3(256), 0, 0, o, rows, cols, cpp, 1.f); });
  timeit("tiles plain: 4 rows x 256 B, 256 B per step", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<256, 256, false>), g, dim3(256), 0, 0, o, rows, cols, cpp, 1.f); });
  timeit("tiles plain: 2 rows x 512 B, 512 B per step", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<512, 512, false>), g, dim3(256), 0, 0, o, rows, cols, cpp, 1.f); });
  timeit("tiles plain: 1 row x 1 KiB, 1 KiB per step", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<1024, 1024, false>), g, dim3(256), 0, 0, o, rows, cols, cpp, 1.f); });
  {
    const dim3 g1((rows + 127) / 128, 1);
    timeit("tiles nt: 4 rows x 256 B, one part (whole rows)", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<256, 256>), g1, dim3(256), 0, 0, o, rows, cols, cols, 1.f); });
    const dim3 g20((rows + 127) / 128, 20);
    timeit("tiles nt: 4 rows x 256 B, 20 parts", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<256, 256>), g20, dim3(256), 0, 0, o, rows, cols, 960, 1.f); });
  }
  timeit("tiles: 8 rows x 128 B, 128 B per step", bytes, [&] { hipLaunchKernelGGL((k_fill_tiles<128, 128>), g, dim3(256), 0, 0, o, rows, cols, cpp, 1.f); });
  return 0;
}
