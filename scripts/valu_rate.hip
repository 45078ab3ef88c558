// What the vector ALU of gfx950 issues per cycle for the instruction kinds of the lazy Adam replay:
//   hipcc -O3 --offload-arch=gfx950 -o scripts/valu_rate scripts/valu_rate.hip && scripts/valu_rate
// One workgroup of 64 x W threads per CU-slot (W waves on each SIMD of every CU), every wave runs N back-to-back
// instructions of one kind on `kChains` independent register chains; cycles from s_memtime around the loop.
// Prints wave-instructions per cycle per SIMD (2-cycle issue = 0.5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int kKind, int kChains>
__global__ __launch_bounds__(1024) void k_rate(float *out, unsigned long long *cyc, int iters, float seed) {
  float x[kChains];
  f32x2 p[kChains];
#pragma unroll
  for (int c = 0; c < kChains; ++c) {
    x[c] = seed + (float)(threadIdx.x + c) * 1e-3f;
    p[c] = {x[c], x[c] + 0.5f};
  }
  const float a = 1.0000001f, b = 1e-9f;
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < kChains; ++c) {
        if (kKind == 0) x[c] = __builtin_fmaf(x[c], a, b);                                  // v_fma_f32
        if (kKind == 1) p[c] = __builtin_elementwise_fma(p[c], (f32x2){a, a}, (f32x2){b, b});  // v_pk_fma_f32
        if (kKind == 2) x[c] = __builtin_amdgcn_sqrtf(x[c]) + 0.0f * x[c];                  // v_sqrt_f32 (+ 1 fma)
        if (kKind == 3) x[c] = __builtin_amdgcn_rcpf(x[c]);                                 // v_rcp_f32
        if (kKind == 4) {                                                                   // v_med3_i32
          int r;
          asm volatile("v_med3_i32 %0, %1, 0, 1" : "=v"(r) : "v"(__float_as_int(x[c])));
          x[c] = __int_as_float(r + 0x3f800000);
        }
        if (kKind == 5) x[c] = fminf(fminf(x[c], a), b + x[c]);                             // v_min3-ish
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kChains; ++c) s += x[c] + p[c].x + p[c].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int kKind, int kChains>
static void run(const char *name, int waves_per_simd, float *out, unsigned long long *cyc) {
  const int threads = 256 * waves_per_simd, blocks = 256, iters = 2048;
  hipLaunchKernelGGL((k_rate<kKind, kChains>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.5f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_rate<kKind, kChains>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.5f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const int nw = blocks * threads / 64;
  std::vector<unsigned long long> h(nw);
  hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (auto v : h) mean += (double)v;
  mean /= nw;
  const double insts = (double)iters * 8 * kChains;      // per wave
  printf("%-14s chains %d waves/SIMD %d : %.3f wave-inst/cycle/SIMD (%.2f cycles per inst per wave), kernel %.3f ms\n", name,
         kChains, waves_per_simd, insts * waves_per_simd / mean, mean / insts, ms);
}

int main() {
  float *out;
  unsigned long long *cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
  for (int w = 1; w <= 4; ++w) {
    run<0, 4>("v_fma_f32", w, out, cyc);
    run<1, 4>("v_pk_fma_f32", w, out, cyc);
    run<2, 4>("v_sqrt+fma", w, out, cyc);
    run<3, 4>("v_rcp_f32", w, out, cyc);
    run<4, 4>("v_med3_i32+add", w, out, cyc);
    run<5, 4>("v_min3", w, out, cyc);
  }
  run<0, 1>("v_fma_f32 dep", 1, out, cyc);
  run<1, 1>("v_pk_fma dep", 1, out, cyc);
  run<0, 2>("v_fma_f32 2ch", 1, out, cyc);
  run<1, 2>("v_pk_fma 2ch", 1, out, cyc);
  return 0;
}
