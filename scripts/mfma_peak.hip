// Practical MFMA ceiling of the box: register-only v_mfma_f32_32x32x16_f16 loops (no memory traffic),
// 2 or 1 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *out;
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512, 1024}) {
    const int iters = 20000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
    printf("blocks=%d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, fl / ms / 1e9);
  }
  return 0;
}
