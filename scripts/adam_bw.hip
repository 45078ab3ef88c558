// Microbenchmark (not product code): what limits the dense Adam stream on MI355X?
// hipcc -O3 --offload-arch=gfx950 scripts/adam_bw.hip -o /tmp/adam_bw && /tmp/adam_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

__device__ __forceinline__ void adam_elem(float &w, float &m, float &v, float g, float alpha) {
#pragma clang fp contract(off)
  m = m + (g - m) * 0.1f;
  v = v + (g * g - v) * 0.001f;
  w = w - (m * alpha) / (sqrtf(v) + 1e-7f);
}
__device__ __forceinline__ void adam4(float4 &w, float4 &m, float4 &v, float tl, float alpha) {
  adam_elem(w.x, m.x, v.x, tl * w.x, alpha); adam_elem(w.y, m.y, v.y, tl * w.y, alpha);
  adam_elem(w.z, m.z, v.z, tl * w.z, alpha); adam_elem(w.w, m.w, v.w, tl * w.w, alpha);
}
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload(const float4 *p) { f4v x = __builtin_nontemporal_load((const f4v *)p); return make_float4(x.x, x.y, x.z, x.w); }
__device__ __forceinline__ void ntstore(float4 v, float4 *p) { f4v x = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(x, (f4v *)p); }
// V1: flat grid-stride float4, U loads in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_flat(float4 *W, float4 *M, float4 *V, size_t n4, float alpha) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 w[U], m[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) { w[u] = ntload(&W[i + u * stride]); m[u] = ntload(&M[i + u * stride]); v[u] = ntload(&V[i + u * stride]); }
      else { w[u] = W[i + u * stride]; m[u] = M[i + u * stride]; v[u] = V[i + u * stride]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      adam4(w[u], m[u], v[u], 2e-4f, alpha);
      if (NT) { ntstore(w[u], &W[i + u * stride]); ntstore(m[u], &M[i + u * stride]); ntstore(v[u], &V[i + u * stride]); }
      else { W[i + u * stride] = w[u]; M[i + u * stride] = m[u]; V[i + u * stride] = v[u]; }
    }
  }
  for (; i < n4; i += stride) { float4 w = W[i], m = M[i], v = V[i]; adam4(w, m, v, 2e-4f, alpha); W[i] = w; M[i] = m; V[i] = v; }
}
// V2: block-contiguous ranges
template <int U>
__global__ __launch_bounds__(256) void k_blockcontig(float4 *W, float4 *M, float4 *V, size_t n4, float alpha) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = min(n4, lo + per);
  size_t i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += U * 256) {
    float4 w[U], m[U], v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { w[u] = W[i + u * 256]; m[u] = M[i + u * 256]; v[u] = V[i + u * 256]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { adam4(w[u], m[u], v[u], 2e-4f, alpha); W[i + u * 256] = w[u]; M[i + u * 256] = m[u]; V[i + u * 256] = v[u]; }
  }
  for (; i < hi; i += 256) { float4 w = W[i], m = M[i], v = V[i]; adam4(w, m, v, 2e-4f, alpha); W[i] = w; M[i] = m; V[i] = v; }
}
// copy with the same bytes: read 3 write 3, no math
__global__ __launch_bounds__(256) void k_copy3(float4 *W, float4 *M, float4 *V, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 w = W[i], m = M[i], v = V[i];
    w.x += 1.f; m.x += 1.f; v.x += 1.f;
    W[i] = w; M[i] = m; V[i] = v;
  }
}
// fast math variant (is ALU the limit?)
__global__ __launch_bounds__(256) void k_fast(float4 *W, float4 *M, float4 *V, size_t n4, float alpha) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 w = W[i], m = M[i], v = V[i];
    float *pw = &w.x, *pm = &m.x, *pv = &v.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) { float g = 2e-4f * pw[k]; pm[k] += (g - pm[k]) * 0.1f; pv[k] += (g * g - pv[k]) * 0.001f; pw[k] -= pm[k] * alpha * __frcp_rn(__fsqrt_rn(pv[k]) + 1e-7f); }
    W[i] = w; M[i] = m; V[i] = v;
  }
}
int main() {
  const size_t rows = 368000, n = rows * 128, n4 = n / 4;
  float *W, *M, *V;
  CK(hipMalloc(&W, n * 4)); CK(hipMalloc(&M, n * 4)); CK(hipMalloc(&V, n * 4));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = 0.05f * ((float)((i * 2654435761u) & 0xffff) / 65536.f - 0.5f);
  CK(hipMemcpy(W, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemset(M, 0, n * 4)); CK(hipMemset(V, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    (void)hipDeviceSynchronize();
    float best = 1e9, tot = 0;
    for (int r = 0; r < 10; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; tot += ms; }
    printf("%-28s avg %.1f us  best %.1f us  moved %.2f TB/s (24B/elem)\n", name, tot * 100, best * 1000, 24.0 * n / (best * 1e-3) / 1e12);
  };
  float4 *W4 = (float4 *)W, *M4 = (float4 *)M, *V4 = (float4 *)V;
  const float alpha = 3.1e-6f;
  for (int g : {1024, 2048, 4096, 8192}) {
    char nm[64];
    snprintf(nm, 64, "copy3 grid=%d", g); run(nm, [&] { hipLaunchKernelGGL(k_copy3, dim3(g), dim3(256), 0, 0, W4, M4, V4, n4); });
    snprintf(nm, 64, "flat U1 grid=%d", g); run(nm, [&] { hipLaunchKernelGGL((k_flat<1, false>), dim3(g), dim3(256), 0, 0, W4, M4, V4, n4, alpha); });
    snprintf(nm, 64, "flat U2 grid=%d", g); run(nm, [&] { hipLaunchKernelGGL((k_flat<2, false>), dim3(g), dim3(256), 0, 0, W4, M4, V4, n4, alpha); });
    snprintf(nm, 64, "flat U4 grid=%d", g); run(nm, [&] { hipLaunchKernelGGL((k_flat<4, false>), dim3(g), dim3(256), 0, 0, W4, M4, V4, n4, alpha); });
    snprintf(nm, 64, "flat U2 NT grid=%d", g); run(nm, [&] { hipLaunchKernelGGL((k_flat<2, true>), dim3(g), dim3(256), 0, 0, W4, M4, V4, n4, alpha); });
    snprintf(nm, 64, "blockcontig U2 grid=%d", g); run(nm, [&] { hipLaunchKernelGGL((k_blockcontig<2>), dim3(g), dim3(256), 0, 0, W4, M4, V4, n4, alpha); });
    snprintf(nm, 64, "fastmath grid=%d", g); run(nm, [&] { hipLaunchKernelGGL(k_fast, dim3(g), dim3(256), 0, 0, W4, M4, V4, n4, alpha); });
  }
  // huge grid: one float4 per thread
  { int g = (int)((n4 + 255) / 256); run("flat U1 grid=n4/256", [&] { hipLaunchKernelGGL((k_flat<1, false>), dim3(g), dim3(256), 0, 0, W4, M4, V4, n4, alpha); });
    run("copy3 grid=n4/256", [&] { hipLaunchKernelGGL(k_copy3, dim3(g), dim3(256), 0, 0, W4, M4, V4, n4); }); }
  return 0;
}
