// Does LDS-DMA (global_load_lds_dword, base in M0) reach LDS offsets above 64 KiB on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint32_t *src, uint32_t *out, uint32_t lds_off) {
  extern __shared__ uint32_t sm[];
  for (int i = threadIdx.x; i < 36 * 1024; i += 64) sm[i] = 0xDEADBEEFu;
  __syncthreads();
  const uint32_t voff = threadIdx.x * 4;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(src), "s"(lds_off) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = sm[lds_off / 4 + threadIdx.x];
  out[64 + threadIdx.x] = sm[(lds_off & 0xFFFF) / 4 + threadIdx.x];  // where a 16-bit M0 would have put it
}
int main() {
  uint32_t *src, *out;
  hipMalloc(&src, 256); hipMalloc(&out, 512);
  std::vector<uint32_t> h(64);
  for (int i = 0; i < 64; ++i) h[i] = 1000 + i;
  hipMemcpy(src, h.data(), 256, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  for (uint32_t off : {4096u, 72u * 1024u, 130u * 1024u}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 144 * 1024, 0, src, out, off);
    std::vector<uint32_t> r(128);
    hipMemcpy(r.data(), out, 512, hipMemcpyDeviceToHost);
    int ok = 0, wrapped = 0;
    for (int i = 0; i < 64; ++i) { ok += r[i] == 1000u + i; wrapped += r[64 + i] == 1000u + i; }
    printf("lds_off=%u: %d/64 at the requested offset, %d/64 at (offset & 0xFFFF)\n", off, ok, wrapped);
  }
  return 0;
}
