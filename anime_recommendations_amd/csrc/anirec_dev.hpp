// Device-side helpers shared by the gfx950 kernels of libanirec.
// Wave = 64 lanes on CDNA4; a 128-float embedding row is one float4 per lane of a
// HALF-wave (32 lanes x 16 B = 512 B, one fully coalesced request).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/anirec.h"

#define ANIREC_HIP_CHECK(expr)                  \
  do {                                          \
    hipError_t _e = (expr);                     \
    if (_e != hipSuccess) return (int)_e;       \
  } while (0)

namespace anirec {

constexpr int kDim = ANIREC_DIM;
constexpr int kRowVec = kDim / 4;  // float4 per row = 32 = half-wave
constexpr float kL2nEps = 1e-12f;  // tf.nn.l2_normalize epsilon (Dot(normalize=True))
constexpr float kBnEps = 1e-3f;    // BatchNormalization epsilon
constexpr float kBnDecay = 0.01f;  // 1 - momentum(0.99)
constexpr float kOneMinusB1 = 0.1f;    // float32(1 - 0.9)
constexpr float kOneMinusB2 = 0.001f;  // float32(1 - 0.999)
constexpr float kAdamEps = 1e-7f;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Sum over the 32 lanes of a half-wave (lanes [0,32) and [32,64) reduce separately);
// every lane of the half receives the total.  Fixed butterfly order -> deterministic.
__device__ __forceinline__ float halfwave_sum(float v) {
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum of N floats per thread (blockDim.x multiple of 64, <= 1024).
// `scratch` needs N*16 floats.  Result broadcast to every thread.
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float *scratch) {
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();  // scratch may still be read from a previous call
  if (lane_id() == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) scratch[i * 16 + w] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float s = 0.f;
    for (int k = 0; k < nw; ++k) s += scratch[i * 16 + k];  // fixed order
    v[i] = s;
  }
}

// IEEE, never contracted into FMA (HIP's __fmul_rn/__fadd_rn are plain operators and hipcc
// defaults to -ffp-contract=fast, so contraction is switched off per block): the fused Adam
// must be bit-identical to the NumPy fp32 oracle given the same gradient.
__device__ __forceinline__ void adam_elem(float &w, float &m, float &v, float g, float alpha) {
#pragma clang fp contract(off)
  m = m + (g - m) * kOneMinusB1;
  v = v + (g * g - v) * kOneMinusB2;
  w = w - (m * alpha) / (sqrtf(v) + kAdamEps);
}

// g_total = (g_sparse - s*w) + two_l2*w with every product and sum rounded once.
__device__ __forceinline__ float grad_total(float gs, float s, float w, float two_l2) {
#pragma clang fp contract(off)
  return (gs - s * w) + two_l2 * w;
}

__device__ __forceinline__ float sigmoidf_stable(float y) {
  float e = __expf(-fabsf(y));
  // expf via the fast path is within 2 ulp; the 1e-5 tolerance on ratings absorbs it
  return y >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

// rating of a pair from its cosine through the folded BN-inference head: ONE definition shared by the
// exact path (k_scores epilogue) and the MFMA path's re-rank, so both produce the same fp32 value
__device__ __forceinline__ float rating_from_cosine(float c, float hs, float hb) {
  return sigmoidf_stable(__fmaf_rn(c, hs, hb));
}

// sigmoid(gamma*(w*c+b-mu)/sqrt(var+eps)+beta) = sigmoid(c*hs + hb); folded in fp32 exactly as
// tf.nn.batch_normalization does: inv = rsqrt(var+eps)*gamma; y = z*inv + (beta - mu*inv)
static inline void head_affine_f32(const anirec_head *h, float *hs, float *hb) {
  const float inv = (1.0f / sqrtf(h->mov_var + kBnEps)) * h->gamma;
  *hs = h->w * inv;
  *hb = h->b * inv + (h->beta - h->mov_mean * inv);
}

}  // namespace anirec
