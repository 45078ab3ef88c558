// Batched rating prediction on the gfx950 matrix cores.
//
// Replaces  model.predict([user x n, anime_idx])  (model_recs/model_recs.py:394) at BASELINE scale
// (100 k query users x 18 k anime = 1.8 G predictions): out[j][a] = sigmoid(hs * <u^_j, a^_a> + hb),
// the BN-inference head folded into (hs, hb).
//
// Precision: ratings must agree with the fp32 reference to 1e-5, i.e. the cosine to ~2e-5.  A single
// f16/bf16 MFMA is 1e-3 off, so every l2-normalised row x (scaled by 2^8 to keep the low part out of
// the fp16 subnormals) is split x = hi + lo, hi = fp16(x), lo = fp16(x - hi), and the product is
// accumulated as hi*hi + hi*lo + lo*hi in the fp32 MFMA accumulator: relative error ~2^-21 per
// term, 3 x the f16 MFMA work (0.46 TFLOP -> 1.4 TFLOP at 100 k x 18 k), still cheaper than
// writing the 7.2 GB fp32 grid (the kernel is HBM-write-bound).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>

#include "anirec_dev.hpp"

namespace anirec {

constexpr int kPM = 128;  // users per workgroup (4 waves x 32 rows)
constexpr int kPN = 64;   // anime per tile
constexpr float kSplitScale = 256.0f;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// l2_normalize (tf epsilon 1e-12) a gathered row, scale by 2^8 and split into fp16 hi / lo planes.
// out layout: [rows][2][128] halves (hi plane then lo plane of the row: 512 B per row)
__global__ __launch_bounds__(256) void k_norm_split(const float *W, const int32_t *rows, int n,
                                                    _Float16 *out) {
  const int l = threadIdx.x & 31;
  const int nhw = gridDim.x * 8;
  for (int r = blockIdx.x * 8 + (threadIdx.x >> 5); r < n; r += nhw) {
    const int src = rows ? rows[r] : r;
    const float4 x = reinterpret_cast<const float4 *>(W)[(size_t)src * kRowVec + l];
    float ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    ss = halfwave_sum(ss);
    const float rinv = kSplitScale / sqrtf(fmaxf(ss, kL2nEps));
    const float v[4] = {x.x * rinv, x.y * rinv, x.z * rinv, x.w * rinv};
    _Float16 hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hi[i] = (_Float16)v[i];
      lo[i] = (_Float16)(v[i] - (float)hi[i]);
    }
    uint2 *o = reinterpret_cast<uint2 *>(out + (size_t)r * 2 * kDim);
    o[l] = *reinterpret_cast<uint2 *>(hi);
    o[kRowVec + l] = *reinterpret_cast<uint2 *>(lo);
  }
}

// sigmoid with the hardware exp2/rcp units (each within 1 ulp): 4 VALU instead of ~35 for the
// IEEE-division form; the epilogue of 1.8 G ratings would otherwise out-cost the MFMAs
__device__ __forceinline__ float sigmoid_fast(float y) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-y));
}

struct PredArgs {
  const uint4 *Ub;  // [n_users][32] 16-B chunks: 16 hi chunks then 16 lo chunks per row
  const uint4 *Ab;  // [n_anime][32]
  int n_users, n_anime;
  float hs, hb;     // sigmoid(c * hs + hb); hs already carries the 2^-16 of the operand scaling
  float *out;       // [n_users][n_anime]
};

__global__ __launch_bounds__(256, 2) void k_predict_mfma(PredArgs a) {
  __shared__ __attribute__((aligned(16))) uint4 Ks[2][kPN * 32];  // 2 x 32 KB: hi+lo of 64 anime rows
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r32 = lane & 31, h = lane >> 5;
  const int u0 = blockIdx.x * kPM;
  f16x8 qh[8], ql[8];
  {
    const int urow = u0 + 32 * w + r32;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      uint4 vh = make_uint4(0, 0, 0, 0), vl = vh;
      if (urow < a.n_users) {
        vh = a.Ub[(size_t)urow * 32 + 2 * ks + h];
        vl = a.Ub[(size_t)urow * 32 + 16 + 2 * ks + h];
      }
      qh[ks] = *reinterpret_cast<f16x8 *>(&vh);
      ql[ks] = *reinterpret_cast<f16x8 *>(&vl);
    }
  }
  const int ntiles = (a.n_anime + kPN - 1) / kPN;
  uint4 stage[8];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i;  // chunk id: row = c >> 5, chunk = c & 31 (0..15 hi, 16..31 lo)
      const int row = t * kPN + (c >> 5);
      stage[i] = row < a.n_anime ? a.Ab[(size_t)row * 32 + (c & 31)] : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i;
      const int row = c >> 5, ch = c & 31;
      // swizzle inside each 16-chunk plane so the 16-lane groups of ds_read_b128 hit distinct slots
      Ks[buf][row * 32 + (ch & 16) + ((ch & 15) ^ (row & 15))] = stage[i];
    }
  };
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) load_tile(t + 1);
    f32x16 acc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[cb][g] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int row = cb * 32 + r32;
        const uint4 bh = Ks[buf][row * 32 + ((2 * ks + h) ^ (row & 15))];
        const uint4 bl = Ks[buf][row * 32 + 16 + ((2 * ks + h) ^ (row & 15))];
        const f16x8 fh = *reinterpret_cast<const f16x8 *>(&bh);
        const f16x8 fl = *reinterpret_cast<const f16x8 *>(&bl);
        // small terms first: hi*lo + lo*hi, then hi*hi
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql[ks], fh, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[ks], fl, acc[cb], 0, 0, 0);
        acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh[ks], fh, acc[cb], 0, 0, 0);
      }
    }
    if (t + 1 < ntiles) store_tile(buf ^ 1);
    // epilogue: BN-inference head + sigmoid; a half-wave writes 32 consecutive anime of one user
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int col = t * kPN + cb * 32 + r32;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int urow = u0 + 32 * w + (g & 3) + 8 * (g >> 2) + 4 * h;
        if (urow < a.n_users && col < a.n_anime)
          a.out[(size_t)urow * a.n_anime + col] = sigmoid_fast(acc[cb][g] * a.hs + a.hb);
      }
    }
    __syncthreads();
  }
}

static inline void head_affine_mfma(const anirec_head *hd, float *hs, float *hb) {
  const float inv = (1.0f / sqrtf(hd->mov_var + kBnEps)) * hd->gamma;
  *hs = hd->w * inv / (kSplitScale * kSplitScale);
  *hb = hd->b * inv + (hd->beta - hd->mov_mean * inv);
}

}  // namespace anirec

using namespace anirec;

extern "C" {

// workspace: split-fp16 copies of A (n_anime * 512 B) and of the query users (n_users * 512 B)
size_t anirec_predict_mfma_workspace_bytes(int32_t n_anime, int32_t n_users) {
  if (n_anime < 1 || n_users < 1) return 0;
  return ((size_t)n_anime + (size_t)n_users) * 512 + 512;
}

int anirec_predict_grid_mfma(const float *U, const float *A, int32_t n_anime, const int32_t *users,
                             int32_t n_users, const anirec_head *head, float *out, void *workspace,
                             size_t workspace_bytes, void *stream) {
  if (!U || !A || !users || !head || !out || !workspace || n_anime < 1 || n_users < 0)
    return ANIREC_EINVAL;
  if (n_users == 0) return ANIREC_OK;
  if (workspace_bytes < anirec_predict_mfma_workspace_bytes(n_anime, n_users)) return ANIREC_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  _Float16 *Ab = (_Float16 *)workspace;
  _Float16 *Ub = Ab + (size_t)n_anime * 2 * kDim;
  int b1 = (n_anime + 7) / 8, b2 = (n_users + 7) / 8;
  if (b1 > 8192) b1 = 8192;
  if (b2 > 8192) b2 = 8192;
  hipLaunchKernelGGL(k_norm_split, dim3(b1), dim3(256), 0, s, A, nullptr, n_anime, Ab);
  hipLaunchKernelGGL(k_norm_split, dim3(b2), dim3(256), 0, s, U, users, n_users, Ub);
  ANIREC_HIP_CHECK(hipGetLastError());
  PredArgs pa;
  pa.Ub = (const uint4 *)Ub;
  pa.Ab = (const uint4 *)Ab;
  pa.n_users = n_users;
  pa.n_anime = n_anime;
  head_affine_mfma(head, &pa.hs, &pa.hb);
  pa.out = out;
  hipLaunchKernelGGL(k_predict_mfma, dim3((n_users + kPM - 1) / kPM), dim3(256), 0, s, pa);
  return (int)hipGetLastError();
}

}  // extern "C"
