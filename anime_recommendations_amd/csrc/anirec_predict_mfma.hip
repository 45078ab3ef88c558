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
#include <stdlib.h>

#include "anirec_dev.hpp"

namespace anirec {

constexpr int kPM = 128;  // users per workgroup (4 waves x 32 rows)
constexpr int kPN = 64;   // anime per tile
constexpr float kSplitScale = 256.0f;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// l2_normalize (tf epsilon 1e-12) a gathered row, scale by 2^8 and split into fp16 hi / lo planes.
// out layout: [rows][2][128] halves (hi plane then lo plane of the row: 512 B per row)
// rows n .. n_pad-1 of the output are zero rows (tile padding of the anime table)
__global__ __launch_bounds__(256) void k_norm_split(const float *W, const int32_t *rows, int n, int n_pad,
                                                    _Float16 *out) {
  const int l = threadIdx.x & 31;
  const int nhw = gridDim.x * 8;
  for (int r = blockIdx.x * 8 + (threadIdx.x >> 5); r < n_pad; r += nhw) {
    if (r >= n) {
      uint2 *o = reinterpret_cast<uint2 *>(out + (size_t)r * 2 * kDim);
      o[l] = make_uint2(0u, 0u);
      o[kRowVec + l] = make_uint2(0u, 0u);
      continue;
    }
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
  int tiles_per_part;  // k_predict_mfma2: blockIdx.y walks anime tiles [y * tiles_per_part, (y+1) * tiles_per_part)
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


// ------------------------------------------------------------------------------------------------
// k_predict_mfma2: the same grid with the operand roles swapped and a pipelined epilogue.
//   * v_mfma_f32_16x16x32_f16 with the ANIME rows as the A (row) operand and the users as the B (column)
//     operand: a lane's four accumulator registers of a 16x16 block are four CONSECUTIVE anime of ONE user,
//     i.e. one 16-byte piece of an output row.  Stored as they stand (16 users x 64 B per instruction) the
//     store path tops out at 3.1 TB/s; the SHAPE of a store instruction is what counts (probed on the GPU:
//     4 rows x 256 B, 2 x 512 B and 1 x 1 KiB per instruction all reach 4.4 TB/s), so the 16 pieces of a
//     user's 64 anime cross a wave-private, XOR-swizzled 4 KB LDS image (4 ds_write_b128 + 4 ds_read_b128 per
//     16 users, no barrier: a wave's LDS operations execute in order) and leave as 4 rows x 256 B per
//     instruction: 8 non-temporal 16-byte buffer stores per tile and wave instead of 32 dword stores, whose
//     range check drops rows past n_users and columns past n_anime without branches.
//   * the anime tile goes global -> LDS by LDS-DMA (no staging registers), two tiles resident;
//   * sigmoid + store of tile t-1 are interleaved with the 96 MFMAs of tile t (second accumulator set):
//     the store tail never runs on its own.
// Requires n_anime % 4 == 0 (16-byte aligned row quads); other shapes take k_predict_mfma.
//
// Round 4, where the 100 k x 18 k grid's time goes (scripts/time_predict.py, interleaved rounds in one process; kDbg 1 =
// epilogue + stores without the MFMAs, kDbg 2 = MFMAs without the stores):
//   box A: whole kernel 1.50 ms (4.8 TB/s written), stores alone 1.30, MFMAs alone 1.43 (0.97 PFLOP/s on the
//   1.38 TFLOP of the three-term split); box B: 1.71 ms (4.2 TB/s), stores alone 1.64, MFMAs alone 1.30.
// The two sides are within 10 % of each other and of the whole (the overlap works), which of them is longer depends
// on the box, and 35 % between boxes on the same binary is more than any kernel change below bought.
// Built, measured and dropped: 64 users per wave (the wave's two LDS reads of an anime fragment feed twelve MFMAs
// instead of six — the MFMA side is LDS-read-bound: every wave re-reads the whole 32 KB tile), a step covering half a
// tile to stay inside 256 VGPRs, a user block's 32 anime leaving as 8 rows x 128 B per store instruction.
// Bit-identical output; MFMAs alone 1.30 -> 1.17 ms, but stores alone 1.64 -> 1.87 ms (128-B row segments store at
// 3.9 TB/s where 256-B segments reach 4.4 on that box) and the whole kernel 1.71 -> 1.92 ms.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Acc2 {
  f32x4 c[4][2];  // [anime block of 16][user block of 16]
};

// kDbg (timing only, wrong output): 1 = epilogue + stores without the MFMAs, 2 = MFMAs without the stores
template <int kDbg>
__global__ __launch_bounds__(256, 2) void k_predict_mfma2(PredArgs a) {
  // one LDS array (a second object beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read):
  // 2 x 32 KB key tiles (hi+lo planes of 64 anime rows), then 4 x 4 KB output staging (one per wave)
  __shared__ __attribute__((aligned(16))) uint4 Ks[2 * kPN * 32 + 4 * 256];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c16 = lane & 15, gq = lane >> 4;
  const int u0 = blockIdx.x * kPM;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  // staging image of 16 users x 64 anime fp32: row = user (256 B = 16 slots of 16 B), slot ^= row (conflict-free
  // ds_write_b128 of the accumulator layout and ds_read_b128 of whole rows)
  uint4 *const stg = &Ks[2 * kPN * 32 + 256 * w];

  // B operand: user (16 ub + c16) of the wave's 32, k = 32 kk + 8 gq + j
  f16x8 uh[2][4], ul[2][4];
#pragma unroll
  for (int ub = 0; ub < 2; ++ub) {
    const int urow = u0 + 32 * w + 16 * ub + c16;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4 vh = {0u, 0u, 0u, 0u}, vl = {0u, 0u, 0u, 0u};
      if (urow < a.n_users) {
        vh = *reinterpret_cast<const u32x4 *>(&a.Ub[(size_t)urow * 32 + 4 * kk + gq]);
        vl = *reinterpret_cast<const u32x4 *>(&a.Ub[(size_t)urow * 32 + 16 + 4 * kk + gq]);
      }
      uh[ub][kk] = __builtin_bit_cast(f16x8, vh);
      ul[ub][kk] = __builtin_bit_cast(f16x8, vl);
    }
  }
  // anime tile t: 64 rows x 512 B, LDS image [row][32 chunks] with slot = plane + ((chunk & 15) ^ (row & 15));
  // LDS-DMA writes lane-linear, so the swizzle is applied to the per-lane SOURCE chunk.  A wave moves 8 pieces
  // of 1 KiB (2 rows each) per tile.  Ab is padded with zero rows to whole tiles: no bounds test.
  uint32_t doff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = 16 * w + 2 * i + (lane >> 5);
    const int sl = lane & 31;
    doff[i] = (uint32_t)(r * 512 + ((sl & 16) + ((sl & 15) ^ (r & 15))) * 16);
  }
  const uint32_t ks_base =
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&Ks[0] + (uint32_t)wu * 8192u;
  auto dma_tile = [&](int t, int buf) {
    const char *base = reinterpret_cast<const char *>(a.Ab) + (size_t)t * (kPN * 512);
    const uint32_t l0 = ks_base + (uint32_t)buf * (kPN * 512);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :
                   : "v"(doff[i]), "s"(base), "s"(l0 + 1024u * i)
                   : "memory", "m0");
  };
  // A operand read addresses: anime row 16 ab + c16 has (row & 15) == c16
  const f16x8 *ka[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    ka[kk] = reinterpret_cast<const f16x8 *>(&Ks[c16 * 32 + ((4 * kk + gq) ^ c16)]);

  // output: per-workgroup buffer descriptor over rows [u0, u0 + valid rows): rows past n_users and the byte
  // offset 0xFFFFFFF0 used for columns past n_anime fail the range check and are dropped by the hardware
  const int rows_valid = min(kPM, a.n_users - u0);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out + (size_t)u0 * a.n_anime, 0, (int)((size_t)rows_valid * a.n_anime * 4), 0x00020000);
  // store instruction j of user block ub: lane -> row 4 j + gq of the block, 16-byte piece c16 of its 256 B
  const uint32_t vrow = (uint32_t)((32 * w + gq) * a.n_anime + 4 * c16) * 4u;
  const uint32_t vrow4 = (uint32_t)(4 * a.n_anime) * 4u;
  // sigmoid(c * hs + hb) = 1 / (1 + 2^(c * nhs + nhb))
  const float nhs = -a.hs * 1.44269504088896341f, nhb = -a.hb * 1.44269504088896341f;

  // this workgroup's anime tiles [tb, tb + ntiles): the grid's y dimension cuts the anime table into parts so that
  // there are several times more workgroups than the 512 the chip holds and the dispatcher balances the tail
  const int tb = blockIdx.y * a.tiles_per_part;
  const int ntiles = min(a.tiles_per_part, (a.n_anime + kPN - 1) / kPN - tb);
  dma_tile(tb, 0);
  if (ntiles > 1) dma_tile(tb + 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // kEpi: interleave the epilogue of `cur` (tile tc) with the MFMAs of `nxt` (tile in buffer `buf`)
  auto step = [&](Acc2 &nxt, const Acc2 &cur, int buf, int tc, bool epi) {
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const f16x8 ah = ka[kk][(buf * kPN + 16 * ab) * 32];
        const f16x8 al = ka[kk][(buf * kPN + 16 * ab) * 32 + 16];
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
          f32x4 c = nxt.c[ab][ub];
          if (kk == 0) c = (f32x4){0.f, 0.f, 0.f, 0.f};
          // small terms first: lo*hi + hi*lo, then hi*hi
          if (kDbg != 1) {
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, uh[ub][kk], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ul[ub][kk], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, uh[ub][kk], c, 0, 0, 0);
          } else {
            c[0] += ah[0] + al[1];
          }
          nxt.c[ab][ub] = c;
        }
        if (epi && (kk & 1)) {  // 8 epilogue blocks per tile, one after every 12 MFMAs
          const int eb = 2 * ab + (kk >> 1);  // 0..7 -> user block eb >> 2, anime block eb & 3
          const int eub = eb >> 2, eab = eb & 3;
          const f32x4 v = cur.c[eab][eub];
          f32x4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            o[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(v[i], nhs, nhb)));
          // piece (anime block eab, quad gq) of user c16
          stg[c16 * 16 + ((4 * eab + gq) ^ c16)] = __builtin_bit_cast(uint4, o);
          if (eab == 3) {  // the user block is complete: 4 rows x 256 B per store instruction
            const bool in = tc * kPN + 4 * c16 < a.n_anime;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int row = 4 * j + gq;
              const u32x4 d = __builtin_bit_cast(u32x4, stg[row * 16 + (c16 ^ row)]);
              const uint32_t voff = in ? vrow + (uint32_t)(16 * eub + 4 * j) * (vrow4 / 4u) + (uint32_t)(tc * kPN) * 4u
                                       : 0xFFFFFFF0u;
              if (kDbg != 2)
                __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, voff, 0, 2 /* nt */);
              else
                asm volatile("" ::"v"(d));
            }
          }
        }
      }
    }
  };
  Acc2 accA, accB;
  step(accA, accA, 0, 0, false);
  __syncthreads();  // everyone is done with buffer 0
  int t = 1;        // tile index relative to tb
  for (; t + 1 < ntiles; t += 2) {
    if (t + 1 < ntiles) dma_tile(tb + t + 1, 0);
    step(accB, accA, 1, tb + t - 1, true);
    // the tile landed.  (Rounds 2-4 waited vmcnt(8) — "the 8 younger stores stay in flight" — which leans on one wave's
    // loads and stores retiring in issue order; the counter is shared but that order is not documented for a mix of the
    // two.  Draining the stores as well measured the same to half a per cent — 1.983 / 1.956 against 1.980 / 1.946 ms,
    // interleaved on one box — so the wait no longer depends on it.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 2 < ntiles) dma_tile(tb + t + 2, 1);
    step(accA, accB, 0, tb + t, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (t < ntiles) {  // one tile left (in buffer 1), results of tile t-1 in accA
    step(accB, accA, 1, tb + t - 1, true);
    step(accA, accB, 0, tb + t, true);  // MFMAs on stale buffer 0 are thrown away; only the epilogue of tile t matters
  } else {           // results of the last tile in accA
    step(accB, accA, 0, tb + t - 1, true);
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
  return (((size_t)n_anime + kPN - 1) / kPN * kPN + (size_t)n_users) * 512 + 512;
}

int anirec_predict_grid_mfma(const float *U, const float *A, int32_t n_anime, const int32_t *users,
                             int32_t n_users, const anirec_head *head, float *out, void *workspace,
                             size_t workspace_bytes, void *stream) {
  if (!U || !A || !users || !head || !out || !workspace || n_anime < 1 || n_users < 0)
    return ANIREC_EINVAL;
  if (n_users == 0) return ANIREC_OK;
  if (workspace_bytes < anirec_predict_mfma_workspace_bytes(n_anime, n_users)) return ANIREC_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int n_pad = (n_anime + kPN - 1) / kPN * kPN;  // whole tiles: the LDS-DMA of k_predict_mfma2 has no bounds test
  _Float16 *Ab = (_Float16 *)workspace;
  _Float16 *Ub = Ab + (size_t)n_pad * 2 * kDim;
  int b1 = (n_anime + 7) / 8, b2 = (n_users + 7) / 8;
  if (b1 > 8192) b1 = 8192;
  if (b2 > 8192) b2 = 8192;
  hipLaunchKernelGGL(k_norm_split, dim3(b1), dim3(256), 0, s, A, nullptr, n_anime, n_pad, Ab);
  hipLaunchKernelGGL(k_norm_split, dim3(b2), dim3(256), 0, s, U, users, n_users, n_users, Ub);
  ANIREC_HIP_CHECK(hipGetLastError());
  PredArgs pa;
  pa.Ub = (const uint4 *)Ub;
  pa.Ab = (const uint4 *)Ab;
  pa.n_users = n_users;
  pa.n_anime = n_anime;
  pa.tiles_per_part = 0;
  head_affine_mfma(head, &pa.hs, &pa.hb);
  pa.out = out;
  // 16-byte row quads need n_anime % 4 == 0 and 32-bit byte offsets inside a workgroup's 128 rows
  const char *ver = getenv("ANIREC_PREDICT_KERNEL");  // "1": force the dword-store kernel (A/B on one box)
  const bool v2 = (n_anime % 4 == 0) && ((size_t)n_anime * kPM * 4 < ((size_t)1 << 31)) && !(ver && ver[0] == '1');
  if (v2) {
    // anime parts: enough workgroups (>= ~8 per resident slot) for the dispatcher to balance the tail, parts of at
    // least 16 tiles so the pipeline fill of a part stays a few per cent
    const int ub = (n_users + kPM - 1) / kPM, ntile = n_pad / kPN;
    const char *pe = getenv("ANIREC_PREDICT_PARTS");
    int parts = pe ? atoi(pe) : (4096 + ub - 1) / ub;
    if (parts > ntile / 16) parts = ntile / 16;
    if (parts < 1) parts = 1;
    pa.tiles_per_part = (ntile + parts - 1) / parts;
    parts = (ntile + pa.tiles_per_part - 1) / pa.tiles_per_part;
    const dim3 grid(ub, parts);
    const char *dbg = getenv("ANIREC_PREDICT_DEBUG");
    const int mode = dbg ? atoi(dbg) : 0;
    if (mode == 1)
      hipLaunchKernelGGL(k_predict_mfma2<1>, grid, dim3(256), 0, s, pa);
    else if (mode == 2)
      hipLaunchKernelGGL(k_predict_mfma2<2>, grid, dim3(256), 0, s, pa);

    else
      hipLaunchKernelGGL(k_predict_mfma2<0>, grid, dim3(256), 0, s, pa);
  } else
    hipLaunchKernelGGL(k_predict_mfma, dim3((n_users + kPM - 1) / kPM), dim3(256), 0, s, pa);
  return (int)hipGetLastError();
}

}  // extern "C"
