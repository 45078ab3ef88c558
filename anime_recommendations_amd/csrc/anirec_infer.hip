// Inference hot path of libanirec for gfx950 (MI355X).
//
//   rownorm        get_weights(): W / ||W||  (similar_anime.py:136-171, similar_users.py:75-101)
//   cosine scores  np.dot(W_hat, W_hat[q])   (similar_anime.py:404, similar_users.py:293)
//   top-k select   np.argsort + slice        (similar_anime.py:408, similar_users.py:294-296,
//                                             model_recs.py:396)
//   predict        model.predict([u, a])     (model_recs.py:394)
//
// Score arithmetic is DEFINED (see oracle/): every dot product is the k-ordered fp32
// fused-multiply-add chain s = fma(x[k], y[k], s), k = 0..127 — the order of
// v_mfma_f32_32x32x2_f32 — so neighbour lists are reproducible bit-for-bit.
#include <hip/hip_runtime.h>
#include <math.h>

#include "anirec_dev.hpp"

namespace anirec {

// ------------------------------------------------------------------------------------
// row normalisation
// ------------------------------------------------------------------------------------
// mode 0: NumPy get_weights  : x / sqrt(sum x^2)            (no epsilon, zero row -> NaN)
// mode 1: tf.nn.l2_normalize : x * (1/sqrt(max(sum x^2, 1e-12)))   (Dot(normalize=True))
// rows: optional gather list (out row j = W[rows[j]]), else identity.
template <int kMode>
__global__ __launch_bounds__(256) void k_rownorm(const float *W, const int32_t *rows, int n,
                                                 float *out) {
  const int l = threadIdx.x & 31;
  const int nhw = gridDim.x * 8;
  for (int r = blockIdx.x * 8 + (threadIdx.x >> 5); r < n; r += nhw) {
    const int src = rows ? rows[r] : r;
    float4 x = reinterpret_cast<const float4 *>(W)[(size_t)src * kRowVec + l];
    float ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    ss = halfwave_sum(ss);
    float4 y;
    if (kMode == 0) {
      const float nrm = sqrtf(ss);
      y.x = x.x / nrm;
      y.y = x.y / nrm;
      y.z = x.z / nrm;
      y.w = x.w / nrm;
    } else {
      const float rinv = 1.0f / sqrtf(fmaxf(ss, kL2nEps));
      y.x = x.x * rinv;
      y.y = x.y * rinv;
      y.z = x.z * rinv;
      y.w = x.w * rinv;
    }
    reinterpret_cast<float4 *>(out)[(size_t)r * kRowVec + l] = y;
  }
}

// ------------------------------------------------------------------------------------
// tiled scores: out[q][j] = epilogue( chain_dot(Q[q], W[j]) )
// 64 queries x 64 rows per workgroup, 4x4 register block per thread, both tiles staged
// in LDS with a 132-float row pitch (ds_read_b128 of rows tx+16r is conflict-free).
// ------------------------------------------------------------------------------------
constexpr int kTile = 64;
constexpr int kPitch = kDim + 4;

struct ScoreArgs {
  const float *Q;         // [*, 128] query matrix
  const int32_t *qrows;   // optional: query j reads Q[qrows[j]]
  int nq;
  const float *W;         // [n, 128]
  int n;
  float *out;             // [nq][ld]
  size_t ld;
  int use_head;           // 1: sigmoid(c*hs + hb)
  float hs, hb;
};

__global__ __launch_bounds__(256) void k_scores(ScoreArgs a) {
  __shared__ __attribute__((aligned(16))) float Qs[kTile * kPitch];
  __shared__ __attribute__((aligned(16))) float Ws[kTile * kPitch];
  const int tid = threadIdx.x;
  const int q0 = blockIdx.y * kTile, j0 = blockIdx.x * kTile;
  // stage: 64 rows x 32 float4 per tile, 256 threads -> 8 float4 each per tile
  for (int e = tid; e < kTile * kRowVec; e += 256) {
    const int r = e >> 5, cidx = e & 31;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f), wv = qv;
    if (q0 + r < a.nq) {
      const int src = a.qrows ? a.qrows[q0 + r] : q0 + r;
      qv = reinterpret_cast<const float4 *>(a.Q)[(size_t)src * kRowVec + cidx];
    }
    if (j0 + r < a.n) wv = reinterpret_cast<const float4 *>(a.W)[(size_t)(j0 + r) * kRowVec + cidx];
    *reinterpret_cast<float4 *>(&Qs[r * kPitch + cidx * 4]) = qv;
    *reinterpret_cast<float4 *>(&Ws[r * kPitch + cidx * 4]) = wv;
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;  // rows tx+16r, queries ty+16q
  float acc[4][4];
#pragma unroll
  for (int qq = 0; qq < 4; ++qq)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[qq][r] = 0.f;
#pragma unroll 4
  for (int k4 = 0; k4 < kRowVec; ++k4) {
    float4 qv[4], wv[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
      qv[qq] = *reinterpret_cast<const float4 *>(&Qs[(ty + 16 * qq) * kPitch + k4 * 4]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      wv[r] = *reinterpret_cast<const float4 *>(&Ws[(tx + 16 * r) * kPitch + k4 * 4]);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = acc[qq][r];
        s = __fmaf_rn(wv[r].x, qv[qq].x, s);
        s = __fmaf_rn(wv[r].y, qv[qq].y, s);
        s = __fmaf_rn(wv[r].z, qv[qq].z, s);
        s = __fmaf_rn(wv[r].w, qv[qq].w, s);
        acc[qq][r] = s;
      }
  }
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int q = q0 + ty + 16 * qq;
    if (q >= a.nq) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + tx + 16 * r;
      if (j >= a.n) continue;
      float s = acc[qq][r];
      if (a.use_head) s = rating_from_cosine(s, a.hs, a.hb);
      a.out[(size_t)q * a.ld + j] = s;
    }
  }
}

// ------------------------------------------------------------------------------------
// exact top-k of each score row: 4-pass MSB radix select on order-preserving keys, then
// ordered collection (ties -> ascending index) and a bitonic sort of the k winners.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t score_key(float s) {
  // larger score -> larger key; NaN -> 1 (ranks after every number); 0 is "not a candidate"
  if (s != s) return 1u;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return u < 2u ? 2u : u;
}

struct SelectArgs {
  const float *scores;   // [nq][ld]
  size_t ld;
  int n, nq, k;
  const int32_t *self;   // optional [nq]: index excluded for query q (or -1)
  const uint8_t *keep;   // optional [n] byte mask shared by all queries
  const uint32_t *wbits; // optional [nq][wwords] bit mask: set bit = excluded (watched)
  int wwords;
  int32_t *out_idx;      // [nq * slices][k]
  float *out_score;      // [nq * slices][k]
  int slices, slice_len; // workgroup b selects among keys [s*slice_len, (s+1)*slice_len) of query b / slices
  const int32_t *src_idx; // merge pass: [nq][ld] real index of list entry j (-1 = empty), masks already applied
};

__device__ __forceinline__ uint32_t cand_key(const SelectArgs &a, const float *row, int q, int j,
                                             int self) {
  if (a.src_idx) return a.src_idx[(size_t)q * a.ld + j] < 0 ? 0u : score_key(row[j]);
  if (j == self) return 0u;
  if (a.keep && !a.keep[j]) return 0u;
  if (a.wbits && ((a.wbits[(size_t)q * a.wwords + (j >> 5)] >> (j & 31)) & 1u)) return 0u;
  return score_key(row[j]);
}

constexpr int kSelThreads = 256;

__global__ __launch_bounds__(kSelThreads) void k_select(SelectArgs a) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh_prefix, sh_want;
  __shared__ uint32_t wsum[kSelThreads / 64];
  __shared__ unsigned long long win[ANIREC_MAX_TOPK];  // (key << 32) | ~idx  -> sort desc
  const int tid = threadIdx.x;
  // Few queries (the reference's literal call is ONE query against every row): a query's keys are cut
  // into slices, one workgroup each, and a second launch of this kernel merges the slice winners.  The
  // winners of a slice are sorted (score desc, index asc) and slices cover ascending index ranges, so
  // list order among equal scores is ascending index — the tie rule survives the merge unchanged.
  const int q = blockIdx.x / a.slices;
  const int j_lo = (blockIdx.x % a.slices) * a.slice_len;
  const int j_hi = min(a.n, j_lo + a.slice_len);
  const size_t orow = blockIdx.x;
  const float *row = a.scores + (size_t)q * a.ld;
  const int self = a.self ? a.self[q] : -1;
  const int k = a.k;

  // MSB-first radix select for the k-th largest key among candidates (key != 0)
  uint32_t prefix = 0, pmask = 0;
  uint32_t want = (uint32_t)k;  // rank (1-based from the top) still to locate inside prefix
  bool short_row = false;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    for (int j = j_lo + tid; j < j_hi; j += kSelThreads) {
      const uint32_t key = cand_key(a, row, q, j, self);
      if (key != 0u && (key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      // digit d (from 255 down) where the running count first reaches `want`: wave scan instead of a
      // 256-step serial walk by one thread (that walk was ~8 us per pass: most of a small select)
      uint32_t h4[4], run = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // lane l covers digits 255-4l .. 252-4l, in that order
        h4[j] = hist[255 - (4 * tid + j)];
        run += h4[j];
      }
      uint32_t inc = run;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(inc, o, 64);
        if (tid >= o) inc += y;
      }
      const uint32_t before = inc - run;  // candidates in digits above this lane's four
      const unsigned long long reach = __ballot(inc >= want);
      if (reach == 0ull) {
        if (tid == 0) {  // fewer than k candidates in total: take them all
          sh_prefix = 0xFFFFFFFFu;
          sh_want = 0;
        }
      } else if (tid == __ffsll((long long)reach) - 1) {
        uint32_t cum = before;
        int j = 0;
        for (; j < 3; ++j) {
          if (cum + h4[j] >= want) break;
          cum += h4[j];
        }
        sh_prefix = prefix | ((uint32_t)(255 - (4 * tid + j)) << shift);
        sh_want = want - cum;
      }
    }
    __syncthreads();
    if (sh_prefix == 0xFFFFFFFFu && sh_want == 0) {
      short_row = true;
      break;
    }
    prefix = sh_prefix;
    want = sh_want;
    pmask |= 255u << shift;
    __syncthreads();
  }
  // threshold T = prefix (exact key of the k-th largest); take all keys > T and the first
  // `want` keys == T in ascending index order.  short_row: take every candidate.
  const uint32_t T = short_row ? 0u : prefix;
  const uint32_t need_eq = short_row ? 0xFFFFFFFFu : want;

  uint32_t n_out = 0;   // winners written so far (block-uniform)
  uint32_t eq_taken = 0;
  constexpr int kPer = 16;
  const int super = kSelThreads * kPer;
  for (int base = j_lo; base < j_hi; base += super) {
    uint32_t keys[kPer];
    uint32_t c_gt = 0, c_eq = 0;
    const int j0 = base + tid * kPer;
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
      const int j = j0 + e;
      keys[e] = j < j_hi ? cand_key(a, row, q, j, self) : 0u;
      if (keys[e] != 0u) {
        if (short_row || keys[e] > T) ++c_gt;
        else if (keys[e] == T) ++c_eq;
      }
    }
    if (__syncthreads_count((c_gt | c_eq) != 0) == 0) continue;
    // ordered ranks of this thread's matches (two scans packed in one: gt in low 16 bits...
    // counts can reach 4096 per super-chunk -> use two separate scans)
    uint32_t tot_gt = 0, tot_eq = 0;
    uint32_t o_gt, o_eq;
    {
      const int lane = tid & 63, w = tid >> 6;
      uint32_t inc = c_gt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      if (lane == 63) wsum[w] = inc;
      __syncthreads();
      uint32_t b = 0;
      for (int kk = 0; kk < kSelThreads / 64; ++kk) {
        if (kk < w) b += wsum[kk];
        tot_gt += wsum[kk];
      }
      o_gt = b + inc - c_gt;
      __syncthreads();
      inc = c_eq;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
      }
      if (lane == 63) wsum[w] = inc;
      __syncthreads();
      b = 0;
      for (int kk = 0; kk < kSelThreads / 64; ++kk) {
        if (kk < w) b += wsum[kk];
        tot_eq += wsum[kk];
      }
      o_eq = b + inc - c_eq;
      __syncthreads();
    }
    const uint32_t eq_room = need_eq == 0xFFFFFFFFu ? 0u : (need_eq - eq_taken);
    const uint32_t eq_use = tot_eq < eq_room ? tot_eq : eq_room;
    // winners land at [n_out, n_out + tot_gt) for gt and after them the eq ones of this chunk
#pragma unroll
    for (int e = 0; e < kPer; ++e) {
      const uint32_t key = keys[e];
      if (key == 0u) continue;
      const int j = j0 + e;
      if (short_row || key > T) {
        const uint32_t slot = n_out + o_gt++;
        if (slot < (uint32_t)ANIREC_MAX_TOPK)
          win[slot] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)j);
      } else if (key == T) {
        const uint32_t r = o_eq++;
        if (r < eq_use) {
          const uint32_t slot = n_out + tot_gt + r;
          if (slot < (uint32_t)ANIREC_MAX_TOPK)
            win[slot] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)j);
        }
      }
    }
    n_out += tot_gt + eq_use;
    eq_taken += eq_use;
    if (n_out > (uint32_t)ANIREC_MAX_TOPK) n_out = ANIREC_MAX_TOPK;
    __syncthreads();
  }
  __syncthreads();
  // pad to 128 with zeros (sort last) and bitonic sort descending
  for (int i = tid; i < ANIREC_MAX_TOPK; i += kSelThreads)
    if ((uint32_t)i >= n_out) win[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= ANIREC_MAX_TOPK; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (tid < ANIREC_MAX_TOPK / 2) {
        const int lo = 2 * tid - (tid & (stride - 1));
        const int hi = lo + stride;
        const bool desc = ((lo & size) == 0);
        const unsigned long long x = win[lo], y = win[hi];
        if ((x < y) == desc) {
          win[lo] = y;
          win[hi] = x;
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += kSelThreads) {
    const unsigned long long v = win[i];
    if ((uint32_t)i < n_out && v != 0ull) {
      const int j = (int)(~(uint32_t)(v & 0xFFFFFFFFull));
      a.out_idx[orow * k + i] = a.src_idx ? a.src_idx[(size_t)q * a.ld + j] : j;
      a.out_score[orow * k + i] = row[j];
    } else {
      a.out_idx[orow * k + i] = -1;
      a.out_score[orow * k + i] = __uint_as_float(0x7FC00000u);
    }
  }
}

// ------------------------------------------------------------------------------------
// predict on explicit pairs (model.predict([user_arr, anime_arr]))
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_predict_pairs(const float *U, const float *A,
                                                       const int32_t *ui, const int32_t *ai, int n,
                                                       float hs, float hb, float *p) {
  const int l = threadIdx.x & 31;
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (i >= n) return;
  const float4 u = reinterpret_cast<const float4 *>(U)[(size_t)ui[i] * kRowVec + l];
  const float4 x = reinterpret_cast<const float4 *>(A)[(size_t)ai[i] * kRowVec + l];
  const float su = halfwave_sum(u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w);
  const float sa = halfwave_sum(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
  const float dd = halfwave_sum(u.x * x.x + u.y * x.y + u.z * x.z + u.w * x.w);
  if (l == 0) {
    const float ru = 1.0f / sqrtf(fmaxf(su, kL2nEps));
    const float ra = 1.0f / sqrtf(fmaxf(sa, kL2nEps));
    p[i] = sigmoidf_stable(dd * ru * ra * hs + hb);
  }
}

__global__ void k_fill_self(const int32_t *queries, int nq, int32_t *self, int enable) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) self[i] = enable ? queries[i] : -1;
}

static inline void head_affine(const anirec_head *h, float *hs, float *hb) { head_affine_f32(h, hs, hb); }

// Few queries (<= 16; the reference's literal call has ONE): a GEMV-shaped kernel.  A workgroup stages 64 key
// rows coalesced into LDS (pitch 132 floats: float4 rows whose 16-lane read groups fall on distinct banks) and
// each thread runs the DEFINED k-ordered fma chain of one (key, query) pair — the same arithmetic as k_scores,
// bit for bit — so the 64 x 64 tile kernel's 63/64 wasted lanes disappear and the pass is HBM-bound.
constexpr int kFewQ = 16;
constexpr int kFewPitch = kDim + 4;
__global__ __launch_bounds__(256) void k_scores_few(ScoreArgs a) {
  __shared__ __attribute__((aligned(16))) float Ws[64 * kFewPitch];
  __shared__ __attribute__((aligned(16))) float Qs[kFewQ * kDim];
  const int tid = threadIdx.x;
  const int j0 = blockIdx.x * 64;
  for (int e = tid; e < 64 * kRowVec; e += 256) {
    const int r = e >> 5, c = e & 31;
    float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j0 + r < a.n) wv = reinterpret_cast<const float4 *>(a.W)[(size_t)(j0 + r) * kRowVec + c];
    *reinterpret_cast<float4 *>(&Ws[r * kFewPitch + c * 4]) = wv;
  }
  for (int e = tid; e < a.nq * kRowVec; e += 256) {
    const int q = e >> 5, c = e & 31;
    const size_t src = a.qrows ? (size_t)a.qrows[q] : (size_t)q;
    *reinterpret_cast<float4 *>(&Qs[q * kDim + c * 4]) = reinterpret_cast<const float4 *>(a.Q)[src * kRowVec + c];
  }
  __syncthreads();
  const int key = tid & 63, j = j0 + key;
  for (int q = tid >> 6; q < a.nq; q += 4) {
    const float4 *w4 = reinterpret_cast<const float4 *>(&Ws[key * kFewPitch]);
    const float4 *q4 = reinterpret_cast<const float4 *>(&Qs[q * kDim]);
    float s = 0.f;
#pragma unroll 8
    for (int k4 = 0; k4 < kRowVec; ++k4) {
      const float4 x = w4[k4], y = q4[k4];
      s = __fmaf_rn(x.x, y.x, s);
      s = __fmaf_rn(x.y, y.y, s);
      s = __fmaf_rn(x.z, y.z, s);
      s = __fmaf_rn(x.w, y.w, s);
    }
    if (a.use_head) s = rating_from_cosine(s, a.hs, a.hb);
    if (j < a.n) a.out[(size_t)q * a.ld + j] = s;
  }
}

// scratch of the sliced select: at most kSelMaxBlocks slice winners lists of ANIREC_MAX_TOPK entries
constexpr int kSelMaxBlocks = 2048;
constexpr size_t kSelTmpBytes = (size_t)kSelMaxBlocks * ANIREC_MAX_TOPK * 8;

// one launch when there are enough queries to fill the chip; otherwise slices + merge
static int launch_select(SelectArgs sa, void *tmp, hipStream_t s) {
  sa.src_idx = nullptr;
  int S = 1;
  if (sa.nq < 1024 && sa.n >= 4096) {
    S = sa.n / 2048;
    if (S > 64) S = 64;
    if (S > kSelMaxBlocks / sa.nq) S = kSelMaxBlocks / sa.nq;
    if (S < 1) S = 1;
  }
  sa.slices = S;
  sa.slice_len = (sa.n + S - 1) / S;
  if (S == 1) {
    hipLaunchKernelGGL(k_select, dim3(sa.nq), dim3(kSelThreads), 0, s, sa);
    return (int)hipGetLastError();
  }
  int32_t *tmp_idx = (int32_t *)tmp;
  float *tmp_score = (float *)((char *)tmp + (size_t)kSelMaxBlocks * ANIREC_MAX_TOPK * 4);
  SelectArgs part = sa;
  part.out_idx = tmp_idx;
  part.out_score = tmp_score;
  hipLaunchKernelGGL(k_select, dim3(sa.nq * S), dim3(kSelThreads), 0, s, part);
  SelectArgs m = sa;
  m.scores = tmp_score;
  m.src_idx = tmp_idx;
  m.ld = (size_t)S * sa.k;
  m.n = S * sa.k;
  m.self = nullptr;
  m.keep = nullptr;
  m.wbits = nullptr;
  m.slices = 1;
  m.slice_len = m.n;
  hipLaunchKernelGGL(k_select, dim3(sa.nq), dim3(kSelThreads), 0, s, m);
  return (int)hipGetLastError();
}

static int launch_scores(const ScoreArgs &a, hipStream_t s) {
  if (a.nq <= kFewQ) {
    hipLaunchKernelGGL(k_scores_few, dim3((a.n + 63) / 64), dim3(256), 0, s, a);
    return (int)hipGetLastError();
  }
  dim3 grid((a.n + kTile - 1) / kTile, (a.nq + kTile - 1) / kTile);
  hipLaunchKernelGGL(k_scores, grid, dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace anirec

using namespace anirec;

extern "C" {

int anirec_rownorm(const float *W, int32_t n, float *What, void *stream) {
  if (!W || !What || n < 0) return ANIREC_EINVAL;
  if (n == 0) return ANIREC_OK;
  int blocks = (n + 7) / 8;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_rownorm<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, nullptr, n,
                     What);
  return (int)hipGetLastError();
}

int anirec_cosine_scores(const float *What, int32_t n, int32_t q, float *scores, void *stream) {
  if (!What || !scores || n < 1 || q < 0 || q >= n) return ANIREC_EINVAL;
  ScoreArgs a;
  a.Q = What + (size_t)q * kDim;
  a.qrows = nullptr;
  a.nq = 1;
  a.W = What;
  a.n = n;
  a.out = scores;
  a.ld = (size_t)n;
  a.use_head = 0;
  a.hs = a.hb = 0.f;
  return launch_scores(a, (hipStream_t)stream);
}

// workspace: self[nq] ints (256-aligned) + score rows for a batch of queries
size_t anirec_topk_workspace_bytes(int32_t n, int32_t nq) {
  if (n < 1 || nq < 1) return 0;
  size_t self_bytes = ((size_t)nq * 4 + 255) / 256 * 256;
  size_t qb = (size_t)nq < 1024 ? (size_t)nq : 1024;
  // cap the score buffer at 4 GiB
  while (qb > 1 && qb * (size_t)n * 4 > ((size_t)4 << 30)) qb >>= 1;
  return self_bytes + kSelTmpBytes + qb * (size_t)n * 4;
}

int anirec_cosine_topk(const float *What, int32_t n, const int32_t *queries, int32_t nq,
                       const uint8_t *keep, int32_t exclude_self, int32_t k, int32_t *out_idx,
                       float *out_score, void *workspace, size_t workspace_bytes, void *stream) {
  if (!What || !queries || !out_idx || !out_score || !workspace) return ANIREC_EINVAL;
  if (n < 1 || nq < 0 || k < 1 || k > ANIREC_MAX_TOPK) return ANIREC_EINVAL;
  if (nq == 0) return ANIREC_OK;
  hipStream_t s = (hipStream_t)stream;
  size_t self_bytes = ((size_t)nq * 4 + 255) / 256 * 256;
  if (workspace_bytes < self_bytes + kSelTmpBytes + (size_t)n * 4) return ANIREC_EWORKSPACE;
  int32_t *self = (int32_t *)workspace;
  void *sel_tmp = (char *)workspace + self_bytes;
  float *buf = (float *)((char *)workspace + self_bytes + kSelTmpBytes);
  size_t qb = (workspace_bytes - self_bytes - kSelTmpBytes) / ((size_t)n * 4);
  if (qb > (size_t)nq) qb = nq;
  hipLaunchKernelGGL(k_fill_self, dim3((nq + 255) / 256), dim3(256), 0, s, queries, nq, self,
                     exclude_self);
  ANIREC_HIP_CHECK(hipGetLastError());
  for (size_t q0 = 0; q0 < (size_t)nq; q0 += qb) {
    const int cnt = (int)((size_t)nq - q0 < qb ? (size_t)nq - q0 : qb);
    ScoreArgs a;
    a.Q = What;
    a.qrows = queries + q0;
    a.nq = cnt;
    a.W = What;
    a.n = n;
    a.out = buf;
    a.ld = (size_t)n;
    a.use_head = 0;
    a.hs = a.hb = 0.f;
    int e = launch_scores(a, s);
    if (e) return e;
    SelectArgs sa;
    sa.scores = buf;
    sa.ld = (size_t)n;
    sa.n = n;
    sa.nq = cnt;
    sa.k = k;
    sa.self = self + q0;
    sa.keep = keep;
    sa.wbits = nullptr;
    sa.wwords = 0;
    sa.out_idx = out_idx + q0 * k;
    sa.out_score = out_score + q0 * k;
    e = launch_select(sa, sel_tmp, s);
    if (e) return e;
  }
  return ANIREC_OK;
}

int anirec_predict_pairs(const float *U, const float *A, const int32_t *user_idx,
                         const int32_t *anime_idx, int32_t n, const anirec_head *head, float *p,
                         void *stream) {
  if (!U || !A || !user_idx || !anime_idx || !head || !p || n < 0) return ANIREC_EINVAL;
  if (n == 0) return ANIREC_OK;
  float hs, hb;
  head_affine(head, &hs, &hb);
  hipLaunchKernelGGL(k_predict_pairs, dim3((n + 7) / 8), dim3(256), 0, (hipStream_t)stream, U, A,
                     user_idx, anime_idx, n, hs, hb, p);
  return (int)hipGetLastError();
}

// workspace of predict_grid / predict_topk: normalised copies of the query users and of A,
// plus (topk) a batch of rating rows.
static size_t norm_bytes(int32_t n_anime, int32_t n_users) {
  return ((size_t)n_anime + (size_t)n_users) * kDim * 4;
}

size_t anirec_predict_workspace_bytes(int32_t n_anime, int32_t n_users, int32_t topk) {
  if (n_anime < 1 || n_users < 1) return 0;
  size_t b = norm_bytes(n_anime, n_users);
  if (topk) {
    size_t qb = (size_t)n_users < 4096 ? (size_t)n_users : 4096;
    b += kSelTmpBytes + qb * (size_t)n_anime * 4;
  }
  return b;
}

int anirec_predict_grid(const float *U, const float *A, int32_t n_anime, const int32_t *users,
                        int32_t n_users, const anirec_head *head, float *out, void *workspace,
                        size_t workspace_bytes, void *stream) {
  if (!U || !A || !users || !head || !out || !workspace || n_anime < 1 || n_users < 0)
    return ANIREC_EINVAL;
  if (n_users == 0) return ANIREC_OK;
  if (workspace_bytes < norm_bytes(n_anime, n_users)) return ANIREC_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float *Ah = (float *)workspace;
  float *Uh = Ah + (size_t)n_anime * kDim;
  int b1 = (n_anime + 7) / 8, b2 = (n_users + 7) / 8;
  if (b1 > 4096) b1 = 4096;
  if (b2 > 4096) b2 = 4096;
  hipLaunchKernelGGL(k_rownorm<1>, dim3(b1), dim3(256), 0, s, A, nullptr, n_anime, Ah);
  hipLaunchKernelGGL(k_rownorm<1>, dim3(b2), dim3(256), 0, s, U, users, n_users, Uh);
  ANIREC_HIP_CHECK(hipGetLastError());
  ScoreArgs a;
  a.Q = Uh;
  a.qrows = nullptr;
  a.nq = n_users;
  a.W = Ah;
  a.n = n_anime;
  a.out = out;
  a.ld = (size_t)n_anime;
  a.use_head = 1;
  head_affine(head, &a.hs, &a.hb);
  return launch_scores(a, s);
}

int anirec_predict_topk(const float *U, const float *A, int32_t n_anime, const int32_t *users,
                        int32_t n_users, const anirec_head *head, const uint32_t *watched,
                        int32_t k, int32_t *out_idx, float *out_p, void *workspace,
                        size_t workspace_bytes, void *stream) {
  if (!U || !A || !users || !head || !out_idx || !out_p || !workspace) return ANIREC_EINVAL;
  if (n_anime < 1 || n_users < 0 || k < 1 || k > ANIREC_MAX_TOPK) return ANIREC_EINVAL;
  if (n_users == 0) return ANIREC_OK;
  const size_t nb = norm_bytes(n_anime, n_users) + kSelTmpBytes;
  if (workspace_bytes < nb + (size_t)n_anime * 4) return ANIREC_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float *Ah = (float *)workspace;
  float *Uh = Ah + (size_t)n_anime * kDim;
  void *sel_tmp = Uh + (size_t)n_users * kDim;
  float *buf = (float *)((char *)sel_tmp + kSelTmpBytes);
  size_t qb = (workspace_bytes - nb) / ((size_t)n_anime * 4);
  if (qb > (size_t)n_users) qb = n_users;
  int b1 = (n_anime + 7) / 8, b2 = (n_users + 7) / 8;
  if (b1 > 4096) b1 = 4096;
  if (b2 > 4096) b2 = 4096;
  hipLaunchKernelGGL(k_rownorm<1>, dim3(b1), dim3(256), 0, s, A, nullptr, n_anime, Ah);
  hipLaunchKernelGGL(k_rownorm<1>, dim3(b2), dim3(256), 0, s, U, users, n_users, Uh);
  ANIREC_HIP_CHECK(hipGetLastError());
  const int wwords = (n_anime + 31) / 32;
  float hs, hb;
  head_affine(head, &hs, &hb);
  for (size_t q0 = 0; q0 < (size_t)n_users; q0 += qb) {
    const int cnt = (int)((size_t)n_users - q0 < qb ? (size_t)n_users - q0 : qb);
    ScoreArgs a;
    a.Q = Uh + q0 * kDim;
    a.qrows = nullptr;
    a.nq = cnt;
    a.W = Ah;
    a.n = n_anime;
    a.out = buf;
    a.ld = (size_t)n_anime;
    a.use_head = 1;
    a.hs = hs;
    a.hb = hb;
    int e = launch_scores(a, s);
    if (e) return e;
    SelectArgs sa;
    sa.scores = buf;
    sa.ld = (size_t)n_anime;
    sa.n = n_anime;
    sa.nq = cnt;
    sa.k = k;
    sa.self = nullptr;
    sa.keep = nullptr;
    sa.wbits = watched ? watched + q0 * wwords : nullptr;
    sa.wwords = wwords;
    sa.out_idx = out_idx + q0 * k;
    sa.out_score = out_p + q0 * k;
    e = launch_select(sa, sel_tmp, s);
    if (e) return e;
  }
  return ANIREC_OK;
}

}  // extern "C"
