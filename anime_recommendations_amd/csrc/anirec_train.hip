// Training hot path of libanirec for gfx950 (MI355X).
//
// Replaces the Keras train step driven by model.fit (reference
// neural_network/neural_network.py:210-217, graph at :66-106):
//
//   prep  one workgroup per (step, table): LDS radix sort of the batch by table row,
//         runs cut into chunks of <= ANIREC_CHUNK ratings  (TF: IndexedSlices ->
//         unsorted_segment_sum densification of the gather gradients)
//   fwd   half-wave per rating: coalesced 512-B gathers of U[ui], A[ai], three dot-128
//         reductions by wavefront shuffles -> c, sum(u^2), sum(a^2)
//   head  ONE workgroup: Dense(1) -> BatchNorm(batch stats) -> sigmoid -> BCE, closed-form
//         backward to d loss/d c, Adam on the 4 scalars, moving stats, History metrics
//   bwd   half-wave per chunk: weighted sum of the OTHER table's rows, accumulated in
//         registers in a fixed order, one coalesced 512-B store per chunk (no float atomics)
//   adam  half-wave per table row, dense: g = chunk sums - s*W + 2*l2*W, Keras-2.12 Adam,
//         emits sum(W_new^2) partials for the L2 loss term.  HBM-bound: 24 B/element
//         (+512 B per touched row) instead of 28 because the dense gradient never exists.
//
// All kernels read the step index from device memory (anirec_state::step_fwd/step_bwd)
// so one captured hipGraph replays for every step.
#include <hip/hip_runtime.h>

#include <new>

#include "anirec_dev.hpp"

namespace anirec {

// ------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------
struct TrainWs {
  int cap, capC, arena_steps;
  float *su, *sa;               // [cap] row square sums from fwd
  float *coef, *selfu, *selfa;  // [cap] backward coefficients from head
  float *regpart;               // [ANIREC_ADAM_BLOCKS]
  float *P;                     // [2*capC][128] chunk partial rows
  float *S;                     // [2*capC]      chunk self-coefficient sums
  // arena slot s: nchunks[2] (4 ints), sidx[2][cap], oth[2][cap], chunks[2][capC] (int4)
  char *arena;
  size_t slot_bytes;
  size_t total;
};

__host__ __device__ inline int chunk_capacity(int cap) {
  int c = cap + cap / ANIREC_CHUNK + 2;
  return (c + 3) & ~3;
}

__host__ inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

__host__ inline TrainWs carve(void *base, int cap, int arena_steps) {
  TrainWs w;
  w.cap = cap;
  w.capC = chunk_capacity(cap);
  w.arena_steps = arena_steps;
  size_t off = 0;
  char *b = (char *)base;
  auto take = [&](size_t bytes) {
    char *p = b + off;
    off += align_up(bytes);
    return p;
  };
  w.su = (float *)take(sizeof(float) * cap);
  w.sa = (float *)take(sizeof(float) * cap);
  w.coef = (float *)take(sizeof(float) * cap);
  w.selfu = (float *)take(sizeof(float) * cap);
  w.selfa = (float *)take(sizeof(float) * cap);
  w.regpart = (float *)take(sizeof(float) * ANIREC_ADAM_BLOCKS);
  w.P = (float *)take(sizeof(float) * 2 * (size_t)w.capC * kDim);
  w.S = (float *)take(sizeof(float) * 2 * (size_t)w.capC);
  w.slot_bytes = align_up(16) + 2 * align_up(sizeof(int32_t) * 2 * (size_t)cap) +
                 align_up(sizeof(int4) * 2 * (size_t)w.capC);
  w.arena = take(w.slot_bytes * (size_t)arena_steps);
  w.total = off;
  return w;
}

struct Slot {
  int32_t *nchunks;  // [2] (+2 pad)
  int32_t *sidx;     // [2][cap]  batch-local rating index in row-sorted order
  int32_t *oth;      // [2][cap]  global W row of the other table, same order
  int4 *chunks;      // [2][capC] {global row, start, len, nch if first chunk of row else 0}
};

__host__ __device__ inline Slot slot_of(char *arena, size_t slot_bytes, int cap, int capC, int s) {
  char *p = arena + slot_bytes * (size_t)s;
  Slot r;
  size_t a0 = 256;
  size_t a1 = ((sizeof(int32_t) * 2 * (size_t)cap) + 255) / 256 * 256;
  r.nchunks = (int32_t *)p;
  r.sidx = (int32_t *)(p + a0);
  r.oth = (int32_t *)(p + a0 + a1);
  r.chunks = (int4 *)(p + a0 + 2 * a1);
  (void)capC;
  return r;
}

// ------------------------------------------------------------------------------------
// prep: stable LSD radix sort (8-bit digits) of one batch in LDS + chunk table
// ------------------------------------------------------------------------------------
constexpr int kSortThreads = 1024;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortMax = ANIREC_MAX_BATCH;       // 16384
constexpr int kPerThread = kSortMax / kSortThreads;  // 16

// exclusive prefix sum over the block; returns this thread's offset, total via *total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum /*[16]*/,
                                                    uint32_t *total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kSortWaves; ++k) {
    uint32_t s = wsum[k];
    if (k < w) base += s;
    tot += s;
  }
  if (total) *total = tot;
  return base + inc - v;
}

struct PrepArgs {
  const int32_t *user_idx, *anime_idx;
  const anirec_step *sched;
  int first_step;
  int n_user_rows, n_anime_rows;
  int cap, capC, arena_steps;
  char *arena;
  size_t slot_bytes;
};

__global__ __launch_bounds__(kSortThreads) void k_prep(PrepArgs a) {
  __shared__ uint32_t keys[kSortMax];             // 64 KB
  __shared__ uint16_t vals[kSortMax];             // 32 KB (later: segment heads)
  __shared__ __attribute__((aligned(16))) uint32_t hist[256 * kSortWaves];  // 16 KB
  __shared__ uint32_t wsum[kSortWaves];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int step = a.first_step + (blockIdx.x >> 1);
  const int T = blockIdx.x & 1;  // 0: sort by user row, 1: by anime row
  const anirec_step sc = a.sched[step];
  const int nb = min(sc.count, a.cap);
  const int base = sc.start;
  const int32_t *ksrc = T == 0 ? a.user_idx : a.anime_idx;
  const int32_t *osrc = T == 0 ? a.anime_idx : a.user_idx;
  const int nrows = T == 0 ? a.n_user_rows : a.n_anime_rows;
  const int row_off = T == 0 ? 0 : a.n_user_rows;     // global W row of key
  const int oth_off = T == 0 ? a.n_user_rows : 0;     // global W row of other index
  Slot sl = slot_of(a.arena, a.slot_bytes, a.cap, a.capC, step % a.arena_steps);

  // elements are dealt to waves in contiguous blocks so (wave, iteration, lane) order
  // == position order (needed for stability)
  const int npad = (nb + kSortThreads - 1) / kSortThreads * kSortThreads;
  const int per_wave = npad / kSortWaves;  // multiple of 64
  const int iters = per_wave / 64;         // <= 16

  for (int e = tid; e < npad; e += kSortThreads) {
    keys[e] = e < nb ? (uint32_t)ksrc[base + e] : 0xFFFFFFFFu;
    vals[e] = (uint16_t)e;
  }
  __syncthreads();

  int bits = 1;
  while (bits < 32 && (1u << bits) < (uint32_t)nrows) ++bits;
  const int npass = (bits + 7) / 8;

  volatile uint32_t *vh = hist;
  for (int pass = 0; pass < npass; ++pass) {
    const int shift = pass * 8;
    for (int i = tid; i < 256 * kSortWaves; i += kSortThreads) hist[i] = 0;
    __syncthreads();
    uint32_t k[kPerThread];
    uint16_t v[kPerThread];
    uint16_t off[kPerThread];
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) {
      if (it < iters) {
        const int e = w * per_wave + it * 64 + lane;
        k[it] = keys[e];
        v[it] = vals[e];
        const uint32_t d = (k[it] >> shift) & 255u;
        // lanes of this wave holding the same digit
        unsigned long long peers = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const bool bit = (d >> b) & 1u;
          const unsigned long long bal = __ballot(bit);
          peers &= bit ? bal : ~bal;
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        const uint32_t rank = __popcll(peers & lt);
        const uint32_t cnt = __popcll(peers);
        const uint32_t old = vh[d * kSortWaves + w];  // all peers read the same word
        if (rank == 0) vh[d * kSortWaves + w] = old + cnt;
        off[it] = (uint16_t)(old + rank);
      }
    }
    __syncthreads();
    // exclusive scan of hist in (digit-major, wave-minor) order: 4 entries per thread
    {
      uint4 h = *reinterpret_cast<uint4 *>(&hist[tid * 4]);
      uint32_t s = h.x + h.y + h.z + h.w;
      uint32_t b0 = block_excl_scan(s, wsum, nullptr);
      uint4 o;
      o.x = b0;
      o.y = b0 + h.x;
      o.z = o.y + h.y;
      o.w = o.z + h.z;
      *reinterpret_cast<uint4 *>(&hist[tid * 4]) = o;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) {
      if (it < iters) {
        const uint32_t d = (k[it] >> shift) & 255u;
        const uint32_t dst = hist[d * kSortWaves + w] + off[it];
        keys[dst] = k[it];
        vals[dst] = v[it];
      }
    }
    __syncthreads();
  }

  // sorted order out: rating index and the other table's global row
  for (int p = tid; p < nb; p += kSortThreads) {
    const int v = vals[p];
    sl.sidx[T * a.cap + p] = v;
    sl.oth[T * a.cap + p] = osrc[base + v] + oth_off;
  }
  __syncthreads();

  // segment heads (first position of each distinct row), compacted into heads[]
  uint16_t *heads = vals;
  uint32_t nseg = 0;
  {
    const int p0 = tid * kPerThread;
    uint32_t flags = 0, cnt = 0;
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
      const int p = p0 + q;
      if (p < nb) {
        const bool h = (p == 0) || (keys[p] != keys[p - 1]);
        if (h) {
          flags |= 1u << q;
          ++cnt;
        }
      }
    }
    uint32_t o = block_excl_scan(cnt, wsum, &nseg);
    __syncthreads();  // everyone has read vals/keys neighbours before heads overwrite vals
#pragma unroll
    for (int q = 0; q < kPerThread; ++q)
      if (flags & (1u << q)) heads[o++] = (uint16_t)(p0 + q);
  }
  __syncthreads();

  // chunks per segment -> exclusive scan -> emit
  {
    const int h0 = tid * kPerThread;
    uint32_t nch[kPerThread];
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
      const uint32_t h = h0 + q;
      nch[q] = 0;
      if (h < nseg) {
        const int s0 = heads[h];
        const int s1 = (h + 1 < nseg) ? heads[h + 1] : nb;
        nch[q] = (uint32_t)((s1 - s0 + ANIREC_CHUNK - 1) / ANIREC_CHUNK);
        cnt += nch[q];
      }
    }
    uint32_t total = 0;
    uint32_t cb = block_excl_scan(cnt, wsum, &total);
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
      const uint32_t h = h0 + q;
      if (h < nseg) {
        const int s0 = heads[h];
        const int s1 = (h + 1 < nseg) ? heads[h + 1] : nb;
        const int row = (int)keys[s0] + row_off;
        for (uint32_t j = 0; j < nch[q]; ++j) {
          const int st = s0 + (int)j * ANIREC_CHUNK;
          int4 rec;
          rec.x = row;
          rec.y = st;
          rec.z = min(ANIREC_CHUNK, s1 - st);
          rec.w = j == 0 ? (int)nch[q] : 0;
          sl.chunks[T * a.capC + cb + j] = rec;
        }
        cb += nch[q];
      }
    }
    if (tid == 0) sl.nchunks[T] = (int)total;
  }
}

// ------------------------------------------------------------------------------------
// fwd: embedding lookup + L2-normalised dot
// ------------------------------------------------------------------------------------
struct FwdArgs {
  const float *W;
  int n_user_rows;
  const int32_t *user_idx, *anime_idx;
  const float *rating;
  const anirec_step *sched;
  const anirec_state *state;
  float *pk_c, *pk_t;
  int32_t *pk_count;
  float *su, *sa;
  int cap;
};

// One half-wave per rating.  Returns dot products through references; all 32 lanes of
// the half hold the totals.
__device__ __forceinline__ void pair_dots(const float4 *W4, int urow, int arow, int l32, float &su,
                                          float &sa, float &dd) {
  const float4 u = W4[(size_t)urow * kRowVec + l32];
  const float4 a = W4[(size_t)arow * kRowVec + l32];
  float s0 = u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
  float s1 = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  float s2 = u.x * a.x + u.y * a.y + u.z * a.z + u.w * a.w;
  su = halfwave_sum(s0);
  sa = halfwave_sum(s1);
  dd = halfwave_sum(s2);
}

__device__ __forceinline__ float cos_from_dots(float su, float sa, float dd) {
  const float ru = 1.0f / sqrtf(fmaxf(su, kL2nEps));
  const float ra = 1.0f / sqrtf(fmaxf(sa, kL2nEps));
  return dd * ru * ra;
}

__global__ __launch_bounds__(256) void k_fwd(FwdArgs a) {
  const anirec_step sc = a.sched[a.state->step_fwd];
  const int nb = min(sc.count, a.cap);
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
  if (blockIdx.x == 0 && threadIdx.x == 0) a.pk_count[0] = nb;
  if (i >= nb) return;
  const int l32 = threadIdx.x & 31;
  const int g = sc.start + i;
  const int ur = a.user_idx[g];
  const int ar = a.anime_idx[g] + a.n_user_rows;
  float su, sa, dd;
  pair_dots(reinterpret_cast<const float4 *>(a.W), ur, ar, l32, su, sa, dd);
  if (l32 == 0) {
    a.pk_c[i] = cos_from_dots(su, sa, dd);
    a.pk_t[i] = a.rating[g];
    a.su[i] = su;
    a.sa[i] = sa;
  }
}

// ------------------------------------------------------------------------------------
// head: everything that couples the batch (one workgroup)
// ------------------------------------------------------------------------------------
struct HeadArgs {
  anirec_state *state;
  const anirec_step *sched;
  const float *packets;  // n_seg packets
  size_t packet_floats;
  int n_seg, my_seg, cap;
  const float *su, *sa;  // local
  float *coef, *selfu, *selfa;
  const float *regpart;
  float l2;
};

__device__ __forceinline__ float bce_logits(float y, float t) {
  return fmaxf(y, 0.f) - y * t + log1pf(expf(-fabsf(y)));
}

__global__ __launch_bounds__(1024) void k_head(HeadArgs a) {
  __shared__ float scratch[4 * 16];
  const int tid = threadIdx.x;
  anirec_state *st = a.state;
  const int step = st->step_fwd;
  const anirec_step sc = a.sched[step];
  const float w = st->w, b = st->b, gamma = st->gamma, beta = st->beta;

  // L2 term: sum(W^2) of the weights this step reads (partials left by adam / init_reg)
  float reg;
  {
    float r[1] = {0.f};
    for (int i = tid; i < ANIREC_ADAM_BLOCKS; i += blockDim.x) r[0] += a.regpart[i];
    block_sum<1>(r, scratch);
    reg = r[0];
  }

  int n_total = 0;
  for (int s = 0; s < a.n_seg; ++s) {
    const float *pk = a.packets + a.packet_floats * s;
    n_total += min(reinterpret_cast<const int32_t *>(pk + 2 * (size_t)a.cap)[0], a.cap);
  }
  const float Bf = (float)n_total;

  // pass 1: mean of z
  float mu;
  {
    float r[1] = {0.f};
    for (int s = 0; s < a.n_seg; ++s) {
      const float *pk = a.packets + a.packet_floats * s;
      const int cnt = min(reinterpret_cast<const int32_t *>(pk + 2 * (size_t)a.cap)[0], a.cap);
      for (int i = tid; i < cnt; i += blockDim.x) r[0] += pk[i] * w + b;
    }
    block_sum<1>(r, scratch);
    mu = r[0] / Bf;
  }
  // pass 2: biased variance (tf.nn.moments)
  float var;
  {
    float r[1] = {0.f};
    for (int s = 0; s < a.n_seg; ++s) {
      const float *pk = a.packets + a.packet_floats * s;
      const int cnt = min(reinterpret_cast<const int32_t *>(pk + 2 * (size_t)a.cap)[0], a.cap);
      for (int i = tid; i < cnt; i += blockDim.x) {
        const float d = (pk[i] * w + b) - mu;
        r[0] += d * d;
      }
    }
    block_sum<1>(r, scratch);
    var = r[0] / Bf;
  }
  const float rs = 1.0f / sqrtf(var + kBnEps);
  const float inv = rs * gamma;
  const float shift = beta - mu * inv;

  // pass 3: sigmoid, loss, first-level sums
  float S1, S2, Lsum, SE;
  {
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < a.n_seg; ++s) {
      const float *pk = a.packets + a.packet_floats * s;
      const float *pt = pk + a.cap;
      const int cnt = min(reinterpret_cast<const int32_t *>(pk + 2 * (size_t)a.cap)[0], a.cap);
      for (int i = tid; i < cnt; i += blockDim.x) {
        const float z = pk[i] * w + b;
        const float t = pt[i];
        const float y = z * inv + shift;
        const float p = sigmoidf_stable(y);
        const float dy = (p - t) / Bf;
        const float zh = (z - mu) * rs;
        r[0] += dy;
        r[1] += dy * zh;
        r[2] += bce_logits(y, t);
        r[3] += (p - t) * (p - t);
      }
    }
    block_sum<4>(r, scratch);
    S1 = r[0];
    S2 = r[1];
    Lsum = r[2];
    SE = r[3];
  }
  const float m1 = gamma * S1 / Bf;  // mean(d zhat)
  const float m2 = gamma * S2 / Bf;  // mean(d zhat * zhat)

  // pass 4: dz, dw, db; coefficients for this rank's ratings
  float dW, dB;
  {
    float r[2] = {0.f, 0.f};
    for (int s = 0; s < a.n_seg; ++s) {
      const float *pk = a.packets + a.packet_floats * s;
      const float *pt = pk + a.cap;
      const int cnt = min(reinterpret_cast<const int32_t *>(pk + 2 * (size_t)a.cap)[0], a.cap);
      for (int i = tid; i < cnt; i += blockDim.x) {
        const float c = pk[i];
        const float z = c * w + b;
        const float t = pt[i];
        const float y = z * inv + shift;
        const float p = sigmoidf_stable(y);
        const float dy = (p - t) / Bf;
        const float zh = (z - mu) * rs;
        const float dz = (dy * gamma - m1 - zh * m2) * rs;
        r[0] += dz * c;
        r[1] += dz;
        if (s == a.my_seg) {
          const float dc = dz * w;
          const float su = a.su[i], sa = a.sa[i];
          const float ru = 1.0f / sqrtf(fmaxf(su, kL2nEps));
          const float ra = 1.0f / sqrtf(fmaxf(sa, kL2nEps));
          a.coef[i] = dc * ru * ra;
          a.selfu[i] = su >= kL2nEps ? dc * c * ru * ru : 0.f;
          a.selfa[i] = sa >= kL2nEps ? dc * c * ra * ra : 0.f;
        }
      }
    }
    block_sum<2>(r, scratch);
    dW = r[0];
    dB = r[1];
  }

  if (tid == 0) {
    const float alpha = sc.alpha;
    float p4[4] = {w, b, gamma, beta};
    const float g4[4] = {dW, dB, S2, S1};  // d w, d b, d gamma, d beta
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float m = st->adam_m[k], v = st->adam_v[k];
      adam_elem(p4[k], m, v, g4[k], alpha);
      st->adam_m[k] = m;
      st->adam_v[k] = v;
    }
    st->w = p4[0];
    st->b = p4[1];
    st->gamma = p4[2];
    st->beta = p4[3];
    st->mov_mean = st->mov_mean - (st->mov_mean - mu) * kBnDecay;
    st->mov_var = st->mov_var - (st->mov_var - var) * kBnDecay;
    st->reg_sumsq = reg;
    st->bn_mu = mu;
    st->bn_var = var;
    const float loss = Lsum / Bf + a.l2 * reg;
    st->last_loss = loss;
    st->last_mse = SE / Bf;
    st->loss_wsum += (double)loss * (double)n_total;
    st->se_sum += (double)SE;
    st->n_seen += (double)n_total;
    st->step_bwd = step;
    st->step_fwd = step + 1;
  }
}

// ------------------------------------------------------------------------------------
// bwd: per-chunk weighted row sums
// ------------------------------------------------------------------------------------
struct BwdArgs {
  const float *W;
  const anirec_state *state;
  char *arena;
  size_t slot_bytes;
  int cap, capC, arena_steps;
  const float *coef, *selfu, *selfa;
  float *P, *S;
  int32_t *rowmap;
};

__global__ __launch_bounds__(256) void k_bwd(BwdArgs a) {
  const int hw = blockIdx.x * 8 + (threadIdx.x >> 5);
  const int T = hw >= a.capC ? 1 : 0;
  const int c = hw - T * a.capC;
  const int step = a.state->step_bwd;
  Slot sl = slot_of(a.arena, a.slot_bytes, a.cap, a.capC, step % a.arena_steps);
  if (c >= sl.nchunks[T]) return;
  const int l = threadIdx.x & 31;
  const int4 rec = sl.chunks[T * a.capC + c];
  const int len = rec.z;
  // lane j < len holds contribution j of the chunk; the rest replicate the last one with
  // weight 0 so every shuffle source is a valid row
  const int pos = rec.y + min(l, len - 1);
  const int i = sl.sidx[T * a.cap + pos];
  const int o = sl.oth[T * a.cap + pos];
  float cf = a.coef[i];
  float sf = (T == 0 ? a.selfu : a.selfa)[i];
  if (l >= len) {
    cf = 0.f;
    sf = 0.f;
  }
  const float4 *W4 = reinterpret_cast<const float4 *>(a.W);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = 0; j < len; j += 4) {
    int oj[4];
    float cj[4];
    float4 r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      oj[q] = __shfl(o, j + q, 32);
      cj[q] = __shfl(cf, j + q, 32);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = W4[(size_t)oj[q] * kRowVec + l];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc.x += cj[q] * r[q].x;
      acc.y += cj[q] * r[q].y;
      acc.z += cj[q] * r[q].z;
      acc.w += cj[q] * r[q].w;
    }
  }
  const float ssum = halfwave_sum(sf);
  const int gc = T * a.capC + c;
  reinterpret_cast<float4 *>(a.P)[(size_t)gc * kRowVec + l] = acc;
  if (l == 0) {
    a.S[gc] = ssum;
    if (rec.w > 0) a.rowmap[rec.x] = ((gc << 10) | (rec.w - 1)) + 1;
  }
}

// densify the anime gradient for the RCCL all-reduce (multi-GPU): anime_grad[row] =
// sum of the row's chunk partials, trailing n_anime floats = self-coefficient sums.
struct DensifyArgs {
  int n_user_rows, n_anime_rows;
  int32_t *rowmap;
  const float *P, *S;
  float *anime_grad;
};

__global__ __launch_bounds__(256) void k_densify(DensifyArgs a) {
  const int l = threadIdx.x & 31;
  const int nhw = gridDim.x * 8;
  for (int r = blockIdx.x * 8 + (threadIdx.x >> 5); r < a.n_anime_rows; r += nhw) {
    const int rm = a.rowmap[a.n_user_rows + r];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    if (rm) {
      const int first = (rm - 1) >> 10, nch = ((rm - 1) & 1023) + 1;
      for (int c = first; c < first + nch; ++c) {
        const float4 p = reinterpret_cast<const float4 *>(a.P)[(size_t)c * kRowVec + l];
        g.x += p.x;
        g.y += p.y;
        g.z += p.z;
        g.w += p.w;
        s += a.S[c];
      }
      if (l == 0) a.rowmap[a.n_user_rows + r] = 0;
    }
    reinterpret_cast<float4 *>(a.anime_grad)[(size_t)r * kRowVec + l] = g;
    if (l == 0) a.anime_grad[(size_t)a.n_anime_rows * kDim + r] = s;
  }
}

// ------------------------------------------------------------------------------------
// adam: dense fused update of every table row
// ------------------------------------------------------------------------------------
struct AdamArgs {
  float *W, *M, *V;
  int n_rows, n_user_rows, n_anime_rows;
  int32_t *rowmap;
  const float *P, *S;
  float *anime_grad;  // non-null: anime rows take their gradient from here (already reduced)
  const anirec_state *state;
  const anirec_step *sched;
  float two_l2;
  float *regpart;
  float fixed_alpha;  // used when sched == nullptr (init_reg passes alpha = 0 path separately)
};

template <bool kUpdate>
__global__ __launch_bounds__(256) void k_adam(AdamArgs a) {
  __shared__ float scratch[16];
  const int l = threadIdx.x & 31;
  const int nhw = gridDim.x * 8;
  float alpha = 0.f;
  if (kUpdate) alpha = a.sched[a.state->step_bwd].alpha;
  float4 *W4 = reinterpret_cast<float4 *>(a.W);
  float4 *M4 = reinterpret_cast<float4 *>(a.M);
  float4 *V4 = reinterpret_cast<float4 *>(a.V);
  const float4 *P4 = reinterpret_cast<const float4 *>(a.P);
  float sq = 0.f;
  for (int r = blockIdx.x * 8 + (threadIdx.x >> 5); r < a.n_rows; r += nhw) {
    const size_t e = (size_t)r * kRowVec + l;
    float4 w = W4[e];
    if (kUpdate) {
      float4 m = M4[e];
      float4 v = V4[e];
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      float s = 0.f;
      if (a.anime_grad != nullptr && r >= a.n_user_rows) {
        const int ar = r - a.n_user_rows;
        float4 *G4 = reinterpret_cast<float4 *>(a.anime_grad);
        g = G4[(size_t)ar * kRowVec + l];
        s = a.anime_grad[(size_t)a.n_anime_rows * kDim + ar];
      } else {
        const int rm = a.rowmap[r];
        if (rm) {
          const int first = (rm - 1) >> 10, nch = ((rm - 1) & 1023) + 1;
          for (int c = first; c < first + nch; ++c) {
            const float4 p = P4[(size_t)c * kRowVec + l];
            g.x += p.x;
            g.y += p.y;
            g.z += p.z;
            g.w += p.w;
            s += a.S[c];
          }
          if (l == 0) a.rowmap[r] = 0;
        }
      }
      g.x = grad_total(g.x, s, w.x, a.two_l2);
      g.y = grad_total(g.y, s, w.y, a.two_l2);
      g.z = grad_total(g.z, s, w.z, a.two_l2);
      g.w = grad_total(g.w, s, w.w, a.two_l2);
      adam_elem(w.x, m.x, v.x, g.x, alpha);
      adam_elem(w.y, m.y, v.y, g.y, alpha);
      adam_elem(w.z, m.z, v.z, g.z, alpha);
      adam_elem(w.w, m.w, v.w, g.w, alpha);
      W4[e] = w;
      M4[e] = m;
      V4[e] = v;
    }
    sq += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
  }
  // block partial of sum(W_new^2), fixed order
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) a.regpart[blockIdx.x] = scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// flat Adam with an explicit gradient (unit-testable bit-exact stage)
__global__ __launch_bounds__(256) void k_adam_flat(float *w, float *m, float *v, const float *g,
                                                   size_t n, float alpha) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float ww = w[i], mm = m[i], vv = v[i];
    adam_elem(ww, mm, vv, g[i], alpha);
    w[i] = ww;
    m[i] = mm;
    v[i] = vv;
  }
}

// ------------------------------------------------------------------------------------
// validation (BN inference mode) and misc
// ------------------------------------------------------------------------------------
struct EvalArgs {
  const float *W;
  int n_user_rows;
  const int32_t *user_idx, *anime_idx;
  const float *rating;
  int n;
  anirec_state *state;
};

__global__ __launch_bounds__(256) void k_eval(EvalArgs a) {
  __shared__ float sh[2][8];
  const anirec_state *st = a.state;
  const float w = st->w, b = st->b;
  const float inv = st->gamma * (1.0f / sqrtf(st->mov_var + kBnEps));
  const float shift = st->beta - st->mov_mean * inv;
  const int l32 = threadIdx.x & 31, h = threadIdx.x >> 5;
  const int i = blockIdx.x * 8 + h;
  float li = 0.f, se = 0.f;
  if (i < a.n) {
    float su, sa, dd;
    pair_dots(reinterpret_cast<const float4 *>(a.W), a.user_idx[i], a.anime_idx[i] + a.n_user_rows,
              l32, su, sa, dd);
    const float c = cos_from_dots(su, sa, dd);
    const float y = (c * w + b) * inv + shift;
    const float p = sigmoidf_stable(y);
    const float t = a.rating[i];
    li = bce_logits(y, t);
    se = (p - t) * (p - t);
  }
  if (l32 == 0) {
    sh[0][h] = li;
    sh[1][h] = se;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double L = 0., E = 0.;
    for (int k = 0; k < 8; ++k) {
      L += sh[0][k];
      E += sh[1][k];
    }
    // fp64 atomics: order-dependent only below 1e-16 relative, far under the fp32 History values
    atomicAdd(&a.state->val_bce_sum, L);
    atomicAdd(&a.state->val_se_sum, E);
    if (blockIdx.x == 0) atomicAdd(&a.state->val_n, (double)a.n);
  }
}

__global__ __launch_bounds__(1024) void k_sum_regpart(anirec_state *st, const float *regpart) {
  __shared__ float scratch[16];
  float r[1] = {0.f};
  for (int i = threadIdx.x; i < ANIREC_ADAM_BLOCKS; i += blockDim.x) r[0] += regpart[i];
  block_sum<1>(r, scratch);
  if (threadIdx.x == 0) st->reg_sumsq = r[0];
}

__global__ __launch_bounds__(256) void k_gather_ratings(const int32_t *ui, const int32_t *ai,
                                                        const float *t, const int64_t *perm,
                                                        size_t n, int32_t *uo, int32_t *ao,
                                                        float *to) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t p = perm[i];
    uo[i] = ui[p];
    ao[i] = ai[p];
    to[i] = t[p];
  }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
static int check_desc(const anirec_train_desc *d) {
  if (!d || !d->W || !d->M || !d->V || !d->rowmap || !d->state || !d->workspace || !d->packets)
    return ANIREC_EINVAL;
  if (d->max_batch < 1 || d->max_batch > ANIREC_MAX_BATCH) return ANIREC_EINVAL;
  if (d->n_user_rows < 1 || d->n_anime_rows < 1 || d->arena_steps < 1) return ANIREC_EINVAL;
  if (d->n_seg < 1 || d->my_seg < 0 || d->my_seg >= d->n_seg) return ANIREC_EINVAL;
  if (d->anime_dense && !d->anime_grad) return ANIREC_EINVAL;
  if (d->workspace_bytes < anirec_train_workspace_bytes(d->max_batch, d->arena_steps))
    return ANIREC_EWORKSPACE;
  return ANIREC_OK;
}

static inline float *packet_ptr(const anirec_train_desc *d, int seg) {
  return d->packets + anirec_packet_floats(d->max_batch) * (size_t)seg;
}

static int launch_fwd(const anirec_train_desc *d, const TrainWs &w, hipStream_t s) {
  FwdArgs a;
  a.W = d->W;
  a.n_user_rows = d->n_user_rows;
  a.user_idx = d->user_idx;
  a.anime_idx = d->anime_idx;
  a.rating = d->rating;
  a.sched = d->sched;
  a.state = d->state;
  float *pk = packet_ptr(d, d->my_seg);
  a.pk_c = pk;
  a.pk_t = pk + d->max_batch;
  a.pk_count = reinterpret_cast<int32_t *>(pk + 2 * (size_t)d->max_batch);
  a.su = w.su;
  a.sa = w.sa;
  a.cap = d->max_batch;
  hipLaunchKernelGGL(k_fwd, dim3((d->max_batch + 7) / 8), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

static int launch_head(const anirec_train_desc *d, const TrainWs &w, hipStream_t s) {
  HeadArgs a;
  a.state = d->state;
  a.sched = d->sched;
  a.packets = d->packets;
  a.packet_floats = anirec_packet_floats(d->max_batch);
  a.n_seg = d->n_seg;
  a.my_seg = d->my_seg;
  a.cap = d->max_batch;
  a.su = w.su;
  a.sa = w.sa;
  a.coef = w.coef;
  a.selfu = w.selfu;
  a.selfa = w.selfa;
  a.regpart = w.regpart;
  a.l2 = d->l2;
  hipLaunchKernelGGL(k_head, dim3(1), dim3(1024), 0, s, a);
  return (int)hipGetLastError();
}

static int launch_bwd(const anirec_train_desc *d, const TrainWs &w, hipStream_t s) {
  BwdArgs a;
  a.W = d->W;
  a.state = d->state;
  a.arena = w.arena;
  a.slot_bytes = w.slot_bytes;
  a.cap = w.cap;
  a.capC = w.capC;
  a.arena_steps = w.arena_steps;
  a.coef = w.coef;
  a.selfu = w.selfu;
  a.selfa = w.selfa;
  a.P = w.P;
  a.S = w.S;
  a.rowmap = d->rowmap;
  hipLaunchKernelGGL(k_bwd, dim3((2 * w.capC + 7) / 8), dim3(256), 0, s, a);
  int e = (int)hipGetLastError();
  if (e) return e;
  if (d->anime_dense) {
    DensifyArgs g;
    g.n_user_rows = d->n_user_rows;
    g.n_anime_rows = d->n_anime_rows;
    g.rowmap = d->rowmap;
    g.P = w.P;
    g.S = w.S;
    g.anime_grad = d->anime_grad;
    int blocks = (d->n_anime_rows + 7) / 8;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_densify, dim3(blocks), dim3(256), 0, s, g);
    e = (int)hipGetLastError();
  }
  return e;
}

static AdamArgs adam_args(const anirec_train_desc *d, const TrainWs &w) {
  AdamArgs a;
  a.W = d->W;
  a.M = d->M;
  a.V = d->V;
  a.n_rows = d->n_user_rows + d->n_anime_rows;
  a.n_user_rows = d->n_user_rows;
  a.n_anime_rows = d->n_anime_rows;
  a.rowmap = d->rowmap;
  a.P = w.P;
  a.S = w.S;
  a.anime_grad = d->anime_dense ? d->anime_grad : nullptr;
  a.state = d->state;
  a.sched = d->sched;
  a.two_l2 = 2.0f * d->l2;
  a.regpart = w.regpart;
  a.fixed_alpha = 0.f;
  return a;
}

static int launch_adam(const anirec_train_desc *d, const TrainWs &w, hipStream_t s) {
  AdamArgs a = adam_args(d, w);
  hipLaunchKernelGGL(k_adam<true>, dim3(ANIREC_ADAM_BLOCKS), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

}  // namespace anirec

using namespace anirec;

extern "C" {

size_t anirec_packet_floats(int32_t max_batch) { return 2 * (size_t)max_batch + 4; }

size_t anirec_train_workspace_bytes(int32_t max_batch, int32_t arena_steps) {
  if (max_batch < 1 || max_batch > ANIREC_MAX_BATCH || arena_steps < 1) return 0;
  return carve(nullptr, max_batch, arena_steps).total;
}

int anirec_train_init_reg(const anirec_train_desc *d, void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  TrainWs w = carve(d->workspace, d->max_batch, d->arena_steps);
  AdamArgs a = adam_args(d, w);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_adam<false>, dim3(ANIREC_ADAM_BLOCKS), dim3(256), 0, s, a);
  ANIREC_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_sum_regpart, dim3(1), dim3(1024), 0, s, d->state, w.regpart);
  return (int)hipGetLastError();
}

int anirec_train_prep(const anirec_train_desc *d, int32_t first_step, int32_t n_steps,
                      void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!d->user_idx || !d->anime_idx || !d->sched) return ANIREC_EINVAL;
  if (first_step < 0 || n_steps < 0 || first_step + n_steps > d->n_steps ||
      n_steps > d->arena_steps)
    return ANIREC_EINVAL;
  if (n_steps == 0) return ANIREC_OK;
  TrainWs w = carve(d->workspace, d->max_batch, d->arena_steps);
  PrepArgs a;
  a.user_idx = d->user_idx;
  a.anime_idx = d->anime_idx;
  a.sched = d->sched;
  a.first_step = first_step;
  a.n_user_rows = d->n_user_rows;
  a.n_anime_rows = d->n_anime_rows;
  a.cap = w.cap;
  a.capC = w.capC;
  a.arena_steps = w.arena_steps;
  a.arena = w.arena;
  a.slot_bytes = w.slot_bytes;
  hipLaunchKernelGGL(k_prep, dim3(2 * n_steps), dim3(kSortThreads), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int anirec_train_fwd(const anirec_train_desc *d, void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!d->user_idx || !d->anime_idx || !d->rating || !d->sched) return ANIREC_EINVAL;
  return launch_fwd(d, carve(d->workspace, d->max_batch, d->arena_steps), (hipStream_t)stream);
}

int anirec_train_head(const anirec_train_desc *d, void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!d->sched) return ANIREC_EINVAL;
  return launch_head(d, carve(d->workspace, d->max_batch, d->arena_steps), (hipStream_t)stream);
}

int anirec_train_bwd(const anirec_train_desc *d, void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  return launch_bwd(d, carve(d->workspace, d->max_batch, d->arena_steps), (hipStream_t)stream);
}

int anirec_train_adam(const anirec_train_desc *d, void *stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  if (!d->sched) return ANIREC_EINVAL;
  return launch_adam(d, carve(d->workspace, d->max_batch, d->arena_steps), (hipStream_t)stream);
}

struct anirec_trainer {
  anirec_train_desc d;
  TrainWs ws;
  hipGraphExec_t exec;
  int graph_steps;
};

int anirec_trainer_create(const anirec_train_desc *d, anirec_trainer **out) {
  if (!out) return ANIREC_EINVAL;
  int rc = check_desc(d);
  if (rc) return rc;
  if (d->n_seg != 1 || d->anime_dense) return ANIREC_EINVAL;  // multi-GPU drives the stages itself
  anirec_trainer *t = new (std::nothrow) anirec_trainer;
  if (!t) return ANIREC_EINVAL;
  t->d = *d;
  t->ws = carve(d->workspace, d->max_batch, d->arena_steps);
  t->exec = nullptr;
  t->graph_steps = 0;
  *out = t;
  return ANIREC_OK;
}

int anirec_trainer_destroy(anirec_trainer *t) {
  if (!t) return ANIREC_EINVAL;
  if (t->exec) (void)hipGraphExecDestroy(t->exec);
  delete t;
  return ANIREC_OK;
}

static int one_step(anirec_trainer *t, hipStream_t s) {
  int e;
  if ((e = launch_fwd(&t->d, t->ws, s))) return e;
  if ((e = launch_head(&t->d, t->ws, s))) return e;
  if ((e = launch_bwd(&t->d, t->ws, s))) return e;
  return launch_adam(&t->d, t->ws, s);
}

int anirec_trainer_run(anirec_trainer *t, int32_t n_steps, int32_t use_graph, void *stream) {
  if (!t || n_steps < 0) return ANIREC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  constexpr int kGraphSteps = 16;
  int done = 0;
  if (use_graph && s != nullptr && n_steps >= kGraphSteps) {
    if (!t->exec) {
      hipGraph_t g = nullptr;
      if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess)
        return ANIREC_ECAPTURE;
      int e = 0;
      for (int i = 0; i < kGraphSteps && !e; ++i) e = one_step(t, s);
      hipError_t ce = hipStreamEndCapture(s, &g);
      if (e || ce != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        return e ? e : ANIREC_ECAPTURE;
      }
      hipError_t ie = hipGraphInstantiate(&t->exec, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      if (ie != hipSuccess) {
        t->exec = nullptr;
        return ANIREC_ECAPTURE;
      }
      t->graph_steps = kGraphSteps;
    }
    while (n_steps - done >= t->graph_steps) {
      ANIREC_HIP_CHECK(hipGraphLaunch(t->exec, s));
      done += t->graph_steps;
    }
  }
  for (; done < n_steps; ++done) {
    int e = one_step(t, s);
    if (e) return e;
  }
  return ANIREC_OK;
}

int anirec_eval(const anirec_train_desc *d, const int32_t *user_idx, const int32_t *anime_idx,
                const float *rating, int32_t n, void *stream) {
  if (!d || !d->W || !d->state || !user_idx || !anime_idx || !rating || n < 0)
    return ANIREC_EINVAL;
  if (n == 0) return ANIREC_OK;
  EvalArgs a;
  a.W = d->W;
  a.n_user_rows = d->n_user_rows;
  a.user_idx = user_idx;
  a.anime_idx = anime_idx;
  a.rating = rating;
  a.n = n;
  a.state = d->state;
  hipLaunchKernelGGL(k_eval, dim3((n + 7) / 8), dim3(256), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

int anirec_adam_flat(float *w, float *m, float *v, const float *g, size_t n, float alpha,
                     void *stream) {
  if (!w || !m || !v || !g) return ANIREC_EINVAL;
  if (n == 0) return ANIREC_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_adam_flat, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, m,
                     v, g, n, alpha);
  return (int)hipGetLastError();
}

int anirec_gather_ratings(const int32_t *user_in, const int32_t *anime_in, const float *rating_in,
                          const int64_t *perm, size_t n, int32_t *user_out, int32_t *anime_out,
                          float *rating_out, void *stream) {
  if (!user_in || !anime_in || !rating_in || !perm || !user_out || !anime_out || !rating_out)
    return ANIREC_EINVAL;
  if (n == 0) return ANIREC_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_gather_ratings, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     user_in, anime_in, rating_in, perm, n, user_out, anime_out, rating_out);
  return (int)hipGetLastError();
}

}  // extern "C"
