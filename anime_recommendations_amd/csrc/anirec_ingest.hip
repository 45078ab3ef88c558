// Rating-table ingest on the GPU: the step immediately before the hot path.
//
// Replaces, for columns already resident in HBM,
//   preprocess/preprocess.py:13-40   drop_useless   (drop_duplicates, dropna, watched/plan filters,
//                                                    users with fewer than num_reviews ratings)
//   preprocess/preprocess.py:52-105  drop_half_watched (per-anime max episodes, keep >= half)
//   preprocess/preprocess.py:108-117 scale_ratings  ((x - min) / (max - min), float64)
//   neural_network/neural_network.py:41-60 get_df   (id -> position in Series.unique(): order of
//                                                    first appearance)
// which the reference does with pandas (dict lookups and Python loops per row).  All of it is
// HBM-bound integer / byte work: flag passes, one open-addressing hash table for the duplicate rows,
// direct-index tables for the per-user / per-anime aggregates, and a 3-pass flag scan that turns
// flags into stable (order-preserving) output positions.  Results are bit-identical to pandas:
// the surviving rows keep their order, duplicates keep their FIRST occurrence, indices follow first
// appearance, and the scaling is the same IEEE double expression.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "anirec_dev.hpp"

namespace anirec {

constexpr int kScanTile = 4096;  // flags per workgroup of the scan (256 threads x 16 flags)

struct IngestCols {
  const int32_t *user, *anime;
  const double *rating;
  const int32_t *status, *episodes;
};

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
// pandas hashes / compares 0.0 and -0.0 as equal; NaN rows never reach the table
__device__ __forceinline__ uint64_t rating_bits(double r) { return (uint64_t)__double_as_longlong(r + 0.0); }

__device__ __forceinline__ uint64_t row_hash(const IngestCols &c, int64_t i) {
  uint64_t h = mix64(((uint64_t)(uint32_t)c.user[i] << 32) | (uint32_t)c.anime[i]);
  h = mix64(h ^ rating_bits(c.rating[i]));
  h = mix64(h ^ (((uint64_t)(uint32_t)c.status[i] << 32) | (uint32_t)c.episodes[i]));
  return h;
}
__device__ __forceinline__ bool row_eq(const IngestCols &c, int64_t i, int64_t j) {
  return c.user[i] == c.user[j] && c.anime[i] == c.anime[j] && c.status[i] == c.status[j] &&
         c.episodes[i] == c.episodes[j] && rating_bits(c.rating[i]) == rating_bits(c.rating[j]);
}

// order-preserving double <-> uint64 (for atomicMin / atomicMax)
__device__ __forceinline__ unsigned long long d2ord(double d) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__host__ __device__ inline double ord2d(unsigned long long o) {
  const unsigned long long b = (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFULL) : ~o;
  double d;
  memcpy(&d, &b, 8);
  return d;
}
// dropna + the two row-local filters (they commute with drop_duplicates: duplicates share their fate)
__global__ __launch_bounds__(256) void k_ing_alive(IngestCols c, int64_t n, int drop_unwatched, int drop_plan,
                                                   int user_bound, int anime_bound, uint8_t *alive,
                                                   uint8_t *keep, int32_t *err) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t u = c.user[i], a = c.anime[i], s = c.status[i], e = c.episodes[i];
    const double r = c.rating[i];
    bool ok = u != ANIREC_NULL_I32 && a != ANIREC_NULL_I32 && s != ANIREC_NULL_I32 && e != ANIREC_NULL_I32 &&
              r == r;
    if (ok && (u < 0 || u >= user_bound || a < 0 || a >= anime_bound)) {
      *err = 1;  // id outside the direct-index tables: the host reports ANIREC_EINVAL
      ok = false;
    }
    if (ok && drop_unwatched && e == 0) ok = false;
    if (ok && drop_plan && s == 6) ok = false;
    alive[i] = ok ? 1 : 0;
    keep[i] = ok ? 1 : 0;  // k_ing_insert clears the later members of every class of identical rows
  }
}

// drop_duplicates(keep='first'): every class of identical rows owns one table slot that ends up
// holding the SMALLEST row index of the class, whatever the insertion order: a row that meets an equal
// row in its slot does atomicMin(slot, i); the larger of (i, previous owner) has lost and its keep
// flag is cleared — the final owner is never cleared, every other member exactly once.  No pass over
// the table is needed afterwards.
// A slot is {upper 32 hash bits, row index}: a probe that lands on another class is rejected by the
// tag without touching that row's five columns (five random sectors).
__global__ __launch_bounds__(256) void k_ing_insert(IngestCols c, int64_t n, const uint8_t *alive,
                                                    unsigned long long *table, uint32_t mask, uint8_t *keep) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (!alive[i]) continue;
    const uint64_t h = row_hash(c, i);
    const unsigned long long entry = (h & 0xFFFFFFFF00000000ULL) | (uint32_t)i;
    uint32_t s = (uint32_t)h & mask;
    for (;;) {
      const unsigned long long old = atomicCAS(&table[s], ~0ULL, entry);
      if (old == ~0ULL) break;
      if ((old >> 32) == (entry >> 32) && row_eq(c, i, (int64_t)(uint32_t)old)) {
        // same tag: the 64-bit minimum is the minimum row index; prev is a member of the same class
        const unsigned long long prev = atomicMin(&table[s], entry);
        keep[prev > entry ? (uint32_t)prev : (uint32_t)i] = 0;
        break;
      }
      s = (s + 1) & mask;
    }
  }
}

// value_counts() of user_id over the surviving rows.  The raw table is grouped by user (ascending
// blocks of user_id), so consecutive rows mostly share their user: a wave adds one atomic per RUN of
// equal ids among its 64 consecutive rows instead of one per row.
__global__ __launch_bounds__(256) void k_ing_count(const int32_t *id, int64_t n, const uint8_t *keep, int32_t *cnt) {
  const int lane = lane_id();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n_round = (n + 63) / 64 * 64;  // whole waves stay converged for the shuffles
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
    const int32_t key = (i < n && keep[i]) ? id[i] : -1;
    const int32_t left = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || key != left;
    const unsigned long long heads = __ballot(head);
    if (head && key >= 0) {
      const unsigned long long later = lane == 63 ? 0ULL : heads >> (lane + 1);
      const int len = later ? __ffsll((long long)later) : 64 - lane;
      atomicAdd(&cnt[key], len);
    }
  }
}
// the num_reviews filter; when it is the last filter it also reduces min / max of the ratings
template <bool kMinMax>
__global__ __launch_bounds__(256) void k_ing_user_filter(const int32_t *user, const double *rating, int64_t n,
                                                         const int32_t *cnt, int num_reviews, uint8_t *keep,
                                                         unsigned long long *mm) {
  unsigned long long lo = ~0ULL, hi = 0ULL;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (!keep[i]) continue;
    if (cnt[user[i]] < num_reviews) {
      keep[i] = 0;
      continue;
    }
    if (kMinMax) {
      const unsigned long long o = d2ord(rating[i]);
      lo = o < lo ? o : lo;
      hi = o > hi ? o : hi;
    }
  }
  if (kMinMax) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned long long l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
    __shared__ unsigned long long red[2][4];
    if (lane_id() == 0) {
      red[0][threadIdx.x >> 6] = lo;
      red[1][threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // one pair of atomics per workgroup
      for (int w = 1; w < 4; ++w) {
        lo = red[0][w] < lo ? red[0][w] : lo;
        hi = red[1][w] > hi ? red[1][w] : hi;
      }
      if (lo <= hi) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
      }
    }
  }
}
// groupby('anime_id')['watched_episodes'].max(), then keep watched >= (max == 1 ? 1 : max * .5)
__global__ __launch_bounds__(256) void k_ing_anime_max(IngestCols c, int64_t n, const uint8_t *keep, int32_t *mx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if (keep[i]) atomicMax(&mx[c.anime[i]], c.episodes[i]);
}
__global__ __launch_bounds__(256) void k_ing_half_filter(IngestCols c, int64_t n, const int32_t *mx, uint8_t *keep) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (!keep[i]) continue;
    const int32_t m = mx[c.anime[i]];
    const double half = m == 1 ? 1.0 : (double)m * .5;
    if (!((double)c.episodes[i] >= half)) keep[i] = 0;
  }
}

// min(df['rating']), max(df['rating']) over the surviving rows
__global__ __launch_bounds__(256) void k_ing_minmax(const double *rating, int64_t n, const uint8_t *keep,
                                                    unsigned long long *mm) {
  unsigned long long lo = ~0ULL, hi = 0ULL;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (!keep[i]) continue;
    const unsigned long long o = d2ord(rating[i]);
    lo = o < lo ? o : lo;
    hi = o > hi ? o : hi;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned long long l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if (lane_id() == 0) {
    atomicMin(&mm[0], lo);
    atomicMax(&mm[1], hi);
  }
}

// ---- exclusive scan of byte flags -> int32 positions (3 passes; flags padded with zeros to a tile) ----
__global__ __launch_bounds__(256) void k_scan_reduce(const uint8_t *flags, int32_t *bsum) {
  __shared__ int wsum[4];
  const uint4 v = reinterpret_cast<const uint4 *>(flags)[(size_t)blockIdx.x * 256 + threadIdx.x];
  // flags are 0/1 bytes: the byte sum of a dword is a popcount
  int s = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(1024) void k_scan_spine(int32_t *bsum, int nb, int64_t *total) {
  __shared__ long long part[1024];
  const int per = (nb + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
  long long s = 0;
  for (int b = b0; b < b1; ++b) s += bsum[b];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long run = 0;
    for (int t = 0; t < 1024; ++t) {
      const long long x = part[t];
      part[t] = run;
      run += x;
    }
    *total = run;
  }
  __syncthreads();
  int run = (int)part[threadIdx.x];
  for (int b = b0; b < b1; ++b) {
    const int x = bsum[b];
    bsum[b] = run;
    run += x;
  }
}
struct IngestOut {
  int32_t *user, *anime;
  double *rating;
  int32_t *status, *episodes;
};
// stable compaction + scale_ratings, fused with the last pass of the flag scan: a workgroup owns one scan tile
// (4 096 rows), turns its flags into output positions in LDS (block base from the spine + in-block exclusive scan)
// and moves its surviving rows — the 4-byte-per-row position array of a separate scan pass (written and gathered
// again: 0.9 GB at 109 M rows) never exists.
__global__ __launch_bounds__(256) void k_ing_compact(IngestCols c, int64_t n, const uint8_t *keep, const int32_t *bsum,
                                                     const unsigned long long *mm, IngestOut o) {
  __shared__ int wsum[4];
  __shared__ int32_t posl[kScanTile];
  const double mn = ord2d(mm[0]), mx = ord2d(mm[1]);
  const double span = mx - mn;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint4 v = reinterpret_cast<const uint4 *>(keep)[t];  // flags are padded with zeros to whole tiles
  const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
  const int mine = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  int incl = mine;  // inclusive scan over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d, 64);
    if (lane_id() >= d) incl += y;
  }
  if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int run = bsum[blockIdx.x] + incl - mine;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wsum[w];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t f = (wds[j >> 2] >> (8 * (j & 3))) & 1u;
    posl[threadIdx.x * 16 + j] = f ? run : -1;
    run += f;
  }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
#pragma unroll 4
  for (int k = 0; k < kScanTile / 256; ++k) {  // coalesced over the tile's rows
    const int r = k * 256 + threadIdx.x;
    const int32_t p = posl[r];
    const int64_t i = base + r;
    if (p < 0 || i >= n) continue;
    o.user[p] = c.user[i];
    o.anime[p] = c.anime[i];
    o.rating[p] = (c.rating[i] - mn) / span;
    o.status[p] = c.status[i];
    o.episodes[p] = c.episodes[i];
  }
}

// the two columns drop_half_watched leaves in the frame (preprocess.py:99-100), for the surviving rows
__global__ __launch_bounds__(256) void k_ing_half_columns(const int32_t *anime, const int64_t *n_out, const int32_t *mx,
                                                          int32_t *max_eps, double *half_eps) {
  const int64_t m = *n_out;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
    const int32_t v = mx[anime[i]];
    max_eps[i] = v;
    half_eps[i] = v == 1 ? 1.0 : (double)v * .5;
  }
}

// ---- Series.unique() encoding: dense index = rank of the id's first appearance ----
// first[v] = smallest row holding id v (atomicMin, run heads only); the RANK of that row among all first rows is
// the id's dense index.  The first rows are marked in a bitmap over the rows (n / 8 bytes), a prefix popcount over
// its words gives every id its rank, and one streaming pass maps the column through the per-id rank table: the
// column is read twice and the index written once (12 B per row) — no per-row flag, scan or rank arrays.
__global__ __launch_bounds__(256) void k_enc_first(const int32_t *id, int64_t n, int bound, int32_t *first, int32_t *err) {
  const int lane = lane_id();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n_round = (n + 63) / 64 * 64;  // whole waves stay converged for the shuffle
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
    int32_t v = i < n ? id[i] : -1;
    if (i < n && (v < 0 || v >= bound)) {
      *err = 1;
      v = -1;
    }
    // A table grouped by user presents runs of equal ids to a wave: only the first lane of a run (the
    // smallest row index of the run) goes to memory.  It reads first (at L2, where the atomics land: an
    // L1 line would stay stale), so that after its first few rows an id costs no atomic at all.
    const int32_t left = __shfl_up(v, 1, 64);
    const bool head = lane == 0 || v != left;
    if (head && v >= 0 && __hip_atomic_load(&first[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (int32_t)i)
      atomicMin(&first[v], (int32_t)i);
  }
}
constexpr int kBitTile = 4096;  // bitmap words per workgroup of the word scan (256 threads x 16 words)
__global__ __launch_bounds__(256) void k_enc_bits(const int32_t *first, int bound, uint32_t *bits) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= bound) return;
  const int32_t f = first[v];
  if (f != 0x7FFFFFFF) atomicOr(&bits[f >> 5], 1u << (f & 31));
}
__global__ __launch_bounds__(256) void k_bits_reduce(const uint32_t *bits, int32_t *bsum) {
  __shared__ int wsum[4];
  const uint4 *p = reinterpret_cast<const uint4 *>(bits) + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  int s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint4 v = p[k];
    s += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane_id() == 0) wsum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// wpre[w] = number of set bits in the words before w
__global__ __launch_bounds__(256) void k_bits_apply(const uint32_t *bits, const int32_t *bsum, int32_t *wpre) {
  __shared__ int wsum[4];
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t wd[16];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint4 v = reinterpret_cast<const uint4 *>(bits)[t * 4 + k];
    wd[4 * k] = v.x, wd[4 * k + 1] = v.y, wd[4 * k + 2] = v.z, wd[4 * k + 3] = v.w;
  }
  int mine = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) mine += __popc(wd[k]);
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d, 64);
    if (lane_id() >= d) incl += y;
  }
  if (lane_id() == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int run = bsum[blockIdx.x] + incl - mine;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wsum[w];
  int32_t out[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    out[k] = run;
    run += __popc(wd[k]);
  }
  int4 *dst = reinterpret_cast<int4 *>(wpre + t * 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) dst[k] = make_int4(out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]);
}
// first[v] (a row number) -> the id's rank; uniques[rank] = v
__global__ __launch_bounds__(256) void k_enc_rank(int32_t *first, int bound, const uint32_t *bits, const int32_t *wpre,
                                                  int32_t *uniques) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= bound) return;
  const int32_t f = first[v];
  if (f == 0x7FFFFFFF) return;
  const int32_t rank = wpre[f >> 5] + __popc(bits[f >> 5] & ((1u << (f & 31)) - 1u));
  first[v] = rank;  // the table now maps id -> dense index
  uniques[rank] = v;
}
__global__ __launch_bounds__(256) void k_enc_emit(const int32_t *id, int64_t n, int bound, const int32_t *rank_of,
                                                  int32_t *idx) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t v = id[i];
    idx[i] = (v < 0 || v >= bound) ? -1 : rank_of[v];
  }
}

static inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }
static inline size_t pad_tile(int64_t n) { return ((size_t)n + kScanTile - 1) / kScanTile * kScanTile; }
static inline uint32_t table_slots(int64_t n) {
  uint64_t s = 1024;
  while (s < (uint64_t)n * 2) s <<= 1;
  return (uint32_t)s;
}
static inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}
// flags (padded, tail zeroed by the caller) -> exclusive tile bases in bsum, total (the in-tile scan is redone by
// the consumer, k_ing_compact)
static int scan_flags(const uint8_t *flags, int64_t n, int32_t *bsum, int64_t *total, hipStream_t s) {
  const int nb = (int)(pad_tile(n) / kScanTile);
  hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(256), 0, s, flags, bsum);
  hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, s, bsum, nb, total);
  return (int)hipGetLastError();
}

}  // namespace anirec

using namespace anirec;

extern "C" {

// alive | keep (padded byte flags) | bsum | hash table | user counts | anime max | minmax + err
size_t anirec_ingest_workspace_bytes(int64_t n, int32_t user_id_bound, int32_t anime_id_bound) {
  if (n < 1 || n >= ((int64_t)1 << 30) || user_id_bound < 1 || anime_id_bound < 1) return 0;
  const size_t np = pad_tile(n);
  return 2 * al256(np) + al256(np / kScanTile * 4) + al256((size_t)table_slots(n) * 8) +
         al256((size_t)user_id_bound * 4) + al256((size_t)anime_id_bound * 4) + 256;
}

int anirec_ingest_preprocess(const int32_t *user_id, const int32_t *anime_id, const double *rating,
                             const int32_t *watching_status, const int32_t *watched_episodes, int64_t n,
                             const anirec_ingest_opts *opts, int32_t *out_user_id, int32_t *out_anime_id,
                             double *out_rating, int32_t *out_status, int32_t *out_episodes, int64_t *n_out,
                             int32_t *err_flag, void *workspace, size_t workspace_bytes, void *stream) {
  if (!user_id || !anime_id || !rating || !watching_status || !watched_episodes || !opts || !out_user_id ||
      !out_anime_id || !out_rating || !out_status || !out_episodes || !n_out || !err_flag || !workspace)
    return ANIREC_EINVAL;
  if (n < 1 || n >= ((int64_t)1 << 30) || opts->user_id_bound < 1 || opts->anime_id_bound < 1) return ANIREC_EINVAL;
  if (workspace_bytes < anirec_ingest_workspace_bytes(n, opts->user_id_bound, opts->anime_id_bound))
    return ANIREC_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const size_t np = pad_tile(n);
  const uint32_t slots = table_slots(n);
  char *p = (char *)workspace;
  uint8_t *alive = (uint8_t *)p;
  p += al256(np);
  uint8_t *keep = (uint8_t *)p;
  p += al256(np);
  int32_t *bsum = (int32_t *)p;
  p += al256(np / kScanTile * 4);
  unsigned long long *table = (unsigned long long *)p;
  p += al256((size_t)slots * 8);
  int32_t *cnt_u = (int32_t *)p;
  p += al256((size_t)opts->user_id_bound * 4);
  int32_t *max_ep = (int32_t *)p;
  p += al256((size_t)opts->anime_id_bound * 4);
  unsigned long long *mm = (unsigned long long *)p;

  const IngestCols c{user_id, anime_id, rating, watching_status, watched_episodes};
  const int g = grid_for(n);
  ANIREC_HIP_CHECK(hipMemsetAsync(err_flag, 0, 4, s));
  ANIREC_HIP_CHECK(hipMemsetAsync(keep + n, 0, np - (size_t)n, s));
  ANIREC_HIP_CHECK(hipMemsetAsync(alive + n, 0, np - (size_t)n, s));
  ANIREC_HIP_CHECK(hipMemsetAsync(table, 0xFF, (size_t)slots * 8, s));
  ANIREC_HIP_CHECK(hipMemsetAsync(cnt_u, 0, (size_t)opts->user_id_bound * 4, s));
  ANIREC_HIP_CHECK(hipMemsetAsync(mm, 0xFF, 8, s));
  ANIREC_HIP_CHECK(hipMemsetAsync(mm + 1, 0, 8, s));
  hipLaunchKernelGGL(k_ing_alive, dim3(g), dim3(256), 0, s, c, n, opts->drop_unwatched, opts->drop_plan,
                     opts->user_id_bound, opts->anime_id_bound, alive, keep, err_flag);
  hipLaunchKernelGGL(k_ing_insert, dim3(g), dim3(256), 0, s, c, n, alive, table, slots - 1, keep);
  hipLaunchKernelGGL(k_ing_count, dim3(g), dim3(256), 0, s, user_id, n, keep, cnt_u);
  if (!opts->drop_half_watched) {
    hipLaunchKernelGGL(k_ing_user_filter<true>, dim3(g > 4096 ? 4096 : g), dim3(256), 0, s, user_id, rating, n, cnt_u,
                       opts->num_reviews, keep, mm);
  } else {
    hipLaunchKernelGGL(k_ing_user_filter<false>, dim3(g), dim3(256), 0, s, user_id, rating, n, cnt_u,
                       opts->num_reviews, keep, mm);
    // episodes can be any int32: start the per-anime maxima at INT32_MIN (0x80 bytes give 0x80808080 < 0,
    // below every non-null value is not guaranteed, so set exactly)
    ANIREC_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)max_ep, (int)0x80000000, (size_t)opts->anime_id_bound, s));
    hipLaunchKernelGGL(k_ing_anime_max, dim3(g), dim3(256), 0, s, c, n, keep, max_ep);
    hipLaunchKernelGGL(k_ing_half_filter, dim3(g), dim3(256), 0, s, c, n, max_ep, keep);
    hipLaunchKernelGGL(k_ing_minmax, dim3(g > 2048 ? 2048 : g), dim3(256), 0, s, rating, n, keep, mm);
  }
  ANIREC_HIP_CHECK(hipGetLastError());
  int rc = scan_flags(keep, n, bsum, n_out, s);
  if (rc) return rc;
  const IngestOut o{out_user_id, out_anime_id, out_rating, out_status, out_episodes};
  hipLaunchKernelGGL(k_ing_compact, dim3((unsigned)(np / kScanTile)), dim3(256), 0, s, c, n, keep, bsum, mm, o);
  return (int)hipGetLastError();
}

int anirec_ingest_half_columns(const int32_t *out_anime_id, const int64_t *n_out, int64_t n,
                               const anirec_ingest_opts *opts, int32_t *out_max_eps, double *out_half_eps,
                               const void *workspace, size_t workspace_bytes, void *stream) {
  if (!out_anime_id || !n_out || !opts || !out_max_eps || !out_half_eps || !workspace) return ANIREC_EINVAL;
  if (n < 1 || n >= ((int64_t)1 << 30) || opts->user_id_bound < 1 || opts->anime_id_bound < 1) return ANIREC_EINVAL;
  if (!opts->drop_half_watched) return ANIREC_EINVAL;  // the per-anime maxima only exist after that pass
  if (workspace_bytes < anirec_ingest_workspace_bytes(n, opts->user_id_bound, opts->anime_id_bound))
    return ANIREC_EWORKSPACE;
  // same carve as anirec_ingest_preprocess: the per-anime maxima sit behind the user counts
  const size_t np = pad_tile(n);
  const char *p = (const char *)workspace;
  p += 2 * al256(np) + al256(np / kScanTile * 4) + al256((size_t)table_slots(n) * 8) +
       al256((size_t)opts->user_id_bound * 4);
  hipLaunchKernelGGL(k_ing_half_columns, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out_anime_id, n_out,
                     (const int32_t *)p, out_max_eps, out_half_eps);
  return (int)hipGetLastError();
}

// first-appearance / rank table [id_bound] | bitmap over the rows (padded to whole word tiles) | word prefix | bsum
static inline size_t bit_words(int64_t n) { return (((size_t)n + 31) / 32 + kBitTile - 1) / kBitTile * kBitTile; }

size_t anirec_ingest_encode_workspace_bytes(int64_t n, int32_t id_bound) {
  if (n < 1 || n >= ((int64_t)1 << 30) || id_bound < 1) return 0;
  const size_t nw = bit_words(n);
  return al256((size_t)id_bound * 4) + 2 * al256(nw * 4) + al256(nw / kBitTile * 4) + 256;
}

int anirec_ingest_encode(const int32_t *id, int64_t n, int32_t id_bound, int32_t *out_index, int32_t *out_uniques,
                         int64_t *n_unique, int32_t *err_flag, void *workspace, size_t workspace_bytes,
                         void *stream) {
  if (!id || !out_index || !out_uniques || !n_unique || !err_flag || !workspace) return ANIREC_EINVAL;
  if (n < 1 || n >= ((int64_t)1 << 30) || id_bound < 1) return ANIREC_EINVAL;
  if (workspace_bytes < anirec_ingest_encode_workspace_bytes(n, id_bound)) return ANIREC_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const size_t nw = bit_words(n);
  char *p = (char *)workspace;
  int32_t *first = (int32_t *)p;
  p += al256((size_t)id_bound * 4);
  uint32_t *bits = (uint32_t *)p;
  p += al256(nw * 4);
  int32_t *wpre = (int32_t *)p;
  p += al256(nw * 4);
  int32_t *bsum = (int32_t *)p;
  const int g = grid_for(n);
  const int gb = (id_bound + 255) / 256;
  const int nb = (int)(nw / kBitTile);
  ANIREC_HIP_CHECK(hipMemsetAsync(err_flag, 0, 4, s));
  ANIREC_HIP_CHECK(hipMemsetAsync(bits, 0, nw * 4, s));
  ANIREC_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)first, 0x7FFFFFFF, (size_t)id_bound, s));
  hipLaunchKernelGGL(k_enc_first, dim3(g), dim3(256), 0, s, id, n, id_bound, first, err_flag);
  hipLaunchKernelGGL(k_enc_bits, dim3(gb), dim3(256), 0, s, first, id_bound, bits);
  hipLaunchKernelGGL(k_bits_reduce, dim3(nb), dim3(256), 0, s, bits, bsum);
  hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, s, bsum, nb, n_unique);
  hipLaunchKernelGGL(k_bits_apply, dim3(nb), dim3(256), 0, s, bits, bsum, wpre);
  hipLaunchKernelGGL(k_enc_rank, dim3(gb), dim3(256), 0, s, first, id_bound, bits, wpre, out_uniques);
  hipLaunchKernelGGL(k_enc_emit, dim3(g), dim3(256), 0, s, id, n, id_bound, first, out_index);
  return (int)hipGetLastError();
}

}  // extern "C"
