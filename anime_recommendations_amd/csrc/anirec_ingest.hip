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
// HBM-bound integer / byte work: one front pass over the five columns (row filters, duplicate rows through an
// LDS hash table per chunk of rows, per-user counts), direct-index tables for the per-user / per-anime
// aggregates, and a flag scan fused into the compaction that turns flags into stable (order-preserving)
// output positions.  Results are bit-identical to pandas:
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
// order-preserving float <-> uint32 (-0.0 below +0.0, like d2ord)
__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) { return __uint_as_float((o >> 31) ? (o & 0x7FFFFFFFu) : ~o); }
// ---- the front pass -------------------------------------------------------------------------------------------
// dropna, the two row-local filters, drop_duplicates(keep='first'), value_counts(user_id) and the num_reviews
// filter in ONE read of the five columns.
//
// Identical rows share their user, so duplicates can only meet inside the rows of one user.  The table is cut into
// chunks of kChunk consecutive rows, one workgroup each; k_ing_span first records every user's first and last row.
// A user whose first row lies in a chunk is OWNED by that chunk's workgroup if its last row lies in the chunk too or
// in the rows the chunk's last user reaches over the boundary (the workgroup reads on to that user's last row, at
// most kExt rows): its duplicates are found in an LDS hash table of the workgroup, its rating count is complete when
// the workgroup is done (counted in LDS, at the slot of the user's first row), so the num_reviews filter (and min /
// max of the surviving ratings) is settled before the workgroup leaves.  The next workgroup skips the rows its
// neighbour owns — both decide from the same two tables, so every row has exactly one owner.  The raw animelist is
// grouped by user: every user is owned, except one whose block is longer than the rows a workgroup may read on.
// Rows of users nobody owns are appended to a list and go through the global hash table (k_nl_*), which is sized
// and cleared for the list, not for the table.  Nothing assumes the grouping: a table in random order just has
// every row on the list.
constexpr int kChunk = 8192;                  // rows whose users a workgroup owns; two workgroups share a CU's LDS
constexpr int kExt = 4096;                    // rows it may read on for its last user
constexpr int kSpanRows = kChunk + kExt;
constexpr int kFrontThreads = 512;
constexpr int kQuads = kSpanRows / (4 * kFrontThreads);  // a lane owns 4 consecutive rows in each of kQuads passes
constexpr int kLdsSlots = 2 * kChunk;         // 4-byte slots {17-bit tag | 14-bit row in the span}: 64 KB
constexpr uint32_t kLdsEmpty = 0xFFFFFFFFu;   // no entry has it: the top tag bit is always 0
constexpr uint32_t kRowMask = 16383;
constexpr int kSpanTiles = kSpanRows / kScanTile;
static_assert(kChunk % kScanTile == 0 && kExt % kScanTile == 0 && (kChunk & (kChunk - 1)) == 0, "whole scan tiles");
static_assert(kQuads * 4 * kFrontThreads == kSpanRows && kChunk / kScanTile == 2 && kQuads % 2 == 0 &&
                  kSpanRows <= (int)kRowMask && kQuads % kSpanTiles == 0,
              "row mapping");

__device__ __forceinline__ int4 ld4(const int32_t *p, int64_t i, int64_t n, int32_t fill) {
  if (i + 3 < n) return *reinterpret_cast<const int4 *>(p + i);
  return make_int4(i < n ? p[i] : fill, i + 1 < n ? p[i + 1] : fill, i + 2 < n ? p[i + 2] : fill, fill);
}

// the same with a 32-bit row number relative to a uniform base pointer
__device__ __forceinline__ int4 ld4r(const int32_t *p, int r, int lim, int32_t fill) {
  if (r + 3 < lim) return *reinterpret_cast<const int4 *>(p + r);
  return make_int4(r < lim ? p[r] : fill, r + 1 < lim ? p[r + 1] : fill, r + 2 < lim ? p[r + 2] : fill, fill);
}

// first / last row of every user (all rows with a valid id, whatever their other columns hold).  A lane reads 4
// consecutive rows; only the first row of a run of equal ids updates the first-row table and only the last row
// of a run the last-row table.
constexpr int kSpanIters = 4;
constexpr int kFewHeads = 8;  // lanes of a wave with a run end among their rows: up to here the atomics go out unread
__global__ __launch_bounds__(256) void k_ing_span(const int32_t *user, int64_t n, int bound, int32_t *ufirst,
                                                  int32_t *ulast) {
  const int lane = lane_id();
  int4 q[kSpanIters];
  int64_t i0[kSpanIters];
#pragma unroll
  for (int k = 0; k < kSpanIters; ++k) {
    i0[k] = (((int64_t)blockIdx.x * kSpanIters + k) * 256 + threadIdx.x) * 4;
    q[k] = ld4(user, i0[k], n, -1);
  }
#pragma unroll
  for (int k = 0; k < kSpanIters; ++k) {
    int32_t v[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v[j] < 0 || v[j] >= bound) v[j] = -1;
    const int32_t left = __shfl_up(v[3], 1, 64), right = __shfl_down(v[0], 1, 64);
    bool head[4], tail[4];
    bool any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      head[j] = v[j] >= 0 && (j ? v[j] != v[j - 1] : (lane == 0 || v[0] != left));
      tail[j] = v[j] >= 0 && (j < 3 ? v[j] != v[j + 1] : (lane == 63 || v[3] != right));
      any |= head[j] | tail[j];
    }
    if (__popcll(__ballot(any)) <= kFewHeads) {
      // a table grouped by user: a handful of run ends per wave — atomics without a return value, nothing to wait for
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int32_t i = (int32_t)(i0[k] + j);
        if (head[j]) atomicMin(&ufirst[v[j]], i);
        if (tail[j]) atomicMax(&ulast[v[j]], i);
      }
    } else {
      // no runs: read before updating (a stale read only costs a redundant atomic), the eight reads in flight together
      int32_t f[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[j] = ufirst[head[j] ? v[j] : 0];
        l[j] = ulast[tail[j] ? v[j] : 0];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int32_t i = (int32_t)(i0[k] + j);
        if (head[j] && f[j] > i) atomicMin(&ufirst[v[j]], i);
        if (tail[j] && l[j] < i) atomicMax(&ulast[v[j]], i);
      }
    }
  }
}

struct RowVals {
  int32_t u, a, s, e;
  double r;
};
// 32 bits for the chunk's LDS table (slot = the low bits, tag = bits 14..30).  Two 32-bit multiplies: the 64-bit
// mixer of the list path is six 64-bit multiplies per row — a quarter of the front pass's issue time on a part whose
// integer multiplier runs at quarter rate.  Rows that meet here share their user; the anime id carries the spread.
__device__ __forceinline__ uint32_t row_hash32(const RowVals &v) {
  const uint64_t rb = rating_bits(v.r);
  uint32_t h = (uint32_t)v.a * 0x9E3779B1u + (uint32_t)v.e;
  h = ((h ^ (h >> 15)) * 0x85EBCA77u) ^ (uint32_t)v.u ^ ((uint32_t)v.s << 24) ^ (uint32_t)rb ^ (uint32_t)(rb >> 32);
  return h ^ (h >> 13) ^ (h >> 21);
}
// a row in registers against a row of the span in memory (scalar column bases + a 32-bit row number)
__device__ __forceinline__ bool row_eq_rel(const int32_t *pu, const int32_t *pa, const double *pr, const int32_t *ps,
                                           const int32_t *pe, const RowVals &v, int j) {
  return v.u == pu[j] && v.a == pa[j] && v.s == ps[j] && v.e == pe[j] && rating_bits(v.r) == rating_bits(pr[j]);
}

struct FrontArgs {
  IngestCols c;
  int64_t n;
  int drop_unwatched, drop_plan, user_bound, anime_bound, num_reviews, n_tiles;
  const int32_t *ufirst, *ulast;
  uint8_t *keep;       // [chunks * kChunk]
  int32_t *tile_cnt;   // zeroed [n_tiles]: surviving rows per scan tile (the list kernels subtract what they drop later)
  int32_t *nl_list;    // rows of non-local users that passed the row filters
  int32_t *nl_count;   // zeroed
  unsigned long long *mm;
  int32_t *err;
};

// per-row state carried from phase A to phase C: bit 15 = passed the row filters, bit 14 = owned user,
// bits 0..12 = the user's first row in the chunk (the slot of its counter)
constexpr uint32_t kOk = 0x8000u, kLocal = 0x4000u;

// rows the last user of the chunk that ends at row `end` reaches past it — the rows that chunk's workgroup reads on
// (0: the chunk ends with its user, the user started before the chunk, or its block is longer than kExt).  The
// workgroup of the chunk and its right neighbour both call this with the same `end`.
__device__ __forceinline__ int reach_over(const FrontArgs &a, int64_t end) {
  if (end <= 0 || end >= a.n) return 0;
  const int32_t u = a.c.user[end - 1];
  if (u < 0 || u >= a.user_bound) return 0;
  const int64_t f = a.ufirst[u], l = a.ulast[u];
  return (f >= end - kChunk && l >= end && l < end + kExt) ? (int)(l - end + 1) : 0;
}

template <bool kMinMax>
__global__ __launch_bounds__(kFrontThreads, 4) void k_ing_front(FrontArgs a) {
  __shared__ uint32_t tab[kLdsSlots];  // phase A: the hash table; afterwards [0, kChunk): the users' counters
  __shared__ uint8_t keepl[kSpanRows];
  __shared__ int32_t tile_l[kSpanTiles];
  __shared__ int32_t nl_total, nl_base, nl_fill;
  __shared__ unsigned long long red[2][kFrontThreads / 64];
  const int tid = threadIdx.x, lane = lane_id();
  const IngestCols &c = a.c;
  const int64_t base = (int64_t)blockIdx.x * kChunk;
  const int base32 = (int)base;  // n < 2^30: row numbers fit 32 bits
  // uniform column bases + 32-bit row numbers inside the span: scalar base + one offset register per access
  const int lim = (int)((a.n - base < 2 * kSpanRows) ? a.n - base : 2 * kSpanRows);  // rows of the table from `base` on
  const int32_t *pu = c.user + base, *pa = c.anime + base, *ps = c.status + base, *pe = c.episodes + base;
  const double *pr = c.rating + base;
  uint8_t *pk = a.keep + base;
  // the rows this workgroup reads on behind its chunk, and the rows at the chunk's front its left neighbour reads
  const int ext = reach_over(a, base + kChunk), ext_prev = reach_over(a, base);

  {
    uint4 *t4 = reinterpret_cast<uint4 *>(tab);
    for (int k = tid; k < kLdsSlots / 4; k += kFrontThreads) t4[k] = make_uint4(kLdsEmpty, kLdsEmpty, kLdsEmpty, kLdsEmpty);
    uint4 *k4 = reinterpret_cast<uint4 *>(keepl);
    for (int k = tid; k < kSpanRows / 16; k += kFrontThreads) k4[k] = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
    if (tid < kSpanTiles) tile_l[tid] = 0;
    if (tid == 0) nl_total = 0, nl_fill = 0;
  }
  __syncthreads();

  // phase A: row filters; rows of owned users into the LDS table, the later members of a class lose their flag
  // per row one word: the low half is the state above; the high half the row's rating for the min / max of phase C,
  // as the upper 16 bits of the order-preserving key of its float value — ratings are small integers in practice; a
  // rating that is not such a float is flagged and read again
  uint32_t mk[kQuads][4], inexact = 0, mine = 0;  // mine: 4 bits per quad, the row's flag is this workgroup's to write
  bool bad = false;
#pragma unroll
  for (int h = 0; h < kQuads; h += 2) {  // two quads of loads in flight
    if (h * kFrontThreads * 4 >= kChunk && (h * kFrontThreads + (tid & ~63)) * 4 >= kChunk + ext) {
      // a wave whose rows all lie behind the last row this workgroup reads (most waves, behind the chunk itself)
#pragma unroll
      for (int j = 0; j < 4; ++j) mk[h][j] = mk[h + 1][j] = 0;
      continue;
    }
    int4 U[2], A[2], S[2], E[2];
    double R[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r0 = ((h + b) * kFrontThreads + tid) * 4;
      U[b] = make_int4(ANIREC_NULL_I32, ANIREC_NULL_I32, ANIREC_NULL_I32, ANIREC_NULL_I32);
      A[b] = S[b] = E[b] = make_int4(0, 0, 0, 0);
      R[b][0] = R[b][1] = R[b][2] = R[b][3] = 0.0;
      if (r0 >= kChunk + ext) continue;  // behind the last row this workgroup reads
      U[b] = ld4r(pu, r0, lim, ANIREC_NULL_I32);
      A[b] = ld4r(pa, r0, lim, 0);
      S[b] = ld4r(ps, r0, lim, 0);
      E[b] = ld4r(pe, r0, lim, 0);
      if (r0 + 3 < lim) {
        const double2 r0v = *reinterpret_cast<const double2 *>(pr + r0);
        const double2 r1v = *reinterpret_cast<const double2 *>(pr + r0 + 2);
        R[b][0] = r0v.x, R[b][1] = r0v.y, R[b][2] = r1v.x, R[b][3] = r1v.y;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) R[b][j] = r0 + j < lim ? pr[r0 + j] : 0.0;
      }
    }
    int32_t F[2][4], L[2][4];
    bool ok[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int32_t u[4] = {U[b].x, U[b].y, U[b].z, U[b].w}, an[4] = {A[b].x, A[b].y, A[b].z, A[b].w};
      const int32_t st[4] = {S[b].x, S[b].y, S[b].z, S[b].w}, ep[4] = {E[b].x, E[b].y, E[b].z, E[b].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bool o = u[j] != ANIREC_NULL_I32 && an[j] != ANIREC_NULL_I32 && st[j] != ANIREC_NULL_I32 &&
                 ep[j] != ANIREC_NULL_I32 && R[b][j] == R[b][j];
        if (o && (u[j] < 0 || u[j] >= a.user_bound || an[j] < 0 || an[j] >= a.anime_bound)) {
          bad = true;  // id outside the direct-index tables: the host reports ANIREC_EINVAL
          o = false;
        }
        if (o && a.drop_unwatched && ep[j] == 0) o = false;
        if (o && a.drop_plan && st[j] == 6) o = false;
        ok[b][j] = o;
        const bool uv = u[j] >= 0 && u[j] < a.user_bound;  // the ownership of a row is its user's, kept or not
        F[b][j] = uv ? a.ufirst[u[j]] - base32 : -0x40000000;  // relative to the chunk's first row
        L[b][j] = uv ? a.ulast[u[j]] - base32 : -0x40000000;
      }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int32_t u[4] = {U[b].x, U[b].y, U[b].z, U[b].w}, an[4] = {A[b].x, A[b].y, A[b].z, A[b].w};
      const int32_t st[4] = {S[b].x, S[b].y, S[b].z, S[b].w}, ep[4] = {E[b].x, E[b].y, E[b].z, E[b].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ridx = ((h + b) * kFrontThreads + tid) * 4 + j;
        const int32_t f = F[b][j], l = L[b][j];
        // the user starts in this chunk and ends inside the rows this workgroup reads / the same for the left
        // neighbour (a row without a valid user id belongs to the chunk it lies in)
        const bool own = (uint32_t)f < (uint32_t)kChunk && l < kChunk + ext;
        const bool prev_owns = (uint32_t)(f + kChunk) < (uint32_t)kChunk && l < ext_prev;
        const bool my = ridx < kChunk ? !prev_owns : (ridx < kChunk + ext && own);
        if (my) mine |= 1u << ((h + b) * 4 + j);
        mk[h + b][j] = 0;
        if (kMinMax) {
          // the double's bits: exact as a float with 7 mantissa bits iff the 45 lower mantissa bits are zero and the
          // exponent is a normal float's (or the value is a zero); the float's upper half follows from the bits
          const uint64_t db = (uint64_t)__double_as_longlong(R[b][j]);
          const uint32_t dh = (uint32_t)(db >> 32), ex = (dh >> 20) & 0x7FFu;
          const bool zero = (db << 1) == 0;
          const bool exact = ((uint32_t)db == 0) & ((dh & 0x1FFFu) == 0) & ((ex - 897u < 254u) | zero);
          const uint32_t fh = zero ? (dh >> 16 & 0x8000u) : ((dh >> 16 & 0x8000u) | ((ex - 896u) << 7) | ((dh >> 13) & 0x7Fu));
          if (!exact) inexact |= 1u << ((h + b) * 4 + j);
          mk[h + b][j] = f2ord(__uint_as_float(fh << 16)) & 0xFFFF0000u;
        }
        if (!my || !ok[b][j]) continue;
        mk[h + b][j] |= kOk;
        if (!own) continue;  // nobody's: the list
        mk[h + b][j] |= kLocal | (uint32_t)f;
        const RowVals v{u[j], an[j], st[j], ep[j], R[b][j]};
        const uint32_t hh = row_hash32(v);
        const uint32_t entry = (hh & 0x7FFFC000u) | (uint32_t)ridx;
        uint32_t s = hh & (kLdsSlots - 1);
        for (;;) {
          const uint32_t old = atomicCAS(&tab[s], kLdsEmpty, entry);
          if (old == kLdsEmpty) break;
          if (((old ^ entry) & ~kRowMask) == 0 && row_eq_rel(pu, pa, pr, ps, pe, v, (int)(old & kRowMask))) {
            // same tag: the minimum entry is the minimum row; prev is a member of the same class (see k_nl_insert)
            const uint32_t prev = atomicMin(&tab[s], entry);
            keepl[prev > entry ? (prev & kRowMask) : (uint32_t)ridx] = 0;
            break;
          }
          s = (s + 1) & (kLdsSlots - 1);
        }
      }
    }
  }
  if (bad) *a.err = 1;
  __syncthreads();
  for (int k = tid; k < kChunk / 4; k += kFrontThreads) reinterpret_cast<uint4 *>(tab)[k] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  // phase B: value_counts of the owned users over the rows that are left, one LDS atomic per run of equal users
  // among a lane's 4 consecutive rows.  Every row of an owned user is one of this workgroup's: after the barrier
  // the counts are final.
  int nl = 0;
#pragma unroll
  for (int q = 0; q < kQuads; ++q) {
    const uint32_t kb = *reinterpret_cast<const uint32_t *>(&keepl[(q * kFrontThreads + tid) * 4]);
    int run = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t m = mk[q][j] & 0xFFFFu;
      if (!((kb >> (8 * j)) & 1u)) m = 0;  // lost to an earlier equal row
      mk[q][j] = (mk[q][j] & 0xFFFF0000u) | m;
      nl += (m & (kOk | kLocal)) == kOk;
      const bool cnt = (m & kLocal) != 0;
      const bool same_next = j < 3 && cnt && ((kb >> (8 * (j + 1))) & 1u) && (mk[q][(j + 1) & 3] & 0xFFFFu) == m;
      run += cnt;
      if (cnt && !same_next) {
        atomicAdd(&tab[m & (kChunk - 1)], (uint32_t)run);
        run = 0;
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) nl += __shfl_xor(nl, o, 64);
  if (lane == 0 && nl) atomicAdd(&nl_total, nl);
  __syncthreads();
  if (tid == 0 && nl_total) nl_base = atomicAdd(a.nl_count, nl_total);
  __syncthreads();

  // phase C: the num_reviews filter for the owned rows, the flags, the list of nobody's rows
  unsigned long long lo = ~0ULL, hi = 0ULL;
  uint32_t klo = 0xFFFFu, khi = 0u;
  int tcnt[kSpanTiles];
#pragma unroll
  for (int t = 0; t < kSpanTiles; ++t) tcnt[t] = 0;
#pragma unroll
  for (int q = 0; q < kQuads; ++q) {
    const int r0 = (q * kFrontThreads + tid) * 4;
    const uint32_t mq = (mine >> (q * 4)) & 0xFu;
    uint32_t kb = 0;
    int nlq = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t m = mk[q][j] & 0xFFFFu;
      bool kept = (m & kOk) != 0;
      if (m & kLocal) {
        if ((int32_t)tab[m & (kChunk - 1)] < a.num_reviews) {
          kept = false;
        } else if (kMinMax) {
          if ((inexact >> (q * 4 + j)) & 1u) {  // not a small float: the rating again, as a double
            const unsigned long long o = d2ord(pr[r0 + j]);
            lo = o < lo ? o : lo;
            hi = o > hi ? o : hi;
          } else {  // the order of the keys is the order of the ratings
            klo = min(klo, mk[q][j] >> 16);
            khi = max(khi, mk[q][j] >> 16);
          }
        }
      } else {
        nlq += kept;
      }
      kb |= (uint32_t)kept << (8 * j);
      tcnt[q / (kQuads / kSpanTiles)] += kept;
    }
    if (mq == 0xFu) {
      *reinterpret_cast<uint32_t *>(pk + r0) = kb;
    } else {  // a quad on the border between this workgroup's rows and a neighbour's
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((mq >> j) & 1u) pk[r0 + j] = (uint8_t)((kb >> (8 * j)) & 1u);
    }
    if (__ballot(nlq != 0)) {  // rare: the rows of a user nobody owns
      int incl = nlq;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d, 64);
        if (lane >= d) incl += y;
      }
      int off = 0;
      if (lane == 63) off = atomicAdd(&nl_fill, incl);
      off = __shfl(off, 63, 64) + nl_base + incl - nlq;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((mk[q][j] & (kOk | kLocal)) == kOk) a.nl_list[off++] = base32 + r0 + j;
    }
  }
#pragma unroll
  for (int t = 0; t < kSpanTiles; ++t) {
    int v = tcnt[t];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0 && v) atomicAdd(&tile_l[t], v);
  }
  if (kMinMax) {
    if (klo <= khi) {
      // a key's lower half: zeros of a non-negative float, ones of a negative one (f2ord flips those)
      const uint32_t kl = klo << 16, kh = khi << 16;
      const unsigned long long l2 = d2ord((double)ord2f(kl | ((kl >> 31) ? 0u : 0xFFFFu)));
      const unsigned long long h2 = d2ord((double)ord2f(kh | ((kh >> 31) ? 0u : 0xFFFFu)));
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const unsigned long long l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
    if (lane == 0) {
      red[0][tid >> 6] = lo;
      red[1][tid >> 6] = hi;
    }
  }
  __syncthreads();
  // the tiles of the chunk, and the tile behind it that the rows read on lie in (zeroed by the host: two
  // workgroups add to it)
  if (tid < kSpanTiles && tile_l[tid]) atomicAdd(&a.tile_cnt[blockIdx.x * (kChunk / kScanTile) + tid], tile_l[tid]);
  if (kMinMax && tid == 0) {  // one pair of atomics per workgroup
    for (int w = 1; w < kFrontThreads / 64; ++w) {
      lo = red[0][w] < lo ? red[0][w] : lo;
      hi = red[1][w] > hi ? red[1][w] : hi;
    }
    if (lo <= hi) {
      atomicMin(&a.mm[0], lo);
      atomicMax(&a.mm[1], hi);
    }
  }
}

// ---- the rows of non-local users (the list) ----
__device__ __forceinline__ uint32_t nl_slots(int32_t count) {  // power of two >= 2 * count
  uint32_t s = 4096;
  while (s < (uint32_t)count * 2u) s <<= 1;
  return s;
}
__global__ __launch_bounds__(256) void k_nl_clear(unsigned long long *table, const int32_t *nl_count) {
  const uint32_t slots = nl_slots(*nl_count);
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += gridDim.x * blockDim.x) table[s] = ~0ULL;
}
// drop_duplicates(keep='first'): every class of identical rows owns one table slot that ends up
// holding the SMALLEST row index of the class, whatever the insertion order: a row that meets an equal
// row in its slot does atomicMin(slot, i); the larger of (i, previous owner) has lost and its keep
// flag is cleared — the final owner is never cleared, every other member exactly once.  No pass over
// the table is needed afterwards.
// A slot is {upper 32 hash bits, row index}: a probe that lands on another class is rejected by the
// tag without touching that row's five columns (five random sectors).
__global__ __launch_bounds__(256) void k_nl_insert(IngestCols c, const int32_t *nl_list, const int32_t *nl_count,
                                                   unsigned long long *table, uint8_t *keep, int32_t *tile_cnt) {
  const int32_t count = *nl_count;
  const uint32_t mask = nl_slots(count) - 1;
  for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
    const int64_t i = nl_list[t];
    const uint64_t h = row_hash(c, i);
    const unsigned long long entry = (h & 0xFFFFFFFF00000000ULL) | (uint32_t)i;
    uint32_t s = (uint32_t)h & mask;
    for (;;) {
      const unsigned long long old = atomicCAS(&table[s], ~0ULL, entry);
      if (old == ~0ULL) break;
      if ((old >> 32) == (entry >> 32) && row_eq(c, i, (int64_t)(uint32_t)old)) {
        // same tag: the 64-bit minimum is the minimum row index; prev is a member of the same class
        const unsigned long long prev = atomicMin(&table[s], entry);
        const uint32_t loser = prev > entry ? (uint32_t)prev : (uint32_t)i;
        keep[loser] = 0;
        atomicSub(&tile_cnt[loser / kScanTile], 1);
        break;
      }
      s = (s + 1) & mask;
    }
  }
}
// value_counts of the non-local users (consecutive list entries mostly share their user: one atomic per run)
__global__ __launch_bounds__(256) void k_nl_count(const int32_t *user, const int32_t *nl_list, const int32_t *nl_count,
                                                  const uint8_t *keep, int32_t *cnt) {
  const int lane = lane_id();
  const int32_t count = *nl_count;
  const int32_t n_round = (count + 63) / 64 * 64;
  for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_round; t += gridDim.x * blockDim.x) {
    int32_t key = -1;
    if (t < count) {
      const int32_t i = nl_list[t];
      if (keep[i]) key = user[i];
    }
    const int32_t left = __shfl_up(key, 1, 64);
    const bool head = lane == 0 || key != left;
    const unsigned long long heads = __ballot(head);
    if (head && key >= 0) {
      const unsigned long long later = lane == 63 ? 0ULL : heads >> (lane + 1);
      atomicAdd(&cnt[key], later ? __ffsll((long long)later) : 64 - lane);
    }
  }
}
// the num_reviews filter; when it is the last filter it also reduces min / max of the ratings
template <bool kMinMax>
__global__ __launch_bounds__(256) void k_nl_filter(const int32_t *user, const double *rating, const int32_t *nl_list,
                                                   const int32_t *nl_count, const int32_t *cnt, int num_reviews,
                                                   uint8_t *keep, int32_t *tile_cnt, unsigned long long *mm) {
  unsigned long long lo = ~0ULL, hi = 0ULL;
  const int32_t count = *nl_count;
  for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
    const int32_t i = nl_list[t];
    if (!keep[i]) continue;
    if (cnt[user[i]] < num_reviews) {
      keep[i] = 0;
      atomicSub(&tile_cnt[i / kScanTile], 1);
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
  __shared__ long long part[16];
  const int per = (nb + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
  long long s = 0;
  for (int b = b0; b < b1; ++b) s += bsum[b];
  long long incl = s;  // inclusive scan over the wave, then over the 16 wave totals
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long long y = __shfl_up(incl, d, 64);
    if (lane_id() >= d) incl += y;
  }
  if (lane_id() == 63) part[threadIdx.x >> 6] = incl;
  __syncthreads();
  long long before = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += part[w];
  if (threadIdx.x == 1023) *total = before + incl;
  int run = (int)(before + incl - s);
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
constexpr int kEncIters = 4;  // a lane reads kEncIters x 4 consecutive rows, all loads issued before the first use
__global__ __launch_bounds__(256) void k_enc_first(const int32_t *id, int64_t n, int bound, int32_t *first, int32_t *err) {
  const int lane = lane_id();
  int4 q[kEncIters];
  int64_t i0[kEncIters];
#pragma unroll
  for (int k = 0; k < kEncIters; ++k) {
    i0[k] = (((int64_t)blockIdx.x * kEncIters + k) * 256 + threadIdx.x) * 4;
    q[k] = ld4(id, i0[k], n, -1);
  }
  bool bad = false;
#pragma unroll
  for (int k = 0; k < kEncIters; ++k) {
    int32_t v[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v[j] < 0 || v[j] >= bound) {
        bad |= i0[k] + j < n;
        v[j] = -1;
      }
    // A table grouped by user presents runs of equal ids: only the first row of a run goes to memory — as an
    // atomic without a return value (a few per wave, nothing waits for them).  A column without runs reads before
    // it updates (a stale read only costs a redundant atomic; after its first few rows an id costs no atomic at
    // all), the four reads in flight together.
    const int32_t left = __shfl_up(v[3], 1, 64);
    bool head[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) head[j] = v[j] >= 0 && (j ? v[j] != v[j - 1] : (lane == 0 || v[0] != left));
    if (__popcll(__ballot(head[0] | head[1] | head[2] | head[3])) <= kFewHeads) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (head[j]) atomicMin(&first[v[j]], (int32_t)(i0[k] + j));
    } else {
      int32_t f[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = first[head[j] ? v[j] : 0];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int32_t i = (int32_t)(i0[k] + j);
        if (head[j] && f[j] > i) atomicMin(&first[v[j]], i);
      }
    }
  }
  if (bad) *err = 1;
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
  int4 q[kEncIters];
  int64_t i0[kEncIters];
#pragma unroll
  for (int k = 0; k < kEncIters; ++k) {
    i0[k] = (((int64_t)blockIdx.x * kEncIters + k) * 256 + threadIdx.x) * 4;
    q[k] = ld4(id, i0[k], n, -1);
  }
#pragma unroll
  for (int k = 0; k < kEncIters; ++k) {
    const int32_t v[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
    int32_t r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (v[j] < 0 || v[j] >= bound) ? -1 : rank_of[v[j]];
    if (i0[k] + 3 < n) {
      *reinterpret_cast<int4 *>(idx + i0[k]) = make_int4(r[0], r[1], r[2], r[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i0[k] + j < n) idx[i0[k] + j] = r[j];
    }
  }
}

// Small id spaces (the anime column: 18 k ids without runs): a gather per row from a 72 KB global table misses the
// CU's vector cache nearly every time, and the column then moves at a tenth of what HBM delivers.  Up to
// kEncLdsIds ids the table lives in LDS instead: one workgroup per CU owns a contiguous, ascending range of the rows,
// finds its own first appearances with LDS atomics and merges the few that beat the global table at the end; the
// emit pass maps the column through an LDS copy of the rank table.
constexpr int kEncLdsIds = 32768;  // 128 KB of LDS
constexpr int kEncLdsThreads = 1024;
constexpr int kEncLdsIters = 4;    // 16-byte loads in flight per lane
__device__ __forceinline__ void enc_lds_range(int64_t n, int64_t &lo, int64_t &hi) {
  const int64_t groups = (n + 3) / 4;
  const int64_t per = (groups + gridDim.x - 1) / gridDim.x;
  const int64_t g0 = (int64_t)blockIdx.x * per, g1 = g0 + per;
  lo = (g0 < groups ? g0 : groups) * 4;
  hi = (g1 < groups ? g1 : groups) * 4;
}
__global__ __launch_bounds__(kEncLdsThreads) void k_enc_first_lds(const int32_t *id, int64_t n, int bound, int32_t *first,
                                                                  int32_t *err) {
  __shared__ int32_t tab[kEncLdsIds];
  for (int v = threadIdx.x; v < bound; v += kEncLdsThreads) tab[v] = 0x7FFFFFFF;
  __syncthreads();
  int64_t lo, hi;
  enc_lds_range(n, lo, hi);
  bool bad = false;
  for (int64_t i = lo + (int64_t)threadIdx.x * 4; i < hi; i += (int64_t)kEncLdsThreads * 4 * kEncLdsIters) {
    int4 q[kEncLdsIters];
#pragma unroll
    for (int k = 0; k < kEncLdsIters; ++k) {
      const int64_t ik = i + (int64_t)k * kEncLdsThreads * 4;
      q[k] = ik < hi ? ld4(id, ik, n, -1) : make_int4(-1, -1, -1, -1);
    }
#pragma unroll
    for (int k = 0; k < kEncLdsIters; ++k) {
      const int64_t ik = i + (int64_t)k * kEncLdsThreads * 4;
      const int32_t v[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (v[j] < 0 || v[j] >= bound)
          bad |= ik < hi && ik + j < n;
        else if (tab[v[j]] > (int32_t)(ik + j))  // (the range ascends: past its first rows an id never wins again — a
          atomicMin(&tab[v[j]], (int32_t)(ik + j));  //  plain LDS read instead of an atomic for nearly every row)
      }
    }
  }
  if (bad) *err = 1;
  __syncthreads();
  for (int v = threadIdx.x; v < bound; v += kEncLdsThreads) {
    const int32_t t = tab[v];
    if (t != 0x7FFFFFFF && first[v] > t) atomicMin(&first[v], t);
  }
}
__global__ __launch_bounds__(kEncLdsThreads) void k_enc_emit_lds(const int32_t *id, int64_t n, int bound,
                                                                 const int32_t *rank_of, int32_t *idx) {
  __shared__ int32_t tab[kEncLdsIds];
  for (int v = threadIdx.x; v < bound; v += kEncLdsThreads) tab[v] = rank_of[v];
  __syncthreads();
  int64_t lo, hi;
  enc_lds_range(n, lo, hi);
  for (int64_t i = lo + (int64_t)threadIdx.x * 4; i < hi; i += (int64_t)kEncLdsThreads * 4 * kEncLdsIters) {
    int4 q[kEncLdsIters];
#pragma unroll
    for (int k = 0; k < kEncLdsIters; ++k) {
      const int64_t ik = i + (int64_t)k * kEncLdsThreads * 4;
      q[k] = ik < hi ? ld4(id, ik, n, -1) : make_int4(-1, -1, -1, -1);
    }
#pragma unroll
    for (int k = 0; k < kEncLdsIters; ++k) {
      const int64_t ik = i + (int64_t)k * kEncLdsThreads * 4;
      if (ik >= hi) continue;
      const int32_t v[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
      int32_t r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = (v[j] < 0 || v[j] >= bound) ? -1 : tab[v[j]];
      if (ik + 3 < n) {
        *reinterpret_cast<int4 *>(idx + ik) = make_int4(r[0], r[1], r[2], r[3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ik + j < n) idx[ik + j] = r[j];
      }
    }
  }
}
// one workgroup per CU, fewer when the column is short (a workgroup then still has a few passes of its own)
static inline unsigned enc_lds_grid(int64_t n) {
  static int cus = 0;
  if (!cus) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
      v = 256;
    cus = v;
  }
  const int64_t per = (int64_t)kEncLdsThreads * 4 * kEncLdsIters;
  const int64_t want = (n + per - 1) / per;
  return (unsigned)(want < 1 ? 1 : want > cus ? cus : want);
}

// max of the two id columns in one pass (the sizes of the direct-index tables; a NULL is the smallest int).  Four
// quads of each column in flight per lane; one pair of atomics per workgroup (thousands of waves on two addresses
// serialise in the L2).
__global__ __launch_bounds__(256) void k_ing_id_max(const int32_t *user, const int32_t *anime, int64_t n, int32_t *out2) {
  __shared__ int32_t red[2][4];
  int32_t mu = ANIREC_NULL_I32, ma = ANIREC_NULL_I32;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += 4 * stride) {
    int4 u[4], a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t ik = i + k * stride;
      u[k] = ik < n ? ld4(user, ik, n, ANIREC_NULL_I32) : make_int4(ANIREC_NULL_I32, ANIREC_NULL_I32, ANIREC_NULL_I32, ANIREC_NULL_I32);
      a[k] = ik < n ? ld4(anime, ik, n, ANIREC_NULL_I32) : make_int4(ANIREC_NULL_I32, ANIREC_NULL_I32, ANIREC_NULL_I32, ANIREC_NULL_I32);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mu = max(mu, max(max(u[k].x, u[k].y), max(u[k].z, u[k].w)));
      ma = max(ma, max(max(a[k].x, a[k].y), max(a[k].z, a[k].w)));
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    mu = max(mu, __shfl_xor(mu, o, 64));
    ma = max(ma, __shfl_xor(ma, o, 64));
  }
  if (lane_id() == 0) red[0][threadIdx.x >> 6] = mu, red[1][threadIdx.x >> 6] = ma;
  __syncthreads();
  if (threadIdx.x < 2) {
    const int32_t *r = red[threadIdx.x];
    atomicMax(&out2[threadIdx.x], max(max(r[0], r[1]), max(r[2], r[3])));
  }
}

// every table a preprocess / an encode call starts from, in ONE launch each (seven / three fills otherwise: the
// launches between them are most of what a fill of a megabyte costs)
__global__ __launch_bounds__(256) void k_ing_init(int32_t *err, int32_t *cnt_u, int32_t *ufirst, int32_t *ulast, int ub,
                                                  int32_t *tile_cnt, int n_tiles, unsigned long long *mm, int32_t *nl_count) {
  const int stride = gridDim.x * 256;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < ub; v += stride) {
    cnt_u[v] = 0;
    ufirst[v] = 0x7FFFFFFF;
    ulast[v] = -1;
  }
  for (int t = blockIdx.x * 256 + threadIdx.x; t < n_tiles; t += stride) tile_cnt[t] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *err = 0;
    mm[0] = ~0ULL;  // min of the surviving ratings (order-preserving keys)
    mm[1] = 0ULL;   // max
    nl_count[0] = 0;
    nl_count[1] = 0;
  }
}
__global__ __launch_bounds__(256) void k_enc_init(int32_t *err, uint32_t *bits, size_t n_words, int32_t *first, int bound) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t x = (size_t)blockIdx.x * 256 + threadIdx.x; x < n_words; x += stride) bits[x] = 0u;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < (size_t)bound; v += stride) first[v] = 0x7FFFFFFF;
  if (blockIdx.x == 0 && threadIdx.x == 0) *err = 0;
}

static inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }
static inline size_t pad_tile(int64_t n) { return ((size_t)n + kChunk - 1) / kChunk * kChunk; }  // whole chunks
static inline uint32_t table_slots(int64_t n) {
  uint64_t s = 4096;  // = nl_slots(n), the list's worst case
  while (s < (uint64_t)n * 2) s <<= 1;
  return (uint32_t)s;
}
// workgroups of 256 lanes x iters quads of 4 rows
static inline unsigned quad_grid(int64_t n, int iters) { return (unsigned)((n + 1024 * iters - 1) / (1024 * iters)); }
static inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }
static inline int grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}
}  // namespace anirec

using namespace anirec;

extern "C" {

// keep (byte flags, whole chunks) | list of non-local rows | per-tile counts | hash table of the list | user counts,
// first row, last row | anime max | minmax, list length.  The list and its table are sized for the worst case
// (a table in random order: every row non-local); only the part the list needs is ever touched.
struct IngestWs {
  uint8_t *keep;
  int32_t *nl_list, *bsum;
  unsigned long long *table;
  int32_t *cnt_u, *ufirst, *ulast, *max_ep;
  unsigned long long *mm;
  int32_t *nl_count;
  size_t bytes;
};
static IngestWs carve_ingest(void *workspace, int64_t n, int32_t user_bound, int32_t anime_bound) {
  const size_t np = pad_tile(n);
  char *p = (char *)workspace;
  IngestWs w;
  w.keep = (uint8_t *)p;
  p += al256(np);
  w.nl_list = (int32_t *)p;
  p += al256((size_t)n * 4);
  w.bsum = (int32_t *)p;
  p += al256(np / kScanTile * 4);
  w.table = (unsigned long long *)p;
  p += al256((size_t)table_slots(n) * 8);
  w.cnt_u = (int32_t *)p;
  p += al256((size_t)user_bound * 4);
  w.ufirst = (int32_t *)p;
  p += al256((size_t)user_bound * 4);
  w.ulast = (int32_t *)p;
  p += al256((size_t)user_bound * 4);
  w.max_ep = (int32_t *)p;
  p += al256((size_t)anime_bound * 4);
  w.mm = (unsigned long long *)p;
  w.nl_count = (int32_t *)(p + 16);
  p += 256;
  w.bytes = (size_t)(p - (char *)workspace);
  return w;
}

int anirec_ingest_id_max(const int32_t *user_id, const int32_t *anime_id, int64_t n, int32_t *out_max2, void *stream) {
  if (!user_id || !anime_id || !out_max2 || n < 1 || n >= ((int64_t)1 << 30)) return ANIREC_EINVAL;
  if (!aligned16(user_id) || !aligned16(anime_id)) return ANIREC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ANIREC_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)out_max2, (int)0x80000000, 2, s));
  const int64_t want = (n + 4095) / 4096;
  hipLaunchKernelGGL(k_ing_id_max, dim3((unsigned)(want > 2048 ? 2048 : want)), dim3(256), 0, s, user_id, anime_id, n, out_max2);
  return (int)hipGetLastError();
}

size_t anirec_ingest_workspace_bytes(int64_t n, int32_t user_id_bound, int32_t anime_id_bound) {
  if (n < 1 || n >= ((int64_t)1 << 30) || user_id_bound < 1 || anime_id_bound < 1) return 0;
  return carve_ingest(nullptr, n, user_id_bound, anime_id_bound).bytes;
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
  if (!aligned16(user_id) || !aligned16(anime_id) || !aligned16(rating) || !aligned16(watching_status) ||
      !aligned16(watched_episodes) || !aligned16(workspace))
    return ANIREC_EINVAL;  // the columns are read 16 bytes at a time
  if (workspace_bytes < anirec_ingest_workspace_bytes(n, opts->user_id_bound, opts->anime_id_bound))
    return ANIREC_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const size_t np = pad_tile(n);
  const IngestWs w = carve_ingest(workspace, n, opts->user_id_bound, opts->anime_id_bound);
  const int n_tiles = (int)(np / kScanTile);
  const size_t ub = (size_t)opts->user_id_bound;

  const IngestCols c{user_id, anime_id, rating, watching_status, watched_episodes};
  const int g = grid_for(n);
  const int gl = 4096;  // the list kernels read the list length on the device
  {
    const size_t most = ub > (size_t)n_tiles ? ub : (size_t)n_tiles;
    const unsigned gi = (unsigned)((most + 255) / 256 > 2048 ? 2048 : (most + 255) / 256);
    hipLaunchKernelGGL(k_ing_init, dim3(gi), dim3(256), 0, s, err_flag, w.cnt_u, w.ufirst, w.ulast, (int)ub, w.bsum, n_tiles,
                       w.mm, w.nl_count);
  }
  hipLaunchKernelGGL(k_ing_span, dim3(quad_grid(n, kSpanIters)), dim3(256), 0, s, user_id, n, opts->user_id_bound,
                     w.ufirst, w.ulast);
  const FrontArgs fa{c,        n,        opts->drop_unwatched, opts->drop_plan, opts->user_id_bound, opts->anime_id_bound,
                     opts->num_reviews, n_tiles, w.ufirst, w.ulast, w.keep,
                     w.bsum,   w.nl_list, w.nl_count, w.mm,   err_flag};
  const dim3 gf((unsigned)(np / kChunk));
  if (!opts->drop_half_watched) {
    hipLaunchKernelGGL(k_ing_front<true>, gf, dim3(kFrontThreads), 0, s, fa);
  } else {
    hipLaunchKernelGGL(k_ing_front<false>, gf, dim3(kFrontThreads), 0, s, fa);
  }
  hipLaunchKernelGGL(k_nl_clear, dim3(gl), dim3(256), 0, s, w.table, w.nl_count);
  hipLaunchKernelGGL(k_nl_insert, dim3(gl), dim3(256), 0, s, c, w.nl_list, w.nl_count, w.table, w.keep, w.bsum);
  hipLaunchKernelGGL(k_nl_count, dim3(gl), dim3(256), 0, s, user_id, w.nl_list, w.nl_count, w.keep, w.cnt_u);
  if (!opts->drop_half_watched) {
    hipLaunchKernelGGL(k_nl_filter<true>, dim3(gl), dim3(256), 0, s, user_id, rating, w.nl_list, w.nl_count, w.cnt_u,
                       opts->num_reviews, w.keep, w.bsum, w.mm);
  } else {
    hipLaunchKernelGGL(k_nl_filter<false>, dim3(gl), dim3(256), 0, s, user_id, rating, w.nl_list, w.nl_count, w.cnt_u,
                       opts->num_reviews, w.keep, w.bsum, w.mm);
    // episodes can be any int32: start the per-anime maxima at INT32_MIN
    ANIREC_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)w.max_ep, (int)0x80000000, (size_t)opts->anime_id_bound, s));
    hipLaunchKernelGGL(k_ing_anime_max, dim3(g), dim3(256), 0, s, c, n, w.keep, w.max_ep);
    hipLaunchKernelGGL(k_ing_half_filter, dim3(g), dim3(256), 0, s, c, n, w.max_ep, w.keep);
    hipLaunchKernelGGL(k_ing_minmax, dim3(g > 2048 ? 2048 : g), dim3(256), 0, s, rating, n, w.keep, w.mm);
    hipLaunchKernelGGL(k_scan_reduce, dim3(n_tiles), dim3(256), 0, s, w.keep, w.bsum);  // the flags changed: recount
  }
  ANIREC_HIP_CHECK(hipGetLastError());
  // per-tile counts -> exclusive tile bases + total; the in-tile scan is k_ing_compact's
  hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, s, w.bsum, n_tiles, n_out);
  const IngestOut o{out_user_id, out_anime_id, out_rating, out_status, out_episodes};
  hipLaunchKernelGGL(k_ing_compact, dim3((unsigned)n_tiles), dim3(256), 0, s, c, n, w.keep, w.bsum, w.mm, o);
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
  const IngestWs w = carve_ingest(const_cast<void *>(workspace), n, opts->user_id_bound, opts->anime_id_bound);
  hipLaunchKernelGGL(k_ing_half_columns, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out_anime_id, n_out,
                     (const int32_t *)w.max_ep, out_max_eps, out_half_eps);
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
  if (!aligned16(id) || !aligned16(out_index) || !aligned16(workspace)) return ANIREC_EINVAL;  // 16-byte accesses
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
  const int gb = (id_bound + 255) / 256;
  const int nb = (int)(nw / kBitTile);
  {
    const size_t most = nw > (size_t)id_bound ? nw : (size_t)id_bound;
    const unsigned gi = (unsigned)((most + 255) / 256 > 2048 ? 2048 : (most + 255) / 256);
    hipLaunchKernelGGL(k_enc_init, dim3(gi), dim3(256), 0, s, err_flag, bits, nw, first, id_bound);
  }
  const bool lds = id_bound <= kEncLdsIds;  // the id tables fit a CU's LDS
  if (lds) {
    hipLaunchKernelGGL(k_enc_first_lds, dim3(enc_lds_grid(n)), dim3(kEncLdsThreads), 0, s, id, n, id_bound, first, err_flag);
  } else {
    hipLaunchKernelGGL(k_enc_first, dim3(quad_grid(n, kEncIters)), dim3(256), 0, s, id, n, id_bound, first, err_flag);
  }
  hipLaunchKernelGGL(k_enc_bits, dim3(gb), dim3(256), 0, s, first, id_bound, bits);
  hipLaunchKernelGGL(k_bits_reduce, dim3(nb), dim3(256), 0, s, bits, bsum);
  hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, s, bsum, nb, n_unique);
  hipLaunchKernelGGL(k_bits_apply, dim3(nb), dim3(256), 0, s, bits, bsum, wpre);
  hipLaunchKernelGGL(k_enc_rank, dim3(gb), dim3(256), 0, s, first, id_bound, bits, wpre, out_uniques);
  if (lds) {
    hipLaunchKernelGGL(k_enc_emit_lds, dim3(enc_lds_grid(n)), dim3(kEncLdsThreads), 0, s, id, n, id_bound, first, out_index);
  } else {
    hipLaunchKernelGGL(k_enc_emit, dim3(quad_grid(n, kEncIters)), dim3(256), 0, s, id, n, id_bound, first, out_index);
  }
  return (int)hipGetLastError();
}

}  // extern "C"
