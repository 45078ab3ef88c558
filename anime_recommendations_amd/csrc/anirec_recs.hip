// Favourites and user-based recommendations on the GPU: the consumer of the similar-users top-k
// (SURVEY.md §8(f) row 4).
//
// Replaces, batched over every user / every query,
//   user_recs/user_recs.py:377-404  fave_genres / fave_sources: a user's favourites are the anime they
//       rated at or above the 80th percentile of their OWN ratings (np.percentile, linear interpolation,
//       float64) — also similar_users/similar_users.py:203-256 get_fave_anime;
//   user_recs/user_recs.py:708-760  similar_user_recs: for a query user, count over its similar users how
//       often each anime is a favourite, drop the anime the query user has favourited itself, rank by count.
// The reference does this with a pandas filter + sort per user and a DataFrame ravel + value_counts per
// query.  Here: ratings are grouped by user (CSR built with one histogram + scan + scatter), one wave per
// user radix-selects the two order statistics np.percentile interpolates between, a pass over the ratings
// sets favourite bits ([n_users][n_anime/32] words), and one workgroup per query adds up its similar
// users' bit rows into an LDS count table and selects the top n.
// Integer / byte work, HBM-bound; bit-identical to NumPy for the thresholds and the favourite sets.
// Ranking ties (pandas value_counts leaves their order to an unstable sort): count desc, then the best
// (lowest) rank of a similar user holding the anime, then ascending anime index.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "anirec_dev.hpp"

namespace anirec {

__device__ __forceinline__ unsigned long long dkey(double d) {  // order-preserving, -0.0 == 0.0
  const unsigned long long b = (unsigned long long)__double_as_longlong(d + 0.0);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double dkey_inv(unsigned long long o) {
  const unsigned long long b = (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFULL) : ~o;
  return __longlong_as_double((long long)b);
}

// Per-user counts.  The rating table usually arrives grouped by user (the raw / preprocessed order): a wave
// adds one atomic per RUN of equal users among its 64 consecutive rows, and `unsorted` records whether any
// row breaks the non-decreasing order — if none does, the table already IS the CSR payload.
// (A lane reads 4 consecutive rows, kRecIters quads in flight: the pass is latency-bound otherwise.)
constexpr int kRecIters = 4;
__device__ __forceinline__ int4 ld4r(const int32_t *p, int64_t i, int64_t n, int32_t fill) {
  if (i + 3 < n) return *reinterpret_cast<const int4 *>(p + i);
  return make_int4(i < n ? p[i] : fill, i + 1 < n ? p[i + 1] : fill, i + 2 < n ? p[i + 2] : fill, fill);
}
__global__ __launch_bounds__(256) void k_rec_count(const int32_t *user, int64_t n, int n_users, int32_t *cnt,
                                                   int32_t *err, int32_t *unsorted) {
  const int lane = lane_id();
  int4 q[kRecIters];
  int32_t prev[kRecIters];  // the row before the lane-0 quad of the wave (the order test crosses wave edges)
  int64_t i0[kRecIters];
#pragma unroll
  for (int k = 0; k < kRecIters; ++k) {
    i0[k] = (((int64_t)blockIdx.x * kRecIters + k) * 256 + threadIdx.x) * 4;
    q[k] = ld4r(user, i0[k], n, -1);
    prev[k] = (lane == 0 && i0[k] > 0 && i0[k] < n) ? user[i0[k] - 1] : -1;
  }
  bool bad = false, uns = false;
#pragma unroll
  for (int k = 0; k < kRecIters; ++k) {
    int32_t v[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
    const int32_t raw3 = v[3];
    int32_t left_raw = __shfl_up(raw3, 1, 64);  // raw ids for the order test
    if (lane == 0) left_raw = prev[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = i0[k] + j < n;
      const int32_t before = j ? v[j - 1] : left_raw;
      if (live && (j || lane || i0[k] > 0) && v[j] < before) uns = true;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0[k] + j < n && (v[j] < 0 || v[j] >= n_users)) {
        bad = true;
        v[j] = -1;
      } else if (i0[k] + j >= n) {
        v[j] = -1;
      }
    // one atomic per RUN of equal users: lanes whose four rows share a user are chained across the wave (a user's
    // ~300 rows are ~78 such lanes), a lane that holds a boundary adds its own pieces
    const bool uni = v[0] >= 0 && v[0] == v[1] && v[1] == v[2] && v[2] == v[3];
    const int32_t key = uni ? v[0] : -2;
    const int32_t lkey = __shfl_up(key, 1, 64);
    const bool head = uni && (lane == 0 || lkey != key);
    const unsigned long long brk = __ballot(head || !uni);  // lanes at which a chain of equal lanes ends
    if (head) {
      const unsigned long long later = lane == 63 ? 0ULL : brk >> (lane + 1);
      const int lanes = later ? __ffsll((long long)later) : 64 - lane;
      atomicAdd(&cnt[key], 4 * lanes);
    }
    if (!uni) {
      int run = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (v[j] < 0) continue;
        ++run;
        if (j == 3 || v[j + 1] != v[j]) {
          atomicAdd(&cnt[v[j]], run);
          run = 0;
        }
      }
    }
  }
  if (bad) *err = 1;
  if (uns) *unsorted = 1;
}
// exclusive scan of the per-user counts -> row pointers, three launches: totals of tiles of 4 096 counts, a
// one-workgroup scan of the tile totals, the tiles again with their bases
constexpr int kRsTile = 4096;  // 256 lanes x 16 counts
__global__ __launch_bounds__(256) void k_rec_scan_reduce(const int32_t *cnt, int n_users, long long *tsum) {
  __shared__ long long ws[4];
  const int base = blockIdx.x * kRsTile + threadIdx.x * 16;
  long long sm = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) sm += base + j < n_users ? cnt[base + j] : 0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sm += __shfl_xor(sm, o, 64);
  if (lane_id() == 0) ws[threadIdx.x >> 6] = sm;
  __syncthreads();
  if (threadIdx.x == 0) tsum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
__global__ __launch_bounds__(1024) void k_rec_scan_spine(long long *tsum, int n_tiles, int64_t *ptr, int n_users) {
  __shared__ long long wtot[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long long carry = 0;
  for (int base = 0; base < n_tiles; base += 1024) {
    const int i = base + threadIdx.x;
    const long long v = i < n_tiles ? tsum[i] : 0;
    long long inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const long long y = __shfl_up(inc, o, 64);
      if (lane >= o) inc += y;
    }
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    long long before = carry, all = 0;
    for (int x = 0; x < 16; ++x) {
      before += x < w ? wtot[x] : 0;
      all += wtot[x];
    }
    if (i < n_tiles) tsum[i] = before + inc - v;
    carry += all;
    __syncthreads();
  }
  if (threadIdx.x == 0) ptr[n_users] = carry;
}
__global__ __launch_bounds__(256) void k_rec_scan_apply(const int32_t *cnt, int n_users, const long long *tsum,
                                                        int64_t *ptr, int64_t *cursor) {
  __shared__ long long ws[4];
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int base = blockIdx.x * kRsTile + threadIdx.x * 16;
  long long c[16], mine = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    c[j] = base + j < n_users ? cnt[base + j] : 0;
    mine += c[j];
  }
  long long inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const long long y = __shfl_up(inc, o, 64);
    if (lane >= o) inc += y;
  }
  if (lane == 63) ws[w] = inc;
  __syncthreads();
  long long run = tsum[blockIdx.x] + inc - mine;
  for (int x = 0; x < w; ++x) run += ws[x];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if (base + j < n_users) {
      ptr[base + j] = run;
      cursor[base + j] = run;
    }
    run += c[j];
  }
}
// CSR payload: a straight copy when the table is grouped by user, an atomic-cursor scatter otherwise
__global__ __launch_bounds__(256) void k_rec_scatter(const int32_t *user, const double *rating, int64_t n, int n_users,
                                                     int64_t *cursor, const int32_t *unsorted, double *csr_rating) {
  // grouped by user: the table already IS the CSR payload and k_rec_percentile reads it in place (out-of-range
  // users were excluded from the counts: positions stay aligned with ptr[] only for valid tables, and err_flag
  // reports the others)
  if (*unsorted == 0) return;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int u = user[i];
    if (u < 0 || u >= n_users) continue;
    const long long p = (long long)atomicAdd((unsigned long long *)&cursor[u], 1ULL);
    csr_rating[p] = rating[i];  // order inside a user's segment is irrelevant to an order statistic
  }
}

// np.percentile(x, pct) with the default 'linear' method, one wave per user:
//   pos = pct/100 * (n - 1); lo = floor(pos); t = pos - lo; a = x_(lo), b = x_(lo+1)
//   result = lerp(a, b, t) = a + (b - a) t, and b - (b - a)(1 - t) when t >= 0.5   (numpy _lerp)
// x_(lo) comes from an 8-pass MSB radix select on the order-preserving 64-bit key of the double
// (wave-private 256-bin histogram in LDS); x_(lo+1) is x_(lo) again if more copies of it follow, else the
// smallest larger value.  NaN ratings never reach here (the preprocess step drops them).
// Measured and dropped (round 2): keeping a user's keys in registers (one global read instead of nine), skipping
// the passes whose digit is shared by every key still in play and adding to the histogram through one leader lane
// per distinct digit — 1 455 us against 1 371 us for 350 k users: the kernel is VALU-issue-bound (PMC: 1 330 VALU +
// 930 SALU instructions per user-wave), not bound by the re-reads, which hit in L2.
// When the table is grouped by user (`*unsorted == 0`: the raw order) the wave reads the user's ratings in place (no
// CSR copy exists) and, with `fav` given, goes on to build the user's favourite-bit row in a wave-private LDS
// bitmap and writes the whole row with plain coalesced stores: no pass over the 109 M rows for the bits, no
// atomics to HBM, no zero-fill of the 788 MB bit matrix (FavFuse; k_rec_favbits then returns at once).
struct FavFuse {
  const double *rating;    // the table's own rating column
  const int32_t *anime;
  const int32_t *unsorted;
  uint32_t *fav;           // nullptr: thresholds only
  int wwords, n_anime;
  int32_t *err;
};
__global__ __launch_bounds__(256) void k_rec_percentile(const double *csr_rating, const int64_t *ptr, int n_users,
                                                        double pct, double *thr, FavFuse f) {
  __shared__ uint32_t hist[4][256];
  extern __shared__ uint32_t bitmaps[];  // [4][wwords] when the bits are fused
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int u = blockIdx.x * 4 + w;
  if (u >= n_users) return;  // whole wave leaves together; no block barrier below
  const bool grouped = *f.unsorted == 0;
  const bool fuse = grouped && f.fav != nullptr;
  if (grouped) csr_rating = f.rating;
  const long long s0 = ptr[u], n = ptr[u + 1] - s0;
  if (n <= 0) {
    if (lane == 0) thr[u] = __longlong_as_double(0x7FF8000000000000LL);  // no ratings: NaN, no favourites
    if (fuse)
      for (int x = lane; x < f.wwords; x += 64) f.fav[(size_t)u * f.wwords + x] = 0u;
    return;
  }
  const double pos = (pct / 100.0) * (double)(n - 1);
  const long long lo = (long long)floor(pos);
  const double t = pos - (double)lo;
  uint32_t *h = hist[w];
  unsigned long long prefix = 0, pmask = 0;
  long long want = lo;  // 0-based rank (ascending) still to locate inside the prefix
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    for (int b = lane; b < 256; b += 64) h[b] = 0;
    __builtin_amdgcn_wave_barrier();
    unsigned long long orv = 0ull, andv = ~0ull;  // OR / AND of the keys still in play, per lane
    bool mine = false;
    for (long long i = lane; i < n; i += 64) {
      const unsigned long long k = dkey(csr_rating[s0 + i]);
      if ((k & pmask) == prefix) {
        atomicAdd(&h[(k >> shift) & 255ull], 1u);
        orv |= k;
        andv &= k;
        mine = true;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // Every key still in play equal (ratings on a grid of 11 values: true after two passes)?  Then it is x_(lo)
    // and the remaining passes would only confirm its bytes one by one.
    {
      const unsigned long long has = __ballot(mine);
      const int first = __ffsll((long long)has) - 1;  // has != 0: want < the number of keys in play
      const unsigned long long k0 = (unsigned long long)__shfl((long long)orv, first, 64);
      const bool differs = mine && (orv != andv || orv != k0);
      if (__ballot(differs) == 0ull) {
        prefix = k0;
        break;
      }
    }
    // ascending digit walk by a wave scan: lane l covers digits 4l .. 4l+3
    uint32_t h4[4], run = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h4[j] = h[4 * lane + j];
      run += h4[j];
    }
    uint32_t inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(inc, o, 64);
      if (lane >= o) inc += y;
    }
    const unsigned long long reach = __ballot((long long)inc > want);
    const int src = __ffsll((long long)reach) - 1;  // reach != 0: want < n
    long long cum = (long long)(inc - run);
    int d = 4 * lane;
    if (lane == src) {
      int j = 0;
      for (; j < 3; ++j) {
        if (cum + (long long)h4[j] > want) break;
        cum += h4[j];
      }
      d = 4 * lane + j;
    }
    d = __shfl(d, src, 64);
    cum = __shfl(cum, src, 64);
    prefix |= (unsigned long long)d << shift;
    pmask |= 255ull << shift;
    want -= cum;
    __builtin_amdgcn_wave_barrier();
  }
  const unsigned long long ka = prefix;  // key of x_(lo)
  // copies of x_(lo) up to and including rank lo: want + 1 of them are needed to reach rank lo; if the
  // segment holds more, x_(lo+1) is the same value
  long long n_eq = 0;
  unsigned long long kmin_gt = ~0ull;
  for (long long i = lane; i < n; i += 64) {
    const unsigned long long k = dkey(csr_rating[s0 + i]);
    if (k == ka) ++n_eq;
    if (k > ka && k < kmin_gt) kmin_gt = k;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    n_eq += __shfl_xor(n_eq, o, 64);
    const unsigned long long y = __shfl_xor(kmin_gt, o, 64);
    kmin_gt = y < kmin_gt ? y : kmin_gt;
  }
  if (lane == 0) {
    const double a = dkey_inv(ka);
    double b = a;
    if (lo + 1 < n && n_eq <= want + 1) b = dkey_inv(kmin_gt);
    // numpy.lib.function_base._lerp (contraction off: plain IEEE double operations)
    double r;
    {
#pragma clang fp contract(off)
      const double diff = b - a;
      r = a + diff * t;
      if (t >= 0.5) r = b - diff * (1.0 - t);
      if (t == 0.0) r = a;  // (numpy: where(t == 0, a, ...) is implied by a + 0; kept explicit for -0.0)
    }
    thr[u] = r;
    if (fuse) hist[w][0] = 0u, reinterpret_cast<double *>(&hist[w][2])[0] = r;  // hand the threshold to the wave
  }
  if (!fuse) return;
  __builtin_amdgcn_wave_barrier();
  const double cut = reinterpret_cast<const double *>(&hist[w][2])[0];
  uint32_t *bm = bitmaps + (size_t)w * f.wwords;
  for (int x = lane; x < f.wwords; x += 64) bm[x] = 0u;
  __builtin_amdgcn_wave_barrier();
  bool bad = false;
  for (long long i = lane; i < n; i += 64) {
    const int a = f.anime[s0 + i];
    if (a < 0 || a >= f.n_anime) {
      bad = true;
      continue;
    }
    if (csr_rating[s0 + i] >= cut) atomicOr(&bm[a >> 5], 1u << (a & 31));  // rating >= percentile (user_recs.py:394)
  }
  if (bad) *f.err = 1;
  __builtin_amdgcn_wave_barrier();
  for (int x = lane; x < f.wwords; x += 64) f.fav[(size_t)u * f.wwords + x] = bm[x];
}

// favourite bits: rating >= the user's own threshold (user_recs.py:394 `watched.rating >= percentile`)
// zero-fill of the bit matrix for the atomicOr path: only a table that is NOT grouped by user needs it
__global__ __launch_bounds__(256) void k_rec_clear(uint4 *fav16, size_t n16, const int32_t *unsorted, int fused) {
  if (fused && *unsorted == 0) return;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    fav16[i] = make_uint4(0u, 0u, 0u, 0u);
}
__global__ __launch_bounds__(256) void k_rec_favbits(const int32_t *user, const int32_t *anime, const double *rating,
                                                     int64_t n, int n_users, int n_anime, const double *thr,
                                                     uint32_t *fav, int wwords, int32_t *err, const int32_t *unsorted,
                                                     int fused) {
  if (fused && *unsorted == 0) return;  // k_rec_percentile wrote the rows
  constexpr int kIt = 2;
  bool bad = false;
  for (int64_t blk = blockIdx.x; blk * (1024 * kIt) < n; blk += gridDim.x) {
  int4 U[kIt], A[kIt];
  double R[kIt][4];
  int64_t i0[kIt];
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    i0[k] = ((blk * kIt + k) * 256 + threadIdx.x) * 4;
    U[k] = ld4r(user, i0[k], n, -1);
    A[k] = ld4r(anime, i0[k], n, 0);
    if (i0[k] + 3 < n) {
      const double2 r0 = *reinterpret_cast<const double2 *>(rating + i0[k]);
      const double2 r1 = *reinterpret_cast<const double2 *>(rating + i0[k] + 2);
      R[k][0] = r0.x, R[k][1] = r0.y, R[k][2] = r1.x, R[k][3] = r1.y;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) R[k][j] = i0[k] + j < n ? rating[i0[k] + j] : 0.0;
    }
  }
#pragma unroll
  for (int k = 0; k < kIt; ++k) {
    const int32_t u[4] = {U[k].x, U[k].y, U[k].z, U[k].w}, a[4] = {A[k].x, A[k].y, A[k].z, A[k].w};
    double t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = thr[(u[j] >= 0 && u[j] < n_users) ? u[j] : 0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i0[k] + j >= n || u[j] < 0 || u[j] >= n_users) continue;
      if (a[j] < 0 || a[j] >= n_anime) {
        bad = true;
        continue;
      }
      if (R[k][j] >= t[j]) atomicOr(&fav[(size_t)u[j] * wwords + (a[j] >> 5)], 1u << (a[j] & 31));
    }
  }
  }
  if (bad) *err = 1;
}

// One workgroup per query: counts[a] = number of similar users holding a as a favourite (a not a favourite
// of the query user), best[a] = lowest rank of such a similar user; then the top n by
// (count desc, best rank asc, anime index asc) with a radix-free threshold search (counts <= k_sim).
struct RecsArgs {
  const uint32_t *fav;
  int n_users, n_anime, wwords;
  const int32_t *query;  // [nq]
  const int32_t *sim;    // [nq][k_sim], -1 = empty slot, best first
  int nq, k_sim, n_recs;
  int32_t *out_anime;    // [nq][n_recs], -1 padded
  int32_t *out_count;    // [nq][n_recs], 0 padded
};
constexpr int kRecsMaxSim = 64;
constexpr int kRecsMaxOut = 256;

// Everything stays BIT-SLICED in registers: a lane owns kR words of 32 anime; plane p of `cp` holds bit p of the
// count of each of a word's 32 anime, so one similar user's 32 favourite bits are added to all 32 counts with a
// 6-step carry chain, and `bp` holds, the same way, the rank of the FIRST similar user holding the anime (they come
// best first).  The selection never unpacks them: "count == c", "count > cut", "best < cut_b" are bitwise
// comparators over the planes, histograms are popcounts of those masks, and only the <= n_recs winners are
// turned into (count, best, anime) entries.  (The first version wrote a 2-byte key per anime to LDS and walked
// the 18 k of them four times: 69 us per query, 0.046 of the HBM rate of the 11 bit rows it reads.)
//
// gt / eq masks of the kP-plane numbers of a word against the scalar v
template <int kP>
__device__ __forceinline__ void cmp_planes(const uint32_t (&pl)[kP], uint32_t v, uint32_t &gt, uint32_t &eq) {
  gt = 0u;
  eq = ~0u;
#pragma unroll
  for (int p = kP - 1; p >= 0; --p) {
    const uint32_t vb = ((v >> p) & 1u) ? ~0u : 0u;
    gt |= eq & pl[p] & ~vb;
    eq &= ~(pl[p] ^ vb);
  }
}
template <int kP>
__device__ __forceinline__ uint32_t plane_value(const uint32_t (&pl)[kP], int b) {
  uint32_t v = 0;
#pragma unroll
  for (int p = 0; p < kP; ++p) v |= ((pl[p] >> b) & 1u) << p;
  return v;
}
// sum of `n` over the workgroup's 256 lanes, the same value in every lane: a wave sum, four LDS words, ONE barrier
// (the caller alternates between two sets of words, so the next sum may start before every wave has read this one)
__device__ __forceinline__ int block_total(int n, int *slot4) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0) slot4[threadIdx.x >> 6] = n;
  __syncthreads();
  return slot4[0] + slot4[1] + slot4[2] + slot4[3];
}

// kR words per lane, kP bit planes per number (4 while k_sim <= 15: the carry chains and comparators are
// what the kernel spends its time on)
template <int kR, int kP>
__global__ __launch_bounds__(256) void k_user_recs(RecsArgs a) {
  __shared__ uint32_t win[kRecsMaxOut];  // winners: count (7 bits) | 63 - best (8 bits) | anime (17 bits)
  __shared__ int sims[kRecsMaxSim];
  __shared__ int n_win;
  __shared__ int wsum[4], tot[2][4];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int qu = a.query[q];
  if (tid < a.k_sim) {
    const int su = a.sim[(size_t)q * a.k_sim + tid];
    sims[tid] = (su < 0 || su >= a.n_users) ? -1 : su;
  }
  if (tid == 0) n_win = 0;
  __syncthreads();

  uint32_t cp[kR][kP], bp[kR][kP], ok[kR];
  {
    uint32_t seen[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      seen[r] = 0u;
#pragma unroll
      for (int p = 0; p < kP; ++p) cp[r][p] = 0u, bp[r][p] = 0u;
    }
    // the words of kB similar users are requested together (an empty slot adds zeros): a workgroup's 11 rows are
    // scattered over the 350 k-user bit matrix, so the loop is bound by how many row reads are in flight
    constexpr int kB = kR <= 3 ? 10 : (kR == 4 ? 8 : (kR == 8 ? 4 : 2));
    for (int j0 = 0; j0 < a.k_sim; j0 += kB) {
      uint32_t batch[kB][kR];
#pragma unroll
      for (int jj = 0; jj < kB; ++jj) {
        const int su = j0 + jj < a.k_sim ? sims[j0 + jj] : -1;
        const uint32_t *row = a.fav + (size_t)(su < 0 ? 0 : su) * a.wwords;
#pragma unroll
        for (int r = 0; r < kR; ++r) {
          const int w = r * 256 + tid;
          batch[jj][r] = (su >= 0 && w < a.wwords) ? row[w] : 0u;
        }
      }
#pragma unroll
      for (int jj = 0; jj < kB; ++jj) {
        const int j = j0 + jj;
        uint32_t bits[kR];
#pragma unroll
        for (int r = 0; r < kR; ++r) bits[r] = batch[jj][r];
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        uint32_t carry = bits[r];
#pragma unroll
        for (int p = 0; p < kP; ++p) {
          const uint32_t t = cp[r][p] & carry;
          cp[r][p] ^= carry;
          carry = t;
        }
        const uint32_t fresh = bits[r] & ~seen[r];
        seen[r] |= bits[r];
#pragma unroll
        for (int p = 0; p < kP; ++p)
          if ((j >> p) & 1) bp[r][p] |= fresh;
      }
      }
    }
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      const int w = r * 256 + tid;
      const uint32_t own = (qu >= 0 && qu < a.n_users && w < a.wwords) ? a.fav[(size_t)qu * a.wwords + w] : 0u;
      const int left = a.n_anime - w * 32;  // bits of the last word past n_anime are not anime
      const uint32_t valid = left >= 32 ? ~0u : (left > 0 ? (1u << left) - 1u : 0u);
      ok[r] = seen[r] & ~own & valid;
    }
  }
  // level 1: the count value `cut` at which the top n_recs end = the largest c with T(c) >= n_recs, where
  // T(c) = #{candidates with count >= c} falls with c: a binary search of block-wide popcounts, every lane
  // following the same path (no serial walk over a histogram).  cut == 0: fewer than n_recs candidates, all taken.
  int step = 0;
  auto count_ge = [&](int c) {  // T(c), c >= 1
    int n = 0;
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      uint32_t gt, eq;
      cmp_planes<kP>(cp[r], (uint32_t)(c - 1), gt, eq);
      n += __popc(gt & ok[r]);
    }
    return block_total(n, tot[step++ & 1]);
  };
  int cut = 0, above = 0;  // above = T(cut + 1): candidates with count > cut
  {
    int lo = 0, hi = a.k_sim + 1;  // T(lo) >= n_recs (T(0) = everything), T(hi) < n_recs (T(k_sim + 1) = 0)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      const int t = count_ge(mid);
      if (t >= a.n_recs) {
        lo = mid;
      } else {
        hi = mid;
        above = t;
      }
    }
    cut = lo;
  }
  const int room = a.n_recs - above;  // entries still to take among those with count == cut
  // level 2: among count == cut, the best similar-user rank `cut_b` at which they end = the smallest b with
  // B(b) = #{count == cut, best <= b} >= room; those with best < cut_b are taken, of those at cut_b the first room2
  uint32_t sure[kR], atcut[kR];
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    uint32_t gt, eq;
    cmp_planes<kP>(cp[r], (uint32_t)cut, gt, eq);
    sure[r] = cut == 0 ? ok[r] : (gt & ok[r]);
    atcut[r] = cut == 0 ? 0u : (eq & ok[r]);
  }
  int cut_b = 0, room2 = 0;
  bool scan = false;
  if (cut > 0) {
    auto best_le = [&](int b) {
      int n = 0;
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        uint32_t g2, e2;
        cmp_planes<kP>(bp[r], (uint32_t)b, g2, e2);
        n += __popc(~g2 & atcut[r]);
      }
      return block_total(n, tot[step++ & 1]);
    };
    int lo = -1, hi = a.k_sim - 1, below = 0, upto = -1;  // B(lo) < room, B(hi) >= room (B(k_sim - 1) = all of them)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      const int t = best_le(mid);
      if (t >= room) {
        hi = mid;
        upto = t;
      } else {
        lo = mid;
        below = t;
      }
    }
    if (upto < 0) upto = best_le(hi);  // hi never moved
    cut_b = hi;
    room2 = room - below;
    scan = upto - below > room2;  // more ties at (cut, cut_b) than room: the anime index decides
  }
  // level 3: exactly min(n_recs, #candidates) winners
  uint32_t tie[kR];
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    uint32_t g2, e2;
    cmp_planes<kP>(bp[r], (uint32_t)cut_b, g2, e2);
    sure[r] |= atcut[r] & ~g2 & ~e2;  // count == cut and best < cut_b
    tie[r] = atcut[r] & e2;
  }
  if (scan) {  // the first room2 ties in anime-index order: words ascend with r, then with the lane
    int seen_ties = 0;
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      const int mine = __popc(tie[r]);
      int incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d, 64);
        if (lane >= d) incl += y;
      }
      if (lane == 63) wsum[tid >> 6] = incl;
      __syncthreads();
      int before = seen_ties + incl - mine;
      for (int x = 0; x < (tid >> 6); ++x) before += wsum[x];
      seen_ties += wsum[0] + wsum[1] + wsum[2] + wsum[3];
      __syncthreads();
      int room = room2 - before;  // ties of this word still to take, lowest bits first
      uint32_t t = tie[r], takem = 0u;
      while (t && room > 0) {
        const uint32_t low = t & (0u - t);
        takem |= low;
        t ^= low;
        --room;
      }
      tie[r] = takem;
    }
  }
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    uint32_t take = sure[r] | tie[r];
    while (take) {
      const int b = __ffs(take) - 1;
      take &= take - 1u;
      const int pos = atomicAdd(&n_win, 1);
      if (pos < kRecsMaxOut)
        win[pos] = (plane_value<kP>(cp[r], b) << 25) | ((63u - plane_value<kP>(bp[r], b)) << 17) | (uint32_t)((r * 256 + tid) * 32 + b);
    }
  }
  for (int i = tid; i < a.n_recs; i += 256) {
    a.out_anime[(size_t)q * a.n_recs + i] = -1;
    a.out_count[(size_t)q * a.n_recs + i] = 0;
  }
  __syncthreads();
  const int n_out = n_win > kRecsMaxOut ? kRecsMaxOut : n_win;
  // exact order of the n_out (<= n_recs) winners: (count, 63 - best) descending, anime index ascending
  for (int i = tid; i < n_out; i += 256) {
    const uint32_t me = win[i];
    const uint32_t mk = me >> 17, ma = me & 0x1FFFFu;
    int rank = 0;
    for (int j = 0; j < n_out; ++j) {
      const uint32_t o = win[j];
      const uint32_t okk = o >> 17, oa = o & 0x1FFFFu;
      rank += (okk > mk || (okk == mk && oa < ma)) ? 1 : 0;
    }
    if (rank < a.n_recs) {
      a.out_anime[(size_t)q * a.n_recs + rank] = (int32_t)ma;
      a.out_count[(size_t)q * a.n_recs + rank] = (int32_t)(me >> 25);
    }
  }
}

static inline size_t al256r(size_t x) { return (x + 255) / 256 * 256; }
static inline int grid_rec(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace anirec

using namespace anirec;

extern "C" {

// cnt | ptr | cursor | csr_rating | 256 spare bytes | tile totals of the scan
static inline size_t scan_tiles(int32_t n_users) { return ((size_t)n_users + kRsTile - 1) / kRsTile; }
size_t anirec_fav_workspace_bytes(int64_t n_ratings, int32_t n_users) {
  if (n_ratings < 1 || n_users < 1) return 0;
  return al256r((size_t)n_users * 4) + 2 * al256r(((size_t)n_users + 1) * 8) + al256r((size_t)n_ratings * 8) + 256 +
         al256r(scan_tiles(n_users) * 8);
}

int anirec_user_favourites(const int32_t *user_idx, const int32_t *anime_idx, const double *rating, int64_t n,
                           int32_t n_users, int32_t n_anime, double percentile, uint32_t *fav_bits,
                           double *threshold, int32_t *err_flag, void *workspace, size_t workspace_bytes,
                           void *stream) {
  if (!user_idx || !anime_idx || !rating || !fav_bits || !threshold || !err_flag || !workspace) return ANIREC_EINVAL;
  if (n < 1 || n_users < 1 || n_anime < 1 || !(percentile >= 0.0 && percentile <= 100.0)) return ANIREC_EINVAL;
  if (workspace_bytes < anirec_fav_workspace_bytes(n, n_users)) return ANIREC_EWORKSPACE;
  if (((uintptr_t)user_idx | (uintptr_t)anime_idx | (uintptr_t)rating | (uintptr_t)workspace) & 15)
    return ANIREC_EINVAL;  // the columns are read 16 bytes at a time
  hipStream_t s = (hipStream_t)stream;
  char *p = (char *)workspace;
  int32_t *cnt = (int32_t *)p;
  p += al256r((size_t)n_users * 4);
  int64_t *ptr = (int64_t *)p;
  p += al256r(((size_t)n_users + 1) * 8);
  int64_t *cursor = (int64_t *)p;
  p += al256r(((size_t)n_users + 1) * 8);
  double *csr = (double *)p;
  const int wwords = (n_anime + 31) / 32;
  int32_t *unsorted = (int32_t *)((char *)csr + al256r((size_t)n * 8));  // the spare 256 B
  long long *tsum = (long long *)((char *)unsorted + 256);
  const int n_tiles = (int)scan_tiles(n_users);
  const unsigned gq = (unsigned)((n + 1024 * kRecIters - 1) / (1024 * kRecIters));
  ANIREC_HIP_CHECK(hipMemsetAsync(err_flag, 0, 4, s));
  ANIREC_HIP_CHECK(hipMemsetAsync(unsorted, 0, 4, s));
  ANIREC_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)n_users * 4, s));
  const int g = grid_rec(n);
  // the bit rows are built inside k_rec_percentile when the table is grouped by user and a row fits the LDS budget
  const int fused = (size_t)wwords * 16 <= 48 * 1024 ? 1 : 0;
  const size_t fav_words = (size_t)n_users * wwords;
  hipLaunchKernelGGL(k_rec_count, dim3(gq), dim3(256), 0, s, user_idx, n, n_users, cnt, err_flag, unsorted);
  hipLaunchKernelGGL(k_rec_scan_reduce, dim3(n_tiles), dim3(256), 0, s, cnt, n_users, tsum);
  hipLaunchKernelGGL(k_rec_scan_spine, dim3(1), dim3(1024), 0, s, tsum, n_tiles, ptr, n_users);
  hipLaunchKernelGGL(k_rec_scan_apply, dim3(n_tiles), dim3(256), 0, s, cnt, n_users, tsum, ptr, cursor);
  if ((fav_words & 3) == 0 && (((uintptr_t)fav_bits) & 15) == 0) {
    hipLaunchKernelGGL(k_rec_clear, dim3(4096), dim3(256), 0, s, (uint4 *)fav_bits, fav_words / 4, unsorted, fused);
  } else {
    ANIREC_HIP_CHECK(hipMemsetAsync(fav_bits, 0, fav_words * 4, s));
  }
  hipLaunchKernelGGL(k_rec_scatter, dim3(g), dim3(256), 0, s, user_idx, rating, n, n_users, cursor, unsorted, csr);
  const FavFuse ff{rating, anime_idx, unsorted, fused ? fav_bits : nullptr, wwords, n_anime, err_flag};
  hipLaunchKernelGGL(k_rec_percentile, dim3((n_users + 3) / 4), dim3(256), fused ? (size_t)wwords * 16 : 0, s, csr, ptr,
                     n_users, percentile, threshold, ff);
  const int64_t fb = (n + 2047) / 2048;
  hipLaunchKernelGGL(k_rec_favbits, dim3((unsigned)(fb < 8192 ? fb : 8192)), dim3(256), 0, s, user_idx, anime_idx, rating, n,
                     n_users, n_anime, threshold, fav_bits, wwords, err_flag, unsorted, fused);
  return (int)hipGetLastError();
}

int anirec_user_recs(const uint32_t *fav_bits, int32_t n_users, int32_t n_anime, const int32_t *query_users,
                     const int32_t *sim_users, int32_t nq, int32_t k_sim, int32_t n_recs, int32_t *out_anime,
                     int32_t *out_count, void *stream) {
  if (!fav_bits || !query_users || !sim_users || !out_anime || !out_count) return ANIREC_EINVAL;
  if (n_users < 1 || n_anime < 1 || n_anime >= (1 << 17) || nq < 0 || k_sim < 1 || k_sim > kRecsMaxSim - 1 ||
      n_recs < 1 || n_recs > kRecsMaxOut)
    return ANIREC_EINVAL;
  if (nq == 0) return ANIREC_OK;
  const int wwords = (n_anime + 31) / 32;
  RecsArgs a;
  a.fav = fav_bits;
  a.n_users = n_users;
  a.n_anime = n_anime;
  a.wwords = wwords;
  a.query = query_users;
  a.sim = sim_users;
  a.nq = nq;
  a.k_sim = k_sim;
  a.n_recs = n_recs;
  a.out_anime = out_anime;
  a.out_count = out_count;
  // a lane holds kR words of 32 anime in registers, as kP bit planes
#define ANIREC_RECS(R)                                                                              \
  do {                                                                                              \
    if (k_sim <= 15)                                                                                \
      hipLaunchKernelGGL((k_user_recs<R, 4>), dim3(nq), dim3(256), 0, (hipStream_t)stream, a);      \
    else                                                                                            \
      hipLaunchKernelGGL((k_user_recs<R, 6>), dim3(nq), dim3(256), 0, (hipStream_t)stream, a);      \
  } while (0)
  if (wwords <= 256) {
    ANIREC_RECS(1);
  } else if (wwords <= 2 * 256) {
    ANIREC_RECS(2);
  } else if (wwords <= 3 * 256) {
    ANIREC_RECS(3);  // 17 560 anime: 549 words
  } else if (wwords <= 4 * 256) {
    ANIREC_RECS(4);
  } else if (wwords <= 8 * 256) {
    ANIREC_RECS(8);
  } else {
    ANIREC_RECS(16);  // n_anime < 2^17
  }
#undef ANIREC_RECS
  return (int)hipGetLastError();
}

}  // extern "C"
