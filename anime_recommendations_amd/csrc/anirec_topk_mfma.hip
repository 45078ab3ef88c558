// All-pairs cosine top-k on the gfx950 matrix cores.
//
// Replaces the reference's per-query  np.dot(W_hat, W_hat[q]) + np.argsort  (similar_anime.py:404-408,
// similar_users.py:293-296) at BASELINE scale (every row a query, top-k of 18 k / 350 k rows) without
// ever materialising the n x n score matrix.
//
// Three kernels (+ k_scatter_log / k_merge_inbox of the all-pairs job, prior_mode 3):
//   k_cand    one SUPER-STEP of the key stream: v_mfma_f32_16x16x32_f16 scores of 128 queries x a range
//             of key tiles per workgroup (fp16 operands: the bf16 MFMA rate with 8x smaller rounding
//             error; the 16x16x32 shape because the chip holds a higher clock on it).  Each query row
//             has a candidate buffer in HBM/L2 and, in registers, a count and a threshold theta that is
//             FIXED for the launch and folded into the MFMA accumulator (C-in = -theta): a candidate
//             is `acc >= 0`; ranks come from wave ballots (no atomics).  The filter of one 32x32 block
//             runs under the MFMAs of the next (software pipeline inside each wave).
//   k_refresh between super-steps, one wave per row: tau = k-th largest MFMA score seen so far is a
//             lower bound of the final one, so theta = tau - 2 eps keeps every key that can still reach
//             the exact top-k; the buffer is compacted.  Super-steps double the keys seen, so a row
//             gains ~k candidates per super-step and no MFMA workgroup ever waits for a compaction.
//   k_rerank  per query: tau = k-th largest MFMA score; MFMA scores of unit vectors are within
//             eps = 2^-10 (+ fp32 accumulation) of the exact score, so the exact top-k is contained in
//             {MFMA score >= tau - 2 eps} — complete because theta <= tau - 2 eps.  Its members
//             are re-scored with the DEFINED fp32 fma chain and ranked (ties: ascending index).
//             Rows whose window overflowed (dense near-duplicates) are flagged and the caller re-runs
//             them through the exact kernels.
// Result: bit-exact neighbour lists at matrix-core speed.
//
// Measured and dropped in round 2 (350 k keys x 65 536 queries, k = 100, same box, interleaved):
//   * 64 query rows per wave (4 row blocks, 240 VGPRs, every B fragment feeding four MFMAs, half the tile-loop
//     overhead per flop): 15.1-15.8 ms per 131 072 queries against 15.4 ms — no gain;
//   * a sampled key schedule (thresholds from a 10 % strided sample, the other 90 % in ONE launch): appends per
//     row 1 429 -> 530, k_refresh 0.58 -> 0.23 ms, k_cand 7.06 -> 6.82 ms — the cost of the filter is how often
//     a 16x16 block holds ANY candidate (~18 % of the blocks at k = 100, set by the k-th-best window itself),
//     not how many candidates are appended; and ~4e-4 of the rows then need the exact fallback.
//   * a ring of 4 key tiles (128 KB of LDS) with a LAGGED barrier (at its sync point of tile t a wave waits, on
//     LDS counters, for every wave to have passed the sync point of tile t-1, so a wave that met candidates does
//     not stall the other seven): bit-identical lists, cycles per wave and tile 3 892 -> 3 644 (-6 %), wait at the
//     sync point 752 -> 632 — and the same wall time to 1 % in interleaved same-box rounds (k = 100: 6.43-6.49 ms
//     against 6.41-6.57; k = 10: 5.21-5.24 against 5.17-5.22): the chip answers fewer cycles with a lower clock.
//     The kernel is POWER-limited; only less energy per MFMA would make it faster.
//   * exploiting the symmetry of the all-pairs job (score(q, k) also is a candidate for row k: half the MFMAs):
//     emulated by adding a column-direction test (one compare per 16x16 block against the key rows' thresholds,
//     then per-register tests and an LDS-staged append on a hit) to every tile of the present kernel — k_cand
//     6.6 -> 13.5 ms at k = 100 and 5.3 -> 8.2 ms at k = 10, i.e. at best 0.98x / 1.29x after halving the tiles.
//     Not built: the second filter costs what the saved MFMAs are worth.
//     (Round 3 built it WITHOUT a second filter — see CandArgs / SymPlan / anirec_cosine_topk_allpairs_plan below: the
//     symmetry is used ACROSS query batches only, where every row of a later batch still sits at the job's learnt
//     prior, so ONE test against the prior serves both rows; the pairs for the key's row leave through per-wave logs
//     (k_scatter_log -> inboxes -> k_merge_inbox / the re-rank).  350 k x 350 k top-100: 31.5 -> 24.9 ms.)
//   Where the k = 100 time goes (ANIREC_TOPK_DEBUG=16, same 65 536-query slice, same box): no filter at all 4.2 ms;
//   the filter with thresholds nothing passes 4.4-4.5 ms; ONE launch with the FINAL thresholds handed in (a row
//   appends only its k + window candidates) 5.0-5.6 ms; the real schedule — 11 launches, each threshold the k-th
//   best of the keys seen so far, ~950 appends per row instead of ~110 — 6.6 ms.  The extra ~850 appends are what
//   an exact sequential top-k pays (k (g - 1) per super-step of growth g, k ln(n / n0) in the limit g -> 1); a
//   conservative threshold from a key sample does not beat it (measured above, and by the same arithmetic).
//   In-kernel stamps: 3 271 (k = 10) / 4 066 (k = 100) cycles per wave and tile against an MFMA floor of 2 048
//   (two waves per SIMD); vmcnt wait at the tile barrier 42-63 cycles (the candidate stores do not stall it),
//   barrier skew 470-790.  Without any filter the loop runs at 1.37 PFLOP/s on that box (0.55 of the peak).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "anirec_dev.hpp"

namespace anirec {

// query rows per workgroup: 32 per wave x kWaves (4 or 8) waves, template parameter of k_cand
constexpr int kBN = 128;        // keys per tile
constexpr int kCap = 512;       // candidate buffer entries per query row
// Key-range SPLITS (small query sets: fewer workgroups than CUs): gridDim.y workgroups share a row block and a
// super-step's key tiles; split s appends into its own region of the row's buffer, [kKept + s kReg, + kReg),
// with its own count (cnt2), and k_refresh folds the regions behind the kept entries [0, cnt).
constexpr int kKept = 160, kReg = 88, kMaxSplit = (kCap - kKept) / kReg;  // 160 + 4 x 88 = 512
static_assert(kMaxSplit == 4, "cnt2 rows are read as one int4");
// |fp16-operand MFMA score - fp32 fma-chain score| for unit-norm rows: each operand is rounded with
// unit roundoff 2^-11 (fp16 has 11 significant bits: 8x tighter than bf16's 2^-8, at the same MFMA
// rate), so a product is off by <= 2^-10 (1 + 2^-12) of |q_k w_k| and sum |q_k w_k| <= 1; fp16
// subnormal components (< 6.1e-5) add <= 2^-25 * sum|w_k| <= 3.4e-7; the two fp32 accumulations add
// <= 2 * 128 * 2^-24 = 1.6e-5; the threshold bias folded into the accumulator adds <= 5e-7.
// 0.000977 + 0.000017 < 0.00101.
constexpr float kEpsMfma = 0.00101f;
// The k-th largest score of a row is found by a bitwise radix select over order-preserving keys.  Every use of it is a
// LOWER bound (a threshold below it, a survivor window below it), so the select stops kSelLow bits early: the prefix
// with its low bits cleared is <= the true k-th key, off by < 2^(kSelLow-23) relative (1.2e-4 at 0.3, an eighth of eps),
// for 20 ballot steps per row instead of 32.  (Walking the row's proven threshold bits for free — no ballot at its
// 1-bits while the prefix still equals it — was measured too: the data-dependent branch in the loop cost more than the
// 7-9 ballots it saved, 0.64 vs 0.62 ms on the 18 k job.)
constexpr int kSelLow = 12;
constexpr float kThetaInit = -4.0f;  // below every cosine; finite so that (score - theta) stays finite

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t f2key(float s) {  // order preserving, NaN -> 0 (never selected)
  if (s != s) return 0u;
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return u == 0u ? 1u : u;
}
__device__ __forceinline__ float key2f(uint32_t u) {
  const uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  return __uint_as_float(b);
}

// fp32 [rows][128] (optionally gathered through `rows`) -> fp16, round to nearest even
// Key rows excluded by the caller's mask become NaN rows: their scores are NaN, which never compares
// >= 0 and is ignored by v_max — the filter needs no per-key test (and no load) in the hot loop.
// rows n .. n_pad-1 of the output are NaN rows (key-tile padding).
// unnorm: set to 1 when a finite row is not unit-norm (| sum x^2 - 1 | > 1e-3): the error window of the MFMA
// scores (kEpsMfma) is proven for unit vectors only, so k_rerank then sends every query to the exact path.
__global__ __launch_bounds__(256) void k_to_f16(const float *W, const int32_t *rows, int n, int n_pad,
                                                const uint8_t *keep, int zero_nan, _Float16 *out, int32_t *unnorm) {
  const int l = threadIdx.x & 31;
  const int nhw = gridDim.x * 8;
  for (int r = blockIdx.x * 8 + (threadIdx.x >> 5); r < n_pad; r += nhw) {
    if (r >= n) {
      reinterpret_cast<uint2 *>(out)[(size_t)r * kRowVec + l] = make_uint2(0x7E007E00u, 0x7E007E00u);
      continue;
    }
    const int src = rows ? rows[r] : r;
    const float4 x = reinterpret_cast<const float4 *>(W)[(size_t)src * kRowVec + l];
    _Float16 o[4] = {(_Float16)x.x, (_Float16)x.y, (_Float16)x.z, (_Float16)x.w};
    uint2 v = *reinterpret_cast<uint2 *>(o);
    const bool masked = keep && !keep[r];
    if (masked) v = make_uint2(0x7E007E00u, 0x7E007E00u);
    const float ss = halfwave_sum(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
    if (!masked && l == 0 && fabsf(ss - 1.0f) > 1e-3f) *unnorm = 1;  // NaN rows compare false: handled by the NaN rules
    // a NaN QUERY row would poison the max over the accumulator registers it shares with other
    // query rows: it is zeroed here and flagged by k_rerank (its exact scores are NaN anyway)
    if (zero_nan && (x.x != x.x || x.y != x.y || x.z != x.z || x.w != x.w)) v = make_uint2(0u, 0u);
    reinterpret_cast<uint2 *>(out)[(size_t)r * kRowVec + l] = v;
  }
}

// model.predict operands: tf l2_normalize (x * rsqrt(max(sum x^2, 1e-12)), the same expressions as
// k_rownorm<1> of anirec_infer.hip) of a gathered row -> fp32 copy for the exact re-rank and fp16 copy,
// multiplied by `sign`, for the MFMA (sign = -1 turns "largest rating" into "largest score" when the
// folded head slope is negative).  Rows n .. n_pad-1 of the fp16 output are NaN rows (tile padding).
__global__ __launch_bounds__(256) void k_norm_f16(const float *W, const int32_t *rows, int n, int n_pad, float sign,
                                                  float *out32, _Float16 *out16) {
  const int l = threadIdx.x & 31;
  const int nhw = gridDim.x * 8;
  for (int r = blockIdx.x * 8 + (threadIdx.x >> 5); r < n_pad; r += nhw) {
    if (r >= n) {
      reinterpret_cast<uint2 *>(out16)[(size_t)r * kRowVec + l] = make_uint2(0x7E007E00u, 0x7E007E00u);
      continue;
    }
    const int src = rows ? rows[r] : r;
    float4 x = reinterpret_cast<const float4 *>(W)[(size_t)src * kRowVec + l];
    float ss = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    ss = halfwave_sum(ss);
    const float rinv = 1.0f / sqrtf(fmaxf(ss, kL2nEps));
    float4 y;
    y.x = x.x * rinv;
    y.y = x.y * rinv;
    y.z = x.z * rinv;
    y.w = x.w * rinv;
    reinterpret_cast<float4 *>(out32)[(size_t)r * kRowVec + l] = y;
    _Float16 o[4] = {(_Float16)(y.x * sign), (_Float16)(y.y * sign), (_Float16)(y.z * sign), (_Float16)(y.w * sign)};
    uint2 v = *reinterpret_cast<uint2 *>(o);
    if (y.x != y.x || y.y != y.y || y.z != y.z || y.w != y.w) v = make_uint2(0u, 0u);  // NaN row: flagged by the re-rank
    reinterpret_cast<uint2 *>(out16)[(size_t)r * kRowVec + l] = v;
  }
}

struct CandArgs {
  const uint4 *Qb;   // [nq][16] 16-B chunks of fp16 query rows
  const uint4 *Wb;   // [n][16]
  int nq, n, k_eff;
  int tile0, tile1;  // this launch scans key tiles [tile0, tile1)
  uint2 *cand;       // [nq][kCap] {score bits, key index}
  int32_t *cnt;      // [nq]
  int32_t *cnt2;     // [nq][kMaxSplit] entries in the split regions (zero outside a split super-step)
  int splits;        // gridDim.y of this launch
  float *theta;      // [nq]
  int32_t *flags;    // [nq] bit0: buffer overflow (dense ties)
  unsigned long long *dbg;  // kDbg == 2: [0] total appends
  const uint32_t *watched;  // kMask: [nq][wwords] per-query key mask, bit set = key excluded (model_recs)
  int wwords;
  // all-pairs job (every key row is a query row): the key stream of a query batch skips the tiles of EARLIER batches
  // — logical tile t is physical tile t (t < map_lo) or t + map_skip — because those batches computed the same dot
  // products and passed every score >= the job's prior on to this batch's rows.  kSym launches (the tiles of LATER
  // batches) do that for the batches to come: a score passes if it reaches the PRIOR theta0 (every later row starts
  // from it): it is written to the wave's own log {score, query row, key row} (ballot-ranked like the row's own
  // appends: no atomic, nothing to wait for inside the MFMA loop), and to the query row's buffer if it also reaches
  // that row's threshold.  k_scatter_log deals the logs to the key rows' inboxes behind the launch.
  int map_lo, map_skip;
  int row0;                 // key-table row of this batch's first query
  const float *theta0_dev;  // the prior (device word)
  uint4 *log;               // [waves of the launch][lcap]
  int32_t *logcnt;          // [waves of the launch] entries a wave wanted to write (may exceed lcap: overflow)
  int lcap;
};
constexpr int kInbox = 256;

// ------------------------------------------------------------------------------------
// k_refresh: one wave per query row, between two super-steps of the key stream.
// tau = k_eff-th largest MFMA score seen so far is a lower bound of the final one, so only keys with
// score >= tau - 2 eps can still matter: that becomes the row's threshold for the next super-step and
// everything below it is dropped from the buffer.  Fully parallel over the rows (no workgroup of the
// MFMA kernel ever waits for a compaction).
// ------------------------------------------------------------------------------------
// kSlots = ceil(entries / 64) register slots per lane: after the first super-step a row holds about
// 2 k_eff entries, so the 32-step radix select runs over 1 slot (k = 10) or 4 (k = 100), not 8.
template <int kSlots>
__device__ __forceinline__ void refresh_row(const CandArgs &a, uint2 *cand_row, int row, int c, int lane) {
  uint2 en[kSlots];
  uint32_t u[kSlots];
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {
    const int e = lane + 64 * j;
    en[j] = e < c ? cand_row[e] : make_uint2(0u, 0u);
    u[j] = e < c ? f2key(__uint_as_float(en[j].x)) : 0u;
  }
  uint32_t prefix = 0;
  for (int bit = 31; bit >= kSelLow; --bit) {
    const uint32_t trial = prefix | (1u << bit);
    int cge = 0;  // wave-uniform: ballots + scalar popcounts, no cross-lane shuffles
#pragma unroll
    for (int j = 0; j < kSlots; ++j) cge += __popcll(__ballot(u[j] >= trial));
    if (cge >= a.k_eff) prefix = trial;
  }
  // never below what the row already uses: thresholds only rise along the key stream, and a caller's prior (a
  // guess of the final threshold; the re-rank proves or refutes it per row) must survive the early refreshes
  const float th = fmaxf(key2f(prefix) - 2.f * kEpsMfma, a.theta[row]);
  int nc = 0;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {
    const bool kp = u[j] != 0u && __uint_as_float(en[j].x) >= th;
    const unsigned long long m = __ballot(kp);
    if (kp) cand_row[nc + __popcll(m & lt)] = en[j];  // nc + rank < c <= kCap
    nc += __popcll(m);
  }
  if (lane == 0) {
    a.cnt[row] = nc;
    a.theta[row] = th;
  }
}

// Fold what a split super-step left in the regions behind the kept entries of a row (all regions are read before the
// first write: the destination may run into them); returns the new entry count.  The entries a lane writes here are
// read back by other lanes of the SAME wave afterwards (the caller's loads follow in program order).
__device__ __forceinline__ int fold_regions(uint2 *cand_row, int32_t *cnt_row, int32_t *cnt2_row, int c, int lane) {
  const int4 c2 = *reinterpret_cast<const int4 *>(cnt2_row);
  if (!(c2.x | c2.y | c2.z | c2.w)) return c;
  const int cs[kMaxSplit] = {c2.x, c2.y, c2.z, c2.w};
  constexpr int kRs = (kReg + 63) / 64;  // register slots per region
  uint2 r[kMaxSplit][kRs];
#pragma unroll
  for (int sp = 0; sp < kMaxSplit; ++sp)
#pragma unroll
    for (int j = 0; j < kRs; ++j) {
      const int e = lane + 64 * j;
      r[sp][j] = e < cs[sp] ? cand_row[kKept + sp * kReg + e] : make_uint2(0u, 0u);
    }
#pragma unroll
  for (int sp = 0; sp < kMaxSplit; ++sp) {
#pragma unroll
    for (int j = 0; j < kRs; ++j) {
      const int e = lane + 64 * j;
      if (e < cs[sp] && c + e < kCap) cand_row[c + e] = r[sp][j];
    }
    c += cs[sp];
  }
  c = min(c, kCap);
  if (lane == 0) {
    *reinterpret_cast<int4 *>(cnt2_row) = make_int4(0, 0, 0, 0);
    *cnt_row = c;
  }
  return c;
}

// all-pairs job: what earlier batches dropped into the inbox of key-table row g goes behind the row's entries.
// More than the inbox or the buffer holds: `ovf` (the caller flags the row; it is re-run without the shortcut).
__device__ __forceinline__ int fold_inbox(uint2 *cand_row, int32_t *cnt_row, const uint2 *inbox_row, const int32_t *icnt_row,
                                          float theta_row, int c, int lane, bool &ovf) {
  int ci = *icnt_row;
  if (ci <= 0) return c;
  if (ci > kInbox) {
    ovf = true;
    ci = kInbox;
  }
  // only what the row's threshold still lets through (the inbox was filled against the prior)
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int e0 = 0; e0 < ci; e0 += 64) {
    const int e = e0 + lane;
    const uint2 en = e < ci ? inbox_row[e] : make_uint2(0u, 0u);
    const bool kp = e < ci && __uint_as_float(en.x) >= theta_row;
    const unsigned long long m = __ballot(kp);
    const int pos = c + __popcll(m & lt);
    if (kp && pos < kCap) cand_row[pos] = en;
    c += __popcll(m);
  }
  if (c > kCap) {
    ovf = true;
    c = kCap;
  }
  if (lane == 0) *cnt_row = c;
  return c;
}

// start of a batch of the all-pairs job: the inbox of the parity whose writers are done (the batches before this one
// on the same chain) becomes the rows' first candidates; the k_refresh that follows turns them into thresholds
__global__ __launch_bounds__(64) void k_merge_inbox(uint2 *cand, int32_t *cnt, int32_t *flags, int nq, int row0,
                                                    const uint2 *inbox, const int32_t *icnt) {
  const int row = blockIdx.x;
  if (row >= nq) return;
  bool ovf = false;
  const size_t g = (size_t)row0 + row;
  (void)fold_inbox(cand + (size_t)row * kCap, cnt + row, inbox + g * kInbox, icnt + g, -INFINITY, 0, threadIdx.x, ovf);
  if (ovf && threadIdx.x == 0) flags[row] |= 1;
}

// behind a kSym launch: the waves' logs go to the key rows' inboxes (slot = atomic counter of the row; the order of
// the entries in an inbox is arbitrary, the re-rank's result does not depend on it).  A log that overflowed lost
// pairs nobody can name: *ovf marks the job, every all-pairs batch then comes out flagged (re-run by the caller).
constexpr int kLogCap = 8192;  // entries per wave and launch; the host bounds the launches' lengths to fit
__global__ __launch_bounds__(256) void k_scatter_log(const uint4 *log, const int32_t *logcnt, int lcap, uint2 *inbox,
                                                     int32_t *icnt, int32_t *ovf) {
  const size_t wv = blockIdx.x;
  int c = logcnt[wv];
  if (c > lcap) {
    if (threadIdx.x == 0) *ovf = 1;
    c = lcap;
  }
  for (int e = threadIdx.x; e < c; e += 256) {
    const uint4 en = log[wv * (size_t)lcap + e];
    const int slot = atomicAdd(icnt + en.z, 1);
    if (slot < kInbox) inbox[(size_t)en.z * kInbox + slot] = make_uint2(en.x, en.y);
  }
}

__device__ __forceinline__ void refresh_one(const CandArgs &a, int row, int lane) {
  uint2 *cand_row = a.cand + (size_t)row * kCap;
  const int c = fold_regions(cand_row, a.cnt + row, a.cnt2 + (size_t)row * kMaxSplit, min(a.cnt[row], kCap), lane);
  if (c < a.k_eff) return;  // not enough candidates yet (tiny tables): keep the threshold
  if (c <= 64)
    refresh_row<1>(a, cand_row, row, c, lane);
  else if (c <= 128)
    refresh_row<2>(a, cand_row, row, c, lane);
  else if (c <= 256)
    refresh_row<4>(a, cand_row, row, c, lane);
  else
    refresh_row<kCap / 64>(a, cand_row, row, c, lane);
}

// One wave per row; the grid is bounded by the host (side_grid) and strides over the rows, so that these waves —
// which run BESIDE another chain's k_cand in a job — never hold more than a few wave slots per CU: an unbounded
// grid of one-wave workgroups takes every slot a finished MFMA workgroup frees, and the next 8-wave k_cand workgroup
// (2 x 160 VGPRs per SIMD + 64 KB of LDS at once) cannot start until the whole side kernel has drained.
// Four rows per 256-thread workgroup (waves that never synchronise) measured the same as one row per workgroup
// (18 k job 0.69 vs 0.69 ms, all-pairs 32.3 vs 32.3 ms): the dispatch rate of one-wave workgroups is not what
// these kernels wait for.
__global__ __launch_bounds__(64) void k_refresh(CandArgs a) {
  for (int row = blockIdx.x; row < a.nq; row += gridDim.x) refresh_one(a, row, threadIdx.x);
}

// ------------------------------------------------------------------------------------
// k_cand: one super-step.  128 query rows per workgroup (4 waves x 32 rows, fragments in registers),
// key tiles of 128 rows double-buffered in XOR-swizzled LDS, thresholds FIXED for the launch.
// ------------------------------------------------------------------------------------
// kWaves = 8 (256 query rows per workgroup, one workgroup per CU) halves the L2 -> LDS key traffic per
// MFMA and is used when the queries fill the chip that way; kWaves = 4 (two workgroups per CU) otherwise.
// kMask: a candidate is dropped at append time when its bit in the query's own mask row is set (the
// "already watched" set of model_recs); the mask words reach a wave-private LDS image by LDS-DMA two tiles ahead.
template <int kDbg, int kWaves, bool kMask = false, bool kSym = false>  // kDbg 0: product; 1: no filter (timing only); 2: count appends; 4: stamps
// (host-side mode 16: after the product run, one launch over all keys with the final / with unreachable thresholds)
__global__ __launch_bounds__(64 * kWaves, 8 / kWaves) void k_cand(CandArgs a) {
  static_assert(!(kSym && (kMask || kDbg != 0)), "the all-pairs variant exists for the product build only");
  constexpr int kBM = 32 * kWaves;
  constexpr int kDma = 32 / kWaves;  // LDS-DMA instructions per wave per key tile (4 key rows each)
  __shared__ __attribute__((aligned(16))) uint4 Ks[2][kBN * 16];  // 2 x 32 KB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c16 = lane & 15, gq = lane >> 4;
  const int q0 = blockIdx.x * kBM;

  // v_mfma_f32_16x16x32_f16 (the shape the chip clocks higher on): a lane holds A[row c16][k = 8 gq + j],
  // B[k = 8 gq + j][col c16], C[row 4 gq + i][col c16].  The wave owns 32 query rows = 2 row blocks (rb);
  // a 32-key block is 2 key blocks (nb); K = 128 is 4 steps (kk) of 32 -> 16 MFMAs in 4 independent
  // accumulation chains per 32x32 block.
  f16x8 qa[2][4];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const int qrow = q0 + 32 * w + 16 * rb + c16;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qrow < a.nq) v = *reinterpret_cast<const u32x4 *>(&a.Qb[(size_t)qrow * 16 + 4 * kk + gq]);
      qa[rb][kk] = __builtin_bit_cast(f16x8, v);
    }
  }
  // Key tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write).
  // One instruction moves 64 lanes x 16 B to 1 KiB of LDS starting at M0 = 4 key rows, lane-linear; the
  // XOR swizzle of the LDS image (slot = chunk ^ (row & 15), conflict-free ds_read_b128) is therefore
  // applied to the per-lane SOURCE chunk.  Wb is padded with NaN rows to whole tiles (host code), so
  // there is no bounds test.  Inline asm: the compiler must not order every LDS read behind the DMA
  // (it would wait vmcnt(0) before each ds_read); the waits are placed by hand next to the barriers.
  const int wu = __builtin_amdgcn_readfirstlane(w);
  uint32_t doff[kDma];  // byte offset of this lane's source chunk inside a tile, per DMA instruction
#pragma unroll
  for (int i = 0; i < kDma; ++i) {
    const int r = 4 * kDma * w + 4 * i + (lane >> 4);
    doff[i] = (uint32_t)((r * 16 + ((lane & 15) ^ (r & 15))) * 16);
  }
  const uint32_t ks_base =
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)&Ks[0][0] + (uint32_t)wu * (1024u * kDma);
  auto dma_tile = [&](int t, int buf) {
    const int pt = t < a.map_lo ? t : t + a.map_skip;
    const char *base = reinterpret_cast<const char *>(a.Wb) + (size_t)pt * (kBN * 256);
    const uint32_t l0 = ks_base + (uint32_t)buf * (kBN * 256);
#pragma unroll
    for (int i = 0; i < kDma; ++i)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                   :
                   : "v"(doff[i]), "s"(base), "s"(l0 + 1024u * i)
                   : "memory", "m0");
  };
  // this workgroup's share of the super-step's key tiles
  const int split = blockIdx.y;
  int tile_lo = a.tile0, tile_hi = a.tile1;
  if (a.splits > 1) {
    const int len = (a.tile1 - a.tile0 + a.splits - 1) / a.splits;
    tile_lo = a.tile0 + split * len;
    tile_hi = min(a.tile1, tile_lo + len);
    if (tile_hi <= tile_lo) return;  // the whole workgroup: nothing left for this split
  }
  const int nt = tile_hi - tile_lo;
  dma_tile(tile_lo, 0);
  if (nt > 1) dma_tile(tile_lo + 1, 1);
  // kMask: the four 32-key mask words of each of the wave's 32 query rows travel global -> LDS by LDS-DMA as
  // well (two 256-B pieces per tile and wave into a wave-private [3][32][4] image, two tiles ahead), so they
  // are covered by the same hand-placed vmcnt waits as the key tiles.  (Plain loads into registers made the
  // compiler wait vmcnt(0) at the end of every tile — for the key-tile DMA issued half a tile earlier too.)
  extern __shared__ uint32_t mask_lds[];
  uint32_t *const mk_l = mask_lds + w * (3 * 128);
  uint32_t mk_voff[2] = {0u, 0u};
  uint32_t mk_base = 0;
  if (kMask) {
    mk_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)mask_lds + (uint32_t)wu * (3u * 512u);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int rg = q0 + 32 * w + 16 * j + (lane >> 2);
      if (rg >= a.nq) rg = a.nq - 1;  // rows past the end never produce candidates: any valid address will do
      mk_voff[j] = (uint32_t)rg * (uint32_t)a.wwords;
    }
  }
  auto dma_mask = [&](int t, int slot) {
    int wd = t * (kBN / 32) + (lane & 3);
    if (wd >= a.wwords) wd = a.wwords - 1;  // words past the end belong to NaN padding keys
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t voff = (mk_voff[j] + (uint32_t)wd) * 4u;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
                   :
                   : "v"(voff), "s"(a.watched), "s"(mk_base + (uint32_t)slot * 512u + 256u * j)
                   : "memory", "m0");
    }
  };
  if (kMask) {
    dma_mask(tile_lo, 0);
    if (nt > 1) dma_mask(tile_lo + 1, 1);
  }

  // Accumulator register i of block (rb, nb) belongs to query row 16 rb + 4 gq + i of the wave's 32
  // rows: the rows are private to the wave, so their thresholds and buffer counts live in registers
  // (replicated over the 16 lanes of a quarter-wave) — no LDS, no atomics in the loop.
  f32x4 nthr[2];  // -theta: loop-invariant C-in of the MFMA chains (no per-tile copies)
  float rthr[2][4];  // kSym: C-in is -theta0 for every row; the row's own test is score - theta0 >= theta - theta0
  uint32_t livem = 0;  // kSym: bit 4 rb + i = the row exists
  const float th0 = kSym ? a.theta0_dev[0] : 0.f;
  int cntr[2][4];
  uint32_t rowoff[2][4];  // byte offset of the row's buffer (nq*kCap*8 < 2^32 is checked on the host)
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = q0 + 32 * w + 16 * rb + 4 * gq + i;
      const bool live = rl < a.nq;
      if (kSym) {
        nthr[rb][i] = -th0;
        rthr[rb][i] = live ? a.theta[rl] - th0 : INFINITY;
        livem |= live ? 1u << (4 * rb + i) : 0u;
      } else {
        nthr[rb][i] = live ? -a.theta[rl] : -INFINITY;
        rthr[rb][i] = 0.f;
      }
      // a split appends to its own (empty) region; the kept entries must then end before the regions begin
      cntr[rb][i] = (live && a.splits == 1) ? a.cnt[rl] : 0;
      if (live && a.splits > 1 && split == 0 && c16 == 0 && a.cnt[rl] > kKept) a.flags[rl] |= 1;
      rowoff[rb][i] = (uint32_t)rl * (uint32_t)(kCap * 8) + (a.splits > 1 ? (uint32_t)((kKept + split * kReg) * 8) : 0u);
    }
  // LDS read addresses of the B fragments: key row 32 cb + 16 nb + c16 has (row & 15) == c16, so the
  // swizzled chunk index depends only on (kk, lane); buffer, cb and nb are constant offsets
  const f16x8 *kb[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    kb[kk] = reinterpret_cast<const f16x8 *>(&Ks[0][c16 * 16 + ((4 * kk + gq) ^ c16)]);
  const uint32_t cap = a.splits > 1 ? (uint32_t)kReg : (uint32_t)kCap;  // entries this workgroup may append per row
  const uint32_t lt16 = (1u << c16) - 1u;
  const int sh16 = 16 * gq;
  // kSym: this wave's log
  const size_t wave_id = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kWaves + wu;
  uint4 *const wlog = kSym ? a.log + wave_id * (size_t)a.lcap : nullptr;
  uint32_t lcnt = 0;
  const unsigned long long lt64 = (1ull << lane) - 1ull;
  char *const cand_bytes = reinterpret_cast<char *>(a.cand);
  // Touch the per-row registers here so the loads above are waited for BEFORE the loop: otherwise the
  // compiler parks an s_waitcnt vmcnt(0) at their first use inside the append path, where it also
  // waits for the key-tile prefetch that is meant to stay in flight under the MFMAs.
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(cntr[rb][i]), "v"(nthr[rb][i]), "v"(rthr[rb][i]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // both DMA tiles have landed
  __syncthreads();

  // Software pipeline over 32x32 blocks (4 per key tile).  Stage j filters block j while the matrix
  // core works on block j+1 of the SAME wave (a second accumulator set), and fetches the B fragments
  // of block j+2 from LDS as the MFMAs consume the current ones.
  f16x8 bv[2][4];
  auto fetch = [&](int kk, int buf, int cb) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bv[nb][kk] = kb[kk][(buf * kBN + cb * 32 + nb * 16) * 16];
  };
  struct Acc { f32x4 c[2][2]; };  // [rb][nb]
  auto mma_step = [&](Acc &x, int kk) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        x.c[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[rb][kk], bv[nb][kk], kk == 0 ? nthr[rb] : x.c[rb][nb], 0, 0, 0);
  };
  // cur: finished chains (score - theta) of the 32-key block starting at key0; nxt: chains being
  // issued from bv; (fbuf, fcb): block whose fragments replace bv as they are consumed
  auto mma1 = [&](Acc &x, int rb, int nb, int kk) {
    x.c[rb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa[rb][kk], bv[nb][kk], kk == 0 ? nthr[rb] : x.c[rb][nb], 0, 0, 0);
  };
  auto stage_fn = [&](const Acc &cur, Acc &nxt, int key0, int fbuf, int fcb, int moff) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {  // quarter qd: MFMA step kk = qd of the next block, filter (rb, nb) of this one
      const int rb = qd >> 1, nb = qd & 1;
      const f32x4 &cv = cur.c[rb][nb];
      // The wave issues in order, so its VALU/LDS instructions only run under its own MFMAs if they sit
      // BETWEEN them: one 4-pass MFMA leaves a 16-cycle gap = the two or three instructions placed
      // after it.  Hierarchical reject: max over the 4 accumulator registers (256 scores) + a ballot.
      // v_max3 by hand: fmaxf() costs two extra canonicalising v_max per quarter.  `cv` was completed a
      // whole stage (16 MFMAs) ago, far beyond the MFMA->VALU hazard window the compiler would pad;
      // NaN scores (masked / padding keys) are quiet NaNs, which v_max3 ignores like fmaxf.
      float m3 = 0.f, mq = 0.f;
      mma1(nxt, 0, 0, qd);
      if (kDbg != 1) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(cv[0]), "v"(cv[1]), "v"(cv[2]));
      __builtin_amdgcn_sched_barrier(0);
      mma1(nxt, 1, 0, qd);
      if (kDbg != 1) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mq) : "v"(m3), "v"(cv[3]));
      __builtin_amdgcn_sched_barrier(0);
      mma1(nxt, 0, 1, qd);
      bv[0][qd] = kb[qd][(fbuf * kBN + fcb * 32) * 16];
      __builtin_amdgcn_sched_barrier(0);
      mma1(nxt, 1, 1, qd);
      bv[1][qd] = kb[qd][(fbuf * kBN + fcb * 32 + 16) * 16];
      __builtin_amdgcn_sched_barrier(0);
      if (kDbg == 1) {
        if (qd == 0) asm volatile("" ::"v"(cur.c[0][0]), "v"(cur.c[0][1]), "v"(cur.c[1][0]), "v"(cur.c[1][1]));
        continue;
      }
      if (__ballot(mq >= 0.f)) {
        const int key = key0 + 16 * nb + c16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bool hit = cv[i] >= (kSym ? rthr[rb][i] : 0.f);
          if (kSym) {  // the key's row will want this pair (every hit of the row's own test is one of these)
            const bool ch = cv[i] >= 0.f && ((livem >> (4 * rb + i)) & 1u);
            const unsigned long long cm = __ballot(ch);
            if (!cm) continue;  // wave-uniform
            const uint32_t lpos = lcnt + (uint32_t)__popcll(cm & lt64);
            if (ch && lpos < (uint32_t)a.lcap)
              wlog[lpos] = make_uint4(__float_as_uint(cv[i] - nthr[rb][i]),
                                      (uint32_t)(a.row0 + q0 + 32 * w + 16 * rb + 4 * gq + i), (uint32_t)key, 0u);
            lcnt += (uint32_t)__popcll(cm);
          }
          if (kMask) {  // the row's 32 mask bits of this key block: one word of the wave's LDS mask image
            const uint32_t wbits = mk_l[moff + 4 * (16 * rb + 4 * gq + i)];
            hit = hit && ((wbits >> (16 * nb + c16)) & 1u) == 0u;
          }
          const unsigned long long mk = __ballot(hit);
          if (mk) {  // wave-uniform
            const uint32_t mh = (uint32_t)(mk >> sh16) & 0xFFFFu;  // the quarter-wave (= row) of this lane
            const uint32_t pos = (uint32_t)cntr[rb][i] + __popc(mh & lt16);
            if (kDbg != 8 && hit && pos < cap)  // (kDbg 8, timing only: everything but the candidate store)
              *reinterpret_cast<uint2 *>(cand_bytes + (rowoff[rb][i] + pos * 8u)) =
                  make_uint2(__float_as_uint(cv[i] - nthr[rb][i]), (uint32_t)key);
            cntr[rb][i] += __popc(mh);
            if (kDbg == 2 && lane == 0) atomicAdd(&a.dbg[0], (unsigned long long)__popcll(mk));
          }
        }
      }
    }
  };

  // prologue: chains of block (tile0, 0), fragments of block (tile0, 1)
  Acc acc0, acc1;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fetch(kk, 0, 0);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) mma_step(acc0, kk);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fetch(kk, 0, 1);

  unsigned long long dbg_store = 0, dbg_barrier = 0, dbg_t0 = 0;
  if (kDbg == 4) dbg_t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < nt; ++it) {
    const int buf = it & 1;
    const int lt = tile_lo + it;
    const int key0 = (lt < a.map_lo ? lt : lt + a.map_skip) * kBN;
    const int ms = (it % 3) * 128;                     // this tile's slot in the wave's LDS mask image
    stage_fn(acc0, acc1, key0, buf, 2, ms + 0);        // filter block 0 | MFMA block 1 | fetch block 2
    stage_fn(acc1, acc0, key0 + 32, buf, 3, ms + 1);   // filter block 1 | MFMA block 2 | fetch block 3
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
    if (kDbg == 4) ts0 = __builtin_amdgcn_s_memtime();
    // tile it+1 was sent to the other buffer one tile ago; its DMA (and this wave's candidate stores,
    // vmcnt counts in order) must have landed before anyone reads it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kDbg == 4) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ts1 = __builtin_amdgcn_s_memtime();
    }
    __syncthreads();
    if (kDbg == 4) {
      ts2 = __builtin_amdgcn_s_memtime();
      dbg_store += ts1 - ts0;
      dbg_barrier += ts2 - ts1;
    }
    // this tile's buffer was last read (fetched) before the barrier: refill it with tile it+2, in
    // flight for a whole tile
    if (kMask && it + 2 < nt) dma_mask(tile_lo + it + 2, (it + 2) % 3);  // its slot was tile it-1's
    if (it + 2 < nt) dma_tile(tile_lo + it + 2, buf);
    // on the last tile the "next tile" blocks are stale LDS: computed and thrown away
    stage_fn(acc0, acc1, key0 + 64, buf ^ 1, 0, ms + 2);  // filter block 2 | MFMA block 3 | fetch next tile's block 0
    stage_fn(acc1, acc0, key0 + 96, buf ^ 1, 1, ms + 3);  // filter block 3 | MFMA next block 0 | fetch next block 1
  }
  if (kDbg == 4 && lane == 0) {  // in-kernel stamps (diagnostic build only): cycles per wave
    unsigned long long *d = a.dbg + 4 * (size_t)(blockIdx.x * kWaves + w);
    d[0] = __builtin_amdgcn_s_memtime() - dbg_t0;
    d[1] = dbg_store;
    d[2] = dbg_barrier;
    d[3] = (unsigned long long)nt;
  }
  if (kSym && lane == 0) a.logcnt[wave_id] = (int32_t)lcnt;
  if (kDbg != 1 && c16 == 0) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = q0 + 32 * w + 16 * rb + 4 * gq + i;
        if (rl < a.nq) {
          if (a.splits > 1)
            a.cnt2[(size_t)rl * kMaxSplit + split] = min(cntr[rb][i], (int)cap);
          else
            a.cnt[rl] = min(cntr[rb][i], kCap);
          if (cntr[rb][i] > (int)cap) a.flags[rl] |= 1;  // more candidates than the buffer / the region holds
        }
      }
  }
}

// theta0_dev (optional): the prior learnt on the device from the first batch (k_learn_prior); else `theta0`
__global__ void k_init_rows(int32_t *cnt, int32_t *cnt2, float *theta, int32_t *flags, int nq, int32_t *unnorm, float theta0,
                            const float *theta0_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && unnorm) *unnorm = 0;
  if (i < nq) {
    cnt[i] = 0;
    *reinterpret_cast<int4 *>(cnt2 + (size_t)i * kMaxSplit) = make_int4(0, 0, 0, 0);
    theta[i] = theta0_dev ? theta0_dev[0] : theta0;  // kThetaInit, or a prior (a lower bound of most rows' final threshold)
    flags[i] = 0;
  }
}

// The threshold prior of the later query batches, learnt on the device from the rows of the first one (no host
// round trip between the batches): tau[r] = score[r * k + k - 1] is the exact k-th best score of row r; the prior is
// the lower edge of the 1/1024-wide score bin that holds their 0.5 % quantile, minus the error window and a margin.
// NaN rows (unproven: re-run later) are skipped; with fewer than 4096 proven rows there is no prior (kThetaInit).
constexpr int kPriorBins = 2048;  // scores in [-1, 1)
__global__ __launch_bounds__(1024) void k_learn_prior(const float *score, int rows, int k, float *theta0_dev) {
  __shared__ int hist[kPriorBins];
  __shared__ int wsum[16];
  __shared__ int total_s, bin_s;
  const int tid = threadIdx.x;
  for (int i = tid; i < kPriorBins; i += 1024) hist[i] = 0;
  if (tid == 0) bin_s = -1;
  __syncthreads();
  int mine = 0;
  for (int r = tid; r < rows; r += 1024) {
    const float t = score[(size_t)r * k + (k - 1)];
    if (t == t) {
      int b = (int)floorf((t + 1.0f) * (kPriorBins / 2));
      b = b < 0 ? 0 : (b >= kPriorBins ? kPriorBins - 1 : b);
      atomicAdd(&hist[b], 1);
      ++mine;
    }
  }
  {
    int v = mine;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = v;
  }
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += wsum[w];
    total_s = t;
  }
  __syncthreads();
  const int total = total_s;
  if (total < 4096) {
    if (tid == 0) theta0_dev[0] = kThetaInit;
    return;
  }
  const int rank = total / 200;  // 0.5 % from below
  // two bins per thread: exclusive prefix of the histogram, the bin where it crosses `rank`
  const int h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
  int inc = h0 + h1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if ((tid & 63) >= o) inc += t;
  }
  __syncthreads();
  if ((tid & 63) == 63) wsum[tid >> 6] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
  const int ex0 = base + inc - (h0 + h1), ex1 = ex0 + h0;
  if (h0 > 0 && ex0 <= rank && rank < ex0 + h0) bin_s = 2 * tid;
  if (h1 > 0 && ex1 <= rank && rank < ex1 + h1) bin_s = 2 * tid + 1;
  __syncthreads();
  if (tid == 0) {
    const float edge = (float)bin_s * (2.0f / kPriorBins) - 1.0f;
    theta0_dev[0] = fmaxf(edge - 8.1e-3f, kThetaInit);
  }
}

// ------------------------------------------------------------------------------------
// exact re-rank of one query row per wave
// ------------------------------------------------------------------------------------
struct RerankArgs {
  const float *What;       // [n][128] fp32 normalised rows (keys)
  const float *Qf;         // [nq][128] fp32 query rows (gathered) or nullptr -> What[qidx[row]]
  const int32_t *qidx;     // [nq] index of the query row in the key table (self exclusion) or -1
  int nq, n, k, k_eff;     // k_eff = k + 1 when the query itself must be dropped
  int exclude_self;
  uint2 *cand;
  int32_t *cnt;
  int32_t *cnt2;           // split regions the last super-step left (folded here: no k_refresh after the last step)
  const uint2 *inbox;      // all-pairs job: the inbox (indexed by key-table row = qidx[row]) still to be merged, or null
  const int32_t *icnt;
  const int32_t *sym_ovf;  // all-pairs job: a log overflowed somewhere (pairs lost): every row is unproven
  const float *theta;
  int32_t *flags;          // bit1: incomplete window / too many survivors / too few candidates
  int32_t *out_idx;        // [nq][k]
  float *out_score;        // [nq][k]
  float hs, hb, sign;      // kPredict: rating = sigmoid(c * hs + hb); MFMA scores are sign * c
  const int32_t *unnorm;   // cosine path: non-zero when some row was not unit-norm (window unproven) -> flags bit 2
};

constexpr int kMaxSurv = 256;

// kPredict (model_recs): the buffer holds MFMA values of sign * cosine; survivors are re-scored with the
// exact cosine chain, mapped through the BN-inference head exactly as the exact path does
// (rating_from_cosine) and ranked by (rating desc, index asc).  Ratings are a non-decreasing function of
// sign * c, so every key that is not a survivor has rating <= rating(tau - eps): the row is complete iff
// k survivors lie strictly above that bound (a few ulps of slack for the fast exp); saturated heads and
// worst-case MFMA errors fail the test and fall back to the exact path.
//
// Measured in round 3 (rocprofv3, 350 k keys x 65 536 queries): at k = 100 the kernel re-reads ~110 fp32 rows of
// 512 B per query — 56 KB per query, 3.7 GB per 65 536 queries, 19.7 GB for the all-pairs job — and runs at
// 6.4-6.9 TB/s of that gather (580 us per 65 536 queries): it is bound by the HBM row gather the EXACT scores need
// (the key table does not stay in the Infinity Cache between two uses of a row), not by its instructions.  A rewrite
// with the survivor rows fetched coalesced into a 16-row LDS tile one tile ahead, the chain walked from LDS and a
// bitonic sort of 64-bit {score, index} keys in place of the O(survivors^2) rank reached 537 us (-7 %) at k = 100 and
// was 1.6-2.5x SLOWER at k = 10 (a dozen survivors: the sort's 21 dependent cross-lane stages and the tile's LDS
// footprint cost more than the old rank loop); removing its fetch, its chain or its sort one at a time changed nothing
// (533-548 us).  Dropped: the bytes are the floor.
// Early-exit builds of this kernel (rocprofv3, 18 000 x 18 000 / 350 000 x 65 536 queries in two batches, k = 100):
// empty kernel 5 / 9 us, + region fold 12 / 21 us, + buffer load, select and survivor scan 25 / 49 us, + the exact
// score chain 132 / 282 us, + rank and output 137 / 290 us.  The chain phase is the row gather: 110 rows x 512 B per
// query = 1.0 GB from the Infinity Cache in 107 us at 18 k (9.5 TB/s; the 9.2 MB table does not fit one XCD's 4 MB L2)
// and 3.7 GB from HBM and Infinity Cache hits in 2 x 233 us at 350 k (7.9 TB/s).  Fetching the same rows as 128-B slices through an LDS tile
// (4 line requests per row instead of 32) changed neither case (0.599 vs 0.598 ms for the 18 k job).
// kSlots = ceil(entries / 64) register slots per lane, as in refresh_row: after the last refresh a row holds about
// 2 k entries, so the 32-step select and the survivor scan run over 2-4 slots, not kCap / 64.
template <bool kPredict, int kSlots>
__device__ __forceinline__ void rerank_row(const RerankArgs &a, int row, int c, bool ovf, int lane, float *qs, int32_t *sidx,
                                           float *sval,
                                           unsigned long long *skey) {
  const size_t base = (size_t)row * kCap;
  const int qrow = kPredict ? -1 : a.qidx[row];
  const int self = (!kPredict && a.exclude_self) ? qrow : -1;
  // query row (fp32) into LDS
  {
    const float *q = a.Qf ? a.Qf + (size_t)row * kDim : a.What + (size_t)qrow * kDim;
    qs[lane] = q[lane];
    qs[lane + 64] = q[lane + 64];
  }
  float sc[kSlots];
  int32_t id[kSlots];
  uint32_t u[kSlots];
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {
    const int e = lane + 64 * j;
    const uint2 en = e < c ? a.cand[base + e] : make_uint2(0u, 0u);
    sc[j] = __uint_as_float(en.x);
    id[j] = e < c ? (int32_t)en.y : -1;
    u[j] = e < c ? f2key(sc[j]) : 0u;
  }
  bool bad = ovf || (a.flags[row] & 1) != 0;
  const bool unnorm = a.unnorm != nullptr && a.unnorm[0] != 0;
  if (unnorm) bad = true;
  if (__ballot(qs[lane] != qs[lane] || qs[lane + 64] != qs[lane + 64])) bad = true;  // NaN query row
  const int kk = min(a.k_eff, a.n);
  if (c < kk) bad = true;
  // tau = kk-th largest bf16 score
  uint32_t prefix = 0;
  for (int bit = 31; bit >= kSelLow; --bit) {
    const uint32_t trial = prefix | (1u << bit);
    int cge = 0;
#pragma unroll
    for (int j = 0; j < kSlots; ++j) cge += __popcll(__ballot(u[j] >= trial));
    if (cge >= kk) prefix = trial;
  }
  const float tau = key2f(prefix);
  const float lo = tau - 2.f * kEpsMfma;
  if (!(a.theta[row] <= lo)) bad = true;  // the buffer is complete only down to theta
  // survivors: bf16 score >= tau - 2 eps
  int ns = 0;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < kSlots; ++j) {
    const bool kp = u[j] != 0u && sc[j] >= lo;
    const unsigned long long m = __ballot(kp);
    if (kp) {
      const int pos = ns + __popcll(m & lt);
      if (pos < kMaxSurv) sidx[pos] = id[j];
    }
    ns += __popcll(m);
  }
  if (ns > kMaxSurv) bad = true;
  __syncthreads();
  if (bad) {
    if (lane == 0) a.flags[row] |= unnorm ? 6 : 2;
    for (int i = lane; i < a.k; i += 64) {
      a.out_idx[(size_t)row * a.k + i] = -1;
      a.out_score[(size_t)row * a.k + i] = __uint_as_float(0x7FC00000u);
    }
    return;
  }
  // exact scores: k-ordered fp32 fma chain (the library's defined order)
  for (int i = lane; i < ns; i += 64) {
    const float4 *r4 = reinterpret_cast<const float4 *>(a.What + (size_t)sidx[i] * kDim);
    float s = 0.f;
#pragma unroll 8
    for (int k4 = 0; k4 < kRowVec; ++k4) {
      const float4 x = r4[k4];
      s = __fmaf_rn(x.x, qs[4 * k4 + 0], s);
      s = __fmaf_rn(x.y, qs[4 * k4 + 1], s);
      s = __fmaf_rn(x.z, qs[4 * k4 + 2], s);
      s = __fmaf_rn(x.w, qs[4 * k4 + 3], s);
    }
    sval[i] = kPredict ? rating_from_cosine(s, a.hs, a.hb) : s;
  }
  __syncthreads();
  if (kPredict) {
    // k-th best rating among the survivors vs the best any excluded key could reach
    // keys that are not survivors have MFMA value < tau - 2 eps, i.e. sign * c < tau - eps
    const float p_bound = rating_from_cosine(a.sign * (lo + kEpsMfma), a.hs, a.hb) * (1.0f + 6e-7f);
    int above = 0;  // survivors strictly above the bound
    for (int i0 = 0; i0 < ns; i0 += 64) {
      const int i = i0 + lane;
      above += __popcll(__ballot(i < ns && sval[i] > p_bound));
    }
    if (above < min(a.k, ns) || !(p_bound == p_bound)) {  // (ns >= k here: c >= kk was checked)
      if (lane == 0) a.flags[row] |= 2;
      for (int i = lane; i < a.k; i += 64) {
        a.out_idx[(size_t)row * a.k + i] = -1;
        a.out_score[(size_t)row * a.k + i] = __uint_as_float(0x7FC00000u);
      }
      return;
    }
  }
  // rank by (score desc, index asc); the query itself is dropped.  One 64-bit key per survivor — the exact path's
  // order-preserving score key (NaN after every number) above the inverted index — turns the O(survivors^2) rank into
  // one LDS broadcast read, one compare and one add per pair: this loop was 7x the instructions of the score chain.
  for (int i = lane; i < ns; i += 64) {
    const float sv = sval[i];
    uint32_t hi = __float_as_uint(sv);
    hi = (hi & 0x80000000u) ? ~hi : (hi | 0x80000000u);
    hi = sv != sv ? 1u : (hi < 2u ? 2u : hi);
    const int mi = sidx[i];
    skey[i] = mi == self ? 0ull : ((unsigned long long)hi << 32) | (uint32_t)~mi;
  }
  __builtin_amdgcn_wave_barrier();
  {
    constexpr int kOwn = kMaxSurv / 64;
    unsigned long long mk[kOwn];
    int rank[kOwn];
#pragma unroll
    for (int t = 0; t < kOwn; ++t) {
      const int i = lane + 64 * t;
      mk[t] = i < ns ? skey[i] : ~0ull;
      rank[t] = 0;
    }
    const int own = (ns + 63) >> 6;  // wave-uniform
    if (own <= 2) {
      for (int j = 0; j < ns; ++j) {
        const unsigned long long kj = skey[j];
        rank[0] += kj > mk[0] ? 1 : 0;
        rank[1] += kj > mk[1] ? 1 : 0;
      }
    } else {
      for (int j = 0; j < ns; ++j) {
        const unsigned long long kj = skey[j];
#pragma unroll
        for (int t = 0; t < kOwn; ++t) rank[t] += kj > mk[t] ? 1 : 0;
      }
    }
#pragma unroll
    for (int t = 0; t < kOwn; ++t) {
      const int i = lane + 64 * t;
      if (i < ns && mk[t] != 0ull && rank[t] < a.k) {
        a.out_idx[(size_t)row * a.k + rank[t]] = sidx[i];
        a.out_score[(size_t)row * a.k + rank[t]] = sval[i];
      }
    }
  }
  // fewer than k real neighbours (tiny tables): pad
  int have = 0;
  for (int i0 = 0; i0 < ns; i0 += 64) {
    const int i = i0 + lane;
    have += __popcll(__ballot(i < ns && sidx[i] != self));
  }
  for (int i = have + lane; i < a.k; i += 64) {
    a.out_idx[(size_t)row * a.k + i] = -1;
    a.out_score[(size_t)row * a.k + i] = __uint_as_float(0x7FC00000u);
  }
}

template <bool kPredict>
__device__ __forceinline__ void rerank_one(const RerankArgs &a, int row, int lane, float *qs, int32_t *sidx, float *sval,
                                           unsigned long long *skey) {
  int c = fold_regions(a.cand + (size_t)row * kCap, a.cnt + row, a.cnt2 + (size_t)row * kMaxSplit,
                       min(a.cnt[row], kCap), lane);
  bool ovf = !kPredict && a.sym_ovf != nullptr && a.sym_ovf[0] != 0;
  if (!kPredict && a.inbox != nullptr) {
    const size_t g = (size_t)a.qidx[row];
    c = fold_inbox(a.cand + (size_t)row * kCap, a.cnt + row, a.inbox + g * kInbox, a.icnt + g, a.theta[row], c, lane, ovf);
  }
  if (c <= 128)
    rerank_row<kPredict, 2>(a, row, c, ovf, lane, qs, sidx, sval, skey);
  else if (c <= 256)
    rerank_row<kPredict, 4>(a, row, c, ovf, lane, qs, sidx, sval, skey);
  else
    rerank_row<kPredict, kCap / 64>(a, row, c, ovf, lane, qs, sidx, sval, skey);
}

template <bool kPredict>
__global__ __launch_bounds__(64) void k_rerank(RerankArgs a) {  // bounded grid, strides over the rows (see k_refresh)
  __shared__ float qs[kDim];
  __shared__ int32_t sidx[kMaxSurv];
  __shared__ float sval[kMaxSurv];
  __shared__ unsigned long long skey[kMaxSurv];
  for (int row = blockIdx.x; row < a.nq; row += gridDim.x) {
    rerank_one<kPredict>(a, row, threadIdx.x, qs, sidx, sval, skey);
    __syncthreads();  // the next row rewrites qs / sidx / sval
  }
}

__global__ void k_flag_all(int32_t *flags, int nq, int32_t *out_idx, float *out_p, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  flags[i] = 2;
  for (int j = 0; j < k; ++j) {
    out_idx[(size_t)i * k + j] = -1;
    out_p[(size_t)i * k + j] = __uint_as_float(0x7FC00000u);
  }
}

__global__ void k_count_flags(const int32_t *flags, int nq, int32_t *count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq && flags[i]) atomicAdd(count, 1);
}

}  // namespace anirec

using namespace anirec;

// HIP-event timing of the MFMA kernel (bench.py's roofline leg): armed by anirec_topk_mfma_timing(1, ...)
static bool g_time_cand = false;
static float g_cand_ms = 0.f;     // sum of the k_cand launch durations of the last call
static int g_cand_launches = 0;

// Super-steps of the key stream: thresholds are fixed inside a launch and refreshed between launches;
// each super-step doubles the number of keys seen, so a row gains about k_eff new candidates per
// super-step (the first one, with no threshold yet, must fit the buffer).
// compute units of the current device (the split heuristic below smooths the wave quantisation of a launch over them)
static int cu_count() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
    else
      cus = 256;
  }
  return cus;
}

// mid_ev (optional): recorded on s behind the first super-step that ends at or beyond mid_pct % of the key tiles (the
// job staggers its chains with it)
// grid of the one-wave-per-row side kernels: one workgroup per row, or — ANIREC_TOPK_SIDE_WAVES > 0, an experiment
// knob — at most that many waves per CU striding over the rows (measured and NOT adopted: see the job's comment)
static int side_grid(int nq) {
  const char *e = getenv("ANIREC_TOPK_SIDE_WAVES");
  const int per_cu = e ? atoi(e) : 0;
  if (per_cu <= 0) return nq;
  const long long g = (long long)cu_count() * per_cu;
  return (int)(g < nq ? g : nq);
}

// one batch of the all-pairs job (see CandArgs): its key stream and its inboxes
struct SymPlan {
  int map_lo, map_skip;  // logical -> physical key tiles
  int sym_from;          // logical tile where the keys of later batches begin (== the stream's end: none)
  int row0;
  const float *theta0_dev;
  uint2 *inbox_w;        // written (through the logs) by this batch's kSym launches
  int32_t *icnt_w;
  uint4 *log;            // this chain's log buffer: log_waves x kLogCap entries
  int32_t *logcnt;
  int32_t *ovf;          // job-wide: a log overflowed
  const uint2 *inbox_start;  // merged before the first super-step (writers: earlier batches of this chain)
  const int32_t *icnt_start;
  const uint2 *inbox_end;    // merged by the re-rank (writers: the other chain, the batch before this one last)
  const int32_t *icnt_end;
  hipEvent_t done_ev;    // recorded behind this batch's last k_cand
  hipEvent_t wait_ev;    // the previous batch's done_ev (other chain), or null
};

// a chain's logs: one per wave of its largest launch (32 rows per wave, up to kMaxSplit key ranges)
static size_t log_waves(size_t rows) { return (rows + 255) / 256 * 8 * kMaxSplit; }  // (>= ceil(rows / 128) * 4 too)

static void cand_defaults(CandArgs &ca) {
  ca.map_lo = 0x7fffffff;
  ca.map_skip = 0;
  ca.row0 = 0;
  ca.theta0_dev = nullptr;
  ca.log = nullptr;
  ca.logcnt = nullptr;
  ca.lcap = 0;
}

static int run_super_steps(CandArgs &ca, int n, int nq, bool masked, int mode, void *dbg_words,
                           unsigned long long *stamps, size_t n_waves, hipStream_t s, hipEvent_t mid_ev = nullptr,
                           int mid_pct = 0, bool has_prior = false, const SymPlan *sp = nullptr) {
  int ntiles = (n + kBN - 1) / kBN;
  int sym_from = 0x7fffffff;
  if (sp) {
    ca.map_lo = sp->map_lo;
    ca.map_skip = sp->map_skip;
    ca.row0 = sp->row0;
    ca.theta0_dev = sp->theta0_dev;
    ca.log = sp->log;
    ca.logcnt = sp->logcnt;
    ca.lcap = kLogCap;
    ntiles -= sp->map_skip;
    sym_from = sp->sym_from;
  }
  // 256-row workgroups once they give every CU one (8 waves per CU either way); 128-row otherwise
  const char *wv = getenv("ANIREC_TOPK_WAVES");
  // (an all-pairs batch of 24 k rows or more: its launches are cut in three key ranges anyway)
  const bool wide = wv ? atoi(wv) == 8 : (nq >= 49152 || (sp != nullptr && nq >= 24576));
  const dim3 grid(wide ? (nq + 255) / 256 : (nq + 127) / 128);
  const dim3 block(wide ? 512 : 256);
  int n_launch = 0;
  if (masked) {  // 64 KB of key tiles (static) + the mask images (dynamic) exceed the default 64 KB cap
    static bool attr_set = false;
    if (!attr_set) {
      ANIREC_HIP_CHECK(hipFuncSetAttribute((const void *)k_cand<0, 8, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1024));
      ANIREC_HIP_CHECK(hipFuncSetAttribute((const void *)k_cand<0, 4, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1024));
      attr_set = true;
    }
  }
  std::vector<hipEvent_t> timed;  // event pairs around the k_cand launches (timing mode only)
  // each super-step sees growth_pct % of the keys seen so far: a row gains about k_eff * growth candidates per
  // step, which must fit the buffer next to the ~2 k_eff it already holds — tripling for small k (fewer
  // launches and refreshes), doubling otherwise
  const char *gp = getenv("ANIREC_TOPK_GROWTH");
  // (rows that start from a prior append little before their own threshold takes over: longer steps, fewer refreshes)
  const char *gpp = getenv("ANIREC_TOPK_GROWTH_PRIOR");
  const int growth_pct = has_prior && gpp ? atoi(gpp) : gp ? atoi(gp) : (ca.k_eff <= 32 ? 200 : 100);
  // the first super-step runs without a threshold and appends every key it sees: keep it as short as
  // the k-th-best estimate allows (>= 4 k_eff keys), at most what the buffer holds
  int first = (4 * ca.k_eff + kBN - 1) / kBN;
  if (first > (kCap - kBN) / kBN) first = (kCap - kBN) / kBN;
  if (first < 1) first = 1;
  // Rows that start from a prior append next to nothing in the first tiles (a prior near the final threshold passes
  // ~k keys of the WHOLE stream), and their own threshold only overtakes the prior once about half the keys have been
  // seen: the tiny first super-steps and their refreshes buy nothing, so the first one takes a share of the stream.
  const char *fpp = getenv("ANIREC_TOPK_FIRST_PRIOR_PCT");
  const int first_pct = fpp ? atoi(fpp) : 0;
  if (has_prior && first_pct > 0 && (long long)ntiles * first_pct / 100 > first) first = (int)((long long)ntiles * first_pct / 100);
  // Few queries: fewer workgroups than the chip holds.  From the second super-step on the key tiles of a super-
  // step are then split over up to kMaxSplit workgroups per row block (the first one, with no threshold yet,
  // needs the whole buffer of a row).
  // The same splits smooth the wave quantisation of bigger launches: 391 workgroups on 256 CUs run as two full
  // rounds; cut in three they run as five rounds of a third of the length (1.67).
  const char *spe = getenv("ANIREC_TOPK_SPLITS");
  const int wg_slots = cu_count() * (wide ? 1 : 2);
  // A row gains about k_eff * growth candidates per super-step, spread over the splits: a region (kReg entries)
  // must hold its share with a wide margin, so few splits are not an option for large k.
  const int min_split = (int)((2.5 * ca.k_eff * growth_pct / 100.0 + kReg - 1) / kReg);
  int max_split = 1;
  if (spe) {
    max_split = atoi(spe);
  } else {
    double best = (double)(((long long)grid.x + wg_slots - 1) / wg_slots);  // rounds without splits
    for (int sp = min_split > 2 ? min_split : 2; sp <= kMaxSplit; ++sp) {
      const double rounds = (double)(((long long)grid.x * sp + wg_slots - 1) / wg_slots) / sp;
      if (rounds < best * 0.97) {  // a larger split must buy at least 3 %
        best = rounds;
        max_split = sp;
      }
    }
  }
  if (max_split > kMaxSplit) max_split = kMaxSplit;
  if (max_split < 1 || ca.k_eff + 16 > kKept) max_split = 1;  // the kept entries (k_eff + the 2 eps window) must fit
  for (int t0 = 0, step = first; t0 < ntiles;) {
    int t1 = t0 + step < ntiles ? t0 + step : ntiles;
    if (t0 < sym_from && t1 > sym_from) t1 = sym_from;  // a launch is all-pairs (keys of later batches) or it is not
    const bool sym = t0 >= sym_from;
    if (sym) {
      // a wave logs about 32 rows x 128 keys x P(score >= prior) pairs per tile, the prior passing ~2.5 k_eff of a
      // row's n scores: the launch is cut so that four times that fits the wave's log
      const double per_tile = 32.0 * kBN * 2.5 * ca.k_eff / (double)n;
      int lim = (int)(kLogCap / 4 / (per_tile > 1e-9 ? per_tile : 1e-9)) * max_split;
      if (lim < 2 * max_split) lim = 2 * max_split;
      if (t1 - t0 > lim) t1 = t0 + lim;
    }
    ca.tile0 = t0;
    ca.tile1 = t1;
    int splits = (t0 == 0 && !(has_prior && first_pct > 0)) ? 1 : max_split;
    while (splits > 1 && (t1 - t0) < 2 * splits) --splits;  // at least two tiles per workgroup
    if (splits > 1 && splits < min_split) splits = 1;
    ca.splits = splits;
    const dim3 grid2(grid.x, splits);
#define ANIREC_LAUNCH_CAND(D, M)                                                                    \
  do {                                                                                              \
    const size_t shm = (M) ? (size_t)(wide ? 8 : 4) * 3 * 512 : 0;  /* the waves' LDS mask images */   \
    if (wide)                                                                                       \
      hipLaunchKernelGGL((k_cand<D, 8, M>), grid2, block, shm, s, ca);                              \
    else                                                                                            \
      hipLaunchKernelGGL((k_cand<D, 4, M>), grid2, block, shm, s, ca);                              \
  } while (0)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (g_time_cand) {
      ANIREC_HIP_CHECK(hipEventCreate(&ev0));
      ANIREC_HIP_CHECK(hipEventCreate(&ev1));
      ANIREC_HIP_CHECK(hipEventRecord(ev0, s));
    }
    if (sym) {
      const size_t lw = (size_t)grid2.x * grid2.y * (wide ? 8 : 4);
      if (lw > log_waves((size_t)nq)) return ANIREC_EINVAL;  // (cannot happen: the workspace was sized by the same formula)
      (void)hipMemsetAsync(ca.logcnt, 0, lw * 4, s);  // (workgroups of an empty split write nothing)
      if (wide)
        hipLaunchKernelGGL((k_cand<0, 8, false, true>), grid2, block, 0, s, ca);
      else
        hipLaunchKernelGGL((k_cand<0, 4, false, true>), grid2, block, 0, s, ca);
      hipLaunchKernelGGL(k_scatter_log, dim3((unsigned)lw), dim3(256), 0, s, ca.log, ca.logcnt, kLogCap, sp->inbox_w,
                         sp->icnt_w, sp->ovf);
    } else if (masked)
      ANIREC_LAUNCH_CAND(0, true);
    else if (mode == 1)
      ANIREC_LAUNCH_CAND(1, false);
    else if (mode == 2)
      ANIREC_LAUNCH_CAND(2, false);
    else if (mode == 4)
      ANIREC_LAUNCH_CAND(4, false);
    else if (mode == 8)
      ANIREC_LAUNCH_CAND(8, false);
    else
      ANIREC_LAUNCH_CAND(0, false);
#undef ANIREC_LAUNCH_CAND
    if (g_time_cand) {
      ANIREC_HIP_CHECK(hipEventRecord(ev1, s));
      timed.push_back(ev0);
      timed.push_back(ev1);
    }
    // (none after the last super-step: the re-rank folds the split regions itself)
    if (t1 < ntiles) hipLaunchKernelGGL(k_refresh, dim3(side_grid(nq)), dim3(64), 0, s, ca);
    ANIREC_HIP_CHECK(hipGetLastError());
    if (mid_ev && (long long)t1 * 100 >= (long long)ntiles * mid_pct) {
      ANIREC_HIP_CHECK(hipEventRecord(mid_ev, s));
      mid_ev = nullptr;
    }
    step = (int)((long long)t1 * growth_pct / 100);  // next super-step: growth_pct % of the tiles seen so far
    if (step < 1) step = 1;
    t0 = t1;
    ++n_launch;
  }
  if (g_time_cand) {  // blocking: only bench.py's roofline leg arms this
    g_cand_launches += (int)timed.size() / 2;  // summed over the batches of a job until the next arm / disarm
    for (size_t i = 0; i + 1 < timed.size(); i += 2) {
      float ms = 0.f;
      (void)hipEventSynchronize(timed[i + 1]);
      (void)hipEventElapsedTime(&ms, timed[i], timed[i + 1]);
      g_cand_ms += ms;
      (void)hipEventDestroy(timed[i]);
      (void)hipEventDestroy(timed[i + 1]);
    }
  }
  if (mode == 16 && !masked) {
    // diagnostic: ONE launch over all keys with the FINAL thresholds (what the filter costs when a row only ever
    // sees its k + window candidates), and one with thresholds nothing passes (filter without any hit)
    for (int pass = 0; pass < 2; ++pass) {
      (void)hipMemsetAsync(ca.cnt, 0, (size_t)nq * 4, s);
      if (pass == 1) {
        std::vector<float> big((size_t)nq, 2.0f);
        (void)hipMemcpyAsync(ca.theta, big.data(), (size_t)nq * 4, hipMemcpyHostToDevice, s);
      }
      ca.tile0 = 0;
      ca.tile1 = ntiles;
      ca.splits = 1;
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, s);
      if (wide)
        hipLaunchKernelGGL((k_cand<0, 8, false>), grid, block, 0, s, ca);
      else
        hipLaunchKernelGGL((k_cand<0, 4, false>), grid, block, 0, s, ca);
      (void)hipEventRecord(e1, s);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      fprintf(stderr, "[anirec topk debug] one launch over %d tiles with %s thresholds: %.3f ms\n", ntiles,
              pass == 0 ? "the final" : "unreachable", ms);
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
  }
  if (mode == 2 && !masked) {
    unsigned long long hv[2] = {0, 0};
    (void)hipMemcpyAsync(hv, dbg_words, 16, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    fprintf(stderr, "[anirec topk debug] nq=%d n=%d appends/row=%.1f super-steps=%d\n", nq, n,
            (double)hv[0] / nq, n_launch);
  }
  if (mode == 4 && !masked) {  // stamps of the LAST super-step
    std::vector<unsigned long long> hv(n_waves * 4);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(hv.data(), stamps, n_waves * 32, hipMemcpyDeviceToHost);
    (void)hipFree(stamps);
    double tot = 0, st = 0, br = 0;
    for (size_t i = 0; i < n_waves; ++i) {
      tot += (double)hv[4 * i];
      st += (double)hv[4 * i + 1];
      br += (double)hv[4 * i + 2];
    }
    const double ntl = (double)hv[3];
    fprintf(stderr, "[anirec topk stamps] last super-step: %d tiles; per wave per tile: total %.0f cycles, "
            "vmcnt wait %.0f, barrier %.0f\n", (int)ntl, tot / n_waves / ntl, st / n_waves / ntl,
            br / n_waves / ntl);
  }
  return ANIREC_OK;
}

extern "C" {

int anirec_topk_mfma_timing(int32_t enable, float *cand_ms, int32_t *launches) {
  if (cand_ms) *cand_ms = g_cand_ms;
  if (launches) *launches = g_cand_launches;
  g_cand_ms = 0.f;
  g_cand_launches = 0;
  g_time_cand = enable != 0;
  return ANIREC_OK;
}

// Wb holds whole key tiles: rows n .. padded_keys(n)-1 are NaN rows (never candidates)
static inline size_t padded_keys(int32_t n) { return ((size_t)n + kBN - 1) / kBN * kBN; }

// ---- one batch of query rows against the converted keys --------------------------------------------------
// per-batch buffers ("lane"): Qb (rows*256 B) | cand (rows*kCap*8) | cnt, theta (rows*4 each) | cnt2 (rows*16)
struct LaneBufs {
  _Float16 *Qb;
  uint2 *cand;
  int32_t *cnt;
  float *theta;
  int32_t *cnt2;
};
static inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }
static size_t lane_bytes(size_t rows) {
  return al256(rows * 256) + al256(rows * kCap * 8) + 2 * al256(rows * 4) + al256(rows * kMaxSplit * 4);
}
static LaneBufs carve_lane(char *p, size_t rows) {
  LaneBufs b;
  b.Qb = (_Float16 *)p;
  p += al256(rows * 256);
  b.cand = (uint2 *)p;
  p += al256(rows * kCap * 8);
  b.cnt = (int32_t *)p;
  p += al256(rows * 4);
  b.theta = (float *)p;
  p += al256(rows * 4);
  b.cnt2 = (int32_t *)p;
  return b;
}

// rows init -> query rows to fp16 -> super-steps of the key stream -> exact re-rank, all on stream s
static int run_batch(const float *What, const _Float16 *Wb, int n, const int32_t *queries, int nq, int exclude_self, int k,
                     float theta0, const float *theta0_dev, const LaneBufs &lb, int32_t *unnorm, bool zero_unnorm,
                     int32_t *out_idx, float *out_score, int32_t *flags, int mode, void *dbg_words, hipStream_t s,
                     hipEvent_t mid_ev = nullptr, int mid_pct = 0, const SymPlan *sp = nullptr) {
  if ((size_t)nq * kCap * 8 >= ((size_t)1 << 32)) return ANIREC_EINVAL;  // 32-bit candidate offsets: batch the queries
  int b2 = (nq + 7) / 8;
  if (b2 > 8192) b2 = 8192;
  hipLaunchKernelGGL(k_init_rows, dim3((nq + 255) / 256), dim3(256), 0, s, lb.cnt, lb.cnt2, lb.theta, flags, nq,
                     zero_unnorm ? unnorm : nullptr, theta0, theta0_dev);
  hipLaunchKernelGGL(k_to_f16, dim3(b2), dim3(256), 0, s, What, queries, nq, nq, nullptr, 1, lb.Qb, unnorm);
  ANIREC_HIP_CHECK(hipGetLastError());
  CandArgs ca;
  cand_defaults(ca);
  ca.Qb = (const uint4 *)lb.Qb;
  ca.Wb = (const uint4 *)Wb;
  ca.nq = nq;
  ca.n = n;
  ca.k_eff = exclude_self ? k + 1 : k;
  ca.cand = lb.cand;
  ca.cnt = lb.cnt;
  ca.cnt2 = lb.cnt2;
  ca.splits = 1;
  ca.theta = lb.theta;
  ca.flags = flags;
  ca.dbg = nullptr;
  ca.watched = nullptr;
  ca.wwords = 0;
  if (mode == 2) {
    ca.dbg = (unsigned long long *)dbg_words;
    (void)hipMemsetAsync(dbg_words, 0, 16, s);
  }
  unsigned long long *stamps = nullptr;
  const size_t n_waves = ((size_t)nq + 255) / 256 * 8;
  if (mode == 4) {  // diagnostic build with in-kernel stamps
    ANIREC_HIP_CHECK(hipMalloc((void **)&stamps, n_waves * 32));
    ca.dbg = stamps;
  }
  if (sp && sp->inbox_start) {  // what the earlier batches of this chain found for these rows: first candidates
    hipLaunchKernelGGL(k_merge_inbox, dim3(nq), dim3(64), 0, s, lb.cand, lb.cnt, flags, nq, sp->row0, sp->inbox_start,
                       sp->icnt_start);
    hipLaunchKernelGGL(k_refresh, dim3(side_grid(nq)), dim3(64), 0, s, ca);
    ANIREC_HIP_CHECK(hipGetLastError());
  }
  {
    const int rc = run_super_steps(ca, n, nq, false, mode, dbg_words, stamps, n_waves, s, mid_ev, mid_pct,
                                   theta0_dev != nullptr || theta0 > kThetaInit, sp);
    if (rc) return rc;
  }
  if (sp) {  // the other chain's batches must have delivered before the re-rank merges their inbox
    if (sp->done_ev) ANIREC_HIP_CHECK(hipEventRecord(sp->done_ev, s));
    if (sp->wait_ev) ANIREC_HIP_CHECK(hipStreamWaitEvent(s, sp->wait_ev, 0));
  }
  RerankArgs ra;
  ra.What = What;
  ra.Qf = nullptr;
  ra.qidx = queries;
  ra.nq = nq;
  ra.n = n;
  ra.k = k;
  ra.k_eff = exclude_self ? k + 1 : k;
  ra.exclude_self = exclude_self ? 1 : 0;
  ra.cand = lb.cand;
  ra.cnt = lb.cnt;
  ra.cnt2 = ca.cnt2;
  ra.inbox = sp ? sp->inbox_end : nullptr;
  ra.icnt = sp ? sp->icnt_end : nullptr;
  ra.sym_ovf = sp ? sp->ovf : nullptr;
  ra.theta = lb.theta;
  ra.flags = flags;
  ra.out_idx = out_idx;
  ra.out_score = out_score;
  // self exclusion is by key index; without it the query index is only used to fetch the row
  ra.hs = ra.hb = 0.f;
  ra.sign = 1.f;
  ra.unnorm = unnorm;
  hipLaunchKernelGGL(k_rerank<false>, dim3(side_grid(nq)), dim3(64), 0, s, ra);
  return (int)hipGetLastError();
}

// workspace: Wb (padded_keys(n)*256 B) | 256 B {dbg words, unnorm @64, learnt prior @128} | one lane of nq rows
size_t anirec_topk_mfma_workspace_bytes(int32_t n, int32_t nq) {
  if (n < 1 || nq < 1) return 0;
  return al256(padded_keys(n) * 256) + 256 + lane_bytes((size_t)nq);
}

int anirec_cosine_topk_mfma(const float *What, int32_t n, const int32_t *queries, int32_t nq,
                            const uint8_t *keep, int32_t exclude_self, int32_t k, int32_t *out_idx,
                            float *out_score, int32_t *flags_out, void *workspace,
                            size_t workspace_bytes, void *stream) {
  return anirec_cosine_topk_mfma_prior(What, n, queries, nq, keep, exclude_self, k, kThetaInit, out_idx, out_score,
                                       flags_out, workspace, workspace_bytes, stream);
}

int anirec_cosine_topk_mfma_prior(const float *What, int32_t n, const int32_t *queries, int32_t nq,
                                  const uint8_t *keep, int32_t exclude_self, int32_t k, float theta0,
                                  int32_t *out_idx, float *out_score, int32_t *flags_out, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  if (!What || !queries || !out_idx || !out_score || !flags_out || !workspace) return ANIREC_EINVAL;
  if (!(theta0 >= kThetaInit && theta0 <= 1.0f)) return ANIREC_EINVAL;
  if (n < 1 || nq < 0 || k < 1 || k > ANIREC_MAX_TOPK - 1) return ANIREC_EINVAL;
  if (nq == 0) return ANIREC_OK;
  if (workspace_bytes < anirec_topk_mfma_workspace_bytes(n, nq)) return ANIREC_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char *p = (char *)workspace;
  _Float16 *Wb = (_Float16 *)p;
  p += al256(padded_keys(n) * 256);
  char *misc = p;
  int32_t *unnorm = (int32_t *)(misc + 64);
  p += 256;
  const LaneBufs lb = carve_lane(p, (size_t)nq);
  int b1 = (n + 7) / 8;
  if (b1 > 8192) b1 = 8192;
  ANIREC_HIP_CHECK(hipMemsetAsync(misc, 0, 256, s));  // unnorm (and the debug words) before the key pass sets it
  hipLaunchKernelGGL(k_to_f16, dim3(b1), dim3(256), 0, s, What, nullptr, n, (int)padded_keys(n), keep, 0, Wb, unnorm);
  ANIREC_HIP_CHECK(hipGetLastError());
  const char *dbg = getenv("ANIREC_TOPK_DEBUG");
  const int mode = dbg ? atoi(dbg) : 0;
  return run_batch(What, Wb, n, queries, nq, exclude_self, k, theta0, nullptr, lb, unnorm, false, out_idx, out_score,
                   flags_out, mode, misc, s);
}

// ---- the whole job: many query rows in batches, two (or more) batches in flight ---------------------------
// The matrix cores idle while a batch's k_refresh / k_rerank run (one wave per row, 15 % of a batch's time at
// 350 k keys, more than half at 18 k).  Batches are independent, so they are dealt to `lanes` stream-ordered chains
// (the caller's stream + side streams of a process-wide pool, forked and joined by events: one cross-stream edge
// per JOB, not per step).
// What the hardware does with that (round 3, rocprofv3 traces of the 350 k x 350 k top-100 job, interleaved A/B on
// one box): two chains that start together stay in LOCKSTEP — both in k_cand (each at half speed: an 8-wave k_cand
// workgroup takes 2 x 160 VGPRs per SIMD and 64 KB of LDS, one per CU), then both in k_refresh.  The gain is the
// filled tails of the k_cand launches and the overlap of the odd re-rank: 33.2 -> 31.8 ms.  Forcing the chains out
// of phase (ANIREC_TOPK_STAGGER_PCT: chain 2 starts when chain 1 has passed 20 / 33 / 66 % of its key stream) is
// SLOWER, 32.4-32.8 ms, and so are three or four chains (32.5-32.7): one-wave side workgroups take every wave slot a
// finished k_cand workgroup frees, so the next k_cand workgroup of the other chain cannot start on that CU until the
// side kernel has drained — the kernels serialise per CU instead of overlapping.  Bounding the side kernels' grids
// (ANIREC_TOPK_SIDE_WAVES = 4 / 8 / 16 waves per CU, rows strided) leaves k_cand its room but makes the side
// kernels — chains of dependent loads, ballots and an O(survivors^2) rank per wave — 2x slower than they are hidden:
// 35-39 ms.  And k_cand itself gains nothing from a second resident workgroup: a 128-VGPR build of the no-filter loop
// (two workgroups, 4 waves per SIMD) ran 7.82 ms per 131 072 queries against 7.80 ms (1.50 PFLOP/s either way): the
// MFMA loop is pipe- / power-bound, not latency-bound.  Longer super-steps for rows that start from a prior
// (ANIREC_TOPK_GROWTH_PRIOR 150-1000 %) cost 10-15 % (38-40 ms against 34.8 on that box), shorter ones (35-70 %)
// 3-8 %: a row's own threshold, refreshed at every doubling, is worth more than the launches it costs.
// Last, the re-rank of batch i taken OUT of its chain — a low-priority stream, grid bounded to 2 / 4 / 8 / 16 / 64
// waves per CU, beside the key stream of batch i + 1 (two sets of buffers): 39-44 ms against 33.0 — the re-rank's
// 3.7 GB of random row gathers per 65 536 queries push the fp16 key table (90 MB, re-streamed by every k_cand
// workgroup) out of L2 / Infinity Cache, and k_cand pays for every tile with an HBM round trip.  The side kernels
// stay where they are: in their chain, alone on the chip.
constexpr int kMaxLanes = 4;
struct LanePool {
  int device = -1;
  hipStream_t side[kMaxLanes - 1] = {nullptr, nullptr, nullptr};
  hipEvent_t fork = nullptr;
  hipEvent_t join[kMaxLanes - 1] = {nullptr, nullptr, nullptr};
  hipEvent_t mid[kMaxLanes - 1] = {nullptr, nullptr, nullptr};  // chain l passed the stagger point of its first batch
  hipEvent_t cand_done[ANIREC_TOPK_MAX_BATCHES] = {};  // all-pairs job: batch b's last k_cand is behind this
};
// One pool per device of the process (a process that runs jobs on cuda:0 and then on cuda:1 gets two).  The side
// streams and events of a pool are shared by every job on its device: g_pool_mu is held for the WHOLE enqueue of a
// job (anirec_cosine_topk_job takes it), so two host threads — or two caller streams — never interleave their
// event records; their jobs then run one after the other on the side streams, which is all a shared pool can offer.
constexpr int kMaxPoolDevices = 16;
static LanePool g_pools[kMaxPoolDevices];
static std::mutex g_pool_mu;

// (caller holds g_pool_mu)
static int pool_get(LanePool **out) {
  int dev = 0;
  ANIREC_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxPoolDevices) return ANIREC_ENODEVICE;
  LanePool &pl = g_pools[dev];
  if (pl.device == dev) {
    *out = &pl;
    return ANIREC_OK;
  }
  for (int i = 0; i < kMaxLanes - 1; ++i) {
    ANIREC_HIP_CHECK(hipStreamCreateWithFlags(&pl.side[i], hipStreamNonBlocking));
    ANIREC_HIP_CHECK(hipEventCreateWithFlags(&pl.join[i], hipEventDisableTiming));
    ANIREC_HIP_CHECK(hipEventCreateWithFlags(&pl.mid[i], hipEventDisableTiming));
  }
  ANIREC_HIP_CHECK(hipEventCreateWithFlags(&pl.fork, hipEventDisableTiming));
  for (int i = 0; i < ANIREC_TOPK_MAX_BATCHES; ++i)
    ANIREC_HIP_CHECK(hipEventCreateWithFlags(&pl.cand_done[i], hipEventDisableTiming));
  pl.device = dev;
  *out = &pl;
  return ANIREC_OK;
}

static int default_lanes() {
  const char *e = getenv("ANIREC_TOPK_LANES");
  int l = e ? atoi(e) : 2;
  return l < 1 ? 1 : (l > kMaxLanes ? kMaxLanes : l);
}

int anirec_cosine_topk_job_plan(int32_t nq, int32_t k, int32_t prior_auto, int32_t max_batch, int32_t lanes,
                                int32_t *starts_host, int32_t *n_batches_host, int32_t *learn_batches_host) {
  if (!starts_host || !n_batches_host || !learn_batches_host || nq < 0 || k < 1) return ANIREC_EINVAL;
  if (max_batch < 1) max_batch = 131072;
  if (max_batch > 1000000) max_batch = 1000000;  // rows * kCap * 8 < 2^32
  if (lanes < 1) lanes = default_lanes();
  if (lanes > kMaxLanes) lanes = kMaxLanes;
  const char *mk = getenv("ANIREC_TOPK_PRIOR_MIN_K");
  const char *pe = getenv("ANIREC_TOPK_PRIOR");
  // (small k: a row appends few candidates anyway and the prior buys nothing)
  const bool may_learn = prior_auto && k >= (mk ? atoi(mk) : 32) && !(pe && atoi(pe) == 0);
  int nb = 0, learn = 0, q0 = 0;
  starts_host[0] = 0;
  // the learning batch is small — 16 384 rows: 64 workgroups, which the key-range splits spread over the chip — so
  // that as many rows as possible run with the prior; a caller's smaller `max_batch` learns from its first batch
  if (may_learn && nq >= 49152 && max_batch >= 65536) {
    q0 = 16384;
  } else if (may_learn && max_batch < nq && max_batch >= 16384) {
    q0 = max_batch;
  }
  if (q0 > 0) {
    learn = 1;
    starts_host[++nb] = q0;
  }
  const int rem = nq - q0;
  if (rem > 0) {
    int cnt = (rem + max_batch - 1) / max_batch;
    // every chain the same number of batches — when there is enough work for two batches in flight to pay (measured:
    // 350 k x 350 k top-100 33.2 -> 31.8 ms on two chains; the 18 k x 18 k job is faster as ONE batch, 0.66 vs 0.69 ms)
    if (lanes > 1 && rem >= lanes * 32768) cnt = (cnt + lanes - 1) / lanes * lanes;
    if (nb + cnt > ANIREC_TOPK_MAX_BATCHES) return ANIREC_EINVAL;
    int rows = (rem + cnt - 1) / cnt;
    rows = (rows + 255) / 256 * 256;  // whole 256-row workgroups
    if (rows > max_batch) rows = max_batch;
    while (q0 < nq) {
      q0 = q0 + rows < nq ? q0 + rows : nq;
      starts_host[++nb] = q0;
    }
  }
  *n_batches_host = nb;
  *learn_batches_host = learn;
  return ANIREC_OK;
}

// (Measured and dropped: the batch's OWN block as a triangle too — a row block starts its stream at its diagonal tile
// and logs the pairs with the batch's later rows, merged by the re-rank from the batch's own inbox: 0.55 instead of 0.61
// of the flops, and 25.2-26.0 ms against 24.5-25.0 for 4-10 batches: the own-block launches then run as the dearer
// all-pairs instantiation, with workgroups of very different stream lengths.)
// The all-pairs job's plan: the learning batch, then `main_batches` batches of EQUAL WORK — batch b streams the
// learning batch's keys, its own and those of the later batches, so the later a batch the more rows it takes (at
// 350 k rows in 4 batches: 54 k, 65 k, 83 k, 130 k) and the two chains finish together.  Falls back to the default
// plan (anirec_cosine_topk_job_plan) when the job is too small to learn a prior.
int anirec_cosine_topk_allpairs_plan(int32_t n, int32_t k, int32_t lanes, int32_t main_batches, int32_t *starts_host,
                                     int32_t *n_batches_host, int32_t *learn_batches_host) {
  if (!starts_host || !n_batches_host || !learn_batches_host || n < 0 || k < 1) return ANIREC_EINVAL;
  const char *mk = getenv("ANIREC_TOPK_PRIOR_MIN_K");
  const char *pe = getenv("ANIREC_TOPK_PRIOR");
  const char *be = getenv("ANIREC_TOPK_SYM_BATCHES");
  // (as many batches as leave the first, smallest one ~24 k rows — enough 256-row workgroups, cut in three key ranges,
  // for every CU; more batches waste less on the batches' own diagonal blocks, which are computed in full)
  if (main_batches < 1) {
    main_batches = be ? atoi(be) : 2 * (int)((double)n / 88000.0 + 0.5);  // an even number: two chains
    if (!be) main_batches = main_batches < 2 ? 2 : (main_batches > 8 ? 8 : main_batches);
  }
  // (the plain job learns a prior from k = 32 on; here the prior is what makes the shortcut possible, and it pays at
  // k = 10 too: 350 k x 350 k top-10 28.3 -> 18.8 ms)
  const bool may_learn = k >= (mk ? atoi(mk) : 8) && !(pe && atoi(pe) == 0);
  // (below ~200 k rows the shortcut saves less than its extra launches and the caller's pilot cost: measured 4.7 vs
  // 4.6 ms at 100 k rows, 8.6 vs 8.2 at 150 k, 10.3-15.8 vs 12.6-17.3 at 222 k, 24-25 vs 31.5 ms at 350 k)
  if (!may_learn || n < 196608 || main_batches < 2 || main_batches + 1 > ANIREC_TOPK_MAX_BATCHES)
    return anirec_cosine_topk_job_plan(n, k, 1, 0, lanes, starts_host, n_batches_host, learn_batches_host);
  const double k0 = 16384.0, rest = (double)n - k0;
  double lo = 0.0, hi = rest * (double)n;
  double r[ANIREC_TOPK_MAX_BATCHES];
  for (int it = 0; it < 100; ++it) {
    const double w = 0.5 * (lo + hi);
    double later = 0.0;
    for (int b = main_batches - 1; b >= 0; --b) {
      const double c = k0 + later;
      r[b] = 0.5 * (-c + sqrt(c * c + 4.0 * w));
      later += r[b];
    }
    if (later < rest)
      lo = w;
    else
      hi = w;
  }
  starts_host[0] = 0;
  starts_host[1] = 16384;
  int q0 = 16384;
  for (int b = 0; b < main_batches; ++b) {
    int rows = ((int)(r[b] + 0.5) + 255) / 256 * 256;
    if (rows < 256) rows = 256;
    if ((size_t)rows * kCap * 8 >= ((size_t)1 << 32)) return ANIREC_EINVAL;
    q0 = (b + 1 == main_batches || q0 + rows >= n) ? n : q0 + rows;
    starts_host[2 + b] = q0;
    if (q0 == n && b + 1 < main_batches) {  // (rounding used the rows up early)
      main_batches = b + 1;
      break;
    }
  }
  if ((size_t)(n - starts_host[main_batches]) * kCap * 8 >= ((size_t)1 << 32)) return ANIREC_EINVAL;
  *n_batches_host = main_batches + 1;
  *learn_batches_host = 1;
  return ANIREC_OK;
}

static int max_batch_rows(const int32_t *starts, int nb) {
  int m = 0;
  for (int b = 0; b < nb; ++b) m = starts[b + 1] - starts[b] > m ? starts[b + 1] - starts[b] : m;
  return m;
}

// Wb | 256 B misc | `lanes` lanes of `rows` rows
size_t anirec_cosine_topk_job_workspace_bytes(int32_t n, int32_t max_batch_rows_, int32_t lanes) {
  if (n < 1 || max_batch_rows_ < 1 || lanes < 1 || lanes > kMaxLanes) return 0;
  return al256(padded_keys(n) * 256) + 256 + (size_t)lanes * lane_bytes((size_t)max_batch_rows_);
}

// all-pairs mode (prior_mode 3): + two inboxes of n rows (entries, slot counters)
static size_t inbox_bytes(size_t n) { return 2 * al256(n * kInbox * 8) + 2 * al256(n * 4); }
static size_t log_bytes(size_t rows) { return al256(log_waves(rows) * kLogCap * 16) + al256(log_waves(rows) * 4); }
size_t anirec_cosine_topk_allpairs_workspace_bytes(int32_t n, int32_t max_batch_rows_, int32_t lanes) {
  const size_t base = anirec_cosine_topk_job_workspace_bytes(n, max_batch_rows_, lanes);
  return base ? base + inbox_bytes((size_t)n) + (size_t)lanes * log_bytes((size_t)max_batch_rows_) : 0;
}

// queries[i] == i for every i, or *bad = 1 (the all-pairs shortcut is only valid for the identity)
__global__ void k_check_identity(const int32_t *queries, int nq, int32_t *bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq && queries[i] != i) *bad = 1;
}

int anirec_cosine_topk_job(const float *What, int32_t n, const int32_t *queries, int32_t nq, const uint8_t *keep,
                           int32_t exclude_self, int32_t k, int32_t prior_mode, float theta0,
                           const int32_t *starts_host, int32_t n_batches, int32_t learn_batches, int32_t lanes,
                           int32_t *out_idx, float *out_score, int32_t *flags_out, void *workspace,
                           size_t workspace_bytes, void *stream) {
  if (!What || !queries || !out_idx || !out_score || !flags_out || !workspace || !starts_host) return ANIREC_EINVAL;
  if (n < 1 || nq < 0 || k < 1 || k > ANIREC_MAX_TOPK - 1) return ANIREC_EINVAL;
  if (prior_mode < 0 || prior_mode > 3 || lanes < 1 || lanes > kMaxLanes) return ANIREC_EINVAL;
  if (prior_mode == 2 && !(theta0 >= kThetaInit && theta0 <= 1.0f)) return ANIREC_EINVAL;
  if (n_batches < 0 || n_batches > ANIREC_TOPK_MAX_BATCHES || learn_batches < 0 || learn_batches > 1 ||
      learn_batches > n_batches)
    return ANIREC_EINVAL;
  if (nq == 0 || n_batches == 0) return nq == 0 ? ANIREC_OK : ANIREC_EINVAL;
  if (starts_host[0] != 0 || starts_host[n_batches] != nq) return ANIREC_EINVAL;
  for (int b = 0; b < n_batches; ++b)
    if (starts_host[b + 1] <= starts_host[b]) return ANIREC_EINVAL;
  // all-pairs mode: every key row is a query row, in order; it needs the learnt prior, whole key tiles per batch and
  // at most two chains (an inbox per chain) — anything else runs as mode 1, with the same results
  bool sym = prior_mode == 3;
  if (sym) {
    prior_mode = 1;
    if (nq != n || keep != nullptr || learn_batches != 1 || n_batches < 3 || lanes > 2) sym = false;
    for (int b = 1; sym && b < n_batches; ++b)
      if (starts_host[b] % kBN != 0) sym = false;
    const char *se = getenv("ANIREC_TOPK_SYM");
    if (se && atoi(se) == 0) sym = false;
    const char *dbg0 = getenv("ANIREC_TOPK_DEBUG");
    if (dbg0 && atoi(dbg0) != 0) sym = false;
  }
  if (prior_mode != 1) learn_batches = 0;
  const int rows = max_batch_rows(starts_host, n_batches);
  if (g_time_cand) lanes = 1;  // the roofline leg times k_cand launches that run alone
  if (n_batches - learn_batches < lanes) lanes = n_batches - learn_batches > 0 ? n_batches - learn_batches : 1;
  const size_t ws_base = anirec_cosine_topk_job_workspace_bytes(n, rows, lanes);
  if (workspace_bytes < ws_base + (sym ? inbox_bytes((size_t)n) + (size_t)lanes * log_bytes((size_t)rows) : 0))
    return ANIREC_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  LanePool *pool = nullptr;
  std::unique_lock<std::mutex> pool_lock(g_pool_mu, std::defer_lock);
  if (lanes > 1 || sym) {
    pool_lock.lock();  // until the job is enqueued (see g_pools)
    const int rc = pool_get(&pool);
    if (rc) return rc;
  }
  char *p = (char *)workspace;
  uint2 *inbox[2] = {nullptr, nullptr};
  int32_t *icnt[2] = {nullptr, nullptr};
  uint4 *logs[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  int32_t *logcnts[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  if (sym) {
    char *q = p + ws_base;
    for (int par = 0; par < 2; ++par) {
      inbox[par] = (uint2 *)q;
      q += al256((size_t)n * kInbox * 8);
    }
    for (int par = 0; par < 2; ++par) {
      icnt[par] = (int32_t *)q;
      q += al256((size_t)n * 4);
    }
    for (int l = 0; l < lanes; ++l) {
      logs[l] = (uint4 *)q;
      q += al256(log_waves((size_t)rows) * kLogCap * 16);
      logcnts[l] = (int32_t *)q;
      q += al256(log_waves((size_t)rows) * 4);
    }
  }
  _Float16 *Wb = (_Float16 *)p;
  p += al256(padded_keys(n) * 256);
  char *misc = p;
  int32_t *unnorm = (int32_t *)(misc + 64);
  float *theta0_dev = (float *)(misc + 128);
  p += 256;
  LaneBufs lb[kMaxLanes];
  for (int l = 0; l < lanes; ++l) lb[l] = carve_lane(p + (size_t)l * lane_bytes((size_t)rows), (size_t)rows);
  int b1 = (n + 7) / 8;
  if (b1 > 8192) b1 = 8192;
  ANIREC_HIP_CHECK(hipMemsetAsync(misc, 0, 256, s));
  hipLaunchKernelGGL(k_to_f16, dim3(b1), dim3(256), 0, s, What, nullptr, n, (int)padded_keys(n), keep, 0, Wb, unnorm);
  ANIREC_HIP_CHECK(hipGetLastError());
  if (sym) {
    ANIREC_HIP_CHECK(hipMemsetAsync(icnt[0], 0, 2 * al256((size_t)n * 4), s));
    // a query list that is not the identity flags every row (the `unnorm` word: rows fall to the caller's exact path)
    hipLaunchKernelGGL(k_check_identity, dim3((nq + 255) / 256), dim3(256), 0, s, queries, nq, unnorm);
    ANIREC_HIP_CHECK(hipGetLastError());
  }
  const float th_imm = prior_mode == 2 ? theta0 : kThetaInit;
  const float *th_dev = nullptr;
  int b = 0;
  for (; b < learn_batches; ++b) {  // the learning batch runs alone, without a prior
    const int q0 = starts_host[b], cnt = starts_host[b + 1] - q0;
    const int rc = run_batch(What, Wb, n, queries + q0, cnt, exclude_self, k, kThetaInit, nullptr, lb[0], unnorm, false,
                             out_idx + (size_t)q0 * k, out_score + (size_t)q0 * k, flags_out + q0, 0, nullptr, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_learn_prior, dim3(1), dim3(1024), 0, s, out_score + (size_t)q0 * k, cnt, (int)k, theta0_dev);
    ANIREC_HIP_CHECK(hipGetLastError());
    th_dev = theta0_dev;
  }
  // from here on the side streams may hold work of this job: EVERY way out joins them back into the caller's
  // stream first (an error return with un-joined side streams would let the caller reuse the workspace under them)
  auto join_all = [&]() -> int {
    if (lanes > 1) {
      for (int l = 1; l < lanes; ++l) {
        ANIREC_HIP_CHECK(hipEventRecord(pool->join[l - 1], pool->side[l - 1]));
        ANIREC_HIP_CHECK(hipStreamWaitEvent(s, pool->join[l - 1], 0));
      }
    }
    return ANIREC_OK;
  };
  if (lanes > 1) {
    ANIREC_HIP_CHECK(hipEventRecord(pool->fork, s));
    for (int l = 1; l < lanes; ++l) {
      const hipError_t we = hipStreamWaitEvent(pool->side[l - 1], pool->fork, 0);
      if (we != hipSuccess) {
        (void)join_all();
        return (int)we;
      }
    }
  }
  // ANIREC_TOPK_STAGGER_PCT > 0 (experiment knob, default off): chain l starts its first batch when chain l-1 has
  // passed that share of the key stream of ITS first batch.
  const char *sg = getenv("ANIREC_TOPK_STAGGER_PCT");
  const int stagger = sg ? atoi(sg) : 0;
  const char *dbg = getenv("ANIREC_TOPK_DEBUG");
  // diagnostic builds of k_cand: 1 without the filter, 8 without the candidate stores (timing only), 4 with in-kernel
  // stamps of the last super-step of every batch (printed to stderr; results valid)
  const int mode = dbg && (atoi(dbg) == 1 || atoi(dbg) == 8 || atoi(dbg) == 4) ? atoi(dbg) : 0;
  for (int i = 0; b < n_batches; ++b, ++i) {
    const int l = i % lanes;
    hipStream_t st = l == 0 ? s : pool->side[l - 1];
    const int q0 = starts_host[b], cnt = starts_host[b + 1] - q0;
    hipEvent_t mid = nullptr;
    if (i < lanes && stagger > 0 && lanes > 1) {
      if (l > 0) {
        const hipError_t we = hipStreamWaitEvent(st, pool->mid[l - 1], 0);
        if (we != hipSuccess) {
          (void)join_all();
          return (int)we;
        }
      }
      if (l + 1 < lanes && i + 1 < n_batches - learn_batches) mid = pool->mid[l];
    }
    SymPlan sp;
    if (sym) {
      // batch i streams the learning batch's keys, its own, and those of the later batches (all-pairs launches);
      // the keys of the batches between arrive through the inboxes: parity = chain of the writer
      const int par = i & 1;
      sp.map_lo = starts_host[learn_batches] / kBN;
      sp.map_skip = (q0 - starts_host[learn_batches]) / kBN;
      sp.sym_from = b + 1 < n_batches ? sp.map_lo + cnt / kBN : 0x7fffffff;
      sp.row0 = q0;
      sp.theta0_dev = th_dev;
      sp.inbox_w = inbox[par];
      sp.icnt_w = icnt[par];
      sp.log = logs[l];
      sp.logcnt = logcnts[l];
      sp.ovf = (int32_t *)(misc + 192);
      sp.inbox_start = i >= 2 ? inbox[par] : nullptr;
      sp.icnt_start = icnt[par];
      sp.inbox_end = i >= 1 ? inbox[par ^ 1] : nullptr;
      sp.icnt_end = icnt[par ^ 1];
      sp.done_ev = pool->cand_done[b];
      sp.wait_ev = (i >= 1 && lanes > 1) ? pool->cand_done[b - 1] : nullptr;
    }
    const int rc = run_batch(What, Wb, n, queries + q0, cnt, exclude_self, k, th_imm, th_dev, lb[l], unnorm, false,
                             out_idx + (size_t)q0 * k, out_score + (size_t)q0 * k, flags_out + q0, mode, nullptr, st, mid,
                             stagger, sym ? &sp : nullptr);
    if (rc) {
      (void)join_all();
      return rc;
    }
  }
  return join_all();
}

// ------------------------------------------------------------------------------------------------
// model_recs batched: top-k unwatched anime by predicted rating for many users, on the matrix cores.
// ------------------------------------------------------------------------------------------------
// Ah (n_anime fp32 rows) | Uh (n_users fp32 rows) | Wb | Qb | cand | cnt | theta | 256 B | cnt2
size_t anirec_predict_topk_mfma_workspace_bytes(int32_t n_anime, int32_t n_users) {
  if (n_anime < 1 || n_users < 1) return 0;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  return al((size_t)n_anime * 512) + al((size_t)n_users * 512) + al(padded_keys(n_anime) * 256) +
         al((size_t)n_users * 256) + al((size_t)n_users * kCap * 8) + 2 * al((size_t)n_users * 4) + 256 +
         al((size_t)n_users * kMaxSplit * 4);
}

int anirec_predict_topk_mfma(const float *U, const float *A, int32_t n_anime, const int32_t *users,
                             int32_t n_users, const anirec_head *head, const uint32_t *watched, int32_t k,
                             int32_t *out_idx, float *out_p, int32_t *flags_out, void *workspace,
                             size_t workspace_bytes, void *stream) {
  if (!U || !A || !users || !head || !out_idx || !out_p || !flags_out || !workspace) return ANIREC_EINVAL;
  if (n_anime < 1 || n_users < 0 || k < 1 || k > ANIREC_MAX_TOPK - 1) return ANIREC_EINVAL;
  if (n_users == 0) return ANIREC_OK;
  if (workspace_bytes < anirec_predict_topk_mfma_workspace_bytes(n_anime, n_users)) return ANIREC_EWORKSPACE;
  if ((size_t)n_users * kCap * 8 >= ((size_t)1 << 32)) return ANIREC_EINVAL;  // batch the users
  if ((size_t)n_users * ((n_anime + 31) / 32) >= ((size_t)1 << 30)) return ANIREC_EINVAL;  // 32-bit mask offsets
  hipStream_t s = (hipStream_t)stream;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  char *p = (char *)workspace;
  float *Ah = (float *)p;
  p += al((size_t)n_anime * 512);
  float *Uh = (float *)p;
  p += al((size_t)n_users * 512);
  _Float16 *Wb = (_Float16 *)p;
  p += al(padded_keys(n_anime) * 256);
  _Float16 *Qb = (_Float16 *)p;
  p += al((size_t)n_users * 256);
  uint2 *cand = (uint2 *)p;
  p += al((size_t)n_users * kCap * 8);
  int32_t *cnt = (int32_t *)p;
  p += al((size_t)n_users * 4);
  float *theta = (float *)p;
  p += al((size_t)n_users * 4);
  int32_t *cnt2 = (int32_t *)(p + 256);
  // sigmoid(gamma*(w*c+b-mu)/sqrt(var+eps)+beta) = sigmoid(c*hs + hb), folded exactly as the exact path does
  float hs, hb;
  head_affine_f32(head, &hs, &hb);
  const float sign = hs < 0.f ? -1.f : 1.f;
  int b1 = (n_anime + 7) / 8, b2 = (n_users + 7) / 8;
  if (b1 > 8192) b1 = 8192;
  if (b2 > 8192) b2 = 8192;
  hipLaunchKernelGGL(k_norm_f16, dim3(b1), dim3(256), 0, s, A, nullptr, n_anime, (int)padded_keys(n_anime), 1.0f, Ah, Wb);
  hipLaunchKernelGGL(k_norm_f16, dim3(b2), dim3(256), 0, s, U, users, n_users, n_users, sign, Uh, Qb);
  hipLaunchKernelGGL(k_init_rows, dim3((n_users + 255) / 256), dim3(256), 0, s, cnt, cnt2, theta, flags_out, n_users,
                     nullptr, kThetaInit, nullptr);
  ANIREC_HIP_CHECK(hipGetLastError());
  CandArgs ca;
  cand_defaults(ca);
  ca.Qb = (const uint4 *)Qb;
  ca.Wb = (const uint4 *)Wb;
  ca.nq = n_users;
  ca.n = n_anime;
  ca.k_eff = k;
  ca.cand = cand;
  ca.cnt = cnt;
  ca.cnt2 = cnt2;
  ca.splits = 1;
  ca.theta = theta;
  ca.flags = flags_out;
  ca.dbg = nullptr;
  ca.watched = watched;
  ca.wwords = (n_anime + 31) / 32;
  const bool masked = watched != nullptr;
  {
    const int rc = run_super_steps(ca, n_anime, n_users, masked, 0, nullptr, nullptr, 0, s);
    if (rc) return rc;
  }
  RerankArgs ra;
  ra.What = Ah;
  ra.Qf = Uh;
  ra.qidx = nullptr;
  ra.nq = n_users;
  ra.n = n_anime;
  ra.k = k;
  ra.k_eff = k;
  ra.exclude_self = 0;
  ra.cand = cand;
  ra.cnt = cnt;
  ra.cnt2 = ca.cnt2;
  ra.inbox = nullptr;
  ra.icnt = nullptr;
  ra.sym_ovf = nullptr;
  ra.theta = theta;
  ra.flags = flags_out;
  ra.out_idx = out_idx;
  ra.out_score = out_p;
  ra.hs = hs;
  ra.hb = hb;
  ra.sign = sign;
  ra.unnorm = nullptr;  // k_norm_f16 normalises the rows itself
  // a zero / non-finite slope makes every rating equal (or NaN): nothing to rank on the MFMA side
  if (!(hs != 0.f) || !(hs == hs) || !(hb == hb)) {
    hipLaunchKernelGGL(k_flag_all, dim3((n_users + 255) / 256), dim3(256), 0, s, flags_out, n_users, out_idx, out_p, k);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(k_rerank<true>, dim3(side_grid(n_users)), dim3(64), 0, s, ra);
  return (int)hipGetLastError();
}

}  // extern "C"
