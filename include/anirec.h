/*
 * anirec.h — C ABI of libanirec.so, the MI355X (gfx950) kernel library behind the
 * anime_recommendations hot path.
 *
 * The reference (Dyrutter/anime_recommendations) is pure Python; its "plugin API" for
 * this path is the MLflow component surface (neural_network, similar_anime,
 * similar_users, model_recs) and the arithmetic lives in TensorFlow/Keras 2.12 and
 * NumPy calls made from those components.  Every entry point below replaces one such
 * call site (cited as reference file:line).  A maintainer of the reference would bind
 * them with ctypes (INTEGRATION.md shows the stubs).
 *
 * Conventions
 *  - extern "C", plain pointers + sizes, no C++/torch types.
 *  - every pointer is a DEVICE pointer unless the name ends in _host.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - return value: 0 = ok, negative = ANIREC_E*, positive = hipError_t.
 *  - the library never allocates device memory and never synchronises the stream
 *    (exception: the *_create/_destroy calls, which touch no stream work);
 *    all work is stream-ordered and graph-capturable.
 *  - embedding rows are fp32, row-major, exactly ANIREC_DIM (=128) wide
 *    (reference: config/config.yaml:63 embedding_size: 128).
 */
#ifndef ANIREC_H
#define ANIREC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANIREC_ABI_VERSION 4
#define ANIREC_DIM 128          /* embedding width (floats) */
#define ANIREC_MAX_BATCH 16384  /* ratings per rank per step handled by one sort workgroup */
#define ANIREC_CHUNK 32         /* max gradient contributions summed by one half-wave */
#define ANIREC_ADAM_BLOCKS 8192 /* grid of the dense Adam kernel == length of reg partials */
#define ANIREC_MAX_TOPK 128     /* k limit of the fused top-k kernels */
#define ANIREC_MAX_SEG 16       /* max head packets (ranks of one node) */
#define ANIREC_TOPK_MAX_BATCHES 64 /* query batches of one anirec_cosine_topk_job */
#define ANIREC_LAZY_WINDOW 8    /* steps between two flushes of the lazy dense Adam (anirec_trainer_run) */

enum {
  ANIREC_OK = 0,
  ANIREC_EINVAL = -1,     /* bad argument (null pointer, size out of range, dim != 128) */
  ANIREC_ENODEVICE = -2,  /* no HIP device / wrong architecture */
  ANIREC_EWORKSPACE = -3, /* workspace too small */
  ANIREC_ECAPTURE = -4,   /* graph capture / instantiate failed */
  ANIREC_ECOMM = -5       /* an RCCL call failed */
};

int anirec_abi_version(void);
/* Human-readable text for a status code returned by this library. */
const char *anirec_status_string(int status);
/* Name of the device the library would run on; returns status. */
int anirec_device_name(char *buf_host, size_t buf_len);

/* ------------------------------------------------------------------------- *
 *  TRAINING  — replaces model.fit's train step, neural_network/neural_network.py:210-217
 *  (graph built at :66-106: Embedding x2 -> Dot(normalize) -> Dense(1) -> BatchNorm
 *   -> sigmoid; binary_crossentropy + whole-table L2; Keras-2.12 Adam, dense update).
 * ------------------------------------------------------------------------- */

/* One optimiser step of the schedule.  `alpha` is the bias-corrected Adam step size
 * lr*sqrt(1-b2^t)/(1-b1^t) for this step, computed on the host from lrfn(epoch)
 * (neural_network.py:109-125) so host and oracle agree bit-for-bit. */
typedef struct anirec_step {
  int32_t start; /* first rating of this rank's part of the batch in the epoch arrays */
  int32_t count; /* ratings of this rank in the batch (<= max_batch) */
  float alpha;
  int32_t global_count; /* ratings of the batch over all ranks (== count on one GPU) */
} anirec_step;

/* Device-resident scalar state; one instance per model.  Layout is ABI. */
typedef struct anirec_state {
  /* trainable scalars: Dense(1) kernel/bias (neural_network.py:97), BN gamma/beta (:99) */
  float w, b, gamma, beta;
  float adam_m[4], adam_v[4]; /* Adam slots of the four scalars, same order */
  float mov_mean, mov_var;    /* BatchNormalization moving statistics */
  float reg_sumsq;            /* sum(U_local^2)+sum(A^2) of the tables this step read (L2 term / lambda) */
  float bn_mu, bn_var;        /* batch statistics of the last training step */
  float last_loss, last_mse;  /* total loss (incl. L2) and mse of the last training step */
  int32_t step_fwd;           /* schedule cursor of the next fwd+head */
  int32_t step_bwd;           /* schedule cursor of the next bwd+adam */
  int32_t pad0;
  /* epoch accumulators (Keras History semantics: sample-weighted means) */
  double loss_wsum;   /* sum over steps of count*total_loss */
  double se_sum;      /* sum of squared errors (mse numerator) */
  double n_seen;      /* ratings seen */
  /* validation accumulators, BN inference mode (neural_network.py:216) */
  double val_bce_sum; /* sum of per-row BCE */
  double val_se_sum;
  double val_n;
  /* the same epoch loss split for user-partitioned multi-GPU runs: the caller adds the
   * other ranks' user-table terms.  sums over steps of count * {bce, sum(U_local^2), sum(A^2)} */
  double bce_wsum, reg_user_wsum, reg_anime_wsum;
  float reg_user_sumsq, reg_anime_sumsq; /* split of reg_sumsq */
} anirec_state;

typedef struct anirec_train_desc {
  int32_t n_user_rows;  /* user rows held by this rank */
  int32_t n_anime_rows; /* anime rows (replicated) */
  int32_t max_batch;    /* capacity of per-step buffers, 1..ANIREC_MAX_BATCH */
  int32_t arena_steps;  /* number of steps the prep arena holds */
  int32_t dense_mode;   /* 0: one GPU, gradients stay chunked.
                           1: user-sharded data parallelism — bwd also writes the ANIME rows' gradient into
                              `dense_grad`, the caller all-reduces it (RCCL), adam reads it;
                           2: replicated tables (the reference's TPUStrategy shape, neural_network.py:173-178)
                              — EVERY row's gradient goes through `dense_grad` (all-reduce, or reduce-scatter
                              with adam_row_lo/hi = this rank's row shard) */
  int32_t n_seg;        /* head packets to read: 1 on one GPU, world size when all-gathered */
  int32_t my_seg;       /* this rank's packet */
  int32_t dense_rows;   /* rows of dense_grad: >= the rows it carries (n_anime_rows in mode 1, all rows in
                           mode 2); extra rows are zero padding up to a multiple of the world size */
  float l2;             /* lambda of embeddings_regularizer L2 (neural_network.py:73) */
  int32_t adam_row_lo;  /* mode 2: adam updates table rows [adam_row_lo, adam_row_hi) only (the caller */
  int32_t adam_row_hi;  /* all-gathers W afterwards); 0,0 = every row */
  int32_t lazy;         /* != 0 (`lazy_state` set): the dense update of the rows a batch does not touch is deferred —
                           dense_mode 0: both tables, inside anirec_trainer_run; dense_mode 1: the rank's user rows,
                           inside anirec_dist_run / the stepper calls (LAZY USER ROWS); ignored in dense_mode 2 */
  /* tables: rows [0,n_user_rows) users, then n_anime_rows anime; [rows][128] fp32.
   * W = embeddings, M/V = Adam first/second moments. */
  float *W, *M, *V;
  int32_t *rowmap;      /* [2][n_user_rows+n_anime_rows] (one map per step parity); zero before first use, left zero */
  anirec_state *state;
  /* this rank's ratings in epoch (shuffled) order; user_idx are LOCAL user rows */
  const int32_t *user_idx;
  const int32_t *anime_idx;
  const float *rating;
  const anirec_step *sched; /* [n_steps] */
  int32_t n_steps;
  int32_t pad2;
  /* head packets: n_seg packets of anirec_packet_floats(max_batch) floats each; packet
   * my_seg is written by the fwd kernel, the others by the caller's all-gather. */
  float *packets;
  float *dense_grad;    /* [dense_rows*128] gradients then [dense_rows] self-coefficient sums, or NULL */
  void *workspace;      /* >= anirec_train_workspace_bytes(max_batch, arena_steps); zero before first use */
  size_t workspace_bytes;
  void *lazy_state;     /* lazy != 0: anirec_train_lazy_bytes(rows) bytes, zero before first use; else NULL */
} anirec_train_desc;

/* floats in one head packet: c[pcap], t[pcap], 4 ints {count,0,0,0}; pcap = max_batch rounded
 * up to a multiple of 4 */
size_t anirec_packet_floats(int32_t max_batch);
size_t anirec_train_workspace_bytes(int32_t max_batch, int32_t arena_steps);
/* per-row state of the lazy dense Adam: the step each row has been updated to + its sum(W^2) of the window's steps */
size_t anirec_train_lazy_bytes(int32_t table_rows);

/* state.reg_sumsq <- sum(W^2) (both tables).  Call once after (re)loading weights. */
int anirec_train_init_reg(const anirec_train_desc *d, void *stream);

/* Sort each batch of steps [first_step, first_step+n_steps) by table row and cut the
 * per-row runs into chunks (<= ANIREC_CHUNK ratings) for the backward pass.  Results
 * land in arena slot (step % arena_steps).  Replaces TF's IndexedSlices ->
 * unsorted_segment_sum densification inside model.fit. */
int anirec_train_prep(const anirec_train_desc *d, int32_t first_step, int32_t n_steps, void *stream);

/* The four stages of one step.  They read the step index from device memory (state->step_fwd /
 * state->step_bwd / a workspace word head publishes) so that a captured graph can be replayed for every step;
 * the per-step scratch is double-buffered by step parity:
 *   fwd  : gather U[ui], A[ai]; c = <l2n(u), l2n(a)>      -> packet, su, sa
 *   head : Dense(1) + BatchNorm(batch stats over ALL packets) + sigmoid + BCE,
 *          d loss / d y per rating and the batch partial sums
 *   bwd  : closed-form backward to d loss / d c, per-chunk weighted row sums of the OTHER
 *          table -> chunk partials (+rowmap)
 *   adam : dense fused Adam over every row of both tables, g = sparse + 2*l2*W, emits
 *          sum(W_new^2) partials; then Adam on the 4 scalars, moving stats, epoch
 *          metrics, step_fwd++                                                           */
int anirec_train_fwd(const anirec_train_desc *d, void *stream);
int anirec_train_head(const anirec_train_desc *d, void *stream);
int anirec_train_bwd(const anirec_train_desc *d, void *stream);
int anirec_train_adam(const anirec_train_desc *d, void *stream);
/* Measurement hook (bench.py): while armed, the training kernels stamp each workgroup's first / last instruction
 * with the 100 MHz constant clock and every launch is followed by a synchronisation that turns the stamps into one
 * duration.  Returns per kernel — 0 fwd, 1 head, 2 bwd, 3 adam, 4 lazy catch-up, 5 lazy adam, 6 lazy flush, 7 lazy
 * reduce — the mean duration [us] of the launches since the last call (-1 = none) and their count (either pointer
 * may be NULL), then arms (enable != 0) or disarms.  Armed steps never replay the captured graph. */
int anirec_train_stage_ticks(const anirec_train_desc *d, int32_t enable, float *us8_host, int32_t *launches8_host,
                             void *stream);

/* adam as two launches (dense_mode 1): which == 1 updates the user rows (may run while the anime gradient is
 * still in the all-reduce), which == 2 the anime rows and finishes the step. */
int anirec_train_adam_part(const anirec_train_desc *d, int32_t which, void *stream);

/* Multi-GPU step as three C calls and two collectives (dense_mode 1 or 2; the stepper owns a side stream):
 *     anirec_train_fwd                      -> all-gather of the head packets (BatchNorm sees the global batch)
 *     anirec_dist_step_mid  (head, bwd, densify; mode 1 forks the user-row adam onto the side stream)
 *                                           -> all-reduce / reduce-scatter of dense_grad
 *     anirec_dist_step_back (joins the side stream; adam of the rows that needed the collective; step finish)
 *   [mode 2 with a row shard: all-gather of W] */
typedef struct anirec_dist_stepper anirec_dist_stepper;
int anirec_dist_stepper_create(const anirec_train_desc *d, anirec_dist_stepper **out_host);
int anirec_dist_stepper_destroy(anirec_dist_stepper *h);
/* anirec_dist_step_front = anirec_train_fwd, preceded — lazy user rows, below — by the catch-up of the batch's rows
 * when the step is the first of its prepared block. */
int anirec_dist_step_front(anirec_dist_stepper *h, void *stream);
int anirec_dist_step_mid(anirec_dist_stepper *h, void *stream);
int anirec_dist_step_back(anirec_dist_stepper *h, void *stream);
/* LAZY USER ROWS (desc->lazy != 0 with dense_mode 1; see LAZY DENSE ADAM below).  In the user-sharded step the dense
 * Adam stream over the rank's user rows becomes: sparse step of the rows the batch touched (forked beside densify +
 * all-reduce; the catch-up of the next batch's rows rides partly in the head launch, partly in that forked launch), a
 * flush of the user rows every ANIREC_LAZY_WINDOW steps and at the end of a run.  The replicated anime rows keep their dense update behind the all-reduce.  Tables, Adam moments
 * and scalar state stay bit-identical to the dense step; reg_user_wsum / loss_wsum receive the user rows' L2 term at
 * the flush.  A caller that drives the steps itself must tell the stepper where it is: _begin(first_step, n_steps)
 * once per run (the tables are current there; stream-ordered) and _block(n) after every anirec_train_prep of n steps;
 * without them the step calls of a lazy descriptor return ANIREC_EINVAL.  anirec_dist_run does both itself. */
int anirec_dist_stepper_begin(anirec_dist_stepper *h, int32_t first_step, int32_t n_steps, void *stream);
int anirec_dist_stepper_block(anirec_dist_stepper *h, int32_t n_steps);

/* The same loop inside the library, the collectives issued to RCCL from C on the engine's stream (one call per
 * block of steps instead of three C calls and two torch.distributed calls per step).  Replaces the reference's only
 * data-parallel construct, neural_network/neural_network.py:142-147,173-178 (replicated variables under a strategy
 * scope: a dense gradient all-reduce per step).  RCCL is bound at run time: anirec_rccl_load(NULL) takes the copy the
 * process has mapped already (torch's) or loads librccl.so.1; ANIREC_ENODEVICE if there is none.
 *   rank 0: anirec_rccl_unique_id(id) -> the caller broadcasts the ANIREC_RCCL_ID_BYTES bytes -> every rank:
 *   anirec_dist_comm_create(id, rank, world) (collective; the current HIP device) -> anirec_dist_run(...) per epoch. */
#define ANIREC_RCCL_ID_BYTES 128
int anirec_rccl_load(const char *path_host);
int anirec_rccl_unique_id(char *id_host);
typedef struct anirec_dist_comm anirec_dist_comm;
int anirec_dist_comm_create(const char *id_host, int32_t rank, int32_t world, anirec_dist_comm **out_host);
int anirec_dist_comm_destroy(anirec_dist_comm *c);
/* steps [first_step, first_step + n_steps) of the stepper's descriptor (n_seg = world, my_seg = rank); the packet
 * all-gather, the gradient all-reduce / reduce-scatter and the W all-gather run in place on the descriptor's buffers.
 * first_step must equal the device cursor.  use_graph != 0: blocks of min(32, arena_steps/2) steps, collectives
 * included, are captured once and replayed (every rank must pass the same value); a failed capture falls back to
 * eager launches. */
int anirec_dist_run(anirec_dist_stepper *h, anirec_dist_comm *c, int32_t first_step, int32_t n_steps, int32_t use_graph,
                    void *stream);

/* LAZY DENSE ADAM (desc->lazy).  Keras' Adam updates every row of both tables every step, because the L2 regulariser
 * gives every row a gradient (2 lambda W) — 28 B/element/step of HBM traffic for rows the batch never touched.  Each
 * element's update sequence is independent of every other element's, so the rows a batch does not touch can take
 * their pure-L2 steps LATER, several at a time, in registers, with the same fp32 operations in the same order:
 *   catch-up(t)  brings the rows batch t touches up to step t (their pending L2-only steps), before fwd(t) reads them
 *                (extra workgroups of step t-1's head and sparse-adam launches; fwd(t-1) marks which rows batch t-1
 *                itself will bring up to date);
 *   sparse adam(t) applies step t (chunk gradient + 2 lambda W) to those rows only;
 *   every ANIREC_LAZY_WINDOW steps (and at the end of every anirec_trainer_run call) a flush replays the pending
 *   steps of every row — one streaming pass over W, M, V per window instead of one per step — and a reduce kernel
 *   assembles the per-step sum(W^2) of the loss's L2 term from the per-row values the replays recorded.
 * Tables, Adam moments and the scalar state are BIT-IDENTICAL to the dense path; the History loss agrees to fp32
 * rounding of its L2 sum (another summation order).  Outside anirec_trainer_run the tables are always up to date. */

/* Steps [first_step, first_step + n_steps) — prep, fwd, head, bwd, adam — on one GPU; first_step
 * must equal the device cursor state->step_fwd.  use_graph != 0 replays a captured hipGraph of
 * G = min(32, arena_steps/2) steps whose first node prepares the G steps after it, so no host
 * work is needed between replays. */
typedef struct anirec_trainer anirec_trainer; /* host-side handle: descriptor copy + graph cache */
int anirec_trainer_create(const anirec_train_desc *d, anirec_trainer **out_host);
int anirec_trainer_destroy(anirec_trainer *t);
int anirec_trainer_run(anirec_trainer *t, int32_t first_step, int32_t n_steps, int32_t use_graph,
                       void *stream);

/* Validation pass on n rows (BN inference, moving stats): accumulates
 * state->val_* ; val_loss = val_bce_sum/val_n + l2*reg_sumsq  (neural_network.py:216). */
int anirec_eval(const anirec_train_desc *d, const int32_t *user_idx, const int32_t *anime_idx,
                const float *rating, int32_t n, void *stream);

/* Standalone fused Adam on a flat fp32 array with an explicit dense gradient
 * (Keras-2.12 Adam dense branch; bit-exact to the oracle given the same g). */
int anirec_adam_flat(float *w, float *m, float *v, const float *g, size_t n, float alpha,
                     void *stream);

/* Self-test of the lazy update's short arithmetic sequences (tests only; no reference call site — it guards the claim
 * that the lazy dense Adam performs the dense kernel's fp32 operations): the correctly rounded square root the replay
 * uses is compared with sqrtf on EVERY float of [2^-96, 2^96], its divide with IEEE `/` on n_div pseudo-random operand
 * pairs of the admitted ranges.  counts2[0] / counts2[1] (device, uint64) receive the numbers of mismatches. */
int anirec_selftest_lazy_math(uint64_t n_div, uint64_t *counts2, void *stream);

/* Epoch shuffle: out[i] = in[perm[i]] for the three rating columns (model.fit shuffle=True). */
int anirec_gather_ratings(const int32_t *user_in, const int32_t *anime_in, const float *rating_in,
                          const int64_t *perm, size_t n, int32_t *user_out, int32_t *anime_out,
                          float *rating_out, void *stream);

/* ------------------------------------------------------------------------- *
 *  SIMILARITY — replaces get_weights + np.dot + np.argsort,
 *  similar_anime/similar_anime.py:136-171,404-408 ; similar_users/similar_users.py:75-101,293-296
 * ------------------------------------------------------------------------- */

/* What = W / ||W||_2 row-wise, no epsilon (zero row -> NaN like NumPy). */
int anirec_rownorm(const float *W, int32_t n, float *What, void *stream);

/* scores[j] = <What[j], What[q]> for every row j (k-ordered fp32 fma chain). */
int anirec_cosine_scores(const float *What, int32_t n, int32_t q, float *scores, void *stream);

/* Top-k rows by descending score for a batch of query rows of the same table.
 *   queries[nq]   : query row indices
 *   keep[n]       : optional byte mask (1 = candidate allowed), NULL = all
 *   exclude_self  : drop the query row itself (similar_users.py:303, similar_anime.py:459)
 *   out_idx[nq*k] : row indices (-1 padded), out_score[nq*k] : fp32 scores (NaN padded)
 * Ties: ascending row index.  NaN scores rank last.  k <= ANIREC_MAX_TOPK.
 * workspace: anirec_topk_workspace_bytes(n, nq) bytes. */
size_t anirec_topk_workspace_bytes(int32_t n, int32_t nq);
int anirec_cosine_topk(const float *What, int32_t n, const int32_t *queries, int32_t nq,
                       const uint8_t *keep, int32_t exclude_self, int32_t k, int32_t *out_idx,
                       float *out_score, void *workspace, size_t workspace_bytes, void *stream);

/* Same result as anirec_cosine_topk on the matrix cores: fp16 MFMA candidate scores for all
 * keys with a rigorous error window, exact fp32 fma-chain re-rank of the survivors.
 * The error window is proven for UNIT-NORM rows (the output of anirec_rownorm, as every reference call
 * site passes: similar_users.py:293 takes get_weights() output).  The conversion pass checks it: if any
 * finite key or query row has | ||row||^2 - 1 | > 1e-3 every query is flagged (flags bit 2) and falls to the
 * caller's exact re-run, so un-normalised input is slow, never silently incomplete.
 * flags[nq] (device) is non-zero for the rare query whose window could not be proven
 * complete (dense ties / more than 256 survivors); its output row is -1/NaN and the caller
 * re-runs it through anirec_cosine_topk.  k <= ANIREC_MAX_TOPK - 1. */
size_t anirec_topk_mfma_workspace_bytes(int32_t n, int32_t nq);
int anirec_cosine_topk_mfma(const float *What, int32_t n, const int32_t *queries, int32_t nq,
                            const uint8_t *keep, int32_t exclude_self, int32_t k, int32_t *out_idx,
                            float *out_score, int32_t *flags, void *workspace,
                            size_t workspace_bytes, void *stream);
/* The same with a PRIOR theta0 for every row's threshold (anirec_cosine_topk_mfma uses -4, below every cosine): a
 * candidate must score >= max(theta0, what the keys seen so far prove).  A good guess of the rows' final k-th best
 * score (minus a margin) spares most of the ~k ln(n) early appends per row.  Results stay exact: a row whose true
 * threshold lies below theta0 ends with too few candidates, is flagged like any other unproven row, and the caller
 * re-runs it (without a prior, or through anirec_cosine_topk).  -4 <= theta0 <= 1. */
int anirec_cosine_topk_mfma_prior(const float *What, int32_t n, const int32_t *queries, int32_t nq,
                                  const uint8_t *keep, int32_t exclude_self, int32_t k, float theta0,
                                  int32_t *out_idx, float *out_score, int32_t *flags, void *workspace,
                                  size_t workspace_bytes, void *stream);
/* The whole similar_anime / similar_users job (every row a query: similar_users.py:290-312 at BASELINE configs[3]
 * scale) as ONE call: the keys are converted once, the queries run in batches [starts[b], starts[b+1]), and the
 * batches are dealt to `lanes` stream-ordered chains (the caller's stream + side streams the library keeps, forked and
 * joined by events inside this call; 1 <= lanes <= 4) so that one batch's per-row refresh / re-rank waves run beside
 * another batch's MFMA kernel.  prior_mode 0: no prior; 1: the first `learn_batches` (0 or 1) batches run alone and
 * the k-th best scores of their rows give the others a prior, computed on the device (no host round trip); 2: theta0.
 * flags[nq] as for anirec_cosine_topk_mfma: the caller re-runs flagged rows (without a prior, then through
 * anirec_cosine_topk).  Results are identical to anirec_cosine_topk whatever the plan.
 * prior_mode 3 = mode 1 for the ALL-PAIRS job (queries[i] == i for all i < nq == n, keep == NULL; checked on the
 * device: any other query list flags every row): cosine(i, j) == cosine(j, i), so a batch computes the dot products of
 * its rows with the rows of LATER batches once, for both — a score that reaches the learnt prior is also dropped into
 * the later row's inbox (through per-wave logs a side kernel deals out) — and skips the key tiles of EARLIER batches
 * (their pairs are in its inboxes already): ~0.65 of the MFMA work at 350 k rows.  An inbox that overflows flags its row
 * like any unproven row.  Needs the learning batch, whole 128-row key tiles per batch (the default plan has them) and lanes <= 2;
 * otherwise the call runs as mode 1.  Results are identical either way.  Workspace:
 * anirec_cosine_topk_allpairs_workspace_bytes (the job's + two inboxes of n x 256 entries + the chains' logs).
 * anirec_cosine_topk_job_plan fills the library's default plan: starts_host[ANIREC_TOPK_MAX_BATCHES + 1];
 * max_batch <= 0 and lanes <= 0 select the defaults (131072 rows; env ANIREC_TOPK_LANES or 2).
 * workspace: anirec_cosine_topk_job_workspace_bytes(n, rows of the largest batch, lanes). */
int anirec_cosine_topk_job_plan(int32_t nq, int32_t k, int32_t prior_auto, int32_t max_batch, int32_t lanes,
                                int32_t *starts_host, int32_t *n_batches_host, int32_t *learn_batches_host);
size_t anirec_cosine_topk_job_workspace_bytes(int32_t n, int32_t max_batch_rows, int32_t lanes);
size_t anirec_cosine_topk_allpairs_workspace_bytes(int32_t n, int32_t max_batch_rows, int32_t lanes);
/* the all-pairs plan: the learning batch + `main_batches` (<= 0: env ANIREC_TOPK_SYM_BATCHES or 4) batches of equal
 * WORK (a later batch streams fewer keys and takes more rows); the default plan when the job is too small to learn */
int anirec_cosine_topk_allpairs_plan(int32_t n, int32_t k, int32_t lanes, int32_t main_batches, int32_t *starts_host,
                                     int32_t *n_batches_host, int32_t *learn_batches_host);
int anirec_cosine_topk_job(const float *What, int32_t n, const int32_t *queries, int32_t nq, const uint8_t *keep,
                           int32_t exclude_self, int32_t k, int32_t prior_mode, float theta0,
                           const int32_t *starts_host, int32_t n_batches, int32_t learn_batches, int32_t lanes,
                           int32_t *out_idx, float *out_score, int32_t *flags, void *workspace,
                           size_t workspace_bytes, void *stream);

/* Measurement hook (bench.py's roofline leg): returns the summed HIP-event duration [ms] and the number of the
 * MFMA candidate-kernel launches of the calls made since it was last armed, then arms (enable != 0) or disarms
 * the timing.  Armed calls block until the stream drains, and an armed job runs its batches on ONE chain so that
 * every timed launch runs alone. */
int anirec_topk_mfma_timing(int32_t enable, float *cand_ms, int32_t *launches);

/* ------------------------------------------------------------------------- *
 *  PREDICTION — replaces model.predict([user_arr, anime_arr]), model_recs/model_recs.py:394
 * ------------------------------------------------------------------------- */

/* Inference head: sigmoid(gamma*(w*c+b-mov_mean)/sqrt(mov_var+1e-3)+beta). */
typedef struct anirec_head {
  float w, b, gamma, beta, mov_mean, mov_var;
} anirec_head;

/* p[i] = model(user_idx[i], anime_idx[i]) for n explicit pairs. */
int anirec_predict_pairs(const float *U, const float *A, const int32_t *user_idx,
                         const int32_t *anime_idx, int32_t n, const anirec_head *head_host,
                         float *p, void *stream);

/* Workspace of predict_grid / predict_topk: l2-normalised copies of A and of the query
 * users (+ a batch of rating rows when topk != 0). */
size_t anirec_predict_workspace_bytes(int32_t n_anime, int32_t n_users, int32_t topk);

/* out[j*n_anime + a] = model(users[j], a) for every anime a (the full rating grid). */
int anirec_predict_grid(const float *U, const float *A, int32_t n_anime, const int32_t *users,
                        int32_t n_users, const anirec_head *head_host, float *out,
                        void *workspace, size_t workspace_bytes, void *stream);

/* The same grid on the matrix cores: rows are split x = hi + lo in fp16 and accumulated as
 * hi*hi + hi*lo + lo*hi by v_mfma_f32_32x32x16_f16 (ratings within 1e-5 of the fp32 path). */
size_t anirec_predict_mfma_workspace_bytes(int32_t n_anime, int32_t n_users);
int anirec_predict_grid_mfma(const float *U, const float *A, int32_t n_anime, const int32_t *users,
                             int32_t n_users, const anirec_head *head_host, float *out,
                             void *workspace, size_t workspace_bytes, void *stream);

/* Per query user: top-k anime by descending predicted rating among anime whose
 * watched bit is clear.  watched: optional [n_users][ceil(n_anime/32)] bitmask words.
 * (model_recs.py:144-155 candidate set, :396 ranking, :451-454 cut). */
int anirec_predict_topk(const float *U, const float *A, int32_t n_anime, const int32_t *users,
                        int32_t n_users, const anirec_head *head_host, const uint32_t *watched,
                        int32_t k, int32_t *out_idx, float *out_p, void *workspace,
                        size_t workspace_bytes, void *stream);

/* The same top-k on the matrix cores (the batched model_recs path: 100 k users x 18 k anime):
 * fp16 MFMA cosine candidates with a rigorous error window, the watched mask applied when a
 * candidate is appended, exact fp32 re-rank through the head.  Same results as
 * anirec_predict_topk; flags[n_users] (device) is non-zero for the rare user whose window could
 * not be proven complete (saturated head, > 256 survivors, fewer than k unwatched anime): its row
 * is -1/NaN and the caller re-runs it through anirec_predict_topk.  k <= ANIREC_MAX_TOPK - 1. */
size_t anirec_predict_topk_mfma_workspace_bytes(int32_t n_anime, int32_t n_users);
int anirec_predict_topk_mfma(const float *U, const float *A, int32_t n_anime, const int32_t *users,
                             int32_t n_users, const anirec_head *head_host, const uint32_t *watched,
                             int32_t k, int32_t *out_idx, float *out_p, int32_t *flags, void *workspace,
                             size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 *  INGEST — the step before the hot path, columns resident in HBM.
 *  Replaces preprocess/preprocess.py:13-40 (drop_useless: drop_duplicates keep-first, dropna,
 *  watched/plan filters, users with < num_reviews ratings), :52-105 (drop_half_watched),
 *  :108-117 (scale_ratings, float64) and the id -> Series.unique() position encoding of
 *  neural_network/neural_network.py:41-60.  Surviving rows keep their order; results are
 *  bit-identical to pandas.
 * ------------------------------------------------------------------------- */
#define ANIREC_NULL_I32 INT32_MIN /* missing value of an integer column (pandas NaN) */

typedef struct anirec_ingest_opts {
  int32_t num_reviews;       /* keep users with at least this many surviving ratings */
  int32_t drop_unwatched;    /* drop rows with watched_episodes == 0 */
  int32_t drop_plan;         /* drop rows with watching_status == 6 */
  int32_t drop_half_watched; /* drop rows with watched < half of the anime's max watched */
  int32_t user_id_bound;     /* ids must lie in [0, bound): sizes of the direct-index tables */
  int32_t anime_id_bound;
} anirec_ingest_opts;

/* rating: float64, NaN = missing.  Outputs hold up to n rows; *n_out (device) receives the row
 * count; *err_flag (device) becomes 1 if a non-missing id is outside its bound (that row is
 * dropped).  1 <= n < 2^30.  The five input columns and the workspace must be 16-byte aligned
 * (ANIREC_EINVAL otherwise: the kernels read four rows per lane).  Any row order is accepted; a table
 * grouped by user (the raw animelist) finds its duplicate rows in LDS, chunk by chunk. */
/* Largest user_id and anime_id of the two columns (ANIREC_NULL_I32 if a column holds nothing else) in one pass:
 * out_max2[0] + 1 / out_max2[1] + 1 are the id bounds anirec_ingest_opts asks for.  Device pointers, 16-byte aligned
 * columns, stream-ordered. */
int anirec_ingest_id_max(const int32_t *user_id, const int32_t *anime_id, int64_t n, int32_t *out_max2, void *stream);
size_t anirec_ingest_workspace_bytes(int64_t n, int32_t user_id_bound, int32_t anime_id_bound);
int anirec_ingest_preprocess(const int32_t *user_id, const int32_t *anime_id, const double *rating,
                             const int32_t *watching_status, const int32_t *watched_episodes, int64_t n,
                             const anirec_ingest_opts *opts, int32_t *out_user_id, int32_t *out_anime_id,
                             double *out_rating, int32_t *out_status, int32_t *out_episodes, int64_t *n_out,
                             int32_t *err_flag, void *workspace, size_t workspace_bytes, void *stream);

/* With opts->drop_half_watched the reference's frame keeps two extra columns (preprocess.py:99-100,104):
 * max_eps = the anime's largest watched_episodes over the rows that survived drop_useless, and
 * half_eps = max_eps == 1 ? 1 : max_eps * .5.  Call right after anirec_ingest_preprocess with the SAME
 * n, opts and workspace (the per-anime maxima are still in it); rows [0, *n_out) are written. */
int anirec_ingest_half_columns(const int32_t *out_anime_id, const int64_t *n_out, int64_t n,
                               const anirec_ingest_opts *opts, int32_t *out_max_eps, double *out_half_eps,
                               const void *workspace, size_t workspace_bytes, void *stream);

/* out_index[i] = position of id[i] in the order of first appearance (pandas Series.unique());
 * out_uniques[j] = the j-th distinct id; *n_unique (device) = number of distinct ids.
 * id, out_index and the workspace must be 16-byte aligned. */
size_t anirec_ingest_encode_workspace_bytes(int64_t n, int32_t id_bound);
int anirec_ingest_encode(const int32_t *id, int64_t n, int32_t id_bound, int32_t *out_index,
                         int32_t *out_uniques, int64_t *n_unique, int32_t *err_flag, void *workspace,
                         size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 *  FAVOURITES + USER-BASED RECS — the consumer of the similar-users top-k.
 *  user_recs/user_recs.py:377-404 (fave_genres: a user's favourites are the anime rated at or above
 *  the 80th percentile of their own ratings, np.percentile 'linear', float64; also
 *  similar_users.py:203-256 get_fave_anime) and :708-760 (similar_user_recs: per query user, count how
 *  many of its similar users hold each anime as a favourite, drop the query's own favourites, rank).
 * ------------------------------------------------------------------------- */

/* fav_bits[n_users][ceil(n_anime/32)]: bit a of row u set iff rating(u, a) >= threshold[u];
 * threshold[u] = np.percentile(ratings of u, percentile) bit for bit (NaN for users without ratings).
 * Ratings must not contain NaN.  *err_flag (device) becomes 1 on an out-of-range index.
 * The three columns and the workspace must be 16-byte aligned (ANIREC_EINVAL otherwise).  A table grouped by user
 * (non-decreasing user_idx: the raw / preprocessed order) is detected on the device and takes the fast path: no CSR
 * copy, the bit rows built per user and written once. */
size_t anirec_fav_workspace_bytes(int64_t n_ratings, int32_t n_users);
int anirec_user_favourites(const int32_t *user_idx, const int32_t *anime_idx, const double *rating, int64_t n,
                           int32_t n_users, int32_t n_anime, double percentile, uint32_t *fav_bits,
                           double *threshold, int32_t *err_flag, void *workspace, size_t workspace_bytes,
                           void *stream);

/* sim_users[nq][k_sim]: similar users of query_users[q], best first, -1 = empty (k_sim <= 63, n_recs <= 256).
 * out_anime/out_count[nq][n_recs]: anime by (count desc, best similar-user rank asc, anime index asc),
 * -1 / 0 padded; anime that are favourites of the query user are skipped.  n_anime < 131072. */
int anirec_user_recs(const uint32_t *fav_bits, int32_t n_users, int32_t n_anime, const int32_t *query_users,
                     const int32_t *sim_users, int32_t nq, int32_t k_sim, int32_t n_recs, int32_t *out_anime,
                     int32_t *out_count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ANIREC_H */
