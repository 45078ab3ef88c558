/*
 * anirec_oracle.c — plain-C restatement of the reference's CPU path for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY (checker + bench.py's cpu_baseline "port" leg).  The product
 * never links or loads this file.  PARITY UNPINNED except lrfn (see anirec_oracle.py).
 *
 * It follows the same reference lines as oracle/anirec_oracle.py:
 *   train step   neural_network/neural_network.py:66-106 (graph), :210-217 (model.fit) —
 *                what TensorFlow's CPU runtime does per batch: gather, normalised dot,
 *                Dense(1), BatchNorm(batch stats), sigmoid, BCE(+whole-table L2), the
 *                gather gradients densified (zeros + scatter-add + 2*l2*W) and the
 *                Keras-2.12 Adam dense update over BOTH full tables.
 *   cosine       similar_anime/similar_anime.py:404, similar_users/similar_users.py:293
 *                (np.dot(W, W[q])) with the build's defined order: k-ordered fmaf chain.
 *
 * Build: gcc -O3 -march=native -ffp-contract=off -fopenmp -shared -fPIC (oracle/build_oracle.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define D 128
#define L2N_EPS 1e-12f
#define BN_EPS 1e-3f
#define ONE_B1 0.1f
#define ONE_B2 0.001f
#define ADAM_EPS 1e-7f

typedef struct {
  float w, b, gamma, beta;
  float m[4], v[4];
  float mov_mean, mov_var;
} orc_head;

typedef struct {
  double loss, bce, reg, mse, mu, var;
} orc_metrics;

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static inline void adam1(float *w, float *m, float *v, float g, float alpha) {
  *m = *m + (g - *m) * ONE_B1;
  *v = *v + (g * g - *v) * ONE_B2;
  *w = *w - (*m * alpha) / (sqrtf(*v) + ADAM_EPS);
}

static inline float sigmoidf_(float y) {
  float e = expf(-fabsf(y));
  return y >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

/* scratch owned by the caller: c,su,sa,z,coef,selfu,selfa [B]; gU [n_u*D], gA [n_a*D] (zeroed) */
typedef struct {
  float *c, *su, *sa, *coef, *selfu, *selfa, *dz;
  float *gU, *gA;
} orc_scratch;

/* One model.fit step on one batch.  Dense gradient buffers gU/gA must be zero on entry and
 * are zero again on return. */
void orc_train_step(float *U, float *A, float *mU, float *vU, float *mA, float *vA, int n_u, int n_a,
                    orc_head *h, const int32_t *ui, const int32_t *ai, const float *t, int B,
                    float alpha, float l2, orc_scratch *s, orc_metrics *out) {
  const float w = h->w, b = h->b, gamma = h->gamma, beta = h->beta;
  double reg = 0.0;
  /* forward: gather + normalised dot */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < B; ++i) {
    const float *u = U + (size_t)ui[i] * D, *a = A + (size_t)ai[i] * D;
    float su = 0.f, sa = 0.f;
    for (int k = 0; k < D; ++k) {
      su += u[k] * u[k];
      sa += a[k] * a[k];
    }
    const float ru = 1.0f / sqrtf(fmaxf(su, L2N_EPS)), ra = 1.0f / sqrtf(fmaxf(sa, L2N_EPS));
    float c = 0.f;
    for (int k = 0; k < D; ++k) c += (u[k] * ru) * (a[k] * ra);
    s->c[i] = c;
    s->su[i] = su;
    s->sa[i] = sa;
  }
  /* whole-table L2 term of the loss */
#pragma omp parallel for reduction(+ : reg) schedule(static)
  for (size_t e = 0; e < (size_t)n_u * D; ++e) reg += (double)(U[e] * U[e]);
#pragma omp parallel for reduction(+ : reg) schedule(static)
  for (size_t e = 0; e < (size_t)n_a * D; ++e) reg += (double)(A[e] * A[e]);
  /* head: Dense(1) -> BatchNorm(batch stats) -> sigmoid -> BCE, closed-form backward */
  const float Bf = (float)B;
  double zs = 0.0;
  for (int i = 0; i < B; ++i) zs += (double)(s->c[i] * w + b);
  const float mu = (float)(zs / B);
  double vs = 0.0;
  for (int i = 0; i < B; ++i) {
    const float d = (s->c[i] * w + b) - mu;
    vs += (double)(d * d);
  }
  const float var = (float)(vs / B);
  const float rs = 1.0f / sqrtf(var + BN_EPS);
  const float inv = rs * gamma, shift = beta - mu * inv;
  double S1 = 0., S2 = 0., L = 0., SE = 0.;
  for (int i = 0; i < B; ++i) {
    const float z = s->c[i] * w + b, y = z * inv + shift, p = sigmoidf_(y);
    const float dy = (p - t[i]) / Bf, zh = (z - mu) * rs;
    S1 += dy;
    S2 += (double)(dy * zh);
    L += (double)(fmaxf(y, 0.f) - y * t[i] + log1pf(expf(-fabsf(y))));
    SE += (double)((p - t[i]) * (p - t[i]));
  }
  const float m1 = gamma * (float)S1 / Bf, m2 = gamma * (float)S2 / Bf;
  double dW = 0., dB = 0.;
  for (int i = 0; i < B; ++i) {
    const float c = s->c[i], z = c * w + b, y = z * inv + shift, p = sigmoidf_(y);
    const float dy = (p - t[i]) / Bf, zh = (z - mu) * rs;
    const float dz = (dy * gamma - m1 - zh * m2) * rs;
    dW += (double)(dz * c);
    dB += dz;
    const float dc = dz * w;
    const float ru = 1.0f / sqrtf(fmaxf(s->su[i], L2N_EPS)), ra = 1.0f / sqrtf(fmaxf(s->sa[i], L2N_EPS));
    s->coef[i] = dc * ru * ra;
    s->selfu[i] = s->su[i] >= L2N_EPS ? dc * c * ru * ru : 0.f;
    s->selfa[i] = s->sa[i] >= L2N_EPS ? dc * c * ra * ra : 0.f;
  }
  /* gather gradients, scatter-added in batch order (IndexedSlices -> dense) */
  for (int i = 0; i < B; ++i) {
    const float *u = U + (size_t)ui[i] * D, *a = A + (size_t)ai[i] * D;
    float *gu = s->gU + (size_t)ui[i] * D, *ga = s->gA + (size_t)ai[i] * D;
    const float cf = s->coef[i], fu = s->selfu[i], fa = s->selfa[i];
    for (int k = 0; k < D; ++k) {
      gu[k] += cf * a[k] - fu * u[k];
      ga[k] += cf * u[k] - fa * a[k];
    }
  }
  /* dense Adam over both full tables, g = scatter + 2*l2*W */
  const float two_l2 = 2.0f * l2;
#pragma omp parallel for schedule(static)
  for (size_t e = 0; e < (size_t)n_u * D; ++e) {
    const float g = s->gU[e] + two_l2 * U[e];
    adam1(&U[e], &mU[e], &vU[e], g, alpha);
    s->gU[e] = 0.f;
  }
#pragma omp parallel for schedule(static)
  for (size_t e = 0; e < (size_t)n_a * D; ++e) {
    const float g = s->gA[e] + two_l2 * A[e];
    adam1(&A[e], &mA[e], &vA[e], g, alpha);
    s->gA[e] = 0.f;
  }
  float hp[4] = {w, b, gamma, beta};
  const float hg[4] = {(float)dW, (float)dB, (float)S2, (float)S1};
  for (int k = 0; k < 4; ++k) adam1(&hp[k], &h->m[k], &h->v[k], hg[k], alpha);
  h->w = hp[0];
  h->b = hp[1];
  h->gamma = hp[2];
  h->beta = hp[3];
  h->mov_mean = h->mov_mean - (h->mov_mean - mu) * 0.01f;
  h->mov_var = h->mov_var - (h->mov_var - var) * 0.01f;
  if (out) {
    out->bce = L / B;
    out->reg = reg;
    out->loss = L / B + (double)l2 * reg;
    out->mse = SE / B;
    out->mu = mu;
    out->var = var;
  }
}

/* n_steps consecutive batches of size B from (ui, ai, t); alphas[n_steps].  Returns 0. */
int orc_train_run(float *U, float *A, float *mU, float *vU, float *mA, float *vA, int n_u, int n_a,
                  orc_head *h, const int32_t *ui, const int32_t *ai, const float *t, int n_ratings,
                  int B, const float *alphas, int n_steps, float l2, orc_metrics *last) {
  orc_scratch s;
  s.c = (float *)malloc(sizeof(float) * B * 7);
  s.su = s.c + B;
  s.sa = s.su + B;
  s.coef = s.sa + B;
  s.selfu = s.coef + B;
  s.selfa = s.selfu + B;
  s.dz = s.selfa + B;
  s.gU = (float *)calloc((size_t)n_u * D, sizeof(float));
  s.gA = (float *)calloc((size_t)n_a * D, sizeof(float));
  if (!s.c || !s.gU || !s.gA) return -1;
  for (int k = 0; k < n_steps; ++k) {
    const int st = k * B;
    int nb = n_ratings - st;
    if (nb <= 0) break;
    if (nb > B) nb = B;
    orc_train_step(U, A, mU, vU, mA, vA, n_u, n_a, h, ui + st, ai + st, t + st, nb, alphas[k], l2, &s,
                   last);
  }
  free(s.c);
  free(s.gU);
  free(s.gA);
  return 0;
}

/* scores[j] = sum_k fma(W[j][k], q[k]) in k order (exactly rounded fmaf chain). */
void orc_cosine_scores(const float *Wh, int n, const float *q, float *scores) {
#pragma omp parallel for schedule(static)
  for (int j = 0; j < n; ++j) {
    const float *r = Wh + (size_t)j * D;
    float s = 0.f;
    for (int k = 0; k < D; ++k) s = fmaf(r[k], q[k], s);
    scores[j] = s;
  }
}

/* np.linalg.norm row normalisation, plain left-to-right sum (1-2 ulp from NumPy's pairwise) */
void orc_rownorm(const float *W, int n, float *out) {
#pragma omp parallel for schedule(static)
  for (int j = 0; j < n; ++j) {
    const float *r = W + (size_t)j * D;
    float ss = 0.f;
    for (int k = 0; k < D; ++k) ss += r[k] * r[k];
    const float nrm = sqrtf(ss);
    for (int k = 0; k < D; ++k) out[(size_t)j * D + k] = r[k] / nrm;
  }
}

/* Top-k of one score row: descending score, ties ascending index, NaN last, `exclude`
 * dropped, optional keep mask.  O(n*k) insertion — checker speed is irrelevant. */
int orc_topk(const float *scores, int n, int k, int exclude, const uint8_t *keep, int32_t *idx,
             float *val) {
  int cnt = 0;
  for (int j = 0; j < n; ++j) {
    if (j == exclude || (keep && !keep[j])) continue;
    const float s = scores[j];
    const int s_nan = s != s;
    int pos = cnt;
    while (pos > 0) {
      const float p = val[pos - 1];
      const int p_nan = p != p;
      /* does candidate j rank before entry pos-1 ?  (index order breaks ties; j > idx) */
      int before = (!s_nan && p_nan) || (!s_nan && !p_nan && s > p);
      if (!before) break;
      --pos;
    }
    if (pos >= k) continue;
    const int last = cnt < k ? cnt : k - 1;
    for (int m = last; m > pos; --m) {
      idx[m] = idx[m - 1];
      val[m] = val[m - 1];
    }
    idx[pos] = j;
    val[pos] = s;
    if (cnt < k) ++cnt;
  }
  return cnt;
}

/* all listed queries: scores by fmaf chain then orc_topk; out arrays [nq][k], padded -1/NaN */
void orc_cosine_topk(const float *Wh, int n, const int32_t *queries, int nq, int k, int exclude_self,
                     const uint8_t *keep, int32_t *out_idx, float *out_val) {
#pragma omp parallel
  {
    float *sc = (float *)malloc(sizeof(float) * (size_t)n);
#pragma omp for schedule(dynamic, 4)
    for (int qi = 0; qi < nq; ++qi) {
      const float *q = Wh + (size_t)queries[qi] * D;
      for (int j = 0; j < n; ++j) {
        const float *r = Wh + (size_t)j * D;
        float s = 0.f;
        for (int kk = 0; kk < D; ++kk) s = fmaf(r[kk], q[kk], s);
        sc[j] = s;
      }
      int32_t *oi = out_idx + (size_t)qi * k;
      float *ov = out_val + (size_t)qi * k;
      const int c = orc_topk(sc, n, k, exclude_self ? queries[qi] : -1, keep, oi, ov);
      for (int m = c; m < k; ++m) {
        oi[m] = -1;
        ov[m] = NAN;
      }
    }
    free(sc);
  }
}

/* model.predict on a user x all-anime grid (BN inference): out[j][a] */
void orc_predict_grid(const float *U, const float *A, int n_a, const int32_t *users, int n_q,
                      const orc_head *h, float *out) {
  const float inv = (1.0f / sqrtf(h->mov_var + BN_EPS)) * h->gamma;
  const float shift = h->beta - h->mov_mean * inv;
#pragma omp parallel for schedule(static)
  for (int j = 0; j < n_q; ++j) {
    const float *u = U + (size_t)users[j] * D;
    float su = 0.f;
    for (int k = 0; k < D; ++k) su += u[k] * u[k];
    const float ru = 1.0f / sqrtf(fmaxf(su, L2N_EPS));
    for (int a = 0; a < n_a; ++a) {
      const float *x = A + (size_t)a * D;
      float sa = 0.f;
      for (int k = 0; k < D; ++k) sa += x[k] * x[k];
      const float ra = 1.0f / sqrtf(fmaxf(sa, L2N_EPS));
      float c = 0.f;
      for (int k = 0; k < D; ++k) c += (u[k] * ru) * (x[k] * ra);
      out[(size_t)j * n_a + a] = sigmoidf_((c * h->w + h->b) * inv + shift);
    }
  }
}
