/* C restatement of the LF-MMI arithmetic -- TEST / BASELINE INFRASTRUCTURE ONLY.
 *
 * Same algorithm as oracle/chain_ref.py (Kaldi's
 * DenominatorComputation / NumeratorComputation / ComputeChainObjfAndDeriv, reached by the
 * reference through kaldi.chain.compute_chain_objf_and_deriv, reference ops/ops.py:265), written
 * in C so that bench.py's cpu_baseline leg can time it on the GPU box's host cores (the numpy
 * version is ~100x slower).  PARITY UNPINNED at the Kaldi boundary (see chain_ref.py); this file
 * is pinned against chain_ref.py in tests/test_oracle_chain_c.py.  float32 state like Kaldi's
 * BaseFloat, double for the log-sums.  The product never links or calls this.
 *
 * Build: gcc -O3 -fopenmp -shared -fPIC chain_oracle.c -o _build/libchain_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Denominator forward-backward of ONE sequence.  Arcs: src,dst,pdf,prob.  gamma[T*P] += occupancy
 * * scale.  Returns log p_den; *check receives sum_h alpha'[0,h] beta'[0,h]. */
/* State type of the denominator recursion: float = Kaldi's BaseFloat (the build bench.py's cpu_baseline times);
 * -DORC_REAL=double builds libchain_oracle_f64.so, the same recursion without float32 rounding, which the parity
 * checks at bench size compare the device with (at T ~ 600 frames the float build itself is ~1e-4 away from it). */
#ifndef ORC_REAL
#define ORC_REAL float
#endif
typedef ORC_REAL real;

/* Test hook: beta'[T] = seed / tot instead of 1 / tot (scales the alpha-beta product of the consistency check: the tests
 * of the abandon rule drive it on real inputs through this).  1.0 = Kaldi. */
static double g_beta_seed = 1.0;
void orc_set_beta_seed(double v) { g_beta_seed = v > 0.0 ? v : 1.0; }

double orc_den_fb(int S, int P, int64_t A, const int32_t* src, const int32_t* dst, const int32_t* pdf,
                  const float* prob, const float* pi, const float* logits, int64_t row_stride, int T,
                  float leaky, float scale, float* gamma, int64_t grow_stride, double* check) {
  real* alpha = (real*)malloc(sizeof(real) * (size_t)(T + 1) * S); /* alpha' (with leaky term) */
  real* asum = (real*)malloc(sizeof(real) * (size_t)(T + 1));
  real* x = (real*)malloc(sizeof(real) * (size_t)P);
  real* bcur = (real*)malloc(sizeof(real) * (size_t)S);
  real* bnext = (real*)malloc(sizeof(real) * (size_t)S);
  real* a = (real*)malloc(sizeof(real) * (size_t)S);
  real* grow = (real*)malloc(sizeof(real) * (size_t)P);
  for (int h = 0; h < S; ++h) a[h] = pi[h];
  double logp = 0.0;
  for (int t = 0; t <= T; ++t) {
    double s = 0.0;
    for (int h = 0; h < S; ++h) s += a[h];
    asum[t] = (real)s;
    real* ad = alpha + (size_t)t * S;
    for (int h = 0; h < S; ++h) ad[h] = a[h] + (real)leaky * asum[t] * (real)pi[h];
    if (t == T) break;
    const float* row = logits + (int64_t)t * row_stride;
    for (int p = 0; p < P; ++p) {
      float v = row[p];
      v = v < -30.f ? -30.f : (v > 30.f ? 30.f : v);
      x[p] = sizeof(real) == sizeof(float) ? (real)expf(v) : (real)exp((double)v);
    }
    memset(a, 0, sizeof(real) * S);
    const real inv = (real)1.0 / asum[t];
    for (int64_t k = 0; k < A; ++k) a[dst[k]] += ad[src[k]] * (real)prob[k] * x[pdf[k]] * inv;
    logp += log((double)asum[t]);
  }
  double tot = 0.0;
  for (int h = 0; h < S; ++h) tot += alpha[(size_t)T * S + h];
  logp += log(tot);
  /* beta */
  double pb = 0.0;
  for (int h = 0; h < S; ++h) { bnext[h] = (real)(g_beta_seed / tot); pb += (double)pi[h] * bnext[h]; }
  for (int h = 0; h < S; ++h) bnext[h] += (real)leaky * (real)pb;
  for (int t = T - 1; t >= 0; --t) {
    const float* row = logits + (int64_t)t * row_stride;
    for (int p = 0; p < P; ++p) {
      float v = row[p];
      v = v < -30.f ? -30.f : (v > 30.f ? 30.f : v);
      x[p] = sizeof(real) == sizeof(float) ? (real)expf(v) : (real)exp((double)v);
    }
    const real* ad = alpha + (size_t)t * S;
    float* gout = gamma + (int64_t)t * grow_stride;
    const real inv = (real)1.0 / asum[t];
    memset(bcur, 0, sizeof(real) * S);
    memset(grow, 0, sizeof(real) * P);
    for (int64_t k = 0; k < A; ++k) {
      const real f = (real)prob[k] * x[pdf[k]] * bnext[dst[k]] * inv;
      bcur[src[k]] += f;
      grow[pdf[k]] += ad[src[k]] * f;
    }
    for (int p = 0; p < P; ++p) gout[p] += scale * (float)grow[p];
    pb = 0.0;
    for (int h = 0; h < S; ++h) pb += (double)pi[h] * bcur[h];
    if (t == 0 && check) {
      double c = 0.0;
      for (int h = 0; h < S; ++h) c += (double)alpha[h] * bcur[h];
      *check = c;
    }
    for (int h = 0; h < S; ++h) bnext[h] = bcur[h] + (real)leaky * (real)pb;
  }
  free(alpha); free(asum); free(x); free(bcur); free(bnext); free(a); free(grow);
  return logp;
}

static inline double log_add(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  return a > b ? a + log1p(exp(b - a)) : b + log1p(exp(a - b));
}

/* Numerator forward-backward (log domain) of ONE sequence; arcs sorted by source frame with
 * frame_off[T+1]; post[T*P] += scale * posterior.  Returns log p_num. */
double orc_num_fb(int NS, const int32_t* src, const int32_t* dst, const int32_t* pdf, const float* w,
                  const int32_t* frame_off, int T, const int32_t* finals, const float* final_w, int nfinal,
                  const float* logits, int64_t row_stride, float scale, float* post, int64_t prow_stride) {
  double* la = (double*)malloc(sizeof(double) * NS);
  double* lb = (double*)malloc(sizeof(double) * NS);
  for (int s = 0; s < NS; ++s) { la[s] = -INFINITY; lb[s] = -INFINITY; }
  la[0] = 0.0;
  for (int t = 0; t < T; ++t) {
    const float* row = logits + (int64_t)t * row_stride;
    for (int a = frame_off[t]; a < frame_off[t + 1]; ++a)
      la[dst[a]] = log_add(la[dst[a]], la[src[a]] - (double)w[a] + (double)row[pdf[a]]);
  }
  double lp = -INFINITY;
  for (int k = 0; k < nfinal; ++k) {
    lp = log_add(lp, la[finals[k]] - (double)final_w[k]);
    lb[finals[k]] = -(double)final_w[k];
  }
  for (int t = T - 1; t >= 0; --t) {
    const float* row = logits + (int64_t)t * row_stride;
    float* prow = post + (int64_t)t * prow_stride;
    for (int a = frame_off[t]; a < frame_off[t + 1]; ++a) {
      const double v = -(double)w[a] + (double)row[pdf[a]] + lb[dst[a]];
      lb[src[a]] = log_add(lb[src[a]], v);
      prow[pdf[a]] += scale * (float)exp(la[src[a]] + v - lp);
    }
  }
  free(la); free(lb);
  return lp;
}

/* Minibatch of N sequences (OpenMP over sequences).  grad must be zero on entry; receives
 * weight*((1+xent)*num_post - den_post).  out[3*N] = objf, num_lp, den_lp (Kaldi NaN guard applied). */
void orc_chain_batch(int S, int P, int64_t A, const int32_t* src, const int32_t* dst, const int32_t* pdf,
                     const float* prob, const float* pi, const float* logits, int64_t seq_stride,
                     int64_t row_stride, const int32_t* lengths, int N, const int32_t* n_src,
                     const int32_t* n_dst, const int32_t* n_pdf, const float* n_w, const int32_t* n_frame_off,
                     const int32_t* n_arc_base, const int32_t* n_state_cnt, const int32_t* n_finals,
                     const float* n_final_w, const int32_t* n_final_off, float leaky, float xent, float weight,
                     float* grad, double* out) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int n = 0; n < N; ++n) {
    const float* lg = logits + (int64_t)n * seq_stride;
    float* g = grad + (int64_t)n * seq_stride;
    const int T = lengths[n];
    int64_t fo = 0;
    for (int m = 0; m < n; ++m) fo += lengths[m] + 1;
    /* frame offsets are relative to the sequence's first arc */
    int32_t* foff = (int32_t*)malloc(sizeof(int32_t) * (T + 1));
    for (int t = 0; t <= T; ++t) foff[t] = n_frame_off[fo + t] - n_arc_base[n];
    const int ab = n_arc_base[n];
    double num = orc_num_fb(n_state_cnt[n], n_src + ab, n_dst + ab, n_pdf + ab, n_w + ab, foff, T,
                            n_finals + n_final_off[n], n_final_w + n_final_off[n],
                            n_final_off[n + 1] - n_final_off[n], lg, row_stride, weight * (1.0f + xent), g,
                            row_stride);
    double check = 0.0;
    double den = orc_den_fb(S, P, A, src, dst, pdf, prob, pi, lg, row_stride, T, leaky, -weight, g, row_stride,
                            &check);
    double objf = weight * (num - den);
    /* Kaldi: abandon when |alpha-beta product - num_sequences| > 2.0 (BetaGeneralFrameDebug), num_sequences = 1 here */
    if (!isfinite(objf) || !(fabs(check - 1.0) <= 2.0)) {
      objf = -10.0 * weight * T;
      for (int t = 0; t < T; ++t) memset(g + (int64_t)t * row_stride, 0, sizeof(float) * P);
    }
    out[n] = objf; out[N + n] = num; out[2 * N + n] = den;
    free(foff);
  }
}
