/* Plain-C statement of the VQ distance / nearest-code assignment.  TEST INFRASTRUCTURE
 * (oracle/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may call it.
 *
 * Follows decomp/nerfvq_nfr3/nerfactor/networks/vq_layers.py:277-301,346-349
 *     distances = sum(x^2,1) - 2*x@C + sum(C^2,0)            (:279-282)
 *     thres:  distances = distances*sel + max(distances)*(1-sel)   (:284-290)
 *     idx = argmax(-distances, 1)   (first index on ties)     (:292)
 *     quantized = C^T[idx]                                    (:346-349)
 * TensorFlow leaves the fp32 summation order of reduce_sum / matmul unspecified; this file
 * pins ONE order -- the order the HIP kernel (vqnerf_release_amd/csrc/vq.hip) uses -- so the
 * kernel can be checked bit for bit (indices AND distances):
 *     dot[n][k]  : one fmaf chain, acc0 = 0, over d in the sequence
 *                  for t in 0..ceil(D/16): for e in 0..3: for q in 0..3: d = 16t + 4q + e
 *                  (elements d >= D are zeros)
 *     x2[n]      : four fmaf chains p_q (q = 0..3) over (t, e) of x[16t+4q+e]^2,
 *                  then (p0 + p1) + (p2 + p3)
 *     c2[k]      : one fmaf chain over d = 0..D-1
 *     dist       : (x2 - 2*dot) + c2   (2*dot is exact)
 * The fp64 statement of the same formulas lives in oracle/decomp.py (vq_distances).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float dot_strict(const float *x, const float *C, int D, int K, int k) {
  float acc = 0.0f;
  int nt = (D + 15) / 16;
  for (int t = 0; t < nt; ++t)
    for (int e = 0; e < 4; ++e)
      for (int q = 0; q < 4; ++q) {
        int d = 16 * t + 4 * q + e;
        float xv = d < D ? x[d] : 0.0f;
        float cv = d < D ? C[(size_t)d * K + k] : 0.0f;
        acc = fmaf(xv, cv, acc);
      }
  return acc;
}

static float x2_strict(const float *x, int D) {
  float p[4] = {0.f, 0.f, 0.f, 0.f};
  int nt = (D + 15) / 16;
  for (int q = 0; q < 4; ++q)
    for (int t = 0; t < nt; ++t)
      for (int e = 0; e < 4; ++e) {
        int d = 16 * t + 4 * q + e;
        float xv = d < D ? x[d] : 0.0f;
        p[q] = fmaf(xv, xv, p[q]);
      }
  return (p[0] + p[1]) + (p[2] + p[3]);
}

/* tf.linalg.l2_normalize over axis 1 (util/math.py:63-64, called at vq_nfr.py:575): y = x * rsqrt(max(sum x^2, eps)).  Pinned order:
 * sum x^2 = x2_strict above (the order vqn_vq_assign already uses for |x|^2); the scale is 1 / sqrt(max(., eps)) with a
 * correctly rounded sqrtf and division, then one multiplication per element. */
int vq_strict_l2_normalize(const float *x, long N, int D, float eps, float *y) {
  for (long n = 0; n < N; ++n) {
    const float *xr = x + (size_t)n * D;
    float x2 = x2_strict(xr, D);
    float sc = 1.0f / sqrtf(x2 > eps ? x2 : eps);
    for (int d = 0; d < D; ++d) y[(size_t)n * D + d] = xr[d] * sc;
  }
  return 0;
}

/* x [N,D] row-major, C [D,K] row-major, sel [K] (0/1) or NULL.
 * dist_out [N,K] or NULL, idx_out [N] int64, quant_out [N,D] or NULL.  Returns 0. */
int vq_strict_assign(const float *x, long N, int D, const float *C, int K, const float *sel,
                     float *dist_out, int64_t *idx_out, float *quant_out) {
  float *c2 = (float *)malloc(sizeof(float) * K);
  float *dist = (float *)malloc(sizeof(float) * (size_t)N * K);
  if (!c2 || !dist) return -1;
  for (int k = 0; k < K; ++k) {
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc = fmaf(C[(size_t)d * K + k], C[(size_t)d * K + k], acc);
    c2[k] = acc;
  }
  float gmax = -INFINITY;
  for (long n = 0; n < N; ++n) {
    const float *xr = x + (size_t)n * D;
    float x2 = x2_strict(xr, D);
    for (int k = 0; k < K; ++k) {
      float dot = dot_strict(xr, C, D, K, k);
      float t1 = x2 - 2.0f * dot;
      float dv = t1 + c2[k];
      dist[(size_t)n * K + k] = dv;
      if (dv > gmax) gmax = dv;
    }
  }
  for (long n = 0; n < N; ++n) {
    float best = 0.f; int bi = -1;
    for (int k = 0; k < K; ++k) {
      float dv = dist[(size_t)n * K + k];
      if (sel && sel[k] == 0.0f) dv = gmax;
      if (dist_out) dist_out[(size_t)n * K + k] = dv;
      if (bi < 0 || dv < best) { best = dv; bi = k; }
    }
    idx_out[n] = bi;
    if (quant_out)
      for (int d = 0; d < D; ++d) quant_out[(size_t)n * D + d] = C[(size_t)d * K + bi];
  }
  free(c2); free(dist);
  return 0;
}

/* EMA statistics (vq_layers.py:304-309): counts[k] = #rows with idx==k (exact, as float),
 * dw[d][k] = sum_n x[n][d] * [idx[n]==k].  The sum order is unspecified in TF; this statement
 * accumulates in double and rounds once, i.e. it is the correctly-rounded reference that the
 * HIP kernel (float atomics / tree sums) is compared against with a tolerance. */
int vq_strict_ema_stats(const float *x, const int64_t *idx, long N, int D, int K, float *counts, float *dw) {
  double *acc = (double *)calloc((size_t)D * K, sizeof(double));
  double *cnt = (double *)calloc(K, sizeof(double));
  if (!acc || !cnt) return -1;
  for (long n = 0; n < N; ++n) {
    int k = (int)idx[n];
    if (k < 0 || k >= K) { free(acc); free(cnt); return -2; }
    cnt[k] += 1.0;
    for (int d = 0; d < D; ++d) acc[(size_t)d * K + k] += (double)x[(size_t)n * D + d];
  }
  for (int k = 0; k < K; ++k) counts[k] = (float)cnt[k];
  for (size_t i = 0; i < (size_t)D * K; ++i) dw[i] = (float)acc[i];
  free(acc); free(cnt);
  return 0;
}
