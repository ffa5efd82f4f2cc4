/* AddressSanitizer / UBSan check of the oracle's C code (oracle/vq_strict.c): ragged sizes, D not a multiple of 16, K = 1, a masked
 * call, empty input.  Built and run by tests/test_sanitizers.py (gcc -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=all). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int vq_strict_assign(const float *x, long N, int D, const float *C, int K, const float *sel, float *dist_out, int64_t *idx_out,
                     float *quant_out);
int vq_strict_ema_stats(const float *x, const int64_t *idx, long N, int D, int K, float *counts, float *dw);

static float frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float)(*s >> 8) / 16777216.0f; }

int main(void) {
  unsigned seed = 1;
  const int Ds[] = {4, 12, 16, 252, 256}, Ks[] = {1, 8, 15, 64, 128};
  const long Ns[] = {0, 1, 17, 1000};
  int runs = 0;
  for (int a = 0; a < 5; ++a)
    for (int b = 0; b < 5; ++b)
      for (int c = 0; c < 4; ++c) {
        const int D = Ds[a], K = Ks[b];
        const long N = Ns[c];
        float *x = malloc(sizeof(float) * (size_t)(N ? N : 1) * D), *C = malloc(sizeof(float) * (size_t)D * K);
        float *sel = malloc(sizeof(float) * K), *dist = malloc(sizeof(float) * (size_t)(N ? N : 1) * K);
        float *quant = malloc(sizeof(float) * (size_t)(N ? N : 1) * D), *counts = malloc(sizeof(float) * K);
        float *dw = malloc(sizeof(float) * (size_t)D * K);
        int64_t *idx = malloc(sizeof(int64_t) * (size_t)(N ? N : 1));
        for (long i = 0; i < N * D; ++i) x[i] = frand(&seed);
        for (int i = 0; i < D * K; ++i) C[i] = frand(&seed);
        for (int k = 0; k < K; ++k) sel[k] = (k % 3) ? 1.0f : 0.0f;
        if (vq_strict_assign(x, N, D, C, K, NULL, dist, idx, quant)) return 1;
        if (vq_strict_assign(x, N, D, C, K, sel, NULL, idx, NULL)) return 1;
        for (long n = 0; n < N; ++n)
          if (idx[n] < 0 || idx[n] >= K) { fprintf(stderr, "index out of range\n"); return 1; }
        if (vq_strict_ema_stats(x, idx, N, D, K, counts, dw)) return 1;
        free(x); free(C); free(sel); free(dist); free(quant); free(counts); free(dw); free(idx);
        ++runs;
      }
  printf("vq_strict under ASan/UBSan: %d runs ok\n", runs);
  return 0;
}
