// What does v_mfma_f32_32x32x16_bf16 do with bits below the accumulator's ulp?  One wave; A row i = weights, B col n = values.
// Each case sets C and the 16 products of output (0, 0) and prints the result bits.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* a16, const float* b16, float c, float* out) {
  const int lane = threadIdx.x, h = lane >> 5, r = lane & 31;
  // lane (r, h) supplies A[r][k = 8 h + j], B[k = 8 h + j][r]; only row 0 / col 0 are non-zero
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    const float av = r == 0 ? a16[8 * h + j] : 0.f, bv = r == 0 ? b16[8 * h + j] : 0.f;
    a[j] = (__bf16)av; b[j] = (__bf16)bv;
  }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (lane == 0) acc[0] = c;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  if (lane == 0) out[0] = acc[0];
}
static float run(const float* a, const float* b, float c) {
  float *da, *db, *dout, out;
  hipMalloc(&da, 64); hipMalloc(&db, 64); hipMalloc(&dout, 4);
  hipMemcpy(da, a, 64, hipMemcpyHostToDevice); hipMemcpy(db, b, 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, c, dout);
  hipMemcpy(&out, dout, 4, hipMemcpyDeviceToHost);
  hipFree(da); hipFree(db); hipFree(dout);
  return out;
}
static void show(const char* what, float got, double exact) {
  unsigned u; memcpy(&u, &got, 4);
  const float rne = (float)exact;
  printf("%-64s got %.10e (0x%08x)  exact %.10e  RNE %.10e  %s\n", what, got, u, exact, rne, got == rne ? "== RNE" : (fabs(got) < fabs(rne) ? "below RNE (truncated?)" : "other"));
}
int main() {
  float a[16], b[16];
  auto zero = [&] { for (int i = 0; i < 16; ++i) { a[i] = 0; b[i] = 0; } };
  const float u = ldexpf(1.f, -23);     // ulp(1)
  zero(); a[0] = 1.5f; b[0] = ldexpf(1.f, -24);
  show("C=1, one product 0.75 ulp", run(a, b, 1.f), 1.0 + 0.75 * u);
  zero(); a[0] = 1.f; b[0] = ldexpf(1.f, -24);
  show("C=1, one product 0.5 ulp (tie -> even = 1)", run(a, b, 1.f), 1.0 + 0.5 * u);
  zero(); a[0] = 1.25f; b[0] = ldexpf(1.f, -24);
  show("C=1, one product 0.625 ulp", run(a, b, 1.f), 1.0 + 0.625 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.f; b[i] = ldexpf(1.f, -26); }
  show("C=1, 16 products of 1/8 ulp (sum 2 ulp)", run(a, b, 1.f), 1.0 + 2 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.f; b[i] = ldexpf(1.f, -28); }
  show("C=1, 16 products of 1/32 ulp (sum 0.5 ulp, tie)", run(a, b, 1.f), 1.0 + 0.5 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.5f; b[i] = ldexpf(1.f, -28); }
  show("C=1, 16 products of 1.5/32 ulp (sum 0.75 ulp)", run(a, b, 1.f), 1.0 + 0.75 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.f; b[i] = ldexpf(1.f, -30); }
  show("C=1, 16 products of 1/128 ulp (sum 0.125 ulp)", run(a, b, 1.f), 1.0 + 0.125 * u);
  zero(); a[0] = 1.f; b[0] = 1.f; for (int i = 1; i < 16; ++i) { a[i] = 1.5f; b[i] = ldexpf(1.f, -28); }
  show("C=0, product 1 + 15 products of 1.5/32 ulp (sum 0.703 ulp)", run(a, b, 0.f), 1.0 + 15 * 1.5 / 32 * u);
  zero(); a[0] = -1.5f; b[0] = ldexpf(1.f, -24);
  show("C=1, one product -0.75 ulp(1) = -1.5 ulp(below 1)", run(a, b, 1.f), 1.0 - 0.75 * u);
  zero(); a[0] = -1.25f; b[0] = ldexpf(1.f, -25);
  show("C=1, one product -0.3125 ulp(1)", run(a, b, 1.f), 1.0 - 0.3125 * u);
  zero(); a[0] = 1.f; b[0] = 1.f; a[1] = 1.5f; b[1] = ldexpf(1.f, -24);
  show("C=0, products 1 and 0.75 ulp", run(a, b, 0.f), 1.0 + 0.75 * u);
  zero(); a[0] = 1.f; b[0] = 1.f; a[8] = 1.5f; b[8] = ldexpf(1.f, -24);
  show("C=0, products 1 (k=0) and 0.75 ulp (k=8, other half)", run(a, b, 0.f), 1.0 + 0.75 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.f + i / 128.f; b[i] = 1.f + (15 - i) / 128.f; }
  { double e = 0.337; for (int i = 0; i < 16; ++i) e += (double)(float)(__bf16)a[i] * (double)(float)(__bf16)b[i]; show("C=0.337, 16 products ~1 (8-bit operands)", run(a, b, 0.337f), e); }
  zero(); for (int i = 0; i < 16; ++i) { a[i] = -1.5f; b[i] = ldexpf(1.f, -28); }
  show("C=1, 16 products of -1.5/32 ulp (sum -0.75 ulp)", run(a, b, 1.f), 1.0 - 0.75 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = -1.5f; b[i] = ldexpf(1.f, -27); }
  show("C=2, 16 products of -1.5/32 ulp(2) (sum -0.75 ulp(2))", run(a, b, 2.f), 2.0 - 0.75 * 2 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.5f; b[i] = ldexpf(1.f, -28); }
  show("C=-1, 16 products of +1.5/32 ulp", run(a, b, -1.f), -1.0 + 0.75 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = -1.5f; b[i] = ldexpf(1.f, -28); }
  show("C=-1, 16 products of -1.5/32 ulp", run(a, b, -1.f), -1.0 - 0.75 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.75f; b[i] = ldexpf(1.f, -28); }
  show("C=1, 16 products of 1.75/32 ulp (sum 0.875)", run(a, b, 1.f), 1.0 + 0.875 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.f; b[i] = ldexpf(1.f, -29); }
  show("C=1, 16 products of 1/64 ulp (sum 0.25)", run(a, b, 1.f), 1.0 + 0.25 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.f; b[i] = ldexpf(1.f, -27); }
  show("C=1, 16 products of 1/16 ulp (sum 1)", run(a, b, 1.f), 1.0 + 1 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.5f; b[i] = ldexpf(1.f, -27); }
  show("C=1, 16 products of 1.5/16 ulp (sum 1.5 -> tie even 2)", run(a, b, 1.f), 1.0 + 1.5 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.25f; b[i] = ldexpf(1.f, -27); }
  show("C=1, 16 products of 1.25/16 ulp (sum 1.25)", run(a, b, 1.f), 1.0 + 1.25 * u);
  zero(); for (int i = 0; i < 16; ++i) { a[i] = 1.75f; b[i] = ldexpf(1.f, -27); }
  show("C=1, 16 products of 1.75/16 ulp (sum 1.75)", run(a, b, 1.f), 1.0 + 1.75 * u);
  // guard bits of the alignment against C: 8 products (one K half) of (1/16)(1 + 2^-m) ulp: exact sum 0.5 + 2^-(m+1) ulp -> RNE gives 1 + ulp
  for (int m = 1; m <= 7; ++m) {
    zero(); for (int i = 0; i < 8; ++i) { a[i] = 1.f + ldexpf(1.f, -m); b[i] = ldexpf(1.f, -27); }
    char nm[96]; snprintf(nm, sizeof nm, "C=1, 8 products (1/16)(1+2^-%d) ulp: sum 0.5+2^-%d", m, m + 1);
    show(nm, run(a, b, 1.f), 1.0 + (0.5 + ldexp(1.0, -(m + 1))) * u);
  }
  // the same against a LARGE product in the group instead of C (C = 0): product 1 + 7 products of (1/14)(...)?  use k=0: 1.0, k=1..7: (1/16)(1+2^-m) ulp x 8/7 not exact -> use 8 in other half
  for (int m = 1; m <= 7; ++m) {
    zero(); a[0] = 1.f; b[0] = 1.f; for (int i = 8; i < 16; ++i) { a[i] = 1.f + ldexpf(1.f, -m); b[i] = ldexpf(1.f, -27); }
    char nm[96]; snprintf(nm, sizeof nm, "C=0, k0: 1.0; other half: 8 x (1/16)(1+2^-%d) ulp", m);
    show(nm, run(a, b, 0.f), 1.0 + (0.5 + ldexp(1.0, -(m + 1))) * u);
  }
  // intra-group: product 1.0 at k=0 and 7 products at k=1..7 of (1/14)... : use 4 products of (1/8)(1+2^-m): sum 0.5 + 2^-(m+1)
  for (int m = 1; m <= 7; ++m) {
    zero(); a[0] = 1.f; b[0] = 1.f; for (int i = 1; i < 5; ++i) { a[i] = 1.f + ldexpf(1.f, -m); b[i] = ldexpf(1.f, -26); }
    char nm[96]; snprintf(nm, sizeof nm, "C=0, same half: 1.0 + 4 x (1/8)(1+2^-%d) ulp", m);
    show(nm, run(a, b, 0.f), 1.0 + (0.5 + ldexp(1.0, -(m + 1))) * u);
  }
  return 0;
}
