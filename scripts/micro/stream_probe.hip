// Micro-benchmark: how fast can [N,256] f32 rows be streamed into registers in the MFMA A-operand layout of the VQ kernels
// (lane (col, q) holds row `col`, floats 16 i + 4 q .. +3), against the row-contiguous ceiling?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, long N, float* out, int waves_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
  const long n_groups = N >> 4;
  f32x4 acc = {0, 0, 0, 0};
  for (long rg = (long)blockIdx.x * waves_per_wg + wave; rg < n_groups; rg += (long)gridDim.x * waves_per_wg) {
    const float* base = x + (rg << 4) * 256;
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float* p;
      if (MODE == 0) p = base + col * 256 + 16 * i + 4 * q;            // A-operand layout: 16 rows x 64 B per instruction
      else if (MODE == 1) p = base + i * 256 + 4 * lane;              // one whole row per instruction
      else p = base + col * 256 + 128 * (i >> 3) + 32 * q + 4 * (i & 7);  // never mind
      if (NT) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
      else v[i] = *reinterpret_cast<const f32x4*>(p);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <int MODE, int NT>
void run(const char* name, const float* x, long N, float* out, int wg_per_cu, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(threads), 0, 0, x, N, out, threads / 64);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(threads), 0, 0, x, N, out, threads / 64);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
  printf("%-40s wg/cu=%d thr=%d: %.3f ms  %.0f GB/s\n", name, wg_per_cu, threads, ms, N * 1024.0 / ms / 1e6);
}

int main() {
  const long N = 1 << 20;
  float *x, *out;
  hipMalloc(&x, N * 1024); hipMalloc(&out, 64);
  hipMemset(x, 0, N * 1024);
  for (int wg : {4, 8, 16}) {
    run<0, 0>("A-layout", x, N, out, wg, 256);
    run<0, 1>("A-layout nt", x, N, out, wg, 256);
    run<1, 0>("row per instruction", x, N, out, wg, 256);
    run<1, 1>("row per instruction nt", x, N, out, wg, 256);
  }
  return 0;
}
