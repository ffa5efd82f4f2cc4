// Weight-gradient contraction over points for gfx950:  G[o][i] = sum_p A[o][p] * B[i][p]  with A, B in the
// feature-major TFMT layout written by csrc/tile_vm.hip (include/vqn_vm_desc.h).  In the reference these are the
// `grad_weight` GEMMs autograd runs inside loss.backward() for every nn.Linear of fields.py (SDFNetwork,
// RenderingNetwork), once per adjoint stream.
//
// One workgroup = 4 waves owns the whole [<=256 x <=256] output block for a strided subset of the 32-point tiles
// (split-K over points); wave w accumulates output tiles (ot in {w, w+4}) x (it in 0..7) in up to 256 accumulator
// registers.  Operands stream straight from HBM into MFMA fragments: lane (feature, kk) reads 4 consecutive points
// (one dwordx4) per 4 MFMAs; the point -> K-slot assignment is a fixed permutation shared by A and B, which a sum
// over points does not care about.  Partial blocks go to a workspace [n_split][rows][cols]; the caller reduces them
// in a fixed order (deterministic, no float atomics).
#include "common.h"
#include "wgrad_batch.h"

namespace {

// NOT: output tiles per wave (1: a_nt <= 4, 2: a_nt <= 8); BT: B tiles held (4 or 8); D: operand buffers = steps in flight.
#ifndef VQN_WGRAD_D
#define VQN_WGRAD_D 2          // operand ring depth of the 256 x 256 form (see the experiments table of DESIGN.md)
#endif
template <int NOT, int BT, int D>
__device__ __forceinline__ void wgrad_body(const float* __restrict__ A, int a_tiles, int a_t0, int a_nt,
                                           const float* __restrict__ B, int b_tiles, int b_t0, int b_nt,
                                           long n_ptiles, float* __restrict__ ws, float* __restrict__ rowsum_ws) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: scalar guards around the MFMA groups
  const int fi = lane & 31, kk = lane >> 5;
  f32x16 acc[NOT][BT];
#pragma unroll
  for (int a = 0; a < NOT; ++a)
#pragma unroll
    for (int b = 0; b < BT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  f32x4 rs[NOT];                                 // running sum over points of this lane's A values (-> bias gradients)
#pragma unroll
  for (int a = 0; a < NOT; ++a) rs[a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // One register-resident wave per SIMD: nothing else hides the HBM latency of the operand fetches, so they run D - 1 steps
  // ahead of the multiplies through a ring of D register buffers; step q = (point tile, quarter u).  A full 256 x 256 block
  // multiplies for ~1.7 us per step and two buffers cover the latency; a 128 x 128 block (the reflectance stacks) multiplies
  // for 0.4 us per step and needs six, or it runs at the latency, not the bandwidth (13 TFLOP/s measured with two).  The
  // fetches are UNCONDITIONAL (tile indices clamped; an out-of-range tile is fetched and never multiplied): with guarded
  // fetches the compiler falls back to s_waitcnt vmcnt(0) right after issuing them (mlp_prims.h, gemm_tiles).
  const long my_tiles = (n_ptiles - blockIdx.x + gridDim.x - 1) / gridDim.x, n_q = 4 * my_tiles;
  auto fetch = [&](long q, f32x4 (&af)[NOT], f32x4 (&bf)[BT]) {
    long t = blockIdx.x + (q >> 2) * (long)gridDim.x;
    if (t >= n_ptiles) t = n_ptiles - 1;
    const int u = (int)(q & 3);
    const float* At = A + ((t * a_tiles + a_t0) * 32 + fi) * 32 + 4 * kk + 8 * u;
    const float* Bt = B + ((t * b_tiles + b_t0) * 32 + fi) * 32 + 4 * kk + 8 * u;
#pragma unroll
    for (int a = 0; a < NOT; ++a) af[a] = *reinterpret_cast<const f32x4*>(At + (long)min(wave + 4 * a, a_nt - 1) * 1024);
#pragma unroll
    for (int b = 0; b < BT; ++b) bf[b] = *reinterpret_cast<const f32x4*>(Bt + (long)min(b, b_nt - 1) * 1024);
  };
  auto multiply = [&](const f32x4 (&af)[NOT], const f32x4 (&bf)[BT]) {
#pragma unroll
    for (int a = 0; a < NOT; ++a) {
      rs[a] += af[a];
#pragma unroll
      for (int b = 0; b < BT; ++b)
        if (wave + 4 * a < a_nt && b < b_nt) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][j], bf[b][j], acc[a][b], 0, 0, 0);
        }
    }
  };
  f32x4 af[D][NOT], bf[D][BT];
#pragma unroll
  for (int s = 0; s < D - 1; ++s) fetch(s, af[s], bf[s]);
  for (long q = 0; q < n_q; q += D) {
#pragma unroll
    for (int s = 0; s < D; ++s) {
      fetch(q + s + D - 1, af[(s + D - 1) % D], bf[(s + D - 1) % D]);      // the buffer multiplied one step ago
      if (q + s < n_q) multiply(af[s], bf[s]);
    }
  }
  // partial block of this workgroup: ws[blockIdx][a_nt*32][b_nt*32]; accumulator reg e of lane (n, hh): row (e&3) + 8 (e>>2) + 4 hh, col n
  const int cols = b_nt * 32;
  float* w = ws + (size_t)blockIdx.x * (size_t)(a_nt * 32) * cols;
#pragma unroll
  for (int a = 0; a < NOT; ++a) {
    const int ot = wave + 4 * a;
    if (ot >= a_nt) continue;
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b >= b_nt) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * kk;
        w[(size_t)(ot * 32 + row) * cols + b * 32 + fi] = acc[a][b][e];
      }
    }
    if (rowsum_ws != nullptr) {                  // sum over this workgroup's points of A[ot*32 + fi][.]: lanes (fi, 0) and (fi, 1) in a fixed order
      float r = (rs[a][0] + rs[a][1]) + (rs[a][2] + rs[a][3]);
      r += __shfl_xor(r, 32);
      if (kk == 0) rowsum_ws[(size_t)blockIdx.x * (a_nt * 32) + ot * 32 + fi] = r;
    }
  }
}

template <int NOT, int BT, int D>
__global__ __launch_bounds__(256, 1) void wgrad_kernel(const float* __restrict__ A, int a_tiles, int a_t0, int a_nt,
                                                       const float* __restrict__ B, int b_tiles, int b_t0, int b_nt,
                                                       long n_ptiles, float* __restrict__ ws, float* __restrict__ rowsum_ws) {
  wgrad_body<NOT, BT, D>(A, a_tiles, a_t0, a_nt, B, b_tiles, b_t0, b_nt, n_ptiles, ws, rowsum_ws);
}

// many contractions over the same points in one launch (blockIdx.y: the problem): the reference's 2048-point steps are launch-bound
template <int NOT, int BT, int D>
__global__ __launch_bounds__(256, 1) void wgrad_batched_kernel(const WgTable tab, long n_ptiles) {
  const WgProblem& P = tab.p[blockIdx.y];
  wgrad_body<NOT, BT, D>(P.A, P.a_tiles, P.a_t0, P.a_nt, P.B, P.b_tiles, P.b_t0, P.b_nt, n_ptiles, P.ws, P.rs);
}

// out[r][c] (+)= sum_s ws[s][r][c] in a fixed order.  A block owns 16 consecutive float4 of the output; its 16 thread groups
// each sum a contiguous share of the n partial blocks (all loads independent: bandwidth-, not latency-bound), then thread
// group 0 adds the 16 group sums in group order (LDS): the same order for every launch, hence deterministic.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const f32x4* __restrict__ ws, const int n, const long numel4,
                                                              const int cols4, float* __restrict__ out, const long out_ld,
                                                              const int accumulate) {
  __shared__ f32x4 part[16][16];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long i = (long)blockIdx.x * 16 + o;
  const int per = (n + 15) >> 4, s0 = g * per, s1 = min(n, s0 + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (i < numel4)
    for (int s = s0; s < s1; ++s) acc += ws[(size_t)s * numel4 + i];
  part[g][o] = acc;
  __syncthreads();
  if (g == 0 && i < numel4) {
    f32x4 t = part[0][o];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += part[k][o];
    const long r = i / cols4, c4 = i - r * cols4;
    f32x4* op = reinterpret_cast<f32x4*>(out + r * out_ld + 4 * c4);
    if (accumulate) t += *op;
    *op = t;
  }
}

}  // namespace

extern "C" int vqn_reduce_partials(const float* ws, int n, int rows, int cols, float* out, int64_t out_ld, int accumulate,
                                   void* stream) {
  VQN_CHECK_ARG(ws && out, "null pointer");
  VQN_CHECK_ARG(n >= 1 && rows >= 1 && cols >= 4, "n >= 1, rows >= 1, cols >= 4");
  VQN_CHECK_SHAPE((cols & 3) == 0 && (out_ld & 3) == 0 && out_ld >= cols, "cols and out_ld multiples of 4, out_ld >= cols");
  VQN_CHECK_SHAPE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)out & 15) == 0, "ws and out must be 16-byte aligned");
  const long numel4 = (long)rows * (cols / 4);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((numel4 + 15) / 16)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const f32x4*>(ws), n, numel4, cols / 4, out, (long)out_ld, accumulate);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

int vqn_wgrad_partials_f32_internal(const float* A, int a_tiles, int a_t0, int a_nt, const float* B, int b_tiles, int b_t0, int b_nt,
                                    int64_t n_point_tiles, int n_split, float* ws, float* rowsum_ws, void* stream);      // below

extern "C" int vqn_wgrad_partials(const float* A, int a_tiles, int a_t0, int a_nt, const float* B, int b_tiles, int b_t0,
                                  int b_nt, int64_t n_point_tiles, int n_split, float* ws, float* rowsum_ws, void* stream) {
  return vqn_wgrad_partials_f32_internal(A, a_tiles, a_t0, a_nt, B, b_tiles, b_t0, b_nt, n_point_tiles, n_split, ws, rowsum_ws, stream);
}

int vqn_wgrad_partials_f32_internal(const float* A, int a_tiles, int a_t0, int a_nt, const float* B, int b_tiles, int b_t0, int b_nt,
                                    int64_t n_point_tiles, int n_split, float* ws, float* rowsum_ws, void* stream) {
  VQN_CHECK_ARG(A && B && ws, "null pointer");
  VQN_CHECK_ARG(n_point_tiles >= 1 && n_split >= 1, "n_point_tiles >= 1, n_split >= 1");
  VQN_CHECK_SHAPE(a_nt >= 1 && a_nt <= 8 && b_nt >= 1 && b_nt <= 8, "1..8 feature tiles per operand and call");
  VQN_CHECK_SHAPE(a_t0 >= 0 && a_t0 + a_nt <= a_tiles && b_t0 >= 0 && b_t0 + b_nt <= b_tiles, "feature-tile range outside the tensor");
  VQN_CHECK_SHAPE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "operands must be 16-byte aligned");
  long grid = n_split;
  if (grid > n_point_tiles) grid = n_point_tiles;
  hipStream_t s = (hipStream_t)stream;
#define VQN_WGRAD(NOT_, BT_, D_)                                                                                       \
  hipLaunchKernelGGL((wgrad_kernel<NOT_, BT_, D_>), dim3((unsigned)grid), dim3(256), 0, s, A, a_tiles, a_t0, a_nt, B, \
                     b_tiles, b_t0, b_nt, (long)n_point_tiles, ws, rowsum_ws)
  if (a_nt <= 4 && b_nt <= 4) VQN_WGRAD(1, 4, 6);
  else if (a_nt <= 4) VQN_WGRAD(1, 8, 4);
  else VQN_WGRAD(2, 8, VQN_WGRAD_D);
#undef VQN_WGRAD
  VQN_LAUNCH_CHECK();
  return (int)grid;      // number of partial blocks written (>= 1)
}

int vqn_wgrad_f32_batched_internal(const WgProblem* probs, int count, int cls, long n_point_tiles, long grid, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  for (int c0 = 0; c0 < count; c0 += WG_MAX) {
    WgTable tab;
    memset(&tab, 0, sizeof(tab));
    const int k = count - c0 < WG_MAX ? count - c0 : WG_MAX;
    for (int i = 0; i < k; ++i) tab.p[i] = probs[c0 + i];
    const dim3 g((unsigned)grid, (unsigned)k);
    if (cls == 0) hipLaunchKernelGGL((wgrad_batched_kernel<1, 4, 6>), g, dim3(256), 0, s, tab, n_point_tiles);
    else if (cls == 1) hipLaunchKernelGGL((wgrad_batched_kernel<1, 8, 4>), g, dim3(256), 0, s, tab, n_point_tiles);
    else hipLaunchKernelGGL((wgrad_batched_kernel<2, 8, VQN_WGRAD_D>), g, dim3(256), 0, s, tab, n_point_tiles);
    VQN_LAUNCH_CHECK();
  }
  return VQN_OK;
}
