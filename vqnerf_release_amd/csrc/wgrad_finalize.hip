// Batched tail of the weight-gradient contractions, and a batched device-to-device copy: launch-count reducers for the training
// steps, whose small configurations (the reference trains 2048 points a step, decomp/nerfvq_nfr3/nerfactor/trainvali.py:443-486)
// are bound by the number of kernel launches, not by their work.
//
//  * vqn_wgrad_finalize: the split-over-points partial blocks of MANY contractions (vqn_wgrad_partials*, each into its own
//    workspace) are summed in the fixed order of vqn_reduce_partials (bit-identical sums) and written where each result
//    belongs -- a slice of a larger matrix, transposed (the Keras [in, out] layout of the reflectance nets), scaled (the skip
//    layer's 1/sqrt2), two contractions added (the first- and second-order terms of the SDF layers) -- in ONE launch instead of
//    one reduce + transpose + cat + scale kernel sequence per weight.
//  * vqn_multi_copy: many contiguous f32 copies (the per-parameter gradients into the flat gradient bucket) in one launch.
#include "common.h"
#include "vqnerf_hip.h"

namespace {

constexpr int FIN_MAX = 40;

struct FinEntry {
  const f32x4* ws; const f32x4* ws2; float* dst;
  long numel4, dst_sr, dst_sc;
  int n, n2, cols4, rows_valid, cols_valid, blk0;
  float scale; int col0;
};
struct FinTable { FinEntry e[FIN_MAX]; int count; };

// the partial sums of one float4 column group, in vqn_reduce_partials' order: 16 thread groups each sum a contiguous share of the n
// blocks, then group 0 adds the 16 group sums in group order
__device__ __forceinline__ f32x4 ordered_sum(const f32x4* __restrict__ ws, const int n, const long numel4, const long i, const int g, const int o,
                                             f32x4 (*part)[16]) {
  const int per = (n + 15) >> 4, s0 = g * per, s1 = min(n, s0 + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (i < numel4)
    for (int s = s0; s < s1; ++s) acc += ws[(size_t)s * numel4 + i];
  __syncthreads();                                   // (part may still be read from the previous call)
  part[g][o] = acc;
  __syncthreads();
  f32x4 t = part[0][o];
#pragma unroll
  for (int k = 1; k < 16; ++k) t += part[k][o];
  return t;
}

__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const FinTable tab) {
  __shared__ f32x4 part[16][16];
  int ei = 0;
  for (int k = 1; k < tab.count; ++k)
    if ((int)blockIdx.x >= tab.e[k].blk0) ei = k;
  const FinEntry& E = tab.e[ei];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long i = (long)(blockIdx.x - E.blk0) * 16 + o;
  f32x4 t = ordered_sum(E.ws, E.n, E.numel4, i, g, o, part);
  if (E.ws2 != nullptr) {
    const f32x4 t2 = ordered_sum(E.ws2, E.n2, E.numel4, i, g, o, part);
    t = t2 + t;                                      // (vqn_reduce_partials with accumulate: second sum + what the first one left)
  }
  if (g == 0 && i < E.numel4) {
    const long r = i / E.cols4, c0 = 4 * (i - r * E.cols4);
    if (r < E.rows_valid) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c0 + j >= E.col0 && c0 + j < E.cols_valid) E.dst[r * E.dst_sr + (c0 + j) * E.dst_sc] = E.scale == 1.0f ? t[j] : t[j] * E.scale;
    }
  }
}

constexpr int CP_MAX = 96;
struct CopyTable { const float* src[CP_MAX]; float* dst[CP_MAX]; long n[CP_MAX]; int blk0[CP_MAX]; int count; };

__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyTable tab) {
  int ei = 0;
  for (int k = 1; k < tab.count; ++k)
    if ((int)blockIdx.x >= tab.blk0[k]) ei = k;
  const float* __restrict__ s = tab.src[ei];
  float* __restrict__ d = tab.dst[ei];
  const long n = tab.n[ei], base = (long)(blockIdx.x - tab.blk0[ei]) * 1024 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long i = base + 256 * k;
    if (i < n) d[i] = s[i];
  }
}

}  // namespace

extern "C" int vqn_wgrad_finalize(int count, const float* const* ws, const int32_t* n, const float* const* ws2, const int32_t* n2,
                                  const int32_t* src_rows, const int32_t* src_cols, const int32_t* rows_valid, const int32_t* col_first,
                                  const int32_t* cols_valid, float* const* dst, const int64_t* dst_row_stride, const int64_t* dst_col_stride, const float* scale,
                                  void* stream) {
  VQN_CHECK_ARG(count >= 0 && ws && n && ws2 && n2 && src_rows && src_cols && rows_valid && col_first && cols_valid && dst && dst_row_stride &&
                dst_col_stride && scale, "null pointer");
  for (int c0 = 0; c0 < count; c0 += FIN_MAX) {
    FinTable tab;
    memset(&tab, 0, sizeof(tab));
    tab.count = count - c0 < FIN_MAX ? count - c0 : FIN_MAX;
    long blocks = 0;
    for (int k = 0; k < tab.count; ++k) {
      const int i = c0 + k;
      VQN_CHECK_ARG(ws[i] && dst[i] && n[i] >= 1 && (ws2[i] == nullptr || n2[i] >= 1), "entry: ws, dst, n >= 1");
      VQN_CHECK_SHAPE(src_rows[i] >= 1 && src_cols[i] >= 4 && (src_cols[i] & 3) == 0 && rows_valid[i] >= 1 && rows_valid[i] <= src_rows[i] &&
                      cols_valid[i] >= 1 && cols_valid[i] <= src_cols[i] && col_first[i] >= 0 && col_first[i] < cols_valid[i], "entry: partial blocks [src_rows, src_cols], cols a multiple of 4");
      VQN_CHECK_SHAPE(((uintptr_t)ws[i] & 15) == 0 && ((uintptr_t)ws2[i] & 15) == 0, "workspaces must be 16-byte aligned");
      FinEntry& E = tab.e[k];
      E.ws = reinterpret_cast<const f32x4*>(ws[i]);
      E.ws2 = reinterpret_cast<const f32x4*>(ws2[i]);
      E.dst = dst[i];
      E.numel4 = (long)src_rows[i] * (src_cols[i] / 4);
      E.dst_sr = dst_row_stride[i]; E.dst_sc = dst_col_stride[i];
      E.n = n[i]; E.n2 = n2[i]; E.cols4 = src_cols[i] / 4; E.rows_valid = rows_valid[i]; E.cols_valid = cols_valid[i];
      E.blk0 = (int)blocks;
      E.scale = scale[i];
      E.col0 = col_first[i];
      blocks += (E.numel4 + 15) / 16;
    }
    if (blocks == 0) continue;
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tab);
    VQN_LAUNCH_CHECK();
  }
  return VQN_OK;
}

extern "C" int vqn_multi_copy(int count, const float* const* src, float* const* dst, const int64_t* n, void* stream) {
  VQN_CHECK_ARG(count >= 0 && src && dst && n, "null pointer");
  for (int c0 = 0; c0 < count; c0 += CP_MAX) {
    CopyTable tab;
    memset(&tab, 0, sizeof(tab));
    tab.count = count - c0 < CP_MAX ? count - c0 : CP_MAX;
    long blocks = 0;
    for (int k = 0; k < tab.count; ++k) {
      const int i = c0 + k;
      VQN_CHECK_ARG(n[i] >= 0 && (n[i] == 0 || (src[i] && dst[i])), "entry: src, dst, n >= 0");
      tab.src[k] = src[i]; tab.dst[k] = dst[i]; tab.n[k] = n[i]; tab.blk0[k] = (int)blocks;
      blocks += (n[i] + 1023) / 1024;
    }
    if (blocks == 0) continue;
    hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tab);
    VQN_LAUNCH_CHECK();
  }
  return VQN_OK;
}
