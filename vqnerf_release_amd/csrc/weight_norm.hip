// Weight normalisation of all layers of a network in one launch, and its backward in another:
//   w[r, :] = g[r] * v[r, :] / ||v[r, :]||                      (nn.utils.weight_norm(dim=0), fields.py:65-66, :139-140)
//   dg[r]   = <dw[r, :], v[r, :]> / ||v[r, :]||
//   dv[r,:] = g[r] / ||v[r, :]|| * (dw[r, :] - v[r, :] * <dw[r, :], v[r, :]> / ||v[r, :]||^2)
// In the reference these are ~10 framework ops per layer and step (norm, div, mul and their autograd), i.e. ~150 launches for
// the 13 layers of the NeuS networks; here one wave per row, rows of all layers in one grid (layer table passed by value).
#include "common.h"

#define VQN_WN_MAX_LAYERS 24

namespace {

struct WnTable {
  const float* v[VQN_WN_MAX_LAYERS];
  const float* g[VQN_WN_MAX_LAYERS];
  const float* dw[VQN_WN_MAX_LAYERS];      // backward only
  float* out0[VQN_WN_MAX_LAYERS];          // forward: w;  backward: dv
  float* out1[VQN_WN_MAX_LAYERS];          // backward: dg
  int rows[VQN_WN_MAX_LAYERS], cols[VQN_WN_MAX_LAYERS], row0[VQN_WN_MAX_LAYERS + 1];
  int n_layers;
};

__device__ __forceinline__ float wave_sum64(float x) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) x += __shfl_xor(x, m);
  return x;
}

template <bool BWD>
__global__ __launch_bounds__(256) void weight_norm_kernel(const WnTable t) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);                 // one wave per row (wave-uniform below)
  if (row >= t.row0[t.n_layers]) return;
  int l = 0;
  while (row >= t.row0[l + 1]) ++l;
  const int r = row - t.row0[l], C = t.cols[l];
  const float* v = t.v[l] + (size_t)r * C;
  float ss = 0.f, dot = 0.f;
  const float* dw = BWD ? t.dw[l] + (size_t)r * C : nullptr;
  for (int c = lane; c < C; c += 64) {
    const float x = v[c];
    ss = fmaf(x, x, ss);
    if (BWD) dot = fmaf(dw[c], x, dot);
  }
  ss = wave_sum64(ss);
  const float nrm = sqrtf(ss), g = t.g[l][r];
  if (!BWD) {
    float* w = t.out0[l] + (size_t)r * C;
    for (int c = lane; c < C; c += 64) w[c] = (g * v[c]) / nrm;       // the reference's order of operations
  } else {
    dot = wave_sum64(dot);
    const float inv = 1.f / nrm, k = dot * inv * inv;
    float* dv = t.out0[l] + (size_t)r * C;
    for (int c = lane; c < C; c += 64) dv[c] = g * inv * (dw[c] - v[c] * k);
    if (lane == 0) t.out1[l][r] = dot * inv;
  }
}

int fill(WnTable& t, int n_layers, const float* const* v, const float* const* g, const float* const* dw, float* const* out0,
         float* const* out1, const int32_t* rows, const int32_t* cols) {
  t.n_layers = n_layers;
  t.row0[0] = 0;
  for (int l = 0; l < n_layers; ++l) {
    if (!v[l] || !g[l] || !out0[l] || rows[l] < 1 || cols[l] < 1) return 1;
    t.v[l] = v[l]; t.g[l] = g[l]; t.dw[l] = dw ? dw[l] : nullptr; t.out0[l] = out0[l]; t.out1[l] = out1 ? out1[l] : nullptr;
    if (dw && (!dw[l] || !out1[l])) return 1;
    t.rows[l] = rows[l]; t.cols[l] = cols[l];
    t.row0[l + 1] = t.row0[l] + rows[l];
  }
  return 0;
}

}  // namespace

extern "C" int vqn_weight_norm_fwd(int n_layers, const float* const* v, const float* const* g, float* const* w,
                                   const int32_t* rows, const int32_t* cols, void* stream) {
  VQN_CHECK_ARG(n_layers >= 0 && n_layers <= VQN_WN_MAX_LAYERS, "0 <= n_layers <= 24");
  if (n_layers == 0) return VQN_OK;
  VQN_CHECK_ARG(v && g && w && rows && cols, "null table");
  WnTable t;
  VQN_CHECK_ARG(fill(t, n_layers, v, g, nullptr, w, nullptr, rows, cols) == 0, "null layer pointer or empty layer");
  const int total = t.row0[n_layers];
  hipLaunchKernelGGL((weight_norm_kernel<false>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, t);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_weight_norm_bwd(int n_layers, const float* const* v, const float* const* g, const float* const* dw,
                                   float* const* dv, float* const* dg, const int32_t* rows, const int32_t* cols, void* stream) {
  VQN_CHECK_ARG(n_layers >= 0 && n_layers <= VQN_WN_MAX_LAYERS, "0 <= n_layers <= 24");
  if (n_layers == 0) return VQN_OK;
  VQN_CHECK_ARG(v && g && dw && dv && dg && rows && cols, "null table");
  WnTable t;
  VQN_CHECK_ARG(fill(t, n_layers, v, g, dw, dv, dg, rows, cols) == 0, "null layer pointer or empty layer");
  const int total = t.row0[n_layers];
  hipLaunchKernelGGL((weight_norm_kernel<true>), dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, t);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
