// Small element-wise pieces of the reflectance model's training graph, one launch each instead of 3..9 framework launches (the
// captured 2048-point step is launch-count-bound: decomp/nerfvq_nfr3/nerfactor/trainvali.py:443-486 trains 1024 pixel pairs a step).
//  * vqn_clip_preserve: tfp.math.clip_by_value_preserve_gradient in the reference's own arithmetic, x + (clip(x) - x), every step
//    rounded (not always bitwise clip(x)); used on the rendered colours (vq_nfr.py:731) and the light (:735-745).  Its gradient is
//    the identity: no backward kernel.
//  * vqn_ks_split_fwd / _bwd: spec = ks * basecolor, albedo = (1 - ks) * basecolor (vq_nfr.py:590-592) and their adjoints.
#include "common.h"
#include "vqnerf_hip.h"

namespace {

__global__ __launch_bounds__(256) void clip_preserve_kernel(const float* __restrict__ x, const long n, const float lo, const float hi,
                                                            float* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    const float c = fminf(fmaxf(v, lo), hi);                    // (NaN propagates through the sum below: v + (c - v) with v = NaN)
    y[i] = __fadd_rn(v, __fsub_rn(c, v));
  }
}

// ks: [N, kc] with kc = 1 (broadcast over the three channels) or 3
__global__ __launch_bounds__(256) void ks_split_fwd_kernel(const float* __restrict__ bc, const float* __restrict__ ks, const int kc, const long n,
                                                           float* __restrict__ albedo, float* __restrict__ spec) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 3 * n; i += (long)gridDim.x * 256) {
    const long p = i / 3;
    const int c = (int)(i - 3 * p);
    const float k = ks[p * kc + (kc == 1 ? 0 : c)], b = bc[i];
    spec[i] = __fmul_rn(k, b);
    albedo[i] = __fmul_rn(__fsub_rn(1.0f, k), b);
  }
}

__global__ __launch_bounds__(256) void ks_split_bwd_kernel(const float* __restrict__ bc, const float* __restrict__ ks, const int kc, const long n,
                                                           const float* __restrict__ g_alb, const float* __restrict__ g_spec,
                                                           float* __restrict__ g_bc, float* __restrict__ g_ks) {
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long)gridDim.x * 256) {
    float gk[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float k = ks[p * kc + (kc == 1 ? 0 : c)], b = bc[3 * p + c];
      const float ga = g_alb != nullptr ? g_alb[3 * p + c] : 0.f, gs = g_spec != nullptr ? g_spec[3 * p + c] : 0.f;
      g_bc[3 * p + c] = __fadd_rn(__fmul_rn(ga, __fsub_rn(1.0f, k)), __fmul_rn(gs, k));
      gk[c] = __fsub_rn(__fmul_rn(gs, b), __fmul_rn(ga, b));     // d spec / d ks + d albedo / d ks = gs b - ga b
    }
    if (kc == 1) g_ks[p] = __fadd_rn(__fadd_rn(gk[0], gk[1]), gk[2]);
    else { g_ks[3 * p] = gk[0]; g_ks[3 * p + 1] = gk[1]; g_ks[3 * p + 2] = gk[2]; }
  }
}

// loss[i] = ((((rgb + vqrgb) + vqloss) [+ chr]) [+ smooth]) [+ sim]) [+ lambert] in the reference's own order (vq_nfr.py:906-981):
// terms [N, 5] = rgb, vqrgb, chromaticity, chr_smooth, lambert per point; vqloss / sim device scalars.
__global__ __launch_bounds__(256) void loss_total_kernel(const float* __restrict__ terms, const long n, const float* __restrict__ vqloss,
                                                         const float* __restrict__ sim, const int use_chr, const int use_smooth,
                                                         const int use_lambert, float* __restrict__ out) {
  const float vl = vqloss[0], sm = sim != nullptr ? sim[0] : 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float* t = terms + 5 * i;
    float l = __fadd_rn(__fadd_rn(t[0], t[1]), vl);
    if (use_chr) l = __fadd_rn(l, t[2]);
    if (use_smooth) l = __fadd_rn(l, t[3]);
    if (sim != nullptr) l = __fadd_rn(l, sm);
    if (use_lambert) l = __fadd_rn(l, t[4]);
    out[i] = l;
  }
}

long grid_for(long n) {
  long blocks = (n + 255) / 256;
  const long cap = (long)vqn_num_cus() * 16;
  return blocks > cap ? cap : (blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" int vqn_clip_preserve(const float* x, int64_t n, float lo, float hi, float* y, void* stream) {
  VQN_CHECK_ARG(n >= 0, "n >= 0");
  if (n == 0) return VQN_OK;
  VQN_CHECK_ARG(x && y, "null pointer");
  hipLaunchKernelGGL(clip_preserve_kernel, dim3((unsigned)grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (long)n, lo, hi, y);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_ks_split_fwd(const float* basecolor, const float* ks, int ks_channels, int64_t n, float* albedo, float* spec, void* stream) {
  VQN_CHECK_ARG(n >= 0 && (ks_channels == 1 || ks_channels == 3), "n >= 0, ks_channels 1 | 3");
  if (n == 0) return VQN_OK;
  VQN_CHECK_ARG(basecolor && ks && albedo && spec, "null pointer");
  hipLaunchKernelGGL(ks_split_fwd_kernel, dim3((unsigned)grid_for(3 * n)), dim3(256), 0, (hipStream_t)stream, basecolor, ks, ks_channels, (long)n,
                     albedo, spec);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_ks_split_bwd(const float* basecolor, const float* ks, int ks_channels, int64_t n, const float* g_albedo, const float* g_spec,
                                float* g_basecolor, float* g_ks, void* stream) {
  VQN_CHECK_ARG(n >= 0 && (ks_channels == 1 || ks_channels == 3), "n >= 0, ks_channels 1 | 3");
  if (n == 0) return VQN_OK;
  VQN_CHECK_ARG(basecolor && ks && g_basecolor && g_ks, "null pointer");
  hipLaunchKernelGGL(ks_split_bwd_kernel, dim3((unsigned)grid_for(n)), dim3(256), 0, (hipStream_t)stream, basecolor, ks, ks_channels, (long)n,
                     g_albedo, g_spec, g_basecolor, g_ks);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_loss_total(const float* terms, int64_t n, const float* vqloss, const float* sim, int use_chr, int use_smooth, int use_lambert,
                              float* out, void* stream) {
  VQN_CHECK_ARG(n >= 0, "n >= 0");
  if (n == 0) return VQN_OK;
  VQN_CHECK_ARG(terms && vqloss && out, "null pointer");
  hipLaunchKernelGGL(loss_total_kernel, dim3((unsigned)grid_for(n)), dim3(256), 0, (hipStream_t)stream, terms, (long)n, vqloss, sim, use_chr,
                     use_smooth, use_lambert, out);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
