// Per-example loss terms of the VQ reflectance stage in TRAIN mode, forward and backward, one launch each.
// Replaces the ~90 framework launches (and ~120 more in their autograd) of vq_nfr.Model.compute_loss
// (decomp/nerfvq_nfr3/nerfactor/models/vq_nfr.py:876-986, train branch) for the terms that are per surface point:
//     rgb          = combine_weight * mean_c (linear_gt - rgb_pred)^2                          (:906-908)
//     vqrgb        = mean_c (linear_gt - vq_rgb)^2                                               (:909-911)
//     chromaticity = w_chr * mean_c (chr(linear_gt) - chr(vq_rgb))^2,  chr(v) = v / |v| (0 where |v| = 0)   (:917-925, :869-874)
//     chr_smooth   = w_s * exp(-alpha e) (1 - <z_2j, z_2j+1>) for both rows of pair j,
//                    e = |chr(gt_2j) - chr(gt_2j+1)| if that exceeds chr_thres else 0  (sRGB targets)   (:927-953)
//     lambert      = w_l * max_c spec * r',  r' = 0 for rough < 0.5 else 2 rough - 1 (rough detached)      (:970-981)
// with linear_gt = srgb2linear(rgb_gt) for data_type 'nerf' (:896-901, util/img.py:166-186) else rgb_gt.  The scalar terms (vq
// commitment loss, code-separation term) stay with the caller.  Gradients follow TensorFlow's conventions where they differ from
// a naive chain rule: divide_no_nan and SqrtGrad give 0 (not NaN) at |v| = 0.
// One wave per PAIR of points: lanes 0 / 1 do the per-point scalar arithmetic, all 64 lanes the D-long dot product of the pair.
#include "common.h"

namespace {

struct LossArgs {
  const float* rgb_pred; const float* vq_rgb; const float* rgb_gt; const float* z; const float* spec; const float* rough;
  long N; int D; int nerf;
  float w_rgb, w_chr, w_smooth, chr_alpha, chr_thres, w_lambert;
};

__device__ __forceinline__ float srgb2linear(float x) {
  // (the reference's operation order, util/img.py:181: the coefficient is added before 1 is taken off, each step rounded to f32)
  return x <= 0.04045f ? x / 12.92f : powf(((x + 1.055f) - 1.0f) / 1.055f, 2.4f);
}

__device__ __forceinline__ void chroma(const float v[3], float c[3], float& norm) {
  norm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float inv = norm == 0.f ? 0.f : 1.f / norm;
  c[0] = v[0] * inv; c[1] = v[1] * inv; c[2] = v[2] * inv;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// the pair's smoothness weight w_s * exp(-alpha e) (sRGB targets) and dot product <z_a, z_b>; every lane returns both
__device__ __forceinline__ void pair_terms(const LossArgs& a, long i0, bool has_b, int lane, float& wexp, float& dot) {
  wexp = 0.f; dot = 0.f;
  if (a.w_smooth <= 0.f || a.z == nullptr || !has_b) return;
  float ga[3], gb[3], ca[3], cb[3], na, nb;
#pragma unroll
  for (int c = 0; c < 3; ++c) { ga[c] = a.rgb_gt[i0 * 3 + c]; gb[c] = a.rgb_gt[(i0 + 1) * 3 + c]; }
  chroma(ga, ca, na); chroma(gb, cb, nb);
  float e = sqrtf((ca[0] - cb[0]) * (ca[0] - cb[0]) + (ca[1] - cb[1]) * (ca[1] - cb[1]) + (ca[2] - cb[2]) * (ca[2] - cb[2]));
  e = e > a.chr_thres ? e : 0.f;
  wexp = a.w_smooth * expf(-a.chr_alpha * e);
  float s = 0.f;
  if (a.D == 256) {                      // (all eight loads in flight at once: with a run-time trip count the loop below waits for HBM four times in a row)
    float za[4], zb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { za[q] = a.z[i0 * 256 + lane + 64 * q]; zb[q] = a.z[(i0 + 1) * 256 + lane + 64 * q]; }
#pragma unroll
    for (int q = 0; q < 4; ++q) s = fmaf(za[q], zb[q], s);       // the same order as the loop
  } else {
    for (int d = lane; d < a.D; d += 64) s = fmaf(a.z[i0 * a.D + d], a.z[(i0 + 1) * a.D + d], s);
  }
  dot = wave_sum(s);
}

__global__ __launch_bounds__(256) void decomp_loss_fwd_kernel(const LossArgs a, float* __restrict__ terms /* [N,5] */) {
  const int lane = threadIdx.x & 63;
  const long n_pairs = (a.N + 1) >> 1;
  for (long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pair < n_pairs; pair += (long)gridDim.x * 4) {
    const long i0 = 2 * pair;
    const bool has_b = i0 + 1 < a.N;
    float wexp, dot;
    pair_terms(a, i0, has_b, lane, wexp, dot);
    if (lane < 2 && i0 + lane < a.N) {
      const long i = i0 + lane;
      float gt[3], lin[3], p[3], v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gt[c] = a.rgb_gt[i * 3 + c]; p[c] = a.rgb_pred[i * 3 + c]; v[c] = a.vq_rgb[i * 3 + c];
        lin[c] = a.nerf ? srgb2linear(gt[c]) : gt[c];
      }
      float t_rgb = 0.f, t_vq = 0.f, t_chr = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) { t_rgb += (lin[c] - p[c]) * (lin[c] - p[c]); t_vq += (lin[c] - v[c]) * (lin[c] - v[c]); }
      float cl[3], cv[3], nl, nv;
      chroma(lin, cl, nl); chroma(v, cv, nv);
#pragma unroll
      for (int c = 0; c < 3; ++c) t_chr += (cl[c] - cv[c]) * (cl[c] - cv[c]);
      float t_l = 0.f;
      if (a.w_lambert > 0.f && a.spec != nullptr) {
        const float r = a.rough[i], rp = r < 0.5f ? 0.f : 2.f * r - 1.f;
        t_l = a.w_lambert * fmaxf(a.spec[i * 3], fmaxf(a.spec[i * 3 + 1], a.spec[i * 3 + 2])) * rp;
      }
      float* o = terms + i * 5;
      o[0] = a.w_rgb * t_rgb * (1.f / 3.f);
      o[1] = t_vq * (1.f / 3.f);
      o[2] = a.w_chr * t_chr * (1.f / 3.f);
      o[3] = has_b ? wexp * (1.f - dot) : 0.f;
      o[4] = t_l;
    }
  }
}

// go [N,5]: upstream gradient of every term.  Writes d/d rgb_pred, d/d vq_rgb [N,3], d/d z [N,D] (if z), d/d spec [N,3] (if spec).
__global__ __launch_bounds__(256) void decomp_loss_bwd_kernel(const LossArgs a, const float* __restrict__ go, float* __restrict__ g_pred,
                                                              float* __restrict__ g_vq, float* __restrict__ g_z, float* __restrict__ g_spec) {
  const int lane = threadIdx.x & 63;
  const long n_pairs = (a.N + 1) >> 1;
  for (long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pair < n_pairs; pair += (long)gridDim.x * 4) {
    const long i0 = 2 * pair;
    const bool has_b = i0 + 1 < a.N;
    float wexp, dot;
    pair_terms(a, i0, has_b, lane, wexp, dot);
    if (g_z != nullptr) {
      // d (wexp (1 - <za, zb>)) / d za = -wexp zb, and the term sits in BOTH rows of the pair
      const float coef = has_b ? -wexp * (go[i0 * 5 + 3] + go[(i0 + 1) * 5 + 3]) : 0.f;
      if (a.D == 256 && has_b) {
        float za[4], zb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { za[q] = a.z[i0 * 256 + lane + 64 * q]; zb[q] = a.z[(i0 + 1) * 256 + lane + 64 * q]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) { g_z[i0 * 256 + lane + 64 * q] = coef * zb[q]; g_z[(i0 + 1) * 256 + lane + 64 * q] = coef * za[q]; }
      } else
      for (int d = lane; d < a.D; d += 64) {
        const float za = a.z[i0 * a.D + d], zb = has_b ? a.z[(i0 + 1) * a.D + d] : 0.f;
        g_z[i0 * a.D + d] = coef * zb;
        if (has_b) g_z[(i0 + 1) * a.D + d] = coef * za;
      }
    }
    if (lane < 2 && i0 + lane < a.N) {
      const long i = i0 + lane;
      const float* g = go + i * 5;
      float lin[3], p[3], v[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float gt = a.rgb_gt[i * 3 + c];
        p[c] = a.rgb_pred[i * 3 + c]; v[c] = a.vq_rgb[i * 3 + c];
        lin[c] = a.nerf ? srgb2linear(gt) : gt;
      }
      float cl[3], cv[3], nl, nv;
      chroma(lin, cl, nl); chroma(v, cv, nv);
      // chromaticity: t = w/3 sum (cv - cl)^2, cv = v / |v|: d t / d v = (I - cv cv^T) (2 w / 3)(cv - cl) / |v|   (0 at |v| = 0)
      float du[3], proj = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) { du[c] = a.w_chr * (2.f / 3.f) * (cv[c] - cl[c]); proj += du[c] * cv[c]; }
      const float inv = nv == 0.f ? 0.f : 1.f / nv;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        g_pred[i * 3 + c] = g[0] * a.w_rgb * (2.f / 3.f) * (p[c] - lin[c]);
        g_vq[i * 3 + c] = g[1] * (2.f / 3.f) * (v[c] - lin[c]) + g[2] * (du[c] - proj * cv[c]) * inv;
      }
      if (g_spec != nullptr) {
        const float r = a.rough[i], rp = r < 0.5f ? 0.f : 2.f * r - 1.f;
        const float s0 = a.spec[i * 3], s1 = a.spec[i * 3 + 1], s2 = a.spec[i * 3 + 2];
        const int am = (s0 >= s1 && s0 >= s2) ? 0 : (s1 >= s2 ? 1 : 2);          // first maximum, as torch.max(dim) reports it
#pragma unroll
        for (int c = 0; c < 3; ++c) g_spec[i * 3 + c] = (c == am) ? g[4] * a.w_lambert * rp : 0.f;
      }
    }
  }
}

int check(const LossArgs& a) {
  if (a.N < 0 || a.D < 0) return 1;
  if (a.N > 0 && (!a.rgb_pred || !a.vq_rgb || !a.rgb_gt)) return 2;
  if (a.w_lambert > 0.f && a.spec != nullptr && a.rough == nullptr) return 3;
  return 0;
}

unsigned grid_for(long N) {
  long blocks = ((N + 1) / 2 + 3) / 4;
  const long cap = (long)vqn_num_cus() * 16;
  if (blocks > cap) blocks = cap;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" int vqn_decomp_loss_fwd(const float* rgb_pred, const float* vq_rgb, const float* rgb_gt, const float* z, const float* spec,
                                   const float* rough, int64_t N, int D, int nerf, float w_rgb, float w_chr, float w_smooth,
                                   float chr_alpha, float chr_thres, float w_lambert, float* terms, void* stream) {
  const LossArgs a{rgb_pred, vq_rgb, rgb_gt, z, spec, rough, (long)N, D, nerf, w_rgb, w_chr, w_smooth, chr_alpha, chr_thres, w_lambert};
  VQN_CHECK_ARG(check(a) == 0 && (N == 0 || terms != nullptr), "null pointer or negative size");
  if (N == 0) return VQN_OK;
  hipLaunchKernelGGL(decomp_loss_fwd_kernel, dim3(grid_for(N)), dim3(256), 0, (hipStream_t)stream, a, terms);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_decomp_loss_bwd(const float* rgb_pred, const float* vq_rgb, const float* rgb_gt, const float* z, const float* spec,
                                   const float* rough, int64_t N, int D, int nerf, float w_rgb, float w_chr, float w_smooth,
                                   float chr_alpha, float chr_thres, float w_lambert, const float* g_terms, float* g_rgb_pred,
                                   float* g_vq_rgb, float* g_z, float* g_spec, void* stream) {
  const LossArgs a{rgb_pred, vq_rgb, rgb_gt, z, spec, rough, (long)N, D, nerf, w_rgb, w_chr, w_smooth, chr_alpha, chr_thres, w_lambert};
  VQN_CHECK_ARG(check(a) == 0 && (N == 0 || (g_terms && g_rgb_pred && g_vq_rgb)), "null pointer or negative size");
  VQN_CHECK_ARG(g_z == nullptr || z != nullptr, "g_z without z");
  VQN_CHECK_ARG(g_spec == nullptr || spec != nullptr, "g_spec without spec");
  if (N == 0) return VQN_OK;
  hipLaunchKernelGGL(decomp_loss_bwd_kernel, dim3(grid_for(N)), dim3(256), 0, (hipStream_t)stream, a, g_terms, g_rgb_pred, g_vq_rgb, g_z,
                     g_spec);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

// ---- the EMA codebook move of VectorQuantizerEMA in training (vq_layers.py:304-325 + Sonnet's ExponentialMovingAverage, twice) as ONE
// launch instead of ~35 framework launches on [K] / [D, K] tensors:
//     for both averages:  counter += 1;  hidden -= (hidden - value) (1 - decay);  average = hidden / (1 - decay^counter)   (f64 debias)
//     n = sum_k cs_k;  cs'_k = (cs_k + eps) / (n + K eps) n;  w = dw_avg / cs';  update = used ? w : codebook   (used: counts_k > 0)
// One workgroup; the K-long sum in index order (bit-reproducible).
namespace {
__global__ __launch_bounds__(256) void vq_ema_update_kernel(const float* __restrict__ counts, const float* __restrict__ dw,
                                                            const float* __restrict__ cb, const int D, const int K, const double decay,
                                                            const float eps, float* __restrict__ hid_cs, float* __restrict__ avg_cs,
                                                            long long* __restrict__ cnt_cs, float* __restrict__ hid_dw,
                                                            float* __restrict__ avg_dw, long long* __restrict__ cnt_dw,
                                                            float* __restrict__ update) {
  __shared__ float cs_adj[1024];
  __shared__ double deb[2];
  const int tid = threadIdx.x;
  if (tid == 0) {
    const long long c0 = cnt_cs[0] + 1, c1 = cnt_dw[0] + 1;
    cnt_cs[0] = c0; cnt_dw[0] = c1;
    deb[0] = 1.0 - pow(decay, (double)c0);
    deb[1] = 1.0 - pow(decay, (double)c1);
  }
  __syncthreads();
  const float om = (float)(1.0 - decay);        // (1 - decay) taken in double, then rounded: what the framework statement multiplies by
  for (int k = tid; k < K; k += 256) {
    const float hdn = __fsub_rn(hid_cs[k], __fmul_rn(__fsub_rn(hid_cs[k], counts[k]), om));      // (each step rounded as the framework
                                                                                               //  statement rounds it: no contraction into an fma)
    hid_cs[k] = hdn;
    const float a = (float)((double)hdn / deb[0]);
    avg_cs[k] = a;
    cs_adj[k] = a;
  }
  __syncthreads();
  if (tid == 0) {
    float n = 0.f;
    for (int k = 0; k < K; ++k) n += cs_adj[k];
    deb[0] = (double)n;                          // (reuse: the total)
  }
  __syncthreads();
  const float n = (float)deb[0];
  for (int k = tid; k < K; k += 256) cs_adj[k] = (cs_adj[k] + eps) / (n + (float)K * eps) * n;
  __syncthreads();
  const double d1 = deb[1];
  for (int i = tid; i < D * K; i += 256) {
    const int k = i % K;
    const float hdn = __fsub_rn(hid_dw[i], __fmul_rn(__fsub_rn(hid_dw[i], dw[i]), om));
    hid_dw[i] = hdn;
    const float a = (float)((double)hdn / d1);
    avg_dw[i] = a;
    update[i] = counts[k] > 0.f ? a / cs_adj[k] : cb[i];
  }
}
}  // namespace

extern "C" int vqn_vq_ema_update(const float* counts, const float* dw, const float* codebook, int D, int K, double decay, float eps,
                                 float* hidden_cs, float* average_cs, int64_t* counter_cs, float* hidden_dw, float* average_dw,
                                 int64_t* counter_dw, float* update, void* stream) {
  VQN_CHECK_ARG(counts && dw && codebook && hidden_cs && average_cs && counter_cs && hidden_dw && average_dw && counter_dw && update,
                "null pointer");
  VQN_CHECK_SHAPE(D >= 1 && K >= 1 && K <= 1024, "1 <= K <= 1024");
  hipLaunchKernelGGL(vq_ema_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, counts, dw, codebook, D, K, decay, eps, hidden_cs,
                     average_cs, (long long*)counter_cs, hidden_dw, average_dw, (long long*)counter_dw, update);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

// ---- backward of the two row-wise steps around the quantiser (networks/vq_layers.py of the reference: the l2-normalised encoder output,
// :302 / :327 the straight-through estimator + commitment loss), one pass over [N, D] each instead of 8 + 3 framework passes ----------
namespace {

// y = x s, s = max(sum x^2, eps)^(-1/2)  ->  gx = g s - [sum x^2 > eps] x s^3 (x . g).  One wave per row, D <= 1024, D % 4 == 0.
__global__ __launch_bounds__(256) void l2_normalize_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, long N, int D,
                                                                    float eps, float* __restrict__ gx) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * D);
  const f32x4* gr = reinterpret_cast<const f32x4*>(g + row * D);
  f32x4* or_ = reinterpret_cast<f32x4*>(gx + row * D);
  const int n4 = D >> 2;
  f32x4 xv[4], gv[4];
  float x2 = 0.f, xg = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane + 64 * k;
    if (i < n4) {
      xv[k] = xr[i]; gv[k] = gr[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) { x2 = fmaf(xv[k][j], xv[k][j], x2); xg = fmaf(xv[k][j], gv[k][j], xg); }
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { x2 += __shfl_xor(x2, m); xg += __shfl_xor(xg, m); }
  const float s = 1.0f / sqrtf(fmaxf(x2, eps));
  const float t = x2 > eps ? (s * s * s) * xg : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane + 64 * k;
    if (i < n4) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = gv[k][j] * s - xv[k][j] * t;
      or_[i] = o;
    }
  }
}

// g = g_ste + (x - q) (g_loss 2 / numel)   (g_ste may be absent; g_loss a device scalar) -- separately rounded, as the framework ops were
__global__ __launch_bounds__(256) void ste_commit_bwd_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ q, const f32x4* __restrict__ g_ste,
                                                             const float* __restrict__ g_loss, float two_over_numel, long n4, f32x4* __restrict__ out) {
  const float t = __fmul_rn(g_loss[0], two_over_numel);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 a = x[i], b = q[i];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __fmul_rn(__fsub_rn(a[j], b[j]), t);
    if (g_ste != nullptr) {
      const f32x4 c = g_ste[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = __fadd_rn(c[j], o[j]);
    }
    out[i] = o;
  }
}

// the two above in one pass per row (round 4; the training quantiser's backward): g_xn = g_ste + (xn - q) (g_loss 2 / numel), then the
// l2-normalise backward of x at g_xn.  One wave per row, D <= 1024, D % 4 == 0; same roundings as the two-kernel sequence.
__global__ __launch_bounds__(256) void vq_train_bwd_kernel(const float* __restrict__ x, const float* __restrict__ xn, const float* __restrict__ q,
                                                           const float* __restrict__ g_ste, const float* __restrict__ g_loss, float two_over_numel,
                                                           float loss_post, long N, int D, float eps, float* __restrict__ gx) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  // (loss_post: the forward's second factor -- the commitment cost -- applied to the incoming adjoint as autograd's own multiplication would)
  const float t_c = __fmul_rn(__fmul_rn(g_loss[0], loss_post), two_over_numel);
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * D);
  const f32x4* nr = reinterpret_cast<const f32x4*>(xn + row * D);
  const f32x4* qr = reinterpret_cast<const f32x4*>(q + row * D);
  const f32x4* gr = g_ste != nullptr ? reinterpret_cast<const f32x4*>(g_ste + row * D) : nullptr;
  f32x4* or_ = reinterpret_cast<f32x4*>(gx + row * D);
  const int n4 = D >> 2;
  f32x4 xv[4], gv[4];
  float x2 = 0.f, xg = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane + 64 * k;
    if (i < n4) {
      xv[k] = xr[i];
      const f32x4 a = nr[i], b = qr[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) gv[k][j] = __fmul_rn(__fsub_rn(a[j], b[j]), t_c);
      if (gr != nullptr) {
        const f32x4 c = gr[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) gv[k][j] = __fadd_rn(c[j], gv[k][j]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { x2 = fmaf(xv[k][j], xv[k][j], x2); xg = fmaf(xv[k][j], gv[k][j], xg); }
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { x2 += __shfl_xor(x2, m); xg += __shfl_xor(xg, m); }
  const float s = 1.0f / sqrtf(fmaxf(x2, eps));
  const float t = x2 > eps ? (s * s * s) * xg : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = lane + 64 * k;
    if (i < n4) {
      f32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = gv[k][j] * s - xv[k][j] * t;
      or_[i] = o;
    }
  }
}

}  // namespace

extern "C" int vqn_vq_train_bwd(const float* z, const float* xnorm, const float* quant, const float* g_ste, const float* g_loss, float loss_post,
                                int64_t N, int D, float eps, float* g_z, void* stream) {
  VQN_CHECK_ARG(N >= 0 && D > 0, "N >= 0, D > 0");
  if (N == 0) return VQN_OK;
  VQN_CHECK_ARG(z && xnorm && quant && g_loss && g_z, "null pointer");
  VQN_CHECK_SHAPE(D % 4 == 0 && D <= 1024, "D a multiple of 4, at most 1024");
  hipLaunchKernelGGL(vq_train_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, z, xnorm, quant, g_ste, g_loss,
                     (float)(2.0 / ((double)N * D)), loss_post, (long)N, D, eps, g_z);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_l2_normalize_rows_bwd(const float* x, const float* g, int64_t N, int D, float eps, float* gx, void* stream) {
  VQN_CHECK_ARG(N >= 0 && D > 0, "N >= 0, D > 0");
  if (N == 0) return VQN_OK;
  VQN_CHECK_ARG(x && g && gx, "null pointer");
  VQN_CHECK_SHAPE(D % 4 == 0 && D <= 1024, "D a multiple of 4, at most 1024");
  hipLaunchKernelGGL(l2_normalize_rows_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, g, (long)N, D, eps, gx);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_vq_ste_loss_bwd(const float* x, const float* quant, const float* g_ste, const float* g_loss, int64_t numel, float* gx,
                                   void* stream) {
  VQN_CHECK_ARG(numel >= 0, "numel >= 0");
  if (numel == 0) return VQN_OK;
  VQN_CHECK_ARG(x && quant && g_loss && gx, "null pointer");
  VQN_CHECK_SHAPE(numel % 4 == 0, "numel a multiple of 4");
  const long n4 = numel / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 256L * 16) blocks = 256L * 16;
  hipLaunchKernelGGL(ste_commit_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(x),
                     reinterpret_cast<const f32x4*>(quant), reinterpret_cast<const f32x4*>(g_ste), g_loss, (float)(2.0 / (double)numel), n4,
                     reinterpret_cast<f32x4*>(gx));
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

// ---- the codebook as the model uses it (vq_nfr.py: clip with identity gradient to [0, 1], l2-normalise every code) and the
// code-separation term (vq_nfr.py:955-968), each one launch forward and one backward instead of ~8 + ~15 framework kernels and their
// autograd: [D, K] with K <= 64 codes is a few KB ----------------------------------------------------------------------------------
namespace {

// y[:, k] = c s_k, c = x + (clamp(x, 0, 1) - x), s_k = max(sum_d c^2, eps)^(-1/2); one workgroup per code.  BWD: g -> gx.
template <bool BWD>
__global__ __launch_bounds__(256) void codebook_prep_kernel(const float* __restrict__ x, const float* __restrict__ gy, int D, int K, float eps,
                                                            float* __restrict__ out) {
  __shared__ float red[2][256];
  const int k = blockIdx.x, t = threadIdx.x;
  float s2 = 0.f, cg = 0.f;
  for (int d = t; d < D; d += 256) {
    const float v = x[(size_t)d * K + k];
    const float c = __fadd_rn(v, __fsub_rn(fminf(fmaxf(v, 0.f), 1.f), v));
    s2 = fmaf(c, c, s2);
    if (BWD) cg = fmaf(c, gy[(size_t)d * K + k], cg);
  }
  red[0][t] = s2; red[1][t] = cg;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) { red[0][t] += red[0][t + w]; red[1][t] += red[1][t + w]; }
    __syncthreads();
  }
  s2 = red[0][0]; cg = red[1][0];
  const float s = 1.0f / sqrtf(fmaxf(s2, eps));
  const float tt = s2 > eps ? (s * s * s) * cg : 0.f;
  for (int d = t; d < D; d += 256) {
    const float v = x[(size_t)d * K + k];
    const float c = __fadd_rn(v, __fsub_rn(fminf(fmaxf(v, 0.f), 1.f), v));
    out[(size_t)d * K + k] = BWD ? gy[(size_t)d * K + k] * s - c * tt : c * s;
  }
}

// out[0] = -w log(min_{i != j} |c_i - c_j|), out[1] = the min, out[2..3] = its pair (i < j) as floats.  One workgroup; cb [D, K] (codes = columns).
// (round 4: the codebook is staged in LDS first when it fits -- every thread walked its two columns through 256 dependent global
//  loads, 65 us of the captured 2048-point step's 760; same fmaf chain over d, the same value bit for bit)
constexpr int SIM_LDS_FLOATS = 12288;        // 48 KB: K <= 48 at D = 256
// (round 5: 33 -> ~8 us at K = 15 -- as long as the encoder + heads forward of the captured 2048-point step.  Two things took the time:
//  each pair's chain of D dependent fmaf waited for its two LDS operands one step at a time (now 16 steps' operands are read ahead
//  of the chain -- the chain itself, and so every bit of the result, is unchanged), and thread 0 folded the 256 candidates serially
//  from LDS (now a tree over the same total order -- smaller distance, then smaller pair index: associative, so the same pair bit for
//  bit; a NaN distance still beats everything and makes out[0] NaN, but WHICH of several NaN pairs is reported is no longer the
//  serial fold's last one -- the caller's numerics guard raises on the NaN loss either way).)
__device__ __forceinline__ bool sim_later_wins(float da, int ia, float db, int ib) {     // candidate b (later in thread order) against a
  return db < da || (db == da && ib < ia) || !(db == db);
}

__global__ __launch_bounds__(256) void sim_smooth_fwd_kernel(const float* __restrict__ cb, int D, int K, float w, float* __restrict__ out) {
  __shared__ float best[256];
  __shared__ int bi[256];
  __shared__ float cbs[SIM_LDS_FLOATS];
  const int t = threadIdx.x;
  const bool staged = D * K <= SIM_LDS_FLOATS;
  if (staged) {
    const int n = D * K;
    if ((n & 3) == 0 && (((uintptr_t)cb) & 15) == 0) {
      const float4* __restrict__ c4 = reinterpret_cast<const float4*>(cb);
      float4* s4 = reinterpret_cast<float4*>(cbs);
      for (int e = t; e < (n >> 2); e += 256) s4[e] = c4[e];
    } else {
      for (int e = t; e < n; e += 256) cbs[e] = cb[e];
    }
    __syncthreads();
  }
  const float* __restrict__ src = staged ? cbs : cb;
  float bd = INFINITY;
  int bp = 0;
  for (int pr = t; pr < K * K; pr += 256) {
    const int i = pr / K, j = pr - i * K;
    if (i >= j) continue;
    float s = 0.f;
    if (staged) {
      int d = 0;
      for (; d + 16 <= D; d += 16) {
        float a[16], b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { a[u] = cbs[(d + u) * K + i]; b[u] = cbs[(d + u) * K + j]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) { const float dl = a[u] - b[u]; s = fmaf(dl, dl, s); }
      }
      for (; d < D; ++d) { const float dl = cbs[d * K + i] - cbs[d * K + j]; s = fmaf(dl, dl, s); }
    } else {
      for (int d = 0; d < D; ++d) { const float dl = src[(size_t)d * K + i] - src[(size_t)d * K + j]; s = fmaf(dl, dl, s); }
    }
    const float dist = sqrtf(s);
    if (dist < bd || !(dist == dist)) { bd = dist; bp = pr; }
  }
  best[t] = bd; bi[t] = bp;
  __syncthreads();
  // (threads without a pair hold (inf, 0): the serial fold never took them either -- inf is not smaller, and on an all-inf tie pair 0 stays)
  for (int half = 128; half > 0; half >>= 1) {
    if (t < half) {
      const float da = best[t], db = best[t + half];
      const int ia = bi[t], ib = bi[t + half];
      if (sim_later_wins(da, ia, db, ib)) { best[t] = db; bi[t] = ib; }
    }
    __syncthreads();
  }
  if (t == 0) {
    bd = best[0]; bp = bi[0];
    out[0] = w * (-logf(bd));
    out[1] = bd;
    out[2] = (float)(bp / K);
    out[3] = (float)(bp % K);
  }
}

// g_cb = g_loss * d(-w log dmin)/d cb: -w (c_i - c_j) / dmin^2 into column i, the negative into column j, zeros elsewhere
__global__ __launch_bounds__(256) void sim_smooth_bwd_kernel(const float* __restrict__ cb, const float* __restrict__ fwd, const float* __restrict__ g_loss,
                                                             int D, int K, float w, float* __restrict__ g_cb) {
  const float dmin = fwd[1];
  const int i = (int)fwd[2], j = (int)fwd[3];
  const float f = -w * g_loss[0] / (dmin * dmin);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)D * K; e += (long)gridDim.x * 256) {
    const int d = (int)(e / K), k = (int)(e - (long)d * K);
    float v = 0.f;
    if (k == i || k == j) {
      const float dl = cb[(size_t)d * K + i] - cb[(size_t)d * K + j];
      v = k == i ? f * dl : -f * dl;
    }
    g_cb[e] = v;
  }
}

}  // namespace

extern "C" int vqn_codebook_prep(const float* raw, const float* g, int D, int K, float eps, float* out, void* stream) {
  VQN_CHECK_ARG(raw && out, "null pointer");
  VQN_CHECK_SHAPE(D >= 1 && K >= 1 && K <= 65535, "D >= 1, 1 <= K");
  if (g == nullptr) hipLaunchKernelGGL(codebook_prep_kernel<false>, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, raw, g, D, K, eps, out);
  else hipLaunchKernelGGL(codebook_prep_kernel<true>, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, raw, g, D, K, eps, out);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_sim_smooth_fwd(const float* codebook, int D, int K, float weight, float* out4, void* stream) {
  VQN_CHECK_ARG(codebook && out4, "null pointer");
  VQN_CHECK_SHAPE(D >= 1 && K >= 2 && K <= 256, "D >= 1, 2 <= K <= 256");
  hipLaunchKernelGGL(sim_smooth_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, codebook, D, K, weight, out4);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_sim_smooth_bwd(const float* codebook, const float* fwd4, const float* g_loss, int D, int K, float weight, float* g_codebook,
                                  void* stream) {
  VQN_CHECK_ARG(codebook && fwd4 && g_loss && g_codebook, "null pointer");
  VQN_CHECK_SHAPE(D >= 1 && K >= 2 && K <= 256, "D >= 1, 2 <= K <= 256");
  long blocks = ((long)D * K + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sim_smooth_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, codebook, fwd4, g_loss, D, K, weight, g_codebook);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
