// Per-ray NeuS kernels for gfx950: one 64-lane wave per ray, scans by wave shuffles, per-ray arrays
// staged in LDS.  All HBM-bound (a few KB per ray).  Replace the ~100 tiny framework launches of
//   geo/NeuS-ours2/models/renderer.py:131-175 (up_sample) + :39-69 (sample_pdf, det=True)  -> vqn_neus_upsample
//   geo/NeuS-ours2/models/renderer.py:177-191 (cat_z_vals: cat + sort + gather)            -> vqn_neus_merge
//   geo/NeuS-ours2/models/renderer.py:209-213 (dists, mid_z_vals)                           -> vqn_neus_section_mids
//   geo/NeuS-ours2/models/renderer.py:229-282 (alpha, transmittance, colour/surf/depth/eikonal) -> vqn_neus_composite_fwd
//   the reverse of the latter (autograd in the reference)                                     -> vqn_neus_composite_bwd
#include "common.h"
#include <math.h>

namespace {

constexpr int MAXN = 256;          // samples per ray supported (reference: <= 128)
constexpr int IT = MAXN / 64;      // items per lane, blocked layout: element i lives in lane i / IT

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}
// exclusive scans over lanes
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(inc, d);
    if (lane >= d) inc *= o;
  }
  float ex = __shfl_up(inc, 1);
  return lane == 0 ? 1.f : ex;
}
__device__ __forceinline__ float wave_excl_sum(float v, int lane) {
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  float ex = __shfl_up(inc, 1);
  return lane == 0 ? 0.f : ex;
}

struct RayLds {
  float z[MAXN], sdf[MAXN], rad[MAXN], cdf[MAXN];
};

__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ zv, const float* __restrict__ sdfv,
                                                       long B, int n, float r_limit, float inv_s,
                                                       const float* __restrict__ u, int m, float* __restrict__ z_new) {
  __shared__ RayLds sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  RayLds& L = sh[wave];
  const long n_groups = (B + 3) >> 2;
  for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long ray_raw = grp * 4 + wave;
    const bool live = ray_raw < B;
    const long ray = live ? ray_raw : B - 1;
    const float ox = rays_o[ray * 3], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
    const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    for (int s = lane; s < n; s += 64) {
      const float z = zv[ray * n + s];
      L.z[s] = z;
      L.sdf[s] = sdfv[ray * n + s];
      const float px = ox + __fmul_rn(dx, z), py = oy + __fmul_rn(dy, z), pz = oz + __fmul_rn(dz, z);
      L.rad[s] = sqrtf(__fmul_rn(px, px) + __fmul_rn(py, py) + __fmul_rn(pz, pz));
    }
    __syncthreads();
    // ---- interval quantities (renderer.py:138-172), intervals i = lane*IT + k ----
    float alpha[IT], w[IT];
    float lprod = 1.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int i = lane * IT + k;
      alpha[k] = 0.f;
      if (i < n - 1) {
        const float z0 = L.z[i], z1 = L.z[i + 1], s0 = L.sdf[i], s1 = L.sdf[i + 1];
        const bool inside = (L.rad[i] < r_limit) || (L.rad[i + 1] < r_limit);
        const float cosv = (s1 - s0) / (z1 - z0 + 1e-5f);
        float prev = 0.f;
        if (i > 0) prev = (s0 - L.sdf[i - 1]) / (z0 - L.z[i - 1] + 1e-5f);
        float c = fminf(prev, cosv);
        c = fminf(fmaxf(c, -1e3f), 0.f) * (inside ? 1.f : 0.f);
        const float dist = z1 - z0, mid = (s0 + s1) * 0.5f;
        const float pe = mid - c * dist * 0.5f, ne = mid + c * dist * 0.5f;
        const float pc = sigmoidf_(pe * inv_s), nc = sigmoidf_(ne * inv_s);
        alpha[k] = (pc - nc + 1e-5f) / (pc + 1e-5f);
        lprod *= (1.f - alpha[k] + 1e-7f);
      }
    }
    float T = wave_excl_prod(lprod, lane);
    float lsum = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int i = lane * IT + k;
      w[k] = 0.f;
      if (i < n - 1) {
        w[k] = alpha[k] * T + 1e-5f;           // "+1e-5" is sample_pdf's (renderer.py:42)
        T *= (1.f - alpha[k] + 1e-7f);
        lsum += w[k];
      }
    }
    const float total = wave_sum(lsum);
    float lc = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) { w[k] = w[k] / total; lc += w[k]; }
    float run = wave_excl_sum(lc, lane);
    if (lane == 0) L.cdf[0] = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int i = lane * IT + k;
      if (i < n - 1) { run += w[k]; L.cdf[i + 1] = run; }
    }
    __syncthreads();
    // ---- inverse CDF at the deterministic u's (renderer.py:55-67) ----
    for (int j = lane; j < m; j += 64) {
      const float uj = u[j];
      int lo = 0, hi = n;                      // first index with cdf > u  (searchsorted right=True)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (L.cdf[mid] <= uj) lo = mid + 1; else hi = mid;
      }
      const int below = max(0, lo - 1), above = min(n - 1, lo);
      const float cb = L.cdf[below], ca = L.cdf[above];
      float denom = ca - cb;
      if (denom < 1e-5f) denom = 1.f;
      const float t = (uj - cb) / denom;
      const float bb = L.z[below], ba = L.z[above];
      if (live) z_new[ray * m + j] = bb + t * (ba - bb);
    }
    __syncthreads();
  }
}

// stable merge of the sorted old samples with the new ones (old first on ties); sdf follows z
__global__ __launch_bounds__(256) void merge_kernel(const float* __restrict__ zv, const float* __restrict__ sdfv,
                                                    const float* __restrict__ znew, const float* __restrict__ sdfnew,
                                                    long B, int n, int m, float* __restrict__ z_out,
                                                    float* __restrict__ sdf_out) {
  __shared__ float shz[4][MAXN], shn[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long n_groups = (B + 3) >> 2;
  for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long ray_raw = grp * 4 + wave;
    const bool live = ray_raw < B;
    const long ray = live ? ray_raw : B - 1;
    for (int s = lane; s < n; s += 64) shz[wave][s] = zv[ray * n + s];
    if (lane < m) shn[wave][lane] = znew[ray * m + lane];
    __syncthreads();
    const int tot = n + m;
    for (int s = lane; s < n; s += 64) {
      const float z = shz[wave][s];
      int cnt = 0;
      for (int j = 0; j < m; ++j) cnt += (shn[wave][j] < z) ? 1 : 0;
      if (live) {
        z_out[ray * tot + s + cnt] = z;
        if (sdf_out) sdf_out[ray * tot + s + cnt] = sdfv[ray * n + s];
      }
    }
    if (lane < m) {
      const float z = shn[wave][lane];
      int lo = 0, hi = n;                      // #old <= z
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (shz[wave][mid] <= z) lo = mid + 1; else hi = mid;
      }
      int cnt = lo;
      for (int j = 0; j < m; ++j) {
        const float zj = shn[wave][j];
        cnt += (zj < z || (zj == z && j < lane)) ? 1 : 0;
      }
      if (live) {
        z_out[ray * tot + cnt] = z;
        if (sdf_out) sdf_out[ray * tot + cnt] = sdfnew[ray * m + lane];
      }
    }
    __syncthreads();
  }
}

__global__ void section_mids_kernel(const float* __restrict__ zv, long B, int n, float sample_dist,
                                    const float* __restrict__ sample_dist_per_ray, float* __restrict__ mid,
                                    float* __restrict__ dists) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n) return;
  const long ray = i / n;
  const int s = (int)(i - ray * n);
  const float z = zv[i];
  const float d = (s + 1 < n) ? (zv[i + 1] - z) : (sample_dist_per_ray ? sample_dist_per_ray[ray] : sample_dist);
  if (dists) dists[i] = d;
  mid[i] = z + d * 0.5f;
}

struct CompArgs {
  const float *rays_o, *rays_d, *mid_z, *dists, *sdf, *grad, *rgb, *inv_s, *bg;
  long B;
  int n;
  float radius, car;
  float *color, *weights, *cdf, *inside, *surf, *depth, *wsum, *wmax, *gerr;   // gerr [B,2] = (num, den)
  float* alpha;                                                                   // optional [B,n] (for backward)
};

__global__ __launch_bounds__(256) void composite_fwd_kernel(const CompArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long n_groups = (a.B + 3) >> 2;
  const int n = a.n;
  const float inv_s = fminf(fmaxf(*a.inv_s, 1e-6f), 1e6f);
  for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long ray = grp * 4 + wave;
    if (ray >= a.B) continue;                                   // no block-level sync below
    const float ox = a.rays_o[ray * 3], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
    const float dx = a.rays_d[ray * 3], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
    float alpha[IT], px[IT], py[IT], pz[IT], cr[IT], cg[IT], cb[IT];
    float lprod = 1.f, g_num = 0.f, g_den = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int i = lane * IT + k;
      alpha[k] = 0.f; px[k] = py[k] = pz[k] = cr[k] = cg[k] = cb[k] = 0.f;
      if (i < n) {
        const long q = ray * n + i;
        const float mz = a.mid_z[q], dist = a.dists[q], sdf = a.sdf[q];
        const float gx = a.grad[q * 3], gy = a.grad[q * 3 + 1], gz = a.grad[q * 3 + 2];
        cr[k] = a.rgb[q * 3]; cg[k] = a.rgb[q * 3 + 1]; cb[k] = a.rgb[q * 3 + 2];
        px[k] = ox + __fmul_rn(dx, mz); py[k] = oy + __fmul_rn(dy, mz); pz[k] = oz + __fmul_rn(dz, mz);
        const float tc = __fmul_rn(dx, gx) + __fmul_rn(dy, gy) + __fmul_rn(dz, gz);
        const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.f) * (1.f - a.car) + fmaxf(-tc, 0.f) * a.car);
        const float en = sdf + ic * dist * 0.5f, ep = sdf - ic * dist * 0.5f;
        const float pc = sigmoidf_(ep * inv_s), nc = sigmoidf_(en * inv_s);
        const float al = fminf(fmaxf((pc - nc + 1e-5f) / (pc + 1e-5f), 0.f), 1.f);
        alpha[k] = al;
        lprod *= (1.f - al + 1e-7f);
        const float pr = sqrtf(__fmul_rn(px[k], px[k]) + __fmul_rn(py[k], py[k]) + __fmul_rn(pz[k], pz[k]));
        a.cdf[q] = pc;
        a.inside[q] = pr < a.radius ? 1.f : 0.f;
        if (a.alpha) a.alpha[q] = al;
        const float relax = pr < a.radius * 1.1f ? 1.f : 0.f;
        const float gn = sqrtf(__fmul_rn(gx, gx) + __fmul_rn(gy, gy) + __fmul_rn(gz, gz)) - 1.f;
        g_num += relax * gn * gn;
        g_den += relax;
      }
    }
    float T = wave_excl_prod(lprod, lane);
    float sr = 0.f, sg = 0.f, sb = 0.f, sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f, mw = -INFINITY;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int i = lane * IT + k;
      if (i < n) {
        const float w = alpha[k] * T;
        T *= (1.f - alpha[k] + 1e-7f);
        a.weights[ray * n + i] = w;
        sr += cr[k] * w; sg += cg[k] * w; sb += cb[k] * w;
        sx += px[k] * w; sy += py[k] * w; sz += pz[k] * w;
        sw += w; mw = fmaxf(mw, w);
      }
    }
    sr = wave_sum(sr); sg = wave_sum(sg); sb = wave_sum(sb);
    sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
    sw = wave_sum(sw); mw = wave_max(mw);
    g_num = wave_sum(g_num); g_den = wave_sum(g_den);
    if (lane == 0) {
      if (a.bg) { sr += a.bg[0] * (1.f - sw); sg += a.bg[1] * (1.f - sw); sb += a.bg[2] * (1.f - sw); }
      a.color[ray * 3] = sr; a.color[ray * 3 + 1] = sg; a.color[ray * 3 + 2] = sb;
      a.surf[ray * 3] = sx; a.surf[ray * 3 + 1] = sy; a.surf[ray * 3 + 2] = sz;
      const float ex = sx - ox, ey = sy - oy, ez = sz - oz;
      a.depth[ray] = sqrtf(ex * ex + ey * ey + ez * ez);
      a.wsum[ray] = sw; a.wmax[ray] = mw;
      a.gerr[ray * 2] = g_num; a.gerr[ray * 2 + 1] = g_den;
    }
  }
}

// exclusive suffix sum over lanes: result(lane) = sum_{l > lane} v(l)
__device__ __forceinline__ float wave_excl_suffix_sum(float v, int lane) {
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    float o = __shfl_down(inc, d);
    if (lane + d < 64) inc += o;
  }
  float ex = __shfl_down(inc, 1);
  return lane == 63 ? 0.f : ex;
}

struct CompBwdArgs {
  const float *rays_d, *mid_z, *dists, *sdf, *grad, *rgb, *inv_s, *bg;
  const float *g_color, *g_wsum, *g_weights;   // [B,3], [B] or null, [B,n] or null
  const float* g_gerr;                         // device scalar: d loss / d gradient_error (or null)
  const float* gerr_den;                       // device scalar: sum over rays of gerr[:,1]
  const float *rays_o;
  long B;
  int n;
  float radius, car;
  float *g_sdf, *g_grad, *g_rgb, *g_inv_s;     // [B,n], [B,n,3], [B,n,3], [B]
};

// reverse of composite_fwd_kernel for the outputs a loss can touch: color, weight_sum, weights, gradient_error
__global__ __launch_bounds__(256) void composite_bwd_kernel(const CompBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long n_groups = (a.B + 3) >> 2;
  const int n = a.n;
  const float s_raw = *a.inv_s;
  const float inv_s = fminf(fmaxf(s_raw, 1e-6f), 1e6f);
  const float ge_coef = (a.g_gerr != nullptr) ? (*a.g_gerr) / (*a.gerr_den + 1e-5f) : 0.f;
  for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long ray = grp * 4 + wave;
    if (ray >= a.B) continue;
    const float ox = a.rays_o[ray * 3], oy = a.rays_o[ray * 3 + 1], oz = a.rays_o[ray * 3 + 2];
    const float dx = a.rays_d[ray * 3], dy = a.rays_d[ray * 3 + 1], dz = a.rays_d[ray * 3 + 2];
    const float gcr = a.g_color[ray * 3], gcg = a.g_color[ray * 3 + 1], gcb = a.g_color[ray * 3 + 2];
    const float gws = a.g_wsum ? a.g_wsum[ray] : 0.f;
    float bgdot = 0.f;
    if (a.bg) bgdot = gcr * a.bg[0] + gcg * a.bg[1] + gcb * a.bg[2];
    float alpha[IT], q[IT], gw[IT], draw_dpc[IT], draw_dnc[IT], pcv[IT], ncv[IT], epv[IT], env[IT], dist_[IT], dic_dtc[IT];
    bool pass[IT];
    float lprod = 1.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int i = lane * IT + k;
      alpha[k] = 0.f; q[k] = 1.f; gw[k] = 0.f; pass[k] = false;
      draw_dpc[k] = draw_dnc[k] = pcv[k] = ncv[k] = epv[k] = env[k] = dist_[k] = dic_dtc[k] = 0.f;
      if (i < n) {
        const long p = ray * n + i;
        const float dist = a.dists[p], sdf = a.sdf[p];
        const float gx = a.grad[p * 3], gy = a.grad[p * 3 + 1], gz = a.grad[p * 3 + 2];
        const float tc = __fmul_rn(dx, gx) + __fmul_rn(dy, gy) + __fmul_rn(dz, gz);
        const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.f) * (1.f - a.car) + fmaxf(-tc, 0.f) * a.car);
        dic_dtc[k] = ((-tc * 0.5f + 0.5f) > 0.f ? 0.5f * (1.f - a.car) : 0.f) + ((-tc) > 0.f ? a.car : 0.f);
        const float en = sdf + ic * dist * 0.5f, ep = sdf - ic * dist * 0.5f;
        const float pc = sigmoidf_(ep * inv_s), nc = sigmoidf_(en * inv_s);
        const float Dn = pc + 1e-5f;
        const float raw = (pc - nc + 1e-5f) / Dn;
        pass[k] = (raw >= 0.f) && (raw <= 1.f);
        alpha[k] = fminf(fmaxf(raw, 0.f), 1.f);
        q[k] = 1.f - alpha[k] + 1e-7f;
        lprod *= q[k];
        draw_dpc[k] = nc / (Dn * Dn);
        draw_dnc[k] = -1.f / Dn;
        pcv[k] = pc; ncv[k] = nc; epv[k] = ep; env[k] = en; dist_[k] = dist;
        gw[k] = (gcr * a.rgb[p * 3] + gcg * a.rgb[p * 3 + 1] + gcb * a.rgb[p * 3 + 2]) - bgdot + gws +
                (a.g_weights ? a.g_weights[p] : 0.f);
      }
    }
    float T = wave_excl_prod(lprod, lane);
    float Tk[IT], c[IT], lsum = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      Tk[k] = T;
      const float w = alpha[k] * T;
      T *= q[k];
      c[k] = gw[k] * w;
      lsum += c[k];
      const int i = lane * IT + k;
      if (i < n) {
        const long p = ray * n + i;
        a.g_rgb[p * 3] = w * gcr; a.g_rgb[p * 3 + 1] = w * gcg; a.g_rgb[p * 3 + 2] = w * gcb;
      }
    }
    float suf = wave_excl_suffix_sum(lsum, lane);          // sum of c over later lanes
    float g_s = 0.f;
#pragma unroll
    for (int k = IT - 1; k >= 0; --k) {
      const int i = lane * IT + k;
      if (i < n) {
        const long p = ray * n + i;
        // d loss / d alpha_i = gw_i T_i - (sum_{j>i} gw_j w_j) / q_i
        float g_alpha = gw[k] * Tk[k] - suf / q[k];
        const float g_raw = pass[k] ? g_alpha : 0.f;
        const float g_pc = g_raw * draw_dpc[k], g_nc = g_raw * draw_dnc[k];
        const float dpc = pcv[k] * (1.f - pcv[k]), dnc = ncv[k] * (1.f - ncv[k]);
        const float g_ep = g_pc * dpc * inv_s, g_en = g_nc * dnc * inv_s;
        g_s += g_pc * dpc * epv[k] + g_nc * dnc * env[k];
        a.g_sdf[p] = g_ep + g_en;
        const float g_ic = (g_en - g_ep) * dist_[k] * 0.5f;
        const float g_tc = g_ic * dic_dtc[k];
        // eikonal term
        const float gx = a.grad[p * 3], gy = a.grad[p * 3 + 1], gz = a.grad[p * 3 + 2];
        float ex = 0.f, ey = 0.f, ez = 0.f;
        if (ge_coef != 0.f) {
          const float mz = a.mid_z[p];
          const float px = ox + __fmul_rn(dx, mz), py = oy + __fmul_rn(dy, mz), pz = oz + __fmul_rn(dz, mz);
          const float pr = sqrtf(__fmul_rn(px, px) + __fmul_rn(py, py) + __fmul_rn(pz, pz));
          if (pr < a.radius * 1.1f) {
            const float gn = sqrtf(__fmul_rn(gx, gx) + __fmul_rn(gy, gy) + __fmul_rn(gz, gz));
            if (gn > 0.f) {
              const float f = ge_coef * 2.f * (gn - 1.f) / gn;
              ex = f * gx; ey = f * gy; ez = f * gz;
            }
          }
        }
        a.g_grad[p * 3] = g_tc * dx + ex; a.g_grad[p * 3 + 1] = g_tc * dy + ey; a.g_grad[p * 3 + 2] = g_tc * dz + ez;
      }
      suf += c[k];
    }
    g_s = wave_sum(g_s);
    if (lane == 0) a.g_inv_s[ray] = (s_raw >= 1e-6f && s_raw <= 1e6f) ? g_s : 0.f;
  }
}

int ray_grid(long B) {
  long g = (B + 3) / 4;
  const long cap = (long)vqn_num_cus() * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int vqn_neus_upsample(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int64_t B,
                                 int n, float r_limit, float inv_s, const float* u, int n_new, float* z_new,
                                 void* stream) {
  VQN_CHECK_ARG(B >= 0, "B >= 0");
  if (B == 0) return VQN_OK;
  VQN_CHECK_ARG(rays_o && rays_d && z && sdf && u && z_new, "null pointer");
  VQN_CHECK_SHAPE(n >= 2 && n <= MAXN, "2 <= n <= 256 samples per ray");
  VQN_CHECK_SHAPE(n_new >= 1 && n_new <= 64, "1 <= n_new <= 64");
  hipLaunchKernelGGL(upsample_kernel, dim3(ray_grid(B)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, z, sdf,
                     (long)B, n, r_limit, inv_s, u, n_new, z_new);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_neus_merge(const float* z, const float* sdf, const float* z_new, const float* sdf_new, int64_t B,
                              int n, int n_new, float* z_out, float* sdf_out, void* stream) {
  VQN_CHECK_ARG(B >= 0, "B >= 0");
  if (B == 0) return VQN_OK;
  VQN_CHECK_ARG(z && z_new && z_out, "null pointer");
  VQN_CHECK_ARG(sdf_out == nullptr || (sdf && sdf_new), "sdf_out needs sdf and sdf_new");
  VQN_CHECK_SHAPE(n >= 1 && n <= MAXN && n_new >= 1 && n_new <= 64, "n <= 256, n_new <= 64");
  hipLaunchKernelGGL(merge_kernel, dim3(ray_grid(B)), dim3(256), 0, (hipStream_t)stream, z, sdf, z_new, sdf_new,
                     (long)B, n, n_new, z_out, sdf_out);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_neus_section_mids(const float* z, int64_t B, int n, float sample_dist,
                                     const float* sample_dist_per_ray, float* mid_z, float* dists, void* stream) {
  VQN_CHECK_ARG(B >= 0 && n >= 1, "B >= 0, n >= 1");
  if (B == 0) return VQN_OK;
  VQN_CHECK_ARG(z && mid_z, "null pointer");
  const long tot = (long)B * n;
  hipLaunchKernelGGL(section_mids_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z,
                     (long)B, n, sample_dist, sample_dist_per_ray, mid_z, dists);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_neus_composite_fwd(const float* rays_o, const float* rays_d, const float* mid_z, const float* dists,
                                      const float* sdf, const float* grad, const float* rgb, const float* inv_s,
                                      const float* background_rgb, int64_t B, int n, float radius,
                                      float cos_anneal_ratio, float* color, float* weights, float* cdf, float* inside,
                                      float* surf, float* depth, float* weight_sum, float* weight_max, float* gerr,
                                      float* alpha, void* stream) {
  VQN_CHECK_ARG(B >= 0, "B >= 0");
  if (B == 0) return VQN_OK;
  VQN_CHECK_ARG(rays_o && rays_d && mid_z && dists && sdf && grad && rgb && inv_s, "null input pointer");
  VQN_CHECK_ARG(color && weights && cdf && inside && surf && depth && weight_sum && weight_max && gerr, "null output pointer");
  VQN_CHECK_SHAPE(n >= 1 && n <= MAXN, "1 <= n <= 256 samples per ray");
  CompArgs a{rays_o, rays_d, mid_z, dists, sdf, grad, rgb, inv_s, background_rgb, (long)B, n, radius, cos_anneal_ratio,
             color, weights, cdf, inside, surf, depth, weight_sum, weight_max, gerr, alpha};
  hipLaunchKernelGGL(composite_fwd_kernel, dim3(ray_grid(B)), dim3(256), 0, (hipStream_t)stream, a);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_neus_composite_bwd(const float* rays_o, const float* rays_d, const float* mid_z, const float* dists,
                                      const float* sdf, const float* grad, const float* rgb, const float* inv_s,
                                      const float* background_rgb, int64_t B, int n, float radius,
                                      float cos_anneal_ratio, const float* g_color, const float* g_weight_sum,
                                      const float* g_weights, const float* g_gradient_error, const float* gerr_den,
                                      float* g_sdf, float* g_grad, float* g_rgb, float* g_inv_s, void* stream) {
  VQN_CHECK_ARG(B >= 0, "B >= 0");
  if (B == 0) return VQN_OK;
  VQN_CHECK_ARG(rays_o && rays_d && mid_z && dists && sdf && grad && rgb && inv_s && g_color, "null input pointer");
  VQN_CHECK_ARG(g_sdf && g_grad && g_rgb && g_inv_s, "null output pointer");
  VQN_CHECK_ARG(g_gradient_error == nullptr || gerr_den != nullptr, "g_gradient_error needs gerr_den");
  VQN_CHECK_SHAPE(n >= 1 && n <= MAXN, "1 <= n <= 256 samples per ray");
  CompBwdArgs a;
  a.rays_d = rays_d; a.mid_z = mid_z; a.dists = dists; a.sdf = sdf; a.grad = grad; a.rgb = rgb; a.inv_s = inv_s;
  a.bg = background_rgb; a.g_color = g_color; a.g_wsum = g_weight_sum; a.g_weights = g_weights;
  a.g_gerr = g_gradient_error; a.gerr_den = gerr_den; a.rays_o = rays_o; a.B = B; a.n = n; a.radius = radius;
  a.car = cos_anneal_ratio; a.g_sdf = g_sdf; a.g_grad = g_grad; a.g_rgb = g_rgb; a.g_inv_s = g_inv_s;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3(ray_grid(B)), dim3(256), 0, (hipStream_t)stream, a);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
