// Fused light-direction + GGX microfacet BRDF + rendering-equation sum for gfx950: one 64-lane wave per surface
// point, every lane owns L/64 fixed lights (their direction, solid angle and radiance stay in registers), the
// per-point [L] visibility row streams from HBM as float4 per lane, nothing of size [N,L,3] is ever materialised.
// Replaces
//   decomp/nerfvq_nfr3/nerfactor/models/shape.py:103-119        (_calc_ldir, _calc_vdir)
//   decomp/nerfvq_nfr3/nerfactor/models/vq_nfr.py:830-833       (_normal_correct)
//   decomp/nerfvq_nfr3/nerfactor/util/microfacet.py:9-89        (get_brdf and its _get_f/_get_d/_get_g)
//   decomp/nerfvq_nfr3/nerfactor/models/vq_nfr.py:694-723       (_render.integrate: cos, front-lit, lvis, sum over L, gamma, clip)
// Up to two material sets (the continuous branch and the VQ branch of vq_nfr.Model.call, vq_nfr.py:593-627) share
// one pass over the geometry and the visibility row.
// Bound: HBM for data_type == 'nerf' (2 KB of lvis per point) -- VALU otherwise; see DESIGN.md.
#include "common.h"
#include <math.h>

namespace {

constexpr float PI_F = 3.14159265358979323846f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float clip01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
__device__ __forceinline__ float div_no_nan(float a, float b) { return b == 0.f ? 0.f : a / b; }
// tf.linalg.l2_normalize: x * rsqrt(max(sum(x^2), eps))   (util/math.py:63-64, eps = 1e-6)
// v_rsq_f32 (1 ulp): the per-light loop is VALU-bound, an IEEE 1/sqrt costs ~20 instructions
__device__ __forceinline__ float inv_norm(float x, float y, float z) {
  return __builtin_amdgcn_rsqf(fmaxf(x * x + y * y + z * z, 1e-6f));
}
// a / b with tf.math.divide_no_nan semantics on the 1-ulp hardware reciprocal
__device__ __forceinline__ float fdiv_no_nan(float a, float b) { return b == 0.f ? 0.f : a * __builtin_amdgcn_rcpf(b); }

struct ShadeArgs {
  const float *xyz, *normal, *rayo, *lvis, *lxyz, *lareas, *light, *gamma;
  const float *albedo[2], *spec[2], *rough[2];
  float *normal_out, *rgb[2], *rgb_diff, *rgb_spec;
  long N;
  int n_sets;
  int raw;                     // 1: write the plain sums over lights (no gamma, no [0,1] clip) -- the training path applies those in torch
  const float* probes;         // [P][L][3] novel light probes (vq_nfr.py:724-733) or null
  float* rgb_probes;           // [N][P][3]: material set 0 re-lit by every probe in the same pass
  int n_probes;
};

struct Material {
  float a[3], f0[3], a2;       // albedo, spec (f0), alpha^2 with alpha = rough^2 (microfacet.py:24, squared again inside D and G)
  float g1v;                   // G1(v.n) (per point)
};

template <int LQ>
__global__ __launch_bounds__(256) void brdf_shade_kernel(const ShadeArgs a) {
  constexpr int LP = 4 * LQ;                       // lights per lane
  constexpr int L = 64 * LP;
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: per-point scalars by scalar loads
  const long n_waves = (long)gridDim.x * 4;

  // ---- this lane's lights: light index = 256 g + 4 lane + e ----
  float lx[LP], ly[LP], lz[LP], area[LP], Lr[LP], Lg[LP], Lb[LP];
#pragma unroll
  for (int g = 0; g < LQ; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int li = 256 * g + 4 * lane + e, k = 4 * g + e;
      lx[k] = a.lxyz[li * 3 + 0]; ly[k] = a.lxyz[li * 3 + 1]; lz[k] = a.lxyz[li * 3 + 2];
      area[k] = a.lareas[li];
      Lr[k] = a.light[li * 3 + 0]; Lg[k] = a.light[li * 3 + 1]; Lb[k] = a.light[li * 3 + 2];
    }
  float gam_b = 1.f, gam_i = 1.f;
  if (a.gamma) { gam_b = a.gamma[0]; gam_i = a.gamma[1]; }
  const bool split = a.rgb_diff != nullptr;

  for (long n = wave_id; n < a.N; n += n_waves) {
    // visibility row first (longest latency)
    f32x4 vis4[LQ];
#pragma unroll
    for (int g = 0; g < LQ; ++g)
      vis4[g] = a.lvis ? *reinterpret_cast<const f32x4*>(a.lvis + n * L + 256 * g + 4 * lane) : (f32x4){1.f, 1.f, 1.f, 1.f};
    const float px = a.xyz[n * 3], py = a.xyz[n * 3 + 1], pz = a.xyz[n * 3 + 2];
    // view direction (shape.py:112-119) and camera-facing normal (vq_nfr.py:830-833)
    float vx = a.rayo[n * 3] - px, vy = a.rayo[n * 3 + 1] - py, vz = a.rayo[n * 3 + 2] - pz;
    float iv = inv_norm(vx, vy, vz);
    vx *= iv; vy *= iv; vz *= iv;                  // surf2c
    float nx = a.normal[n * 3], ny = a.normal[n * 3 + 1], nz = a.normal[n * 3 + 2];
    if (nx * vx + ny * vy + nz * vz < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
    if (a.normal_out && lane < 3) a.normal_out[n * 3 + lane] = lane == 0 ? nx : (lane == 1 ? ny : nz);
    // get_brdf re-normalises its inputs (microfacet.py:13-16)
    iv = inv_norm(vx, vy, vz);
    const float ux = vx * iv, uy = vy * iv, uz = vz * iv;          // v
    const float in_ = inv_norm(nx, ny, nz);
    const float mx = nx * in_, my = ny * in_, mz = nz * in_;       // n (unit)
    const float v_dot_n = ux * mx + uy * my + uz * mz;
    Material M[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (s < a.n_sets) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { M[s].a[c] = a.albedo[s][n * 3 + c]; M[s].f0[c] = a.spec[s][n * 3 + c]; }
        const float r = a.rough[s][n];
        const float alpha = r * r;
        M[s].a2 = alpha * alpha;
        const float c = clip01(v_dot_n);
        M[s].g1v = div_no_nan(2.f * c, c + sqrtf(fabsf(M[s].a2 + (1.f - M[s].a2) * c * c)));
      }
    float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    float accd[3] = {0.f, 0.f, 0.f}, accs[3] = {0.f, 0.f, 0.f};
    float wbr[LP][3];                                // set 0: brdf_c * vis * cos * area per light = contribution per unit radiance
#pragma unroll
    for (int k = 0; k < LP; ++k) {
      // light direction (shape.py:103-110)
      float dx = lx[k] - px, dy = ly[k] - py, dz = lz[k] - pz;
      float il = inv_norm(dx, dy, dz);
      dx *= il; dy *= il; dz *= il;                                 // surf2l
      const float cosl = dx * nx + dy * ny + dz * nz;               // vq_nfr.py:702 (un-renormalised normal)
      const float vis = (cosl > 0.f ? 1.f : 0.f) * vis4[k >> 2][k & 3];
      il = inv_norm(dx, dy, dz);
      const float wx = dx * il, wy = dy * il, wz = dz * il;          // l (microfacet.py:13)
      float hx = wx + ux, hy = wy + uy, hz = wz + uz;
      const float ih = inv_norm(hx, hy, hz);
      hx *= ih; hy *= ih; hz *= ih;
      const float cos_vh = clip01(hx * ux + hy * uy + hz * uz);
      const float om = 1.f - cos_vh, om2 = om * om, om5 = om2 * om2 * om;
      const float cos_m = clip01(hx * mx + hy * my + hz * mz);
      const float l_dot_n = wx * mx + wy * my + wz * mz;
      const float cl = clip01(l_dot_n);
      const float den = 4.f * fabsf(l_dot_n) * fabsf(v_dot_n);
      const float inv_den = den == 0.f ? 0.f : __builtin_amdgcn_rcpf(den);
      const float lr = vis * Lr[k], lg = vis * Lg[k], lb = vis * Lb[k];
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (s < a.n_sets) {
          const float a2 = M[s].a2;
          const float t = cos_m * cos_m * (a2 - 1.f) + 1.f;
          const float D = fdiv_no_nan(a2, PI_F * t * t);
          const float G = fdiv_no_nan(2.f * cl, cl + __builtin_amdgcn_sqrtf(fabsf(a2 + (1.f - a2) * cl * cl))) * M[s].g1v;
          const float gd = G * D;
          float gl[3], df[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float F = M[s].f0[c] + (1.f - M[s].f0[c]) * om5;
            gl[c] = F * gd * inv_den;
            df[c] = M[s].a[c] * (1.f / PI_F);
          }
          acc[s][0] += (gl[0] + df[0]) * lr * cosl * area[k];
          acc[s][1] += (gl[1] + df[1]) * lg * cosl * area[k];
          acc[s][2] += (gl[2] + df[2]) * lb * cosl * area[k];
          if (s == 0 && a.probes != nullptr) {
            const float gw = vis * cosl * area[k];
            wbr[k][0] = (gl[0] + df[0]) * gw; wbr[k][1] = (gl[1] + df[1]) * gw; wbr[k][2] = (gl[2] + df[2]) * gw;
          }
          if (split && s == 0) {
            accd[0] += df[0] * lr * cosl * area[k]; accd[1] += df[1] * lg * cosl * area[k]; accd[2] += df[2] * lb * cosl * area[k];
            accs[0] += gl[0] * lr * cosl * area[k]; accs[1] += gl[1] * lg * cosl * area[k]; accs[2] += gl[2] * lb * cosl * area[k];
          }
        }
    }
    auto finish = [&](float v) {
      v = wave_sum(v);
      if (a.raw) return v;
      if (a.gamma) v = powf(v * gam_b, gam_i);
      return clip01(v);
    };
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (s < a.n_sets) {
        const float r = finish(acc[s][0]), g = finish(acc[s][1]), b = finish(acc[s][2]);
        if (lane < 3) a.rgb[s][n * 3 + lane] = lane == 0 ? r : (lane == 1 ? g : b);
      }
    if (a.probes != nullptr) {
      // all probes against the SAME per-light weights: the [N,L] x [L,3P] contraction of the relighting loop
      for (int pr = 0; pr < a.n_probes; ++pr) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int g = 0; g < LQ; ++g) {
          const f32x4* pp = reinterpret_cast<const f32x4*>(a.probes + ((size_t)pr * L + 256 * g + 4 * lane) * 3);
          const f32x4 q0 = pp[0], q1 = pp[1], q2 = pp[2];      // 4 lights x rgb, interleaved
          const float rad[12] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3], q2[0], q2[1], q2[2], q2[3]};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s0 += wbr[4 * g + e][0] * rad[3 * e]; s1 += wbr[4 * g + e][1] * rad[3 * e + 1]; s2 += wbr[4 * g + e][2] * rad[3 * e + 2];
          }
        }
        const float r = finish(s0), gg = finish(s1), b = finish(s2);
        if (lane < 3) a.rgb_probes[((size_t)n * a.n_probes + pr) * 3 + lane] = lane == 0 ? r : (lane == 1 ? gg : b);
      }
    }
    if (split) {
      const float r = finish(accd[0]), g = finish(accd[1]), b = finish(accd[2]);
      if (lane < 3) a.rgb_diff[n * 3 + lane] = lane == 0 ? r : (lane == 1 ? g : b);
      const float r2 = finish(accs[0]), g2 = finish(accs[1]), b2 = finish(accs[2]);
      if (lane < 3) a.rgb_spec[n * 3 + lane] = lane == 0 ? r2 : (lane == 1 ? g2 : b2);
    }
  }
}

struct ShadeBwdArgs {
  const float *xyz, *normal, *rayo, *lvis, *lxyz, *lareas, *light;
  const float *albedo[2], *spec[2], *rough[2], *g_sum[2];     // g_sum: d loss / d (plain sum over lights) [N,3]
  float *g_albedo[2], *g_spec[2], *g_rough[2];                // [N,3], [N,3], [N]
  float* g_light_part;                                        // [n_waves_total][L][3] per-wave partials (summed by the caller, in order)
  long N;
  int n_sets;
};

// d/d a2 of G1(c) = 2c / (c + sqrt|a2 + (1 - a2) c^2|), with divide_no_nan semantics
__device__ __forceinline__ void g1_and_da2(float c, float a2, float* g1, float* dg1) {
  const float q = a2 + (1.f - a2) * c * c;
  const float s = __builtin_amdgcn_sqrtf(fabsf(q));
  const float den = c + s;
  if (den == 0.f) { *g1 = 0.f; *dg1 = 0.f; return; }
  const float rden = __builtin_amdgcn_rcpf(den);
  *g1 = 2.f * c * rden;
  const float ds = s > 0.f ? (q >= 0.f ? 1.f : -1.f) * (1.f - c * c) * 0.5f * __builtin_amdgcn_rcpf(s) : 0.f;
  *dg1 = -2.f * c * rden * rden * ds;
}

template <int LQ>
__global__ __launch_bounds__(256) void brdf_shade_bwd_kernel(const ShadeBwdArgs a) {
  constexpr int LP = 4 * LQ;
  constexpr int L = 64 * LP;
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: per-point scalars by scalar loads
  const long n_waves = (long)gridDim.x * 4;
  float lx[LP], ly[LP], lz[LP], area[LP], Lr[LP], Lg[LP], Lb[LP], gL[LP][3];
#pragma unroll
  for (int g = 0; g < LQ; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int li = 256 * g + 4 * lane + e, k = 4 * g + e;
      lx[k] = a.lxyz[li * 3 + 0]; ly[k] = a.lxyz[li * 3 + 1]; lz[k] = a.lxyz[li * 3 + 2];
      area[k] = a.lareas[li];
      Lr[k] = a.light[li * 3 + 0]; Lg[k] = a.light[li * 3 + 1]; Lb[k] = a.light[li * 3 + 2];
      gL[k][0] = gL[k][1] = gL[k][2] = 0.f;
    }
  for (long n = wave_id; n < a.N; n += n_waves) {
    f32x4 vis4[LQ];
#pragma unroll
    for (int g = 0; g < LQ; ++g)
      vis4[g] = a.lvis ? *reinterpret_cast<const f32x4*>(a.lvis + n * L + 256 * g + 4 * lane) : (f32x4){1.f, 1.f, 1.f, 1.f};
    const float px = a.xyz[n * 3], py = a.xyz[n * 3 + 1], pz = a.xyz[n * 3 + 2];
    float vx = a.rayo[n * 3] - px, vy = a.rayo[n * 3 + 1] - py, vz = a.rayo[n * 3 + 2] - pz;
    float iv = inv_norm(vx, vy, vz);
    vx *= iv; vy *= iv; vz *= iv;
    float nx = a.normal[n * 3], ny = a.normal[n * 3 + 1], nz = a.normal[n * 3 + 2];
    if (nx * vx + ny * vy + nz * vz < 0.f) { nx = -nx; ny = -ny; nz = -nz; }
    iv = inv_norm(vx, vy, vz);
    const float ux = vx * iv, uy = vy * iv, uz = vz * iv;
    const float in_ = inv_norm(nx, ny, nz);
    const float mx = nx * in_, my = ny * in_, mz = nz * in_;
    const float v_dot_n = ux * mx + uy * my + uz * mz;
    const float cv = clip01(v_dot_n);
    float alb[2][3], f0[2][3], a2[2], g1v[2], dg1v[2], gs[2][3], rgh[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (s < a.n_sets) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          alb[s][c] = a.albedo[s][n * 3 + c]; f0[s][c] = a.spec[s][n * 3 + c]; gs[s][c] = a.g_sum[s][n * 3 + c];
        }
        rgh[s] = a.rough[s][n];
        const float alpha = rgh[s] * rgh[s];
        a2[s] = alpha * alpha;
        g1_and_da2(cv, a2[s], &g1v[s], &dg1v[s]);
      }
    float acc_alb[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, acc_f0[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, acc_a2[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < LP; ++k) {
      float dx = lx[k] - px, dy = ly[k] - py, dz = lz[k] - pz;
      float il = inv_norm(dx, dy, dz);
      dx *= il; dy *= il; dz *= il;
      const float cosl = dx * nx + dy * ny + dz * nz;
      const float vis = (cosl > 0.f ? 1.f : 0.f) * vis4[k >> 2][k & 3];
      il = inv_norm(dx, dy, dz);
      const float wx = dx * il, wy = dy * il, wz = dz * il;
      float hx = wx + ux, hy = wy + uy, hz = wz + uz;
      const float ih = inv_norm(hx, hy, hz);
      hx *= ih; hy *= ih; hz *= ih;
      const float cos_vh = clip01(hx * ux + hy * uy + hz * uz);
      const float om = 1.f - cos_vh, om2 = om * om, om5 = om2 * om2 * om;
      const float cos_m = clip01(hx * mx + hy * my + hz * mz);
      const float l_dot_n = wx * mx + wy * my + wz * mz;
      const float cl = clip01(l_dot_n);
      const float den = 4.f * fabsf(l_dot_n) * fabsf(v_dot_n);
      const float inv_den = den == 0.f ? 0.f : __builtin_amdgcn_rcpf(den);
      const float wgt = vis * cosl * area[k];                       // geometry weight of this light (without radiance)
      const float Lc[3] = {Lr[k], Lg[k], Lb[k]};
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (s < a.n_sets) {
          const float t = cos_m * cos_m * (a2[s] - 1.f) + 1.f;
          const float pd = PI_F * t * t;
          float D = 0.f, dD = 0.f;
          if (pd != 0.f) {
            const float rpd = __builtin_amdgcn_rcpf(pd);
            D = a2[s] * rpd;
            dD = (t - 2.f * a2[s] * cos_m * cos_m) * rpd * __builtin_amdgcn_rcpf(t);
          }
          float g1l, dg1l;
          g1_and_da2(cl, a2[s], &g1l, &dg1l);
          const float G = g1l * g1v[s];
          const float dG = dg1l * g1v[s] + g1l * dg1v[s];
          const float gd = G * D * inv_den, dgd = (dG * D + G * dD) * inv_den;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float F = f0[s][c] + (1.f - f0[s][c]) * om5;
            const float gw = gs[s][c] * wgt;                        // d loss / d (brdf_c * radiance_c)  per unit radiance
            const float glw = gw * Lc[c];
            acc_alb[s][c] += glw * (1.f / PI_F);
            acc_f0[s][c] += glw * (1.f - om5) * gd;
            acc_a2[s] += glw * F * dgd;
            gL[k][c] += gw * (F * gd + alb[s][c] / PI_F);
          }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (s < a.n_sets) {
        float r[7];
#pragma unroll
        for (int c = 0; c < 3; ++c) { r[c] = wave_sum(acc_alb[s][c]); r[3 + c] = wave_sum(acc_f0[s][c]); }
        r[6] = wave_sum(acc_a2[s]) * 4.f * rgh[s] * rgh[s] * rgh[s];          // a2 = rough^4
        if (lane < 3) { a.g_albedo[s][n * 3 + lane] = r[lane]; a.g_spec[s][n * 3 + lane] = r[3 + lane]; }
        if (lane == 0) a.g_rough[s][n] = r[6];
      }
  }
  float* part = a.g_light_part + (size_t)wave_id * L * 3;
#pragma unroll
  for (int g = 0; g < LQ; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int li = 256 * g + 4 * lane + e, k = 4 * g + e;
      part[li * 3 + 0] = gL[k][0]; part[li * 3 + 1] = gL[k][1]; part[li * 3 + 2] = gL[k][2];
    }
}

}  // namespace

extern "C" int vqn_brdf_shade_fwd(const float* xyz, const float* normal, const float* rayo, const float* lvis,
                                  const float* lxyz, const float* lareas, const float* light, int64_t N, int L,
                                  int n_sets, const float* albedo0, const float* spec0, const float* rough0,
                                  const float* albedo1, const float* spec1, const float* rough1, const float* gamma,
                                  float* normal_out, float* rgb0, float* rgb1, float* rgb0_diff, float* rgb0_spec,
                                  int raw, const float* probes, int n_probes, float* rgb0_probes, void* stream) {
  VQN_CHECK_ARG(N >= 0, "N >= 0");
  if (N == 0) return VQN_OK;
  VQN_CHECK_ARG(xyz && normal && rayo && lxyz && lareas && light, "null geometry / light pointer");
  VQN_CHECK_ARG(n_sets == 1 || n_sets == 2, "n_sets must be 1 or 2");
  VQN_CHECK_ARG(albedo0 && spec0 && rough0 && rgb0, "material set 0 and rgb0 must be non-null");
  VQN_CHECK_ARG(n_sets == 1 || (albedo1 && spec1 && rough1 && rgb1), "material set 1 and rgb1 must be non-null");
  VQN_CHECK_ARG((rgb0_diff == nullptr) == (rgb0_spec == nullptr), "rgb0_diff and rgb0_spec go together");
  VQN_CHECK_SHAPE(L == 256 || L == 512 || L == 1024, "L must be 256, 512 or 1024 lights");
  VQN_CHECK_SHAPE(lvis == nullptr || ((uintptr_t)lvis & 15) == 0, "lvis must be 16-byte aligned");
  ShadeArgs a;
  a.xyz = xyz; a.normal = normal; a.rayo = rayo; a.lvis = lvis; a.lxyz = lxyz; a.lareas = lareas; a.light = light;
  a.gamma = gamma;
  a.albedo[0] = albedo0; a.spec[0] = spec0; a.rough[0] = rough0;
  a.albedo[1] = albedo1; a.spec[1] = spec1; a.rough[1] = rough1;
  a.normal_out = normal_out; a.rgb[0] = rgb0; a.rgb[1] = rgb1; a.rgb_diff = rgb0_diff; a.rgb_spec = rgb0_spec;
  a.N = N; a.n_sets = n_sets; a.raw = raw;
  VQN_CHECK_ARG(probes == nullptr || (n_probes >= 1 && rgb0_probes != nullptr), "probes need n_probes >= 1 and rgb0_probes");
  VQN_CHECK_SHAPE(probes == nullptr || ((uintptr_t)probes & 15) == 0, "probes must be 16-byte aligned");
  a.probes = probes; a.n_probes = probes ? n_probes : 0; a.rgb_probes = rgb0_probes;
  long blocks = (N + 3) / 4;
  const long cap = (long)vqn_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipStream_t s = (hipStream_t)stream;
  if (L == 256) hipLaunchKernelGGL(brdf_shade_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else if (L == 512) hipLaunchKernelGGL(brdf_shade_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(brdf_shade_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int64_t vqn_brdf_shade_bwd_partials(int64_t N) {
  long blocks = (N + 3) / 4;
  const long cap = (long)vqn_num_cus() * 4;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return blocks * 4;
}

extern "C" int vqn_brdf_shade_bwd(const float* xyz, const float* normal, const float* rayo, const float* lvis,
                                  const float* lxyz, const float* lareas, const float* light, int64_t N, int L,
                                  int n_sets, const float* albedo0, const float* spec0, const float* rough0,
                                  const float* g_sum0, const float* albedo1, const float* spec1, const float* rough1,
                                  const float* g_sum1, float* g_albedo0, float* g_spec0, float* g_rough0,
                                  float* g_albedo1, float* g_spec1, float* g_rough1, float* g_light_partials,
                                  void* stream) {
  VQN_CHECK_ARG(N >= 1, "N >= 1");
  VQN_CHECK_ARG(xyz && normal && rayo && lxyz && lareas && light && g_light_partials, "null geometry / light pointer");
  VQN_CHECK_ARG(n_sets == 1 || n_sets == 2, "n_sets must be 1 or 2");
  VQN_CHECK_ARG(albedo0 && spec0 && rough0 && g_sum0 && g_albedo0 && g_spec0 && g_rough0, "material set 0 pointers");
  VQN_CHECK_ARG(n_sets == 1 || (albedo1 && spec1 && rough1 && g_sum1 && g_albedo1 && g_spec1 && g_rough1), "material set 1 pointers");
  VQN_CHECK_SHAPE(L == 256 || L == 512 || L == 1024, "L must be 256, 512 or 1024 lights");
  VQN_CHECK_SHAPE(lvis == nullptr || ((uintptr_t)lvis & 15) == 0, "lvis must be 16-byte aligned");
  ShadeBwdArgs a;
  a.xyz = xyz; a.normal = normal; a.rayo = rayo; a.lvis = lvis; a.lxyz = lxyz; a.lareas = lareas; a.light = light;
  a.albedo[0] = albedo0; a.spec[0] = spec0; a.rough[0] = rough0; a.g_sum[0] = g_sum0;
  a.albedo[1] = albedo1; a.spec[1] = spec1; a.rough[1] = rough1; a.g_sum[1] = g_sum1;
  a.g_albedo[0] = g_albedo0; a.g_spec[0] = g_spec0; a.g_rough[0] = g_rough0;
  a.g_albedo[1] = g_albedo1; a.g_spec[1] = g_spec1; a.g_rough[1] = g_rough1;
  a.g_light_part = g_light_partials; a.N = N; a.n_sets = n_sets;
  const long blocks = vqn_brdf_shade_bwd_partials(N) / 4;
  hipStream_t s = (hipStream_t)stream;
  if (L == 256) hipLaunchKernelGGL(brdf_shade_bwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else if (L == 512) hipLaunchKernelGGL(brdf_shade_bwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(brdf_shade_bwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
