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
#include <stdlib.h>
#include "vqnerf_hip.h"
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

// Two lights per instruction (round 4): the per-light arithmetic is the same instruction stream for every light, so lights 2j and 2j + 1 of
// a lane travel as the two halves of 64-bit register pairs -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do both at the issue cost of one
// (the vector pipe's 157 TFLOP/s IS the packed rate; the scalar forms peak at half of it).  Not packable: min / max / compares and the
// quarter-rate rsq / rcp / sqrt, issued per half.  Component-wise the operations are those of the scalar statement (explicit fma).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 max2(f32x2 a, float b) { return (f32x2){fmaxf(a[0], b), fmaxf(a[1], b)}; }
__device__ __forceinline__ f32x2 clip01_2(f32x2 a) { return (f32x2){clip01(a[0]), clip01(a[1])}; }
__device__ __forceinline__ f32x2 rsq2(f32x2 a) { return (f32x2){__builtin_amdgcn_rsqf(a[0]), __builtin_amdgcn_rsqf(a[1])}; }
__device__ __forceinline__ f32x2 rcp2(f32x2 a) { return (f32x2){__builtin_amdgcn_rcpf(a[0]), __builtin_amdgcn_rcpf(a[1])}; }
__device__ __forceinline__ f32x2 sqrt2(f32x2 a) { return (f32x2){__builtin_amdgcn_sqrtf(a[0]), __builtin_amdgcn_sqrtf(a[1])}; }
__device__ __forceinline__ f32x2 abs2(f32x2 a) { return (f32x2){fabsf(a[0]), fabsf(a[1])}; }
__device__ __forceinline__ f32x2 inv_norm2(f32x2 x, f32x2 y, f32x2 z) { return rsq2(max2(fma2(z, z, fma2(y, y, x * x)), 1e-6f)); }
__device__ __forceinline__ f32x2 dot2(f32x2 ax, f32x2 ay, f32x2 az, float bx, float by, float bz) {
  return fma2(az, splat2(bz), fma2(ay, splat2(by), ax * splat2(bx)));
}

// Sixteen per-lane values summed over the wave in ONE halving butterfly: across lane bits 32, 16, 8, 4 every lane keeps half of its
// values and hands the other half over (8 + 4 + 2 + 1 exchanges), bits 2 and 1 finish the last value: 17 exchanges instead of the 96 of
// sixteen separate six-step reductions.  Returns in lane l the total of value (l >> 2) & 15.
__device__ __forceinline__ float wave_sum16(float (&v)[16], const int lane) {
#pragma unroll
  for (int half = 8; half >= 1; half >>= 1) {
    const int bit = half << 2;
    const bool up = (lane & bit) != 0;
#pragma unroll
    for (int i = 0; i < half; ++i) {
      float lo = v[i], hi = v[i + half];
      asm volatile("" : "+v"(lo), "+v"(hi));      // (values, not addresses: the optimiser otherwise selects the array INDEX and walks a 16-way compare chain per access)
      const float keep = up ? hi : lo;
      const float send = up ? lo : hi;
      v[i] = keep + __shfl_xor(send, bit);
    }
  }
  float t = v[0];
  t += __shfl_xor(t, 2);
  t += __shfl_xor(t, 1);
  return t;
}

struct ShadeArgs {
  const float *xyz, *normal, *rayo, *lvis, *lxyz, *lareas, *light, *gamma;
  const float *albedo[2], *spec[2], *rough[2];
  float *normal_out, *rgb[2], *rgb_diff, *rgb_spec;
  long N;
  int n_sets;
  int raw;                     // 1: write the plain sums over lights (no gamma, no [0,1] clip) -- the training path applies those in torch;
                               // 2: the plain sums through x + (clip(x, 0, 1) - x), tfp's clip_by_value_preserve_gradient as the training path of
                               //    data_type 'nerf' applies it (vq_nfr.py:735-745: no gamma there) -- the same roundings as vqn_clip_preserve
  const long long* rows;       // null, or [N]: point n takes its visibility row from row rows[n] of the FULL-view lvis tensor --
                               // the tf.boolean_mask gather of vq_nfr.py:558-559 (2 KB per point) without the copy
  const float* probes;         // [P][L][3] novel light probes (vq_nfr.py:724-733) or null
  float* rgb_probes;           // [N][P][3]: material set 0 re-lit by every probe in the same pass
  int n_probes;
};

struct Material {
  float a[3], f0[3], omf0[3];  // albedo / pi, spec (f0), 1 - f0
  float a2, a2_pi, oma2;       // alpha^2 with alpha = rough^2 (microfacet.py:24, squared again inside D and G), a2 / pi, 1 - a2
  float kv;                    // G1(v.n) / (2 |v.n|)  (per point; 0 where v.n == 0: divide_no_nan, microfacet.py:33)
};

// One wave per point, LP lights per lane.  Per (point, light) the arithmetic of microfacet.py:9-89 + vq_nfr.py:694-723 is
//     rgb_c = sum_l (glossy_c + albedo_c / pi) * vis_l * L_{l,c} * cos_l * area_l,
//     glossy_c = F_c D G1(l.n) G1(v.n) / (4 |l.n| |v.n|),   G1(c) = 2 c / (c + sqrt|a2 + (1 - a2) c^2|).
// The kernel is vector-ALU bound (the 2 KB visibility row per point costs 0.6 TB/s of the 8 available), so the inner loop is
// written for instruction count, algebraically equal to the reference's op sequence where a light contributes at all:
//   * a light contributes only if it is front-lit (cos_l > 0, vq_nfr.py:704); there 0 < l.n, so
//     G1(l.n) / (4 |l.n|) = 1 / (2 (cl + sqrt|a2 + (1 - a2) cl^2|)), cl = min(l.n, 1): one reciprocal instead of two and no
//     2 cl / (4 |l.n|) round trip; G1(v.n) / (2 |v.n|) is a per-point scalar (kv);
//   * the Lambertian part albedo_c / pi * sum_l w_{l,c} is factored out of the loop (w = vis L cos area is shared by both
//     material sets and by the diffuse / specular split);
//   * get_brdf's re-normalisation of the already unit light direction (microfacet.py:13) is dropped (<= 2 ulp).
// Results move by fp32 rounding only (tests/test_gpu_decomp.py judges them against the float64 oracle).
// PLDS (relighting, <= 24 probes): the probe table lives in LDS, laid out [probe][g][q][lane] float4 -- one copy per 768-thread
// (1024 lights: 512-thread) workgroup, staged once.  Without it every wave re-reads the whole table (6 KB per probe) for every point through the L1: 63 GB per
// 640,000-point view under 16 probes, and that traffic, not the arithmetic, paced the relighting pass.
template <int LQ, bool PROBES, int NS, bool PLDS = false>
__global__ __launch_bounds__(PLDS ? (LQ == 4 ? 512 : 768) : 256) void brdf_shade_kernel(const ShadeArgs a) {
  constexpr int LP = 4 * LQ;                       // lights per lane
  constexpr int L = 64 * LP;
  constexpr int WPB = PLDS ? (LQ == 4 ? 8 : 12) : 4;      // waves per workgroup (PLDS: as many as the registers allow -- 168 at L <= 512)
  extern __shared__ __attribute__((aligned(16))) f32x4 ptab[];
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * WPB + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: per-point scalars by scalar loads
  const long n_waves = (long)gridDim.x * WPB;
  if (PROBES && PLDS) {
    const int n4 = a.n_probes * LQ * 3 * 64;
    for (int i = threadIdx.x; i < n4; i += 64 * WPB) {
      const int l = i & 63, q = (i >> 6) % 3, g = ((i >> 6) / 3) % LQ, pr = (i >> 6) / (3 * LQ);
      ptab[i] = reinterpret_cast<const f32x4*>(a.probes + ((size_t)pr * L + 256 * g + 4 * l) * 3)[q];
    }
    __syncthreads();
  }

  // ---- this lane's lights: light index = 256 g + 4 lane + e ----
  constexpr int LH = LP / 2;                       // light PAIRS per lane: pair j = lights (2 j, 2 j + 1) of the lane
  f32x2 lx[LH], ly[LH], lz[LH], Ar[LH], Ag[LH], Ab[LH];      // A_c = radiance_c * solid angle
#pragma unroll
  for (int g = 0; g < LQ; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int li = 256 * g + 4 * lane + e, k = 4 * g + e;
      lx[k >> 1][k & 1] = a.lxyz[li * 3 + 0]; ly[k >> 1][k & 1] = a.lxyz[li * 3 + 1]; lz[k >> 1][k & 1] = a.lxyz[li * 3 + 2];
      const float area = a.lareas[li];
      Ar[k >> 1][k & 1] = a.light[li * 3 + 0] * area; Ag[k >> 1][k & 1] = a.light[li * 3 + 1] * area; Ab[k >> 1][k & 1] = a.light[li * 3 + 2] * area;
    }
  f32x2 area_k[PROBES ? LH : 1];                    // only the relighting pass needs the bare solid angles
  if (PROBES) {
#pragma unroll
    for (int g = 0; g < LQ; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) area_k[(4 * g + e) >> 1][e & 1] = a.lareas[256 * g + 4 * lane + e];
  }
  float gam_b = 1.f, gam_i = 1.f;
  if (a.gamma) { gam_b = a.gamma[0]; gam_i = a.gamma[1]; }
  const bool split = a.rgb_diff != nullptr;

  for (long n = wave_id; n < a.N; n += n_waves) {
    const long m = a.rows ? (long)a.rows[n] : n;   // visibility row (wave-uniform)
    // visibility row first (longest latency)
    f32x4 vis4[LQ];
#pragma unroll
    for (int g = 0; g < LQ; ++g)
      vis4[g] = a.lvis ? *reinterpret_cast<const f32x4*>(a.lvis + m * L + 256 * g + 4 * lane) : (f32x4){1.f, 1.f, 1.f, 1.f};
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
      if (s < NS) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          M[s].a[c] = a.albedo[s][n * 3 + c] * (1.f / PI_F);
          M[s].f0[c] = a.spec[s][n * 3 + c];
          M[s].omf0[c] = 1.f - M[s].f0[c];
        }
        const float r = a.rough[s][n];
        const float alpha = r * r;
        M[s].a2 = alpha * alpha;
        M[s].a2_pi = M[s].a2 * (1.f / PI_F);
        M[s].oma2 = 1.f - M[s].a2;
        const float c = clip01(v_dot_n);
        const float g1v = div_no_nan(2.f * c, c + sqrtf(fabsf(M[s].a2 + M[s].oma2 * c * c)));
        M[s].kv = div_no_nan(g1v, 2.f * fabsf(v_dot_n));
      }
    f32x2 S2[2][3], W2[3];                                           // sum_l glossy_c * w_c per material set; sum_l w_c (Lambertian part, shared): even / odd lights
#pragma unroll
    for (int c = 0; c < 3; ++c) { S2[0][c] = splat2(0.f); S2[1][c] = splat2(0.f); W2[c] = splat2(0.f); }
    float wsp[PROBES ? LP : 1][3];                                   // relighting: set-0 brdf_c * vis * cos * area per light (per unit radiance)
#pragma unroll
    for (int j = 0; j < LH; ++j) {
      // light direction (shape.py:103-110)
      f32x2 dx = lx[j] - splat2(px), dy = ly[j] - splat2(py), dz = lz[j] - splat2(pz);
      const f32x2 il = inv_norm2(dx, dy, dz);
      dx *= il; dy *= il; dz *= il;                                 // surf2l = l
      const f32x2 cosl = dot2(dx, dy, dz, nx, ny, nz);              // vq_nfr.py:702 (un-renormalised normal)
      const f32x2 l_dot_n = dot2(dx, dy, dz, mx, my, mz);
      const f32x2 visj = {vis4[j >> 1][2 * (j & 1)], vis4[j >> 1][2 * (j & 1) + 1]};
      f32x2 cw = cosl * visj;                                       // vis * cos
      cw[0] = (cosl[0] > 0.f && l_dot_n[0] > 0.f) ? cw[0] : 0.f;
      cw[1] = (cosl[1] > 0.f && l_dot_n[1] > 0.f) ? cw[1] : 0.f;
      f32x2 hx = dx + splat2(ux), hy = dy + splat2(uy), hz = dz + splat2(uz);
      const f32x2 ih = inv_norm2(hx, hy, hz);
      hx *= ih; hy *= ih; hz *= ih;
      const f32x2 cos_vh = clip01_2(dot2(hx, hy, hz, ux, uy, uz));
      const f32x2 om = splat2(1.f) - cos_vh, om2 = om * om, om5 = om2 * om2 * om;
      const f32x2 cos_m = clip01_2(dot2(hx, hy, hz, mx, my, mz));
      const f32x2 cm2 = cos_m * cos_m;
      const f32x2 cl = clip01_2(l_dot_n), cl2 = cl * cl;
      const f32x2 wr = cw * Ar[j], wg = cw * Ag[j], wb = cw * Ab[j];
      W2[0] += wr; W2[1] += wg; W2[2] += wb;
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (s < NS) {
          const f32x2 t = fma2(cm2, splat2(-M[s].oma2), splat2(1.f));   // cos_m^2 (a2 - 1) + 1
          const f32x2 t2 = t * t;
          // divide_no_nan (microfacet.py:57, :71-72): both denominators vanish only together with their numerators (t = 0 needs
          // a2 = 0; cl + sl = 0 needs cl = 0 and a2 = 0, where D = 0 too), so a floor on the denominator gives the same 0
          const f32x2 D = splat2(M[s].a2_pi) * rcp2(max2(t2, 1e-30f));
          const f32x2 sl = sqrt2(abs2(fma2(splat2(M[s].oma2), cl2, splat2(M[s].a2))));
          const f32x2 gdi = D * splat2(M[s].kv) * rcp2(max2(cl + sl, 1e-30f));
          const f32x2 g0 = fma2(splat2(M[s].omf0[0]), om5, splat2(M[s].f0[0])) * gdi, g1 = fma2(splat2(M[s].omf0[1]), om5, splat2(M[s].f0[1])) * gdi,
                      g2 = fma2(splat2(M[s].omf0[2]), om5, splat2(M[s].f0[2])) * gdi;
          S2[s][0] = fma2(g0, wr, S2[s][0]); S2[s][1] = fma2(g1, wg, S2[s][1]); S2[s][2] = fma2(g2, wb, S2[s][2]);
          if (PROBES && s == 0) {
            const f32x2 gw = cw * area_k[j];
            const f32x2 p0 = (g0 + splat2(M[0].a[0])) * gw, p1 = (g1 + splat2(M[0].a[1])) * gw, p2 = (g2 + splat2(M[0].a[2])) * gw;
            wsp[2 * j][0] = p0[0]; wsp[2 * j][1] = p1[0]; wsp[2 * j][2] = p2[0];
            wsp[2 * j + 1][0] = p0[1]; wsp[2 * j + 1][1] = p1[1]; wsp[2 * j + 1][2] = p2[1];
          }
        }
    }
    float S[2][3], W[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { S[0][c] = S2[0][c][0] + S2[0][c][1]; S[1][c] = S2[1][c][0] + S2[1][c][1]; W[c] = W2[c][0] + W2[c][1]; }
    {
      // the (up to) 12 sums over lights of this point -- rgb of each material set, and set 0's diffuse / specular split: each is
      // reduced over the wave, then lane i takes value i and lanes 0..11 apply the gamma curve / clip to their own value (ONE powf
      // per wave instead of twelve: the curve of the 'dtu' / 'hw' data types cost half as much as the shading itself) and store it
      float u[16];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        u[c] = fmaf(M[0].a[c], W[c], S[0][c]);
        u[3 + c] = NS > 1 ? fmaf(M[1].a[c], W[c], S[1][c]) : 0.f;
        u[6 + c] = split ? M[0].a[c] * W[c] : 0.f;
        u[9 + c] = split ? S[0][c] : 0.f;
      }
      u[12] = u[13] = u[14] = u[15] = 0.f;
      float t = wave_sum16(u, lane);                                 // lane l: value (l >> 2) & 15
      if (!a.raw) {
        if (a.gamma) t = powf(t * gam_b, gam_i);
        t = clip01(t);
      } else if (a.raw == 2) {
        t = __fadd_rn(t, __fsub_rn(clip01(t), t));                   // clip_by_value_preserve_gradient's own arithmetic (vqn_clip_preserve)
      }
      // (one store per output array, each from a uniform base pointer)
      const int vi = lane >> 2;
      const bool first = (lane & 3) == 0;
      if (first && vi < 3) a.rgb[0][n * 3 + vi] = t;
      if (NS > 1 && first && vi >= 3 && vi < 6) a.rgb[1][n * 3 + vi - 3] = t;
      if (split && first && vi >= 6 && vi < 9) a.rgb_diff[n * 3 + vi - 6] = t;
      if (split && first && vi >= 9 && vi < 12) a.rgb_spec[n * 3 + vi - 9] = t;
    }
    if (PROBES) {
      // all probes against the SAME per-light weights: the [N,L] x [L,3P] contraction of the relighting loop, 16 probes at a time.
      // Every lane first sums its own lights into 48 accumulators (probe, channel); the 64-lane reduction of all 48 is then ONE
      // halving butterfly (64 -> 32 -> ... -> 1 values per lane, 63 shuffles) instead of 48 separate 6-step reductions (288), and it
      // leaves lane i with the total of value i = 3 probe + channel -- the order they lie in memory: one coalesced 192 B store.
      for (int pb = 0; pb < a.n_probes; pb += 16) {
        float v[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = 0.f;
#pragma unroll
        for (int pi = 0; pi < 16; ++pi) {
          const int pr = min(pb + pi, a.n_probes - 1);         // (clamped: the surplus sums of a ragged last chunk are not stored)
#pragma unroll
          for (int g = 0; g < LQ; ++g) {
            f32x4 q0, q1, q2;                                    // 4 lights x rgb, interleaved
            if (PLDS) {
              const f32x4* pt = ptab + (size_t)(pr * LQ + g) * 3 * 64 + lane;
              q0 = pt[0]; q1 = pt[64]; q2 = pt[128];
            } else {
              const f32x4* pp = reinterpret_cast<const f32x4*>(a.probes + ((size_t)pr * L + 256 * g + 4 * lane) * 3);
              q0 = pp[0]; q1 = pp[1]; q2 = pp[2];
            }
            const float rad[12] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3], q2[0], q2[1], q2[2], q2[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = 4 * g + e;
              v[3 * pi + 0] = fmaf(wsp[k][0], rad[3 * e + 0], v[3 * pi + 0]);
              v[3 * pi + 1] = fmaf(wsp[k][1], rad[3 * e + 1], v[3 * pi + 1]);
              v[3 * pi + 2] = fmaf(wsp[k][2], rad[3 * e + 2], v[3 * pi + 2]);
            }
          }
        }
#pragma unroll
        for (int half = 32; half >= 1; half >>= 1) {
          const bool up = (lane & half) != 0;                  // this lane keeps the upper / lower half of its values, sends the other
#pragma unroll
          for (int i = 0; i < half; ++i) {
            const float keep = up ? v[i + half] : v[i];
            const float send = up ? v[i] : v[i + half];
            v[i] = keep + __shfl_xor(send, half);
          }
        }
        float t = v[0];                                        // total of value `lane`
        if (!a.raw) {
          if (a.gamma) t = powf(t * gam_b, gam_i);
          t = clip01(t);
        }
        const int n_here = min(16, a.n_probes - pb) * 3;
        if (lane < n_here) a.rgb_probes[((size_t)n * a.n_probes + pb) * 3 + lane] = t;
      }
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

// d/d a2 of G1(c) = 2c / (c + sqrt|a2 + (1 - a2) c^2|), with divide_no_nan semantics.  Two quarter-rate instructions (rsq, rcp): the
// root and its reciprocal both come from one v_rsq (round 4: the backward's inner loop was bound by its 14 transcendentals per light)
__device__ __forceinline__ void g1_and_da2(float c, float a2, float* g1, float* dg1) {
  const float q = a2 + (1.f - a2) * c * c;
  const float aq = fabsf(q);
  const float rs = aq > 0.f ? __builtin_amdgcn_rsqf(aq) : 0.f;          // 1 / sqrt|q|   (0 where the root is 0)
  const float s = aq * rs;                                              // sqrt|q|
  const float den = c + s;
  if (den == 0.f) { *g1 = 0.f; *dg1 = 0.f; return; }
  const float rden = __builtin_amdgcn_rcpf(den);
  *g1 = 2.f * c * rden;
  const float ds = (q >= 0.f ? 0.5f : -0.5f) * (1.f - c * c) * rs;
  *dg1 = -2.f * c * rden * rden * ds;
}

template <int LQ>
__global__ __launch_bounds__(256) void brdf_shade_bwd_kernel(const ShadeBwdArgs a) {
  constexpr int LP = 4 * LQ;
  constexpr int L = 64 * LP;
  const int lane = threadIdx.x & 63;
  const long wave_id = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: per-point scalars by scalar loads
  const long n_waves = (long)gridDim.x * 4;
  constexpr int LH = LP / 2;                       // light pairs per lane (see the forward kernel)
  f32x2 lx[LH], ly[LH], lz[LH], area[LH], Lr[LH], Lg[LH], Lb[LH], gL[LH][3];
#pragma unroll
  for (int g = 0; g < LQ; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int li = 256 * g + 4 * lane + e, k = 4 * g + e, j = k >> 1, o = k & 1;
      lx[j][o] = a.lxyz[li * 3 + 0]; ly[j][o] = a.lxyz[li * 3 + 1]; lz[j][o] = a.lxyz[li * 3 + 2];
      area[j][o] = a.lareas[li];
      Lr[j][o] = a.light[li * 3 + 0]; Lg[j][o] = a.light[li * 3 + 1]; Lb[j][o] = a.light[li * 3 + 2];
      gL[j][0][o] = gL[j][1][o] = gL[j][2][o] = 0.f;
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
    const float avn = fabsf(v_dot_n);
    float alb_pi[2][3], f0[2][3], a2[2], g1v[2], dg1v[2], gs[2][3], rgh[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (s < a.n_sets) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          alb_pi[s][c] = a.albedo[s][n * 3 + c] / PI_F;           // (per point: an IEEE division is ten instructions, and this one sat in the light loop)
          f0[s][c] = a.spec[s][n * 3 + c]; gs[s][c] = a.g_sum[s][n * 3 + c];
        }
        rgh[s] = a.rough[s][n];
        const float alpha = rgh[s] * rgh[s];
        a2[s] = alpha * alpha;
        g1_and_da2(cv, a2[s], &g1v[s], &dg1v[s]);
      }
    f32x2 acc_alb[2][3], acc_f0[2][3], acc_a2[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      acc_a2[s] = splat2(0.f);
#pragma unroll
      for (int c = 0; c < 3; ++c) { acc_alb[s][c] = splat2(0.f); acc_f0[s][c] = splat2(0.f); }
    }
#pragma unroll
    for (int j = 0; j < LH; ++j) {
      f32x2 dx = lx[j] - splat2(px), dy = ly[j] - splat2(py), dz = lz[j] - splat2(pz);
      const f32x2 il = inv_norm2(dx, dy, dz);
      dx *= il; dy *= il; dz *= il;
      const f32x2 cosl = dot2(dx, dy, dz, nx, ny, nz);
      const f32x2 visj = {vis4[j >> 1][2 * (j & 1)], vis4[j >> 1][2 * (j & 1) + 1]};
      const f32x2 vis = {cosl[0] > 0.f ? visj[0] : 0.f, cosl[1] > 0.f ? visj[1] : 0.f};
      // (the forward normalises the light direction a second time, as the reference does -- microfacet.py:18 on top of shape.py:103-119;
      //  a unit vector's second normalisation moves it by an ulp, which the GRADIENT does not need: one rsq and eight instructions less)
      f32x2 hx = dx + splat2(ux), hy = dy + splat2(uy), hz = dz + splat2(uz);
      const f32x2 ih = inv_norm2(hx, hy, hz);
      hx *= ih; hy *= ih; hz *= ih;
      const f32x2 cos_vh = clip01_2(dot2(hx, hy, hz, ux, uy, uz));
      const f32x2 om = splat2(1.f) - cos_vh, om2 = om * om, om5 = om2 * om2 * om;
      const f32x2 omo5 = splat2(1.f) - om5;
      const f32x2 cos_m = clip01_2(dot2(hx, hy, hz, mx, my, mz));
      const f32x2 cm2 = cos_m * cos_m;
      const f32x2 l_dot_n = dot2(dx, dy, dz, mx, my, mz);
      const f32x2 cl = clip01_2(l_dot_n);
      const f32x2 den = splat2(4.f) * abs2(l_dot_n) * splat2(avn);
      const f32x2 inv_den = {den[0] == 0.f ? 0.f : __builtin_amdgcn_rcpf(den[0]), den[1] == 0.f ? 0.f : __builtin_amdgcn_rcpf(den[1])};
      const f32x2 wgt = vis * cosl * area[j];                      // geometry weight of this light (without radiance)
      const f32x2 Lc[3] = {Lr[j], Lg[j], Lb[j]};
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (s < a.n_sets) {
          const f32x2 t = fma2(cm2, splat2(a2[s] - 1.f), splat2(1.f));
          // (t = 0: D = dD = 0, as the reference's divide_no_nan -- a zero reciprocal gives both)
          const f32x2 rt = {t[0] != 0.f ? __builtin_amdgcn_rcpf(t[0]) : 0.f, t[1] != 0.f ? __builtin_amdgcn_rcpf(t[1]) : 0.f};
          const f32x2 rpd = rt * rt * splat2(1.f / PI_F);          // 1 / (pi t^2): one reciprocal for D and its derivative
          const f32x2 D = splat2(a2[s]) * rpd;
          const f32x2 dD = fma2(splat2(-2.f * a2[s]), cm2, t) * rpd * rt;
          // G1(l.n) and its derivative in a2 (g1_and_da2, both halves)
          const f32x2 q = fma2(splat2(1.f - a2[s]) * cl, cl, splat2(a2[s]));
          const f32x2 aq = abs2(q);
          const f32x2 rs = {aq[0] > 0.f ? __builtin_amdgcn_rsqf(aq[0]) : 0.f, aq[1] > 0.f ? __builtin_amdgcn_rsqf(aq[1]) : 0.f};
          const f32x2 dn = fma2(aq, rs, cl);                       // c + sqrt|q|
          const f32x2 rden = {dn[0] == 0.f ? 0.f : __builtin_amdgcn_rcpf(dn[0]), dn[1] == 0.f ? 0.f : __builtin_amdgcn_rcpf(dn[1])};
          const f32x2 g1l = splat2(2.f) * cl * rden;
          const f32x2 hs = {q[0] >= 0.f ? 0.5f : -0.5f, q[1] >= 0.f ? 0.5f : -0.5f};
          const f32x2 ds = hs * fma2(-cl, cl, splat2(1.f)) * rs;
          const f32x2 dg1l = splat2(-2.f) * cl * rden * rden * ds;
          const f32x2 G = g1l * splat2(g1v[s]);
          const f32x2 dG = fma2(dg1l, splat2(g1v[s]), g1l * splat2(dg1v[s]));
          const f32x2 gd = G * D * inv_den, dgd = fma2(dG, D, G * dD) * inv_den;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const f32x2 F = fma2(splat2(1.f - f0[s][c]), om5, splat2(f0[s][c]));
            const f32x2 gw = splat2(gs[s][c]) * wgt;               // d loss / d (brdf_c * radiance_c)  per unit radiance
            const f32x2 glw = gw * Lc[c];
            acc_alb[s][c] = fma2(glw, splat2(1.f / PI_F), acc_alb[s][c]);
            acc_f0[s][c] = fma2(glw * omo5, gd, acc_f0[s][c]);
            acc_a2[s] = fma2(glw * F, dgd, acc_a2[s]);
            gL[j][c] = fma2(gw, fma2(F, gd, splat2(alb_pi[s][c])), gL[j][c]);
          }
        }
    }
    {
      // the 14 sums over lights (albedo, f0: 3 each, a2: 1, per set): one butterfly; lane l gets value (l >> 2) & 15 = 7 s + i
      float u[16];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bool on = s < a.n_sets;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          u[7 * s + c] = on ? acc_alb[s][c][0] + acc_alb[s][c][1] : 0.f;
          u[7 * s + 3 + c] = on ? acc_f0[s][c][0] + acc_f0[s][c][1] : 0.f;
        }
        u[7 * s + 6] = on ? (acc_a2[s][0] + acc_a2[s][1]) : 0.f;
      }
      u[14] = u[15] = 0.f;
      const float t = wave_sum16(u, lane);
      const int vi = lane >> 2, s_ = vi >= 7 ? 1 : 0, i = vi - 7 * s_;
      if ((lane & 3) == 0 && vi < 14 && s_ < a.n_sets) {
        if (i < 3) a.g_albedo[s_][n * 3 + i] = t;
        else if (i < 6) a.g_spec[s_][n * 3 + i - 3] = t;
        else a.g_rough[s_][n] = t * 4.f * rgh[s_] * rgh[s_] * rgh[s_];                          // a2 = rough^4
      }
    }
  }
  float* part = a.g_light_part + (size_t)wave_id * L * 3;
#pragma unroll
  for (int g = 0; g < LQ; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int li = 256 * g + 4 * lane + e, k = 4 * g + e;
      part[li * 3 + 0] = gL[k >> 1][0][k & 1]; part[li * 3 + 1] = gL[k >> 1][1][k & 1]; part[li * 3 + 2] = gL[k >> 1][2][k & 1];
    }
}

}  // namespace

extern "C" int vqn_brdf_shade_fwd(const float* xyz, const float* normal, const float* rayo, const float* lvis,
                                  const float* lxyz, const float* lareas, const float* light, int64_t N, int L,
                                  int n_sets, const float* albedo0, const float* spec0, const float* rough0,
                                  const float* albedo1, const float* spec1, const float* rough1, const float* gamma,
                                  float* normal_out, float* rgb0, float* rgb1, float* rgb0_diff, float* rgb0_spec,
                                  int raw, const float* probes, int n_probes, float* rgb0_probes, void* stream) {
  return vqn_brdf_shade_fwd_rows(nullptr, xyz, normal, rayo, lvis, lxyz, lareas, light, N, L, n_sets, albedo0, spec0, rough0, albedo1,
                                 spec1, rough1, gamma, normal_out, rgb0, rgb1, rgb0_diff, rgb0_spec, raw, probes, n_probes, rgb0_probes,
                                 stream);
}

extern "C" int vqn_brdf_shade_fwd_rows(const int64_t* lvis_rows, const float* xyz, const float* normal, const float* rayo, const float* lvis,
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
  VQN_CHECK_ARG(raw >= 0 && raw <= 2, "raw: 0 (gamma / clip as configured), 1 (no clip), 2 (no gamma, no clip)");
  VQN_CHECK_ARG(raw == 0 || gamma == nullptr, "raw = 1 / 2 write the plain sums and ignore gamma: pass NULL");
  VQN_CHECK_SHAPE(L == 256 || L == 512 || L == 1024, "L must be 256, 512 or 1024 lights");
  VQN_CHECK_SHAPE(lvis == nullptr || ((uintptr_t)lvis & 15) == 0, "lvis must be 16-byte aligned");
  ShadeArgs a;
  a.xyz = xyz; a.normal = normal; a.rayo = rayo; a.lvis = lvis; a.lxyz = lxyz; a.lareas = lareas; a.light = light;
  a.gamma = gamma;
  a.albedo[0] = albedo0; a.spec[0] = spec0; a.rough[0] = rough0;
  a.albedo[1] = albedo1; a.spec[1] = spec1; a.rough[1] = rough1;
  a.normal_out = normal_out; a.rgb[0] = rgb0; a.rgb[1] = rgb1; a.rgb_diff = rgb0_diff; a.rgb_spec = rgb0_spec;
  a.N = N; a.n_sets = n_sets; a.raw = raw;
  a.rows = lvis ? reinterpret_cast<const long long*>(lvis_rows) : nullptr;
  VQN_CHECK_ARG(probes == nullptr || (n_probes >= 1 && rgb0_probes != nullptr), "probes need n_probes >= 1 and rgb0_probes");
  VQN_CHECK_SHAPE(probes == nullptr || ((uintptr_t)probes & 15) == 0, "probes must be 16-byte aligned");
  a.probes = probes; a.n_probes = probes ? n_probes : 0; a.rgb_probes = rgb0_probes;
  long blocks = (N + 3) / 4;
  const long cap = (long)vqn_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipStream_t s = (hipStream_t)stream;
  const bool pr = a.probes != nullptr;
  // relighting with the probe table in LDS: one 768-thread workgroup per CU (three waves per SIMD at 168 registers)
  const size_t plds = pr ? (size_t)n_probes * (L / 256) * 3 * 1024 : 0;
  static const int no_plds = [] { const char* e = getenv("VQN_SHADE_NO_PLDS"); return (e != nullptr && atoi(e) != 0) ? 1 : 0; }();
  const bool use_plds = pr && plds <= 144 * 1024 && !no_plds && !(L == 1024 && n_sets == 2);      // (that one form would spill)
  const int plds_waves = L == 1024 ? 8 : 12;
  long blocks8 = (N + plds_waves - 1) / plds_waves;
  if (blocks8 > (long)vqn_num_cus()) blocks8 = vqn_num_cus();
#define VQN_SHADE_PLDS(LQ, NS_)                                                                                                    \
  do {                                                                                                                             \
    VQN_HIP(hipFuncSetAttribute((const void*)brdf_shade_kernel<LQ, true, NS_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds)); \
    hipLaunchKernelGGL((brdf_shade_kernel<LQ, true, NS_, true>), dim3((unsigned)blocks8), dim3(64 * plds_waves), plds, s, a);                    \
  } while (0)
#define VQN_SHADE_LAUNCH(LQ)                                                                                   \
  do {                                                                                                         \
    if (use_plds && n_sets == 1) VQN_SHADE_PLDS(LQ, 1);                                                        \
    else if (use_plds) VQN_SHADE_PLDS(LQ, 2);                                                                  \
    else if (pr && n_sets == 1) hipLaunchKernelGGL((brdf_shade_kernel<LQ, true, 1>), dim3((unsigned)blocks), dim3(256), 0, s, a);   \
    else if (pr) hipLaunchKernelGGL((brdf_shade_kernel<LQ, true, 2>), dim3((unsigned)blocks), dim3(256), 0, s, a);                  \
    else if (n_sets == 1) hipLaunchKernelGGL((brdf_shade_kernel<LQ, false, 1>), dim3((unsigned)blocks), dim3(256), 0, s, a);        \
    else hipLaunchKernelGGL((brdf_shade_kernel<LQ, false, 2>), dim3((unsigned)blocks), dim3(256), 0, s, a);                         \
  } while (0)
  if (L == 256) VQN_SHADE_LAUNCH(1);
  else if (L == 512) VQN_SHADE_LAUNCH(2);
  else VQN_SHADE_LAUNCH(4);
#undef VQN_SHADE_LAUNCH
#undef VQN_SHADE_PLDS
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

// ---- display transfer of the rendered colours (nerfactor/util/img.py:142-186: clip to [0, 1], then the piecewise sRGB curve) as one
// pass instead of the seven elementwise framework launches of its torch statement (clamp, compare, mul, pow, mul, sub, where)
namespace {
__global__ __launch_bounds__(256) void linear2srgb_kernel(const float* __restrict__ x, const long n, float* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    const float t = fminf(fmaxf(v, 0.0f), 1.0f);         // fminf / fmaxf drop a NaN: put it back, as the clip of the statement does
    const float r = t <= 0.0031308f ? t * 12.92f : 1.055f * powf(t, 1.0f / 2.4f) - (1.055f - 1.0f);
    y[i] = (v != v) ? v : r;
  }
}
}  // namespace

extern "C" int vqn_linear2srgb(const float* x, int64_t n, float* y, void* stream) {
  VQN_CHECK_ARG(n >= 0, "n >= 0");
  if (n == 0) return VQN_OK;
  VQN_CHECK_ARG(x && y, "x and y must be non-null");
  long blocks = (n + 255) / 256;
  const long cap = (long)vqn_num_cus() * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(linear2srgb_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, y);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
