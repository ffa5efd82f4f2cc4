// VQ codebook nearest-neighbour assignment + EMA statistics for gfx950 (MI355X).
//
// Replaces the framework-op sequences at
//   decomp/nerfvq_nfr3/nerfactor/networks/vq_layers.py:277-301,346-349  (distances, dropout mask, argmax(-d), lookup)
//   decomp/nerfvq_nfr3/nerfactor/networks/vq_layers.py:304-309          (sum(enc,0), x^T @ enc)
//
// vq_assign: HBM-bound (1 KB/row in at D=256).  One wave owns 16 rows; the dot products run on the
// f32 matrix pipe (v_mfma_f32_16x16x4_f32), whose result is bit-for-bit a k-ordered fmaf chain, so
// the summation order is *defined* (oracle/vq_strict.c states the same order in plain C):
//     dot[n][k] : fmaf chain over d = 16t + 4q + e, (t outer, e, q inner)
//     x2[n]     : four chains p_q, then (p0+p1)+(p2+p3)      (two xor-shuffles)
//     c2[k]     : fmaf chain over d = 0..D-1
//     dist      : (x2 - 2 dot) + c2 ; argmin, lowest index wins ties.
// The codebook lives in LDS already laid out as MFMA B-fragments (one ds_read_b128 per 4 MFMAs).
#include "common.h"
#include "vqnerf_hip.h"
#include <math.h>

namespace {

__device__ __forceinline__ int f2key(float f) {
  int b = __float_as_int(f);
  return b >= 0 ? b : (b ^ 0x7fffffff);
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : (k ^ 0x7fffffff)); }

// FUSE (the inference path of vq_nfr.Model: vq_nfr.py:575-578 + vq_layers.py:277-302, :327 in one pass over the rows, D <= 256):
// the rows arrive UN-normalised and are l2-normalised in registers first (util/math.py:63-64; arithmetic of
// l2_normalize_rows_kernel below, so that this kernel and the three-kernel sequence normalise -> assign -> ste agree bit for
// bit); `quant` then receives the straight-through output x^ + (q - x^) instead of q, `loss_part[block]` the block's share of
// sum (q - x^)^2 (fixed order) and `counts[k]` the code usage (integer adds: exact in any order).
struct VqFuse {
  float eps;
  float* loss_part;
  float* counts;
  float* xnorm;          // optional [N, D]: the l2-normalised rows (the training path keeps them: EMA statistics, backward)
};

template <int KT, bool MAXONLY, bool FUSE>
__global__ __launch_bounds__(256) void vq_assign_kernel(const float* __restrict__ x, long N, int D,
                                                        const float* __restrict__ C, int K,
                                                        const float* __restrict__ sel, int* __restrict__ gmax_key,
                                                        long long* __restrict__ idx, float* __restrict__ quant,
                                                        float* __restrict__ dist, const VqFuse fuse) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D16 = (D + 15) >> 4;
  f32x4* Bf = reinterpret_cast<f32x4*>(smem);             // [KT][D16][64] float4
  float* c2 = smem + (size_t)KT * D16 * 256;              // [KT*16]
  float* selm = c2 + KT * 16;                             // [KT*16]  1 = keep, 0 = dropped
  int* hist = reinterpret_cast<int*>(selm + KT * 16);     // FUSE: [KT*16] code usage of this workgroup, then 4 floats of wave sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, q = lane >> 4;

  // ---- stage the codebook as B fragments: Bf[kt][t][l][e] = C[16t + 4(l>>4) + e][16kt + (l&15)] ----
  for (int i = tid; i < KT * D16 * 64; i += 256) {
    int l = i & 63, t = (i >> 6) % D16, kt = (i >> 6) / D16;
    int code = 16 * kt + (l & 15);
    f32x4 v;
    for (int e = 0; e < 4; ++e) {
      int d = 16 * t + 4 * (l >> 4) + e;
      v[e] = (d < D && code < K) ? C[(size_t)d * K + code] : 0.0f;
    }
    Bf[i] = v;
  }
  __syncthreads();
  // c2[k]: the fmaf chain over d = 0..D-1 read back from the staged fragments (LDS latency; D dependent global loads per
  // thread made this prologue a fifth of the launch at 1 M rows)
  for (int k = tid; k < KT * 16; k += 256) {
    float acc = 0.f;
    if (k < K) {
      const f32x4* col_k = Bf + (size_t)(k >> 4) * D16 * 64 + (k & 15);
      for (int d4 = 0; 4 * d4 < D; ++d4) {
        const f32x4 c = col_k[(d4 >> 2) * 64 + (d4 & 3) * 16];
        acc = fmaf(c[0], c[0], acc);
        if (4 * d4 + 1 < D) acc = fmaf(c[1], c[1], acc);
        if (4 * d4 + 2 < D) acc = fmaf(c[2], c[2], acc);
        if (4 * d4 + 3 < D) acc = fmaf(c[3], c[3], acc);
      }
    }
    c2[k] = acc;
    selm[k] = (k < K) ? ((sel != nullptr && !MAXONLY) ? sel[k] : 1.0f) : 0.0f;
    if (FUSE) hist[k] = 0;
  }
  __syncthreads();
  float wave_loss = 0.f;                                  // FUSE: this wave's rows, summed group by group

  const float gmax = (!MAXONLY && sel != nullptr) ? key2f(*gmax_key) : 0.f;
  float wmax = -INFINITY;
  const long n_groups = (N + 15) >> 4;
  for (long rg = (long)blockIdx.x * 4 + wave; rg < n_groups; rg += (long)gridDim.x * 4) {
    const long row0 = rg << 4;
    const bool rvalid = (row0 + col) < N;
    const float* xr = x + (size_t)(rvalid ? row0 + col : N - 1) * D;          // clamped: every fetch below is in bounds
    f32x4 acc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float p = 0.f;
    // the row's fragments are requested 16 K-steps (one 256-feature row) at a time, all of them before the first use and
    // unconditionally (clamped offsets, zeroed afterwards where out of range): the stream is HBM-bound, so what matters is how
    // many fetches are in flight -- a guarded fetch per step makes the compiler wait for each one before the next is issued.
    // (Requesting the next group's rows before this group is multiplied was measured: 2x the registers, half the waves, -10 %.)
    f32x4 av[16];                                        // (FUSE: D16 <= 16, the one chunk stays live for the straight-through output)
    for (int t0 = 0; t0 < D16; t0 += 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int off = min(16 * (t0 + i) + 4 * q, D - 4);
        av[i] = *reinterpret_cast<const f32x4*>(xr + off);
      }
      if (FUSE) {
        // x^ = x / sqrt(max(sum x^2, eps)) with sum x^2 in the order of x2 below (four chains, then (p0 + p1) + (p2 + p3))
        float pr = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (!(rvalid && i < D16 && (16 * i + 4 * q) < D)) av[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
          pr = fmaf(av[i][0], av[i][0], pr); pr = fmaf(av[i][1], av[i][1], pr); pr = fmaf(av[i][2], av[i][2], pr); pr = fmaf(av[i][3], av[i][3], pr);
        }
        pr = pr + __shfl_xor(pr, 16);
        pr = pr + __shfl_xor(pr, 32);
        const float sc = 1.0f / sqrtf(fmaxf(pr, fuse.eps));
#pragma unroll
        for (int i = 0; i < 16; ++i) av[i] = av[i] * sc;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int t = t0 + i;
        if (t < D16) {
          f32x4 a = av[i];
          if (!(rvalid && (16 * t + 4 * q) < D)) a = (f32x4){0.f, 0.f, 0.f, 0.f};
          p = fmaf(a[0], a[0], p); p = fmaf(a[1], a[1], p); p = fmaf(a[2], a[2], p); p = fmaf(a[3], a[3], p);
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            f32x4 b = Bf[((size_t)kt * D16 + t) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[kt], 0, 0, 0);
          }
        }
      }
    }
    // x2 for row (lane&15): (p0+p1)+(p2+p3)
    p = p + __shfl_xor(p, 16);
    p = p + __shfl_xor(p, 32);
    float best_v[4];
    int best_i[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x2 = __shfl(p, 4 * q + j);          // acc reg j of this lane is row 4q+j
      const bool row_ok = (row0 + 4 * q + j) < N;
      best_v[j] = INFINITY;
      best_i[j] = 0x7fffffff;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int code = 16 * kt + col;
        float t1 = x2 - 2.0f * acc[kt][j];
        float dv = t1 + c2[code];
        if (MAXONLY) {
          if (row_ok && code < K) wmax = fmaxf(wmax, dv);
        } else {
          if (selm[code] == 0.0f) dv = gmax;
          if (code < K) {
            if (dist != nullptr && row_ok) dist[(size_t)(row0 + 4 * q + j) * K + code] = dv;
            if (dv < best_v[j] || best_i[j] == 0x7fffffff) { best_v[j] = dv; best_i[j] = code; }
          }
        }
      }
    }
    if (!MAXONLY) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
          float ov = __shfl_xor(best_v[j], m);
          int oi = __shfl_xor(best_i[j], m);
          bool take = (oi != 0x7fffffff) && (best_i[j] == 0x7fffffff || ov < best_v[j] || (ov == best_v[j] && oi < best_i[j]));
          if (take) { best_v[j] = ov; best_i[j] = oi; }
        }
        if (col == 0 && (row0 + 4 * q + j) < N) idx[row0 + 4 * q + j] = (long long)best_i[j];
      }
      if (FUSE) {
        // this lane's row is `col`; its nearest code sits in the lanes of quarter col >> 2, slot col & 3
        int kr = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kj = __shfl(best_i[j], (col >> 2) * 16);
          if ((col & 3) == j) kr = kj;
        }
        const int kt_r = kr >> 4, kc_r = kr & 15;
        float lr = 0.f;                                   // sum over this lane's 64 features of (q - x^)^2, one fmaf chain
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < D16 && (16 * i + 4 * q) < D) {
            const f32x4 cq = Bf[((size_t)kt_r * D16 + i) * 64 + 16 * q + kc_r];        // C[16 i + 4 q + e][kr], e = 0..3
            const f32x4 dq = cq - av[i];
            lr = fmaf(dq[0], dq[0], lr); lr = fmaf(dq[1], dq[1], lr); lr = fmaf(dq[2], dq[2], lr); lr = fmaf(dq[3], dq[3], lr);
            if (rvalid && quant != nullptr) *reinterpret_cast<f32x4*>(quant + (size_t)(row0 + col) * D + 16 * i + 4 * q) = av[i] + dq;
            if (rvalid && fuse.xnorm != nullptr) *reinterpret_cast<f32x4*>(fuse.xnorm + (size_t)(row0 + col) * D + 16 * i + 4 * q) = av[i];
          }
        }
        if (!rvalid) lr = 0.f;
        lr = lr + __shfl_xor(lr, 16);                     // the row's four quarter sums: (l0 + l1) + (l2 + l3)
        lr = lr + __shfl_xor(lr, 32);
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) lr += __shfl_xor(lr, m);      // the group's 16 rows, fixed xor tree
        wave_loss += lr;
        if (q == 0 && rvalid) atomicAdd(&hist[kr], 1);
      } else if (quant != nullptr) {
        const int nchunk = (D + 255) >> 8;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (row0 + r >= N) continue;                                // wave-uniform
          const int k = __shfl(best_i[r & 3], (r >> 2) * 16);
          const int kt = k >> 4, kc = k & 15;
          for (int c = 0; c < nchunk; ++c) {
            const int d4 = lane + 64 * c;                            // float4 index along D
            if (4 * d4 < D) {
              f32x4 v = Bf[((size_t)kt * D16 + (d4 >> 2)) * 64 + (d4 & 3) * 16 + kc];
              *reinterpret_cast<f32x4*>(quant + (size_t)(row0 + r) * D + 4 * d4) = v;
            }
          }
        }
      }
    }
  }
  if (MAXONLY) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, m));
    if (lane == 0 && wmax > -INFINITY) atomicMax(gmax_key, f2key(wmax));
  }
  if (FUSE && !MAXONLY) {
    float* wsum = reinterpret_cast<float*>(hist + KT * 16);
    if (lane == 0) wsum[wave] = wave_loss;
    __syncthreads();
    if (tid == 0) fuse.loss_part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    for (int k = tid; k < K; k += 256)
      if (hist[k]) atomicAdd(&fuse.counts[k], (float)hist[k]);       // integers below 2^24: exact in any order
  }
}

// ---- K in 17..64: the f32 matrix pipe, not HBM, bounds vq_assign_kernel (2 D K FLOP per 4 D bytes: 164 TFLOP/s at 5 TB/s for K = 64).
// This kernel gets the SAME indices (and straight-through rows) from 1/5 of the matrix-pipe time:
//   1. approximate distances on the f16 pipe with f32-level accuracy: every operand as an f16 pair (hi = f16(v), lo = f16((v - hi) 2^11)),
//      dot ~ hi.hi + 2^-11 (hi.lo + lo.hi) on v_mfma_f32_16x16x32_f16 with f32 accumulation -- |error| <= 3.3e-5 sqrt(|x|^2 |c|^2)
//      (operand split 2^-22 per factor, the dropped lo.lo term, two f32 summations of D <= 256 terms);
//   2. every code whose approximate distance lies within a margin M of the row's minimum is a CANDIDATE, M = twice the largest
//      possible |approximate - defined| distance error: no other code can be the argmin (or tie with it) in the defined arithmetic;
//   3. a row with one candidate is decided; the candidates of the other rows are evaluated EXACTLY, in the defined order (the f32 MFMA
//      chain of vq_assign_kernel, 16x16x4 passes whose B columns are the UNION of the group's undecided rows' candidates: one pass for
//      up to 16 codes, two for up to 32 -- round 5; rounds 2-4: one pass per candidate rank with the result on the diagonal), and
//      compared as (distance, index) -- ties to the lowest index, as tf.argmax(-d).  More than 32 codes in a group's union (or
//      non-finite inputs): the whole group takes the plain f32 path.
// Bit-identical to vq_assign_kernel / oracle/vq_strict.c by construction; D <= 256, no code-dropout mask, no distance output
// (those calls keep the f32 kernel).  One 512-thread workgroup per CU (codebook as f32 + f16 hi / lo fragments: 32 KB per 16 codes).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// DPP lane exchanges inside a row of 16 lanes (no LDS round trip, unlike ds_bpermute): xor 1, xor 2, mirror of 8, mirror of 16 --
// applied in this order they leave the reduction over the 16 lanes in every lane.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
__device__ __forceinline__ float row16_min(float v) {
  v = fminf(v, dpp_f<0xB1>(v)); v = fminf(v, dpp_f<0x4E>(v)); v = fminf(v, dpp_f<0x141>(v)); v = fminf(v, dpp_f<0x140>(v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v = v + dpp_f<0xB1>(v); v = v + dpp_f<0x4E>(v); v = v + dpp_f<0x141>(v); v = v + dpp_f<0x140>(v);
  return v;
}

template <int KT, bool FUSE>
__global__ __launch_bounds__(512, 1) void vq_assign_split_kernel(const float* __restrict__ x, long N, int D, const float* __restrict__ C,
                                                                 int K, long long* __restrict__ idx, float* __restrict__ quant,
                                                                 const VqFuse fuse) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D16 = (D + 15) >> 4;                                  // <= 16
  f32x4* Bf = reinterpret_cast<f32x4*>(smem);                     // [KT][D16][64] float4: the f32 fragments of vq_assign_kernel
  f16x8* Bh = reinterpret_cast<f16x8*>(Bf + (size_t)KT * D16 * 64);   // [KT][8][64] x 8 halves: code 16 kt + (l & 15), d = 32 m + 16 (jj >> 2) + 4 (l >> 4) + (jj & 3)
  f16x8* Bl = Bh + KT * 8 * 64;
  float* c2 = reinterpret_cast<float*>(Bl + KT * 8 * 64);         // [KT*16] + [1] max
  int* hist = reinterpret_cast<int*>(c2 + KT * 16 + 4);           // [KT*16], then 8 floats of wave sums
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, q = lane >> 4;

  for (int i = tid; i < KT * D16 * 64; i += 512) {
    const int l = i & 63, t = (i >> 6) % D16, kt = (i >> 6) / D16;
    const int code = 16 * kt + (l & 15);
    f32x4 v;
    for (int e = 0; e < 4; ++e) {
      const int d = 16 * t + 4 * (l >> 4) + e;
      v[e] = (d < D && code < K) ? C[(size_t)d * K + code] : 0.0f;
    }
    Bf[i] = v;
  }
  for (int i = tid; i < KT * 8 * 64; i += 512) {
    const int l = i & 63, m = (i >> 6) & 7, kt = i >> 9;
    const int code = 16 * kt + (l & 15);
    f16x8 hi, lo;
    for (int jj = 0; jj < 8; ++jj) {
      const int d = 32 * m + 16 * (jj >> 2) + 4 * (l >> 4) + (jj & 3);
      const float c = (d < D && code < K) ? C[(size_t)d * K + code] : 0.0f;
      hi[jj] = (_Float16)c;
      lo[jj] = (_Float16)((c - (float)hi[jj]) * 2048.0f);
    }
    Bh[i] = hi; Bl[i] = lo;
  }
  __syncthreads();
  for (int k = tid; k < KT * 16; k += 512) {
    float acc = 0.f;
    if (k < K) {
      const f32x4* col_k = Bf + (size_t)(k >> 4) * D16 * 64 + (k & 15);
      for (int d4 = 0; 4 * d4 < D; ++d4) {
        const f32x4 c = col_k[(d4 >> 2) * 64 + (d4 & 3) * 16];
        acc = fmaf(c[0], c[0], acc);
        if (4 * d4 + 1 < D) acc = fmaf(c[1], c[1], acc);
        if (4 * d4 + 2 < D) acc = fmaf(c[2], c[2], acc);
        if (4 * d4 + 3 < D) acc = fmaf(c[3], c[3], acc);
      }
    }
    c2[k] = k < K ? acc : INFINITY;                                 // padding codes: never a candidate, never the minimum
    if (FUSE) hist[k] = 0;
  }
  __syncthreads();
  if (tid == 0) {
    float mx = 0.f, ok = 1.f;
    for (int k = 0; k < K; ++k) {
      mx = fmaxf(mx, c2[k]);
      if (!(c2[k] < 4.0e9f)) ok = 0.f;                                // NaN / Inf / beyond f16 range (|c| may exceed 65504): no prefilter
    }
    c2[KT * 16] = mx;
    c2[KT * 16 + 1] = ok;
  }
  __syncthreads();
  const float c2max = c2[KT * 16];
  const bool cb_ok = c2[KT * 16 + 1] != 0.f;
  const float sc2max = __builtin_amdgcn_sqrtf(c2max) * 1.000001f;
  float c2r[KT];                                                    // |c|^2 of this lane's code in each tile
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) c2r[kt] = c2[16 * kt + col];
  float wave_loss = 0.f;

  // PRE (the index-only form): the rows of the NEXT group are in flight while this group computes (two waves per SIMD cannot hide HBM
  // latency otherwise).  Round 5: into a SECOND register buffer, and this group's f32 rows stay where they are -- the f16 pairs are
  // cut one 32-feature step at a time inside the MFMA loop (16 live registers instead of 64).  Rounds 2-4 prefetched into the rows' own
  // registers once they were converted, so a group with an undecided row had to RE-READ its f32 rows for the exact passes, behind the
  // prefetch on the in-order vector-memory counter: ~7 us per such group and wave against 0.85 us of exact matrix work
  // (profiles/r05_engine_diag.txt #4) -- every group on encoder outputs, one in six on uniform rows.
  constexpr bool PRE = !FUSE;
  const long n_groups = (N + 15) >> 4;
  const long rg_step = (long)gridDim.x * 8;
  auto load_rows = [&](long g, f32x4 (&r)[16]) {
    const long r0 = g << 4;
    const float* xr = x + (size_t)((r0 + col) < N ? r0 + col : N - 1) * D;
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = *reinterpret_cast<const f32x4*>(xr + min(16 * i + 4 * q, D - 4));
  };
  auto mask_rows = [&](long g, f32x4 (&r)[16]) {
    const bool ok = ((g << 4) + col) < N;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (!(ok && i < D16 && (16 * i + 4 * q) < D)) r[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // |x|^2 of row `col` in the DEFINED order: four chains over d = 16 t + 4 q + e, then (p0 + p1) + (p2 + p3)
  auto strict_x2 = [&](const f32x4 (&r)[16]) -> float {
    float pp = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      pp = fmaf(r[i][0], r[i][0], pp); pp = fmaf(r[i][1], r[i][1], pp); pp = fmaf(r[i][2], r[i][2], pp); pp = fmaf(r[i][3], r[i][3], pp);
    }
    pp = pp + __shfl_xor(pp, 16);
    pp = pp + __shfl_xor(pp, 32);
    return pp;
  };
  // one group: `av` holds (PRE) or receives (FUSE) its rows; PRE: the next group's rows are requested into `an` first thing and stay in
  // flight through everything below (the two buffers swap roles from group to group: the loop further down is unrolled by two)
  auto one_group = [&](f32x4 (&av)[16], f32x4 (&an)[16], const long rg) __attribute__((always_inline)) {
    const long row0 = rg << 4;
    const bool rvalid = (row0 + col) < N;
    const bool ragged = (row0 + 16 > N) || D != 256;                // wave-uniform: only then are there registers to blank
    if constexpr (PRE) {
      if (rg + rg_step < n_groups) load_rows(rg + rg_step, an);
    } else {
      load_rows(rg, av);
    }
    if (ragged) mask_rows(rg, av);
    float x2b;                                                      // an upper estimate of |x|^2 (the margin and the range checks only)
    if (FUSE) {
      const float pr = strict_x2(av);
      const float sc = 1.0f / sqrtf(fmaxf(pr, fuse.eps));
#pragma unroll
      for (int i = 0; i < 16; ++i) av[i] = av[i] * sc;
      x2b = (pr * sc) * sc;
    } else {
      f32x2 pb = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f32x2 lo2 = {av[i][0], av[i][1]}, hi2 = {av[i][2], av[i][3]};
        pb = __builtin_elementwise_fma(lo2, lo2, pb);
        pb = __builtin_elementwise_fma(hi2, hi2, pb);
      }
      x2b = pb[0] + pb[1];
      x2b = x2b + __shfl_xor(x2b, 16);
      x2b = x2b + __shfl_xor(x2b, 32);
    }
    x2b *= 1.00002f;
    // the margin of row `col` (see 2. below); NaN marks a row the prefilter must not decide
    const float sx = __builtin_amdgcn_sqrtf(x2b);
    float mrow = 1.72e-4f * (sx * sc2max) + 1.2e-6f * (x2b + c2max) + 1.0e-9f * (sx + sc2max);
    if (!(x2b < 4.0e9f)) mrow = NAN;                                // NaN / Inf rows, elements beyond the f16 range

    // ---- 1. approximate dot products on the f16 pipe
    // (the fragment loads are loop-invariant per lane: without the opaque offset the compiler hoists all 64 KT of them out of the
    //  group loop -- 256 VGPRs at KT = 4 -- and spills them to scratch)
    int lofs = lane;
    asm volatile("" : "+v"(lofs));
    f32x4 ahh[KT], axx[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { ahh[kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; axx[kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      // the f16 pair of this 32-feature step, cut here (the f32 rows stay in av for the exact passes)
      f16x8 hm, lm;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float v = av[2 * m + (jj >> 2)][jj & 3];
        const _Float16 h = (_Float16)v;
        hm[jj] = h;
        lm[jj] = (_Float16)fmaf((float)h, -2048.0f, v * 2048.0f);           // (v - hi) 2^11, exactly (one mixed-precision fma)
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const f16x8 bh = Bh[(kt * 8 + m) * 64 + lofs], bl = Bl[(kt * 8 + m) * 64 + lofs];
        ahh[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hm, bh, ahh[kt], 0, 0, 0);
        axx[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hm, bl, axx[kt], 0, 0, 0);
        axx[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lm, bh, axx[kt], 0, 0, 0);
      }
    }
    // ---- 2. candidates: |c|^2 - 2 x.c (the row's |x|^2 is common to its codes) within the margin of the row's minimum.
    // Margin = 2 x the largest |approximate - defined| difference: the f32 summations of both (3.3e-5 sqrt(|x|^2 |c|^2), which also
    // covers the 2^-22 relative split error), the two roundings of the distance formula (1.2e-6 (|x|^2 + |c|^2)), and the ABSOLUTE
    // floor of an f16 pair, 2^-36 per element (1e-9 (|x| + |c|)): tiny rows are decided by the exact passes, not by noise.
    unsigned rm_lo[4], rm_hi[4];                                    // candidate codes of row 4 q + j, bit = code
    bool bad = !cb_ok, multi = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float margin = __shfl(mrow, 4 * q + j);                 // accumulator register j of this lane is row 4 q + j
      float dt[KT], vmin = INFINITY;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const float dot = fmaf(axx[kt][j], 1.0f / 2048.0f, ahh[kt][j]);
        dt[kt] = fmaf(-2.0f, dot, c2r[kt]);
        vmin = fminf(vmin, dt[kt]);
      }
      vmin = row16_min(vmin);
      const float thr = vmin + margin;
      bad = bad || !(fabsf(thr) < INFINITY);
      unsigned f[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const unsigned long long bl64 = __ballot(dt[kt] <= thr);    // bit (16 q' + c): code 16 kt + c is a candidate of row 4 q' + j
        const unsigned half = (q & 2) ? (unsigned)(bl64 >> 32) : (unsigned)bl64;
        f[kt] = (half >> (16 * (q & 1))) & 0xffffu;
      }
      rm_lo[j] = f[0] | (f[1] << 16);
      rm_hi[j] = KT > 2 ? (f[2] | (f[KT > 2 ? 3 : 0] << 16)) : 0u;
      multi = multi || (__popc(rm_lo[j]) + __popc(rm_hi[j])) != 1;
    }
    const bool any_bad = __ballot(bad) != 0ull;
    const bool any_multi = __ballot(multi) != 0ull;
    int best_i[4];
    if (!any_bad && !any_multi) {
      // every row is decided by the margin alone
#pragma unroll
      for (int j = 0; j < 4; ++j) best_i[j] = rm_lo[j] ? __builtin_ctz(rm_lo[j]) : 32 + __builtin_ctz(rm_hi[j]);
    } else {
      const f32x4 (&aw)[16] = av;                                     // (the group's f32 rows: still in registers)
      const float p = strict_x2(aw);
      // this lane's row in the exact passes is row `col`: its mask sits in lane 16 (col >> 2) as entry col & 3
      unsigned long long my_mask = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned lo = __shfl(rm_lo[j], (col >> 2) * 16), hi = __shfl(rm_hi[j], (col >> 2) * 16);
        if ((col & 3) == j) my_mask = ((unsigned long long)hi << 32) | lo;
      }
      // ---- 3. exact evaluation of the candidates of the UNDECIDED rows.  Rounds 2-4 ran one pass per candidate RANK -- column n of the B
      // operand = the pass's candidate of row n, the result on the diagonal: 16 useful outputs of an instruction's 256, and as many
      // passes as the group's worst row has candidates (3.1 on encoder rows, where 99.7 % of the groups hold an undecided row; groups
      // with a row of more than four candidates took the plain path: 6.5 %).  Round 5: the columns are the UNION of the undecided rows'
      // candidates -- a row with one candidate is decided and asks for nothing --: typically 5..15 codes for a whole group, i.e. ONE pass
      // of 16 columns (two for up to 32 codes; beyond that the plain path), every (row, candidate) pair an output element.  An output
      // element is the same k-ordered chain whichever column it stands in: bit-identical distances, compared as (distance, code).
      const bool undecided = __popcll(my_mask) > 1;
      unsigned u_lo = undecided ? (unsigned)my_mask : 0u, u_hi = undecided ? (unsigned)(my_mask >> 32) : 0u;
#pragma unroll
      for (int mm = 1; mm < 16; mm <<= 1) { u_lo |= __shfl_xor(u_lo, mm); u_hi |= __shfl_xor(u_hi, mm); }
      const unsigned long long uni = ((unsigned long long)u_hi << 32) | u_lo;       // (the same in every lane: all 16 rows folded)
      const int n_uni = __popcll(uni);
      if (!any_bad && n_uni <= 32) {
        float bv[4];
        int bi[4];
        unsigned m_lo[4], m_hi[4];
        float x2r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bv[j] = INFINITY; bi[j] = 0x7fffffff;
          m_lo[j] = __shfl((unsigned)my_mask, 4 * q + j);                // row 4 q + j: its mask and |x|^2 live in lane col = 4 q + j
          m_hi[j] = __shfl((unsigned)(my_mask >> 32), 4 * q + j);
          x2r[j] = __shfl(p, 4 * q + j);
        }
        const int n_pass = (n_uni + 15) >> 4;
        for (int ps = 0; ps < n_pass; ++ps) {
          // this lane's column: the (16 ps + col)-th code of the union (none: column unused -- any code, never taken)
          unsigned long long mk = uni;
          const int want = 16 * ps + col;
          for (int i = 0; i < want && mk != 0ull; ++i) mk &= mk - 1;
          const bool has = mk != 0ull;
          const int k = has ? __builtin_ctzll(mk) : __builtin_ctzll(uni);
          const f32x4* bp = Bf + ((size_t)(k >> 4) * D16) * 64 + 16 * q + (k & 15);
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < 16; ++t)
            if (t < D16) {
              const f32x4 b = bp[t * 64];
#pragma unroll
              for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[t][e], b[e], acc, 0, 0, 0);
            }
          const float c2k = c2[k];
          const unsigned kbit = 1u << (k & 31);
          // register j of lane (col, q) = row 4 q + j against this lane's code
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool cand = has && (((k < 32 ? m_lo[j] : m_hi[j]) & kbit) != 0u);
            const float dv = (x2r[j] - 2.0f * acc[j]) + c2k;
            if (cand && (bi[j] == 0x7fffffff || dv < bv[j] || (dv == bv[j] && k < bi[j]))) { bv[j] = dv; bi[j] = k; }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int mm = 1; mm < 16; mm <<= 1) {
            const float ov = __shfl_xor(bv[j], mm);
            const int oi = __shfl_xor(bi[j], mm);
            const bool take = (oi != 0x7fffffff) && (bi[j] == 0x7fffffff || ov < bv[j] || (ov == bv[j] && oi < bi[j]));
            if (take) { bv[j] = ov; bi[j] = oi; }
          }
          // (a decided row -- one candidate -- took part in no column: its candidate is the answer)
          const bool one = (__popc(m_lo[j]) + __popc(m_hi[j])) == 1;
          best_i[j] = one ? (m_lo[j] ? __builtin_ctz(m_lo[j]) : 32 + __builtin_ctz(m_hi[j])) : bi[j];
        }
      } else {
        // ---- plain f32 path for this group (vq_assign_kernel's arithmetic)
        f32x4 acc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 16; ++t)
          if (t < D16) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
              const f32x4 b = Bf[((size_t)kt * D16 + t) * 64 + lofs];
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[t][e], b[e], acc[kt], 0, 0, 0);
            }
          }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x2 = __shfl(p, 4 * q + j);
          float bv = INFINITY;
          int bi = 0x7fffffff;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            const int code = 16 * kt + col;
            const float dv = (x2 - 2.0f * acc[kt][j]) + c2r[kt];
            if (code < K && (dv < bv || bi == 0x7fffffff)) { bv = dv; bi = code; }
          }
#pragma unroll
          for (int mm = 1; mm < 16; mm <<= 1) {
            const float ov = __shfl_xor(bv, mm);
            const int oi = __shfl_xor(bi, mm);
            const bool take = (oi != 0x7fffffff) && (bi == 0x7fffffff || ov < bv || (ov == bv && oi < bi));
            if (take) { bv = ov; bi = oi; }
          }
          best_i[j] = bi;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (col == 0 && (row0 + 4 * q + j) < N) idx[row0 + 4 * q + j] = (long long)best_i[j];
    if (FUSE) {
      int kr = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kj = __shfl(best_i[j], (col >> 2) * 16);
        if ((col & 3) == j) kr = kj;
      }
      const int kt_r = kr >> 4, kc_r = kr & 15;
      float lr = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < D16 && (16 * i + 4 * q) < D) {
          const f32x4 cq = Bf[((size_t)kt_r * D16 + i) * 64 + 16 * q + kc_r];
          const f32x4 dq = cq - av[i];
          lr = fmaf(dq[0], dq[0], lr); lr = fmaf(dq[1], dq[1], lr); lr = fmaf(dq[2], dq[2], lr); lr = fmaf(dq[3], dq[3], lr);
          if (rvalid && quant != nullptr) *reinterpret_cast<f32x4*>(quant + (size_t)(row0 + col) * D + 16 * i + 4 * q) = av[i] + dq;
          if (rvalid && fuse.xnorm != nullptr) *reinterpret_cast<f32x4*>(fuse.xnorm + (size_t)(row0 + col) * D + 16 * i + 4 * q) = av[i];
        }
      }
      if (!rvalid) lr = 0.f;
      lr = lr + __shfl_xor(lr, 16);
      lr = lr + __shfl_xor(lr, 32);
      lr = row16_sum(lr);
      wave_loss += lr;
      if (q == 0 && rvalid) atomicAdd(&hist[kr], 1);
    } else if (quant != nullptr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (row0 + r >= N) continue;                                // wave-uniform
        const int k = __shfl(best_i[r & 3], (r >> 2) * 16);
        const int d4 = lane;                                        // float4 index along D (D <= 256)
        if (4 * d4 < D) {
          const f32x4 v = Bf[((size_t)(k >> 4) * D16 + (d4 >> 2)) * 64 + (d4 & 3) * 16 + (k & 15)];
          *reinterpret_cast<f32x4*>(quant + (size_t)(row0 + r) * D + 4 * d4) = v;
        }
      }
    }
  };
  long rg = (long)blockIdx.x * 8 + wave;
  if constexpr (PRE) {
    f32x4 ra[16], rb[16];
    if (rg < n_groups) load_rows(rg, ra);
    for (; rg < n_groups; rg += 2 * rg_step) {
      one_group(ra, rb, rg);
      if (rg + rg_step < n_groups) one_group(rb, ra, rg + rg_step);
    }
  } else {
    f32x4 ra[16];
    for (; rg < n_groups; rg += rg_step) one_group(ra, ra, rg);
  }
  if (FUSE) {
    float* wsum = reinterpret_cast<float*>(hist + KT * 16);
    if (lane == 0) wsum[wave] = wave_loss;
    __syncthreads();
    // (the loss partial of a workgroup: its eight waves in wave order, two by two -- a fixed order, as in vq_assign_kernel)
    if (tid == 0) fuse.loss_part[blockIdx.x] = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + ((wsum[4] + wsum[5]) + (wsum[6] + wsum[7]));
    for (int k = tid; k < K; k += 512)
      if (hist[k]) atomicAdd(&fuse.counts[k], (float)hist[k]);
  }
}

// y = x / sqrt(max(sum_d x^2, eps)) row by row (tf.linalg.l2_normalize, util/math.py:63-64), with the sum in the DEFINED order of
// vq_assign_kernel's x2 (lane (row, q) runs one fmaf chain over d = 16 t + 4 q + e, t outer; then (p0 + p1) + (p2 + p3)) and a
// correctly rounded sqrt and division: oracle/vq_strict.c states the same in plain C.  16 rows per wave, any D % 4 == 0.
__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* __restrict__ x, long N, int D, float eps, float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
  const int D16 = (D + 15) >> 4;
  const long n_groups = (N + 15) >> 4;
  for (long rg = (long)blockIdx.x * 4 + wave; rg < n_groups; rg += (long)gridDim.x * 4) {
    const long row = (rg << 4) + col;
    const bool rvalid = row < N;
    const float* xr = x + (size_t)(rvalid ? row : N - 1) * D;
    float p = 0.f;
    for (int t = 0; t < D16; ++t) {
      if (16 * t + 4 * q < D) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * t + 4 * q);
        p = fmaf(a[0], a[0], p); p = fmaf(a[1], a[1], p); p = fmaf(a[2], a[2], p); p = fmaf(a[3], a[3], p);
      }
    }
    p = p + __shfl_xor(p, 16);
    p = p + __shfl_xor(p, 32);
    const float sc = 1.0f / sqrtf(fmaxf(p, eps));
    if (rvalid)
      for (int t = 0; t < D16; ++t)
        if (16 * t + 4 * q < D) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(xr + 16 * t + 4 * q);       // (L2-resident second read)
          *reinterpret_cast<f32x4*>(y + (size_t)row * D + 16 * t + 4 * q) = a * sc;
        }
  }
}

__global__ void vq_ema_stats_kernel(const float* __restrict__ x, const long long* __restrict__ idx, long N, int D,
                                    int K, float* __restrict__ counts, float* __restrict__ dw) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* acc = smem;                                  // [K][D]
  int* cnt = reinterpret_cast<int*>(smem + (size_t)K * D);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < K * D; i += blockDim.x) acc[i] = 0.f;
  for (int i = tid; i < K; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  const int D4 = D >> 2;
  for (long row = (long)blockIdx.x * 4 + wave; row < N; row += (long)gridDim.x * 4) {
    const int k = (int)idx[row];
    if (k < 0 || k >= K) continue;                    // wave-uniform
    if (lane == 0) atomicAdd(&cnt[k], 1);
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * D);
    float* a = acc + (size_t)k * D;
    for (int c = lane; c < D4; c += 64) {
      f32x4 v = xr[c];
      atomicAdd(&a[4 * c + 0], v[0]);
      atomicAdd(&a[4 * c + 1], v[1]);
      atomicAdd(&a[4 * c + 2], v[2]);
      atomicAdd(&a[4 * c + 3], v[3]);
    }
  }
  __syncthreads();
  for (int i = tid; i < K * D; i += blockDim.x) {
    const int k = i / D, d = i - k * D;
    const float v = acc[i];
    if (v != 0.f) atomicAdd(&dw[(size_t)d * K + k], v);
  }
  for (int i = tid; i < K; i += blockDim.x)
    if (cnt[i]) atomicAdd(&counts[i], (float)cnt[i]);
}


// ---- EMA statistics, deterministic two-pass form (K * D <= 4096) ---------------------------------------------------
// pass 1: every wave owns a private [K][D] accumulator image in LDS (no atomics): for each of its rows it adds the row
//         (one float4 per lane and 256 features) into the slice of the row's code; 4 rows are in flight per wave.
//         The per-wave images go to the workspace.
// pass 2: counts[k], dw[d][k] = sums over the wave images in index order  -> bit-reproducible.
template <int RU>
__global__ __launch_bounds__(256) void vq_ema_partial_kernel(const float* __restrict__ x, const long long* __restrict__ idx,
                                                             long N, int D, int K, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int KD = K * D;
  float* acc = smem + (size_t)wave * (KD + K);         // [K][D] then [K] counts
  for (int i = lane; i < KD + K; i += 64) acc[i] = 0.f;
  __builtin_amdgcn_wave_barrier();
  const long wid = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
  const int D4 = D >> 2;
  // RU rows per wave and pass.  Every fetch is unconditional (row index clamped; a clamped row is never accumulated) and the
  // next pass's indices are requested before this pass's rows are used: the stream is latency-bound otherwise (guarded
  // fetches make the compiler drain the memory counter before each dependent step).
  auto row_of = [&](long r) { return r < N ? r : N - 1; };
  long long kn[RU];
#pragma unroll
  for (int u = 0; u < RU; ++u) kn[u] = idx[row_of(wid * RU + u)];
  for (long r0 = wid * RU; r0 < N; r0 += nw * RU) {
    int k[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) k[u] = (r0 + u < N) ? (int)kn[u] : -1;
#pragma unroll
    for (int u = 0; u < RU; ++u) kn[u] = idx[row_of(r0 + nw * RU + u)];
    for (int c = lane; c < D4; c += 64) {
      f32x4 v[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (size_t)row_of(r0 + u) * D + 4 * c);
#pragma unroll
      for (int u = 0; u < RU; ++u)
        if (k[u] >= 0 && k[u] < K) {
          f32x4* a = reinterpret_cast<f32x4*>(acc + (size_t)k[u] * D + 4 * c);
          f32x4 t = *a;
          t[0] += v[u][0]; t[1] += v[u][1]; t[2] += v[u][2]; t[3] += v[u][3];
          *a = t;
        }
    }
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < RU; ++u)
        if (k[u] >= 0 && k[u] < K) acc[KD + k[u]] += 1.f;
    }
    __builtin_amdgcn_wave_barrier();
  }
  float* out = part + (size_t)wid * (KD + K);
  for (int i = lane; i < KD + K; i += 64) out[i] = acc[i];
}

__global__ void vq_ema_reduce_kernel(const float* __restrict__ part, long n_part, int D, int K, float* __restrict__ counts,
                                     float* __restrict__ dw) {
  const int KD = K * D;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= KD + K) return;
  float s = 0.f;
  for (long w = 0; w < n_part; ++w) s += part[(size_t)w * (KD + K) + i];
  if (i < KD) { const int k = i / D, d = i - k * D; dw[(size_t)d * K + k] = s; }
  else counts[i - KD] = s;
}

// ---- EMA statistics on the matrix pipe (D % 64 == 0, K <= 64): dw = x^T . onehot(idx) -------------------------------
// This is literally the reference's statement (vq_layers.py:307-309, matmul(x, encodings, transpose_a=True)): a one-hot
// operand is exact in f32 (x * 1, x * 0), so v_mfma_f32_16x16x4_f32 computes the segmented sums bit-for-bit as chains of
// f32 adds, with no LDS read-modify-write per row and no atomics.  One wave step = 4 rows: lane (i, kq) fetches row kq's
// features 64 j + 4 i + (0..3) (one dwordx4 per j: fully coalesced 1 KB rows) as the A operand of tiles (j, e) and builds
// the B operand 1[idx[row kq] == 16 kt + i] on the fly; RU steps are in flight per wave.  Rows map to waves and steps in a
// fixed pattern, the 4 waves of a workgroup are summed in wave order through LDS, and the per-workgroup images are summed
// in index order by vq_ema_reduce2_kernel: bit-reproducible from launch to launch.
template <int KT, int RU>
__global__ __launch_bounds__(256) void vq_ema_mfma_kernel(const f32x4* __restrict__ x4, const long long* __restrict__ idx,
                                                          long N, int DJ /* D / 64 */, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float smem[];        // [16 KT][D + 1] then [16 KT] counts
  constexpr int MAXDJ = 4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int D = DJ * 64, D4 = DJ * 16;
  f32x4 acc[MAXDJ][4][KT];
  float cnt[KT];
#pragma unroll
  for (int j = 0; j < MAXDJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) acc[j][e][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) cnt[kt] = 0.f;

  const long n_steps = (N + 3) >> 2, wid = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
  for (long s0 = wid; s0 < n_steps; s0 += nw * RU) {
    f32x4 v[RU][MAXDJ];
    long long kk[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {                       // unconditional, clamped fetches (a clamped row is zeroed below)
      const long r = 4 * (s0 + u * nw) + kq, rc = r < N ? r : N - 1;
      kk[u] = idx[rc];
#pragma unroll
      for (int j = 0; j < MAXDJ; ++j) v[u][j] = x4[rc * D4 + (j < DJ ? j : DJ - 1) * 16 + i];
    }
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      const bool valid = 4 * (s0 + u * nw) + kq < N;
      float b[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        b[kt] = (valid && kk[u] == (long long)(16 * kt + i)) ? 1.f : 0.f;
        cnt[kt] += b[kt];
      }
#pragma unroll
      for (int j = 0; j < MAXDJ; ++j) {
        if (j < DJ) {
          const f32x4 a = valid ? v[u][j] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) acc[j][e][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[kt], acc[j][e][kt], 0, 0, 0);
        }
      }
    }
  }
  // counts: lane (i, kq) holds the matches of code 16 kt + i among its rows -> sum the four kq groups in a fixed order
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const float c1 = cnt[kt] + __shfl_xor(cnt[kt], 16);
    cnt[kt] = c1 + __shfl_xor(c1, 32);
  }
  // accumulator tile (j, e), lane (n = i, q = kq), reg g  ->  code 16 kt + n, feature 64 j + 4 (4 q + g) + e
  const int ld = D + 1;
  float* cn = smem + (size_t)16 * KT * ld;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        float* row = smem + (size_t)(16 * kt + i) * ld;
#pragma unroll
        for (int j = 0; j < MAXDJ; ++j)
          if (j < DJ) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int f = 64 * j + 4 * (4 * kq + g) + e;
                row[f] = (w == 0 ? 0.f : row[f]) + acc[j][e][kt][g];
              }
          }
        if (kq == 0) cn[16 * kt + i] = (w == 0 ? 0.f : cn[16 * kt + i]) + cnt[kt];
      }
    }
    __syncthreads();
  }
  const int KP = 16 * KT;
  float* out = part + (size_t)blockIdx.x * ((size_t)KP * D + KP);
  for (int t = threadIdx.x; t < KP * D; t += 256) out[t] = smem[(size_t)(t / D) * ld + (t % D)];
  for (int t = threadIdx.x; t < KP; t += 256) out[(size_t)KP * D + t] = cn[t];
}

// counts[k], dw[d][k] = sums over the per-workgroup images in index order.  16 thread groups split the images (independent
// loads), thread group 0 adds the 16 group sums in group order.
__global__ __launch_bounds__(256) void vq_ema_reduce2_kernel(const float* __restrict__ part, int n_part, int D, int K, int KP,
                                                             float* __restrict__ counts, float* __restrict__ dw) {
  __shared__ float sums[16][17];
  const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int tot = K * D + K, t = blockIdx.x * 16 + o;            // output t: (k, d) for t < K D, else count of code t - K D
  const size_t stride = (size_t)KP * D + KP;
  const size_t src = t < K * D ? (size_t)t : (size_t)KP * D + (t - K * D);
  const int per = (n_part + 15) >> 4, p0 = g * per, p1 = min(n_part, p0 + per);
  float a = 0.f;
  if (t < tot)
    for (int p = p0; p < p1; ++p) a += part[(size_t)p * stride + src];
  sums[g][o] = a;
  __syncthreads();
  if (g == 0 && t < tot) {
    float r = sums[0][o];
#pragma unroll
    for (int k = 1; k < 16; ++k) r += sums[k][o];
    if (t < K * D) { const int k = t / D, d = t - k * D; dw[(size_t)d * K + k] = r; }
    else counts[t - K * D] = r;
  }
}

// ---- straight-through output + commitment term (vq_layers.py:302, :327) in one pass ---------------------------------
//   ste = x + (q - x)   (the reference's expression, kept as written: it is q up to one rounding)
//   loss = scale * sum((q - x)^2): per-thread chains over a fixed element pattern, fixed-order tree per workgroup,
//   workgroup sums added in index order by the last kernel  -> bit-reproducible for a given shape.
__global__ __launch_bounds__(256) void vq_ste_loss_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ q, long n4,
                                                          f32x4* __restrict__ ste, float* __restrict__ part) {
  __shared__ float red[256];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long stride = (long)gridDim.x * 256;
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 4 * stride) {
    f32x4 xv[4], qv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = min(i0 + u * stride, n4 - 1);
      xv[u] = x[i]; qv[u] = q[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = i0 + u * stride;
      if (i < n4) {
        const f32x4 d = qv[u] - xv[u];
        if (ste != nullptr) ste[i] = xv[u] + d;
        s0 = fmaf(d[0], d[0], s0); s1 = fmaf(d[1], d[1], s1); s2 = fmaf(d[2], d[2], s2); s3 = fmaf(d[3], d[3], s3);
      }
    }
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// (`post`: a second factor applied AFTER the mean -- the commitment cost of the training quantiser, `commitment_cost * mean(...)` as the
//  reference states it, vq_layers.py:327-330 -- rounded as its own multiplication; 1.0f is exact)
__global__ __launch_bounds__(64) void vq_ste_loss_final_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ loss,
                                                                float post = 1.0f) {
  float s = 0.f;                                         // lane l: part[l], part[l + 64], ... then a fixed xor tree
  for (int i = threadIdx.x; i < n; i += 64) s += part[i];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
  if (threadIdx.x == 0) *loss = __fmul_rn(__fmul_rn(s, scale), post);
}

// counts only (dw == NULL): per-workgroup LDS histogram, then one float add of an integer per code and workgroup (sums of
// integers are exact in f32 below 2^24, so the order of the atomics does not matter)
__global__ __launch_bounds__(256) void vq_counts_kernel(const long long* __restrict__ idx, long N, int K, float* __restrict__ counts) {
  extern __shared__ int hist[];
  for (int k = threadIdx.x; k < K; k += 256) hist[k] = 0;
  __syncthreads();
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < N; r += (long)gridDim.x * 256) {
    const long long k = idx[r];
    if (k >= 0 && k < K) atomicAdd(&hist[(int)k], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 256)
    if (hist[k]) atomicAdd(&counts[k], (float)hist[k]);
}

static bool ema_mfma_ok(int D, int K) { return D > 0 && (D & 63) == 0 && D <= 256 && K > 0 && K <= 64; }
static int ema_mfma_kt(int K) { return K <= 16 ? 1 : K <= 32 ? 2 : 4; }
static long ema_mfma_grid(long N, int K) {
  const long per_cu = ema_mfma_kt(K) == 4 ? 1 : 2;          // KT = 4 keeps 256 accumulator registers: one wave per SIMD
  long blocks = ((N + 3) / 4 + 3) / 4;                      // >= one 4-row step per wave
  const long cap = (long)vqn_num_cus() * per_cu;
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}

static long ema_grid(long N, int D, int K) {
  const size_t lds = (size_t)4 * (K * D + K) * sizeof(float);
  const long per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
  long blocks = (N + 63) / 64;
  const long cap = (long)vqn_num_cus() * per_cu;
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}

static long assign_grid(long N) {
  const long n_groups = (N + 15) >> 4;
  long blocks = (n_groups + 3) / 4;
  const long cap = (long)vqn_num_cus() * 8;
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}

template <int KT, bool FUSE>
int launch_assign(const float* x, long N, int D, const float* C, int K, const float* sel, float* ws, long long* idx,
                  float* quant, float* dist, const VqFuse fuse, hipStream_t s) {
  const int D16 = (D + 15) >> 4;
  const size_t lds = ((size_t)KT * D16 * 256 + 3 * KT * 16 + 4) * sizeof(float);
  const long blocks = assign_grid(N);
  if (lds > 64 * 1024) {
    VQN_HIP(hipFuncSetAttribute((const void*)vq_assign_kernel<KT, true, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    VQN_HIP(hipFuncSetAttribute((const void*)vq_assign_kernel<KT, false, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  if (sel != nullptr) {
    // distances.max() is taken over the UNMASKED distances (vq_layers.py:285) -> extra pass
    VQN_HIP(hipMemsetD32Async((hipDeviceptr_t)ws, (int)0x807fffff /* key(-inf) */, 1, s));
    hipLaunchKernelGGL((vq_assign_kernel<KT, true, FUSE>), dim3((unsigned)blocks), dim3(256), lds, s, x, N, D, C, K, sel,
                       reinterpret_cast<int*>(ws), idx, quant, dist, fuse);
    VQN_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL((vq_assign_kernel<KT, false, FUSE>), dim3((unsigned)blocks), dim3(256), lds, s, x, N, D, C, K, sel,
                     reinterpret_cast<int*>(ws), idx, quant, dist, fuse);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

// The prefiltered kernel (vq_assign_split_kernel) serves K in 17..64 without a code-dropout mask or a distance output; VQN_VQ_SPLIT=0
// keeps the f32 kernel (A/B measurements, bisecting).
static bool split_ok(int K, int D, const float* sel, const float* dist) {
  static const int on = [] { const char* e = getenv("VQN_VQ_SPLIT"); return (e == nullptr || atoi(e) != 0) ? 1 : 0; }();
  return on && K > 16 && K <= 64 && D <= 256 && sel == nullptr && dist == nullptr;
}

static long split_grid(long N) {
  const long n_groups = (N + 15) >> 4;
  long blocks = (n_groups + 7) / 8;
  const long cap = (long)vqn_num_cus();
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}

template <int KT, bool FUSE>
int launch_split(const float* x, long N, int D, const float* C, int K, long long* idx, float* quant, const VqFuse fuse, hipStream_t s) {
  const int D16 = (D + 15) >> 4;
  const size_t lds = (size_t)KT * D16 * 1024 + (size_t)2 * KT * 8 * 1024 + ((size_t)2 * KT * 16 + 4 + 8) * sizeof(float);
  VQN_CHECK_SHAPE(lds <= 160 * 1024, "codebook does not fit in 160 KB of LDS");
  VQN_HIP(hipFuncSetAttribute((const void*)vq_assign_split_kernel<KT, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((vq_assign_split_kernel<KT, FUSE>), dim3((unsigned)split_grid(N)), dim3(512), lds, s, x, N, D, C, K, idx, quant, fuse);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

// B fragments + |c|^2 of a codebook of <= 64 codes (1, 2 or 4 tiles of 16), D = 256, as the fused reflectance kernel (csrc/mlp_chain.hip) reads them:
// exactly what vq_assign_kernel<1, ...> stages into LDS (same element order, the same fmaf chain over d for |c|^2).
__global__ __launch_bounds__(256) void vq_codebook_frags_kernel(const float* __restrict__ C, int D, int K, int KT, float* __restrict__ out) {
  const int D16 = (D + 15) >> 4;
  f32x4* Bf = reinterpret_cast<f32x4*>(out);
  for (int i = threadIdx.x; i < KT * D16 * 64; i += 256) {
    const int l = i & 63, t = (i >> 6) % D16, kt = (i >> 6) / D16;
    const int code = 16 * kt + (l & 15);
    f32x4 v;
    for (int e = 0; e < 4; ++e) {
      const int d = 16 * t + 4 * (l >> 4) + e;
      v[e] = (d < D && code < K) ? C[(size_t)d * K + code] : 0.0f;
    }
    Bf[i] = v;
  }
  if ((int)threadIdx.x < 16 * KT) {
    const int k = threadIdx.x;
    float acc = 0.f;
    if (k < K)
      for (int d = 0; d < D; ++d) { const float c = C[(size_t)d * K + k]; acc = fmaf(c, c, acc); }
    out[(size_t)KT * D16 * 256 + k] = acc;
  }
}

}  // namespace

int vqn_internal_finish_loss(const float* part, int n, float scale, float* loss, hipStream_t s) {
  hipLaunchKernelGGL(vq_ste_loss_final_kernel, dim3(1), dim3(64), 0, s, part, n, scale, loss, 1.0f);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_vq_codebook_frags(const float* codebook, int D, int K, float* frags, void* stream) {
  VQN_CHECK_ARG(codebook && frags, "codebook and frags must be non-null");
  VQN_CHECK_SHAPE(D == 256 && K >= 1 && K <= 64, "D = 256, K <= 64");
  hipLaunchKernelGGL(vq_codebook_frags_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, codebook, D, K, K <= 16 ? 1 : (K <= 32 ? 2 : 4), frags);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_vq_assign_variant(int D, int K, int has_sel_mask, int has_dist) {
  static const float one = 1.f;
  return split_ok(K, D, has_sel_mask ? &one : nullptr, has_dist ? &one : nullptr) ? 1 : 0;
}

extern "C" int vqn_vq_assign(const float* x, int64_t N, int D, const float* codebook, int K, const float* sel_mask,
                             float* ws, int64_t* idx, float* quant, float* dist, void* stream) {
  VQN_CHECK_ARG(N >= 0 && D > 0 && K > 0, "N >= 0, D > 0, K > 0 required");
  if (N == 0) return VQN_OK;
  VQN_CHECK_ARG(x && codebook && idx, "x, codebook and idx must be non-null");
  VQN_CHECK_ARG(sel_mask == nullptr || ws != nullptr, "sel_mask needs a 4-byte device workspace `ws`");
  VQN_CHECK_SHAPE(D % 4 == 0, "D must be a multiple of 4 (16-byte rows)");
  VQN_CHECK_SHAPE(((uintptr_t)x % 16) == 0 && (quant == nullptr || ((uintptr_t)quant % 16) == 0), "x/quant must be 16-byte aligned");
  const int KT = (K + 15) / 16;
  const int D16 = (D + 15) / 16;
  VQN_CHECK_SHAPE(KT <= 8, "K <= 128");
  int KTp = KT <= 1 ? 1 : KT <= 2 ? 2 : KT <= 4 ? 4 : 8;
  VQN_CHECK_SHAPE(((size_t)KTp * D16 * 256 + 3 * KTp * 16 + 4) * 4 <= 160 * 1024, "codebook does not fit in 160 KB of LDS");
  hipStream_t s = (hipStream_t)stream;
  long long* idx_ll = reinterpret_cast<long long*>(idx);
  const VqFuse nf = {0.f, nullptr, nullptr, nullptr};
  if (split_ok(K, D, sel_mask, dist))
    return KTp == 2 ? launch_split<2, false>(x, N, D, codebook, K, idx_ll, quant, nf, s)
                    : launch_split<4, false>(x, N, D, codebook, K, idx_ll, quant, nf, s);
  switch (KTp) {
    case 1: return launch_assign<1, false>(x, N, D, codebook, K, sel_mask, ws, idx_ll, quant, dist, nf, s);
    case 2: return launch_assign<2, false>(x, N, D, codebook, K, sel_mask, ws, idx_ll, quant, dist, nf, s);
    case 4: return launch_assign<4, false>(x, N, D, codebook, K, sel_mask, ws, idx_ll, quant, dist, nf, s);
    default: return launch_assign<8, false>(x, N, D, codebook, K, sel_mask, ws, idx_ll, quant, dist, nf, s);
  }
}

extern "C" int vqn_l2_normalize_rows(const float* x, int64_t N, int D, float eps, float* y, void* stream) {
  VQN_CHECK_ARG(N >= 0 && D > 0, "N >= 0, D > 0 required");
  if (N == 0) return VQN_OK;
  VQN_CHECK_ARG(x && y, "x and y must be non-null");
  VQN_CHECK_SHAPE(D % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "D multiple of 4, x / y 16-byte aligned");
  hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((unsigned)assign_grid(N)), dim3(256), 0, (hipStream_t)stream, x, (long)N, D, eps, y);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

static int quantize_rows_impl(const float* z, int64_t N, int D, const float* codebook, int K, const float* sel_mask, float eps,
                              float loss_scale, float* ws, int64_t* idx, float* ste, float* loss, float* counts, float* xnorm, void* stream,
                              float loss_post = 1.0f) {
  VQN_CHECK_ARG(N >= 0 && D > 0 && K > 0, "N >= 0, D > 0, K > 0 required");
  VQN_CHECK_ARG(loss && counts && ws, "loss, counts and ws (VQN_QUANT_WS_FLOATS floats) must be non-null");
  hipStream_t s = (hipStream_t)stream;
  VQN_HIP(hipMemsetAsync(counts, 0, sizeof(float) * K, s));
  if (N == 0) {                                         /* mean over nothing: the reference yields NaN (0 / 0) */
    VQN_HIP(hipMemsetD32Async((hipDeviceptr_t)loss, 0x7fc00000, 1, s));     // quiet NaN, written on the device (capturable)
    return VQN_OK;
  }
  VQN_CHECK_ARG(z && codebook && idx, "z, codebook and idx must be non-null");
  VQN_CHECK_SHAPE(D % 4 == 0 && D <= 256, "D must be a multiple of 4 and <= 256 (the row stays in registers)");
  VQN_CHECK_SHAPE(((uintptr_t)z % 16) == 0 && (ste == nullptr || ((uintptr_t)ste % 16) == 0), "z / ste must be 16-byte aligned");
  const int KT = (K + 15) / 16;
  const int D16 = (D + 15) / 16;
  VQN_CHECK_SHAPE(KT <= 8, "K <= 128");
  int KTp = KT <= 1 ? 1 : KT <= 2 ? 2 : KT <= 4 ? 4 : 8;
  VQN_CHECK_SHAPE(((size_t)KTp * D16 * 256 + 3 * KTp * 16 + 4) * 4 <= 160 * 1024, "codebook does not fit in 160 KB of LDS");
  long long* idx_ll = reinterpret_cast<long long*>(idx);
  const bool split = split_ok(K, D, sel_mask, nullptr);
  const long blocks = split ? split_grid(N) : assign_grid(N);
  VQN_CHECK_SHAPE(blocks + 1 <= VQN_QUANT_WS_FLOATS, "workspace too small for this device");
  VQN_CHECK_SHAPE(xnorm == nullptr || ((uintptr_t)xnorm % 16) == 0, "xnorm must be 16-byte aligned");
  const VqFuse fz = {eps, ws + 1, counts, xnorm};      // ws[0]: the code-dropout maximum; ws[1 ..]: per-workgroup loss sums
  int rc;
  if (split)
    rc = KTp == 2 ? launch_split<2, true>(z, N, D, codebook, K, idx_ll, ste, fz, s) : launch_split<4, true>(z, N, D, codebook, K, idx_ll, ste, fz, s);
  else switch (KTp) {
    case 1: rc = launch_assign<1, true>(z, N, D, codebook, K, sel_mask, ws, idx_ll, ste, nullptr, fz, s); break;
    case 2: rc = launch_assign<2, true>(z, N, D, codebook, K, sel_mask, ws, idx_ll, ste, nullptr, fz, s); break;
    case 4: rc = launch_assign<4, true>(z, N, D, codebook, K, sel_mask, ws, idx_ll, ste, nullptr, fz, s); break;
    default: rc = launch_assign<8, true>(z, N, D, codebook, K, sel_mask, ws, idx_ll, ste, nullptr, fz, s); break;
  }
  if (rc != VQN_OK) return rc;
  hipLaunchKernelGGL(vq_ste_loss_final_kernel, dim3(1), dim3(64), 0, s, ws + 1, (int)blocks, loss_scale, loss, loss_post);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_vq_quantize_rows(const float* z, int64_t N, int D, const float* codebook, int K, const float* sel_mask, float eps,
                                    float loss_scale, float* ws, int64_t* idx, float* ste, float* loss, float* counts, void* stream) {
  return quantize_rows_impl(z, N, D, codebook, K, sel_mask, eps, loss_scale, ws, idx, ste, loss, counts, nullptr, stream);
}

// ... and the form the TRAINING path uses (round 4): the same pass also leaves the l2-normalised rows, which the EMA statistics
// (vqn_vq_ema_stats) and the backward (vqn_vq_ste_loss_bwd, vqn_l2_normalize_rows_bwd) read.
extern "C" int vqn_vq_quantize_rows_train(const float* z, int64_t N, int D, const float* codebook, int K, const float* sel_mask, float eps,
                                          float loss_scale, float loss_post, float* ws, int64_t* idx, float* ste, float* loss, float* counts,
                                          float* xnorm, void* stream) {
  VQN_CHECK_ARG(N == 0 || xnorm != nullptr, "xnorm must be non-null");
  return quantize_rows_impl(z, N, D, codebook, K, sel_mask, eps, loss_scale, ws, idx, ste, loss, counts, xnorm, stream, loss_post);
}

extern "C" int64_t vqn_vq_ema_stats_ws_bytes(int64_t N, int D, int K) {
  if (N > 0 && ema_mfma_ok(D, K)) {
    const int KP = 16 * ema_mfma_kt(K);
    return ema_mfma_grid(N, K) * ((int64_t)KP * D + KP) * (int64_t)sizeof(float);
  }
  if (N <= 0 || D <= 0 || K <= 0 || (long)K * D > 4096 || (D & 3)) return 0;       // 0: the single-pass (LDS-atomic) form is used
  return ema_grid(N, D, K) * 4 * (int64_t)(K * D + K) * (int64_t)sizeof(float);
}

extern "C" int vqn_vq_ema_stats(const float* x, const int64_t* idx, int64_t N, int D, int K, float* counts, float* dw,
                                float* ws, int64_t ws_bytes, void* stream) {
  VQN_CHECK_ARG(N >= 0 && D > 0 && K > 0, "N >= 0, D > 0, K > 0 required");
  VQN_CHECK_ARG(counts, "counts must be non-null");
  hipStream_t s = (hipStream_t)stream;
  if (dw == nullptr) {                                  /* counts only */
    VQN_HIP(hipMemsetAsync(counts, 0, sizeof(float) * K, s));
    if (N == 0) return VQN_OK;
    VQN_CHECK_ARG(idx, "idx must be non-null");
    long blocks = (N + 4095) / 4096;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(vq_counts_kernel, dim3((unsigned)blocks), dim3(256), sizeof(int) * K, s,
                       reinterpret_cast<const long long*>(idx), (long)N, K, counts);
    VQN_LAUNCH_CHECK();
    return VQN_OK;
  }
  const int64_t need = vqn_vq_ema_stats_ws_bytes(N, D, K);
  if (N > 0 && need > 0 && ws != nullptr && ws_bytes >= need && ema_mfma_ok(D, K)) {
    VQN_CHECK_ARG(x && idx, "x and idx must be non-null");
    VQN_CHECK_SHAPE(((uintptr_t)x % 16) == 0, "x must be 16-byte aligned");
    const int KT = ema_mfma_kt(K), KP = 16 * KT, DJ = D / 64;
    const long blocks = ema_mfma_grid(N, K);
    const size_t lds = ((size_t)KP * (D + 1) + KP) * sizeof(float);
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const long long* il = reinterpret_cast<const long long*>(idx);
#define VQN_EMA_LAUNCH(KT_, RU_)                                                                                              \
    do {                                                                                                                      \
      if (lds > 64 * 1024)                                                                                                    \
        VQN_HIP(hipFuncSetAttribute((const void*)vq_ema_mfma_kernel<KT_, RU_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL((vq_ema_mfma_kernel<KT_, RU_>), dim3((unsigned)blocks), dim3(256), lds, s, x4, il, (long)N, DJ, ws);  \
    } while (0)
    if (KT == 1) VQN_EMA_LAUNCH(1, 8);
    else if (KT == 2) VQN_EMA_LAUNCH(2, 4);
    else VQN_EMA_LAUNCH(4, 4);
#undef VQN_EMA_LAUNCH
    VQN_LAUNCH_CHECK();
    const int tot = K * D + K;
    hipLaunchKernelGGL(vq_ema_reduce2_kernel, dim3((tot + 15) / 16), dim3(256), 0, s, ws, (int)blocks, D, K, KP, counts, dw);
    VQN_LAUNCH_CHECK();
    return VQN_OK;
  }
  if (N > 0 && need > 0 && ws != nullptr && ws_bytes >= need) {
    VQN_CHECK_ARG(x && idx, "x and idx must be non-null");
    VQN_CHECK_SHAPE(((uintptr_t)x % 16) == 0, "x must be 16-byte aligned");
    const long blocks = ema_grid(N, D, K);
    const size_t lds = (size_t)4 * (K * D + K) * sizeof(float);
    if (lds > 64 * 1024)
      VQN_HIP(hipFuncSetAttribute((const void*)vq_ema_partial_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(vq_ema_partial_kernel<8>, dim3((unsigned)blocks), dim3(256), lds, s, x,
                       reinterpret_cast<const long long*>(idx), (long)N, D, K, ws);
    VQN_LAUNCH_CHECK();
    const int tot = K * D + K;
    hipLaunchKernelGGL(vq_ema_reduce_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, ws, blocks * 4, D, K, counts, dw);
    VQN_LAUNCH_CHECK();
    return VQN_OK;
  }
  VQN_HIP(hipMemsetAsync(counts, 0, sizeof(float) * K, s));
  VQN_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)D * K, s));
  if (N == 0) return VQN_OK;
  VQN_CHECK_ARG(x && idx, "x and idx must be non-null");
  VQN_CHECK_SHAPE(D % 4 == 0 && ((uintptr_t)x % 16) == 0, "D multiple of 4 and x 16-byte aligned");
  const size_t lds = ((size_t)K * D + K) * sizeof(float);
  VQN_CHECK_SHAPE(lds <= 160 * 1024, "K*D accumulators do not fit in 160 KB of LDS");
  if (lds > 64 * 1024)
    VQN_HIP(hipFuncSetAttribute((const void*)vq_ema_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long blocks = (N + 63) / 64;                        // >= 16 rows per wave
  const long cap = (long)vqn_num_cus() * 4;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(vq_ema_stats_kernel, dim3((unsigned)blocks), dim3(256), lds, s, x,
                     reinterpret_cast<const long long*>(idx), (long)N, D, K, counts, dw);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_vq_ste_loss(const float* x, const float* quant, int64_t numel, float scale, float* ste, float* loss,
                               float* ws, void* stream) {
  VQN_CHECK_ARG(numel >= 0 && loss && ws, "numel >= 0, loss and ws (VQN_STE_WS_FLOATS floats) must be non-null");
  hipStream_t s = (hipStream_t)stream;
  if (numel == 0) {                                     /* mean over nothing: the reference yields NaN (0 / 0) */
    VQN_HIP(hipMemsetD32Async((hipDeviceptr_t)loss, 0x7fc00000, 1, s));     // quiet NaN, written on the device (capturable)
    return VQN_OK;
  }
  VQN_CHECK_ARG(x && quant, "x and quant must be non-null");
  VQN_CHECK_SHAPE((numel & 3) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)quant % 16) == 0 && (ste == nullptr || ((uintptr_t)ste % 16) == 0),
                  "numel multiple of 4 and 16-byte aligned tensors");
  const long n4 = numel / 4;
  long blocks = (n4 + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;                     /* = VQN_STE_WS_FLOATS */
  hipLaunchKernelGGL(vq_ste_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const f32x4*>(x),
                     reinterpret_cast<const f32x4*>(quant), n4, reinterpret_cast<f32x4*>(ste), ws);
  VQN_LAUNCH_CHECK();
  hipLaunchKernelGGL(vq_ste_loss_final_kernel, dim3(1), dim3(64), 0, s, ws, (int)blocks, scale, loss, 1.0f);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
