// Fused NeuS network evaluation for gfx950: ray -> points -> posenc -> SDF MLP [-> analytic
// d sdf / d x -> colour MLP], one launch, activations never leave the CU except for the per-tile
// backward stash.  Replaces the framework-op sequences at
//   geo/NeuS-ours2/models/renderer.py:337-338,180-185   (coarse / up-sampling SDF evaluations)   -> vqn_neus_sdf_points
//   geo/NeuS-ours2/models/renderer.py:216-227 + fields.py:72-107,147-172
//        (sdf_network(pts), sdf_network.gradient(pts) = autograd wrt the input, color_network)   -> vqn_neus_fine_points
// The reference's gradient() re-runs the whole SDF forward (fields.py:98) and then calls autograd;
// here the input gradient is an explicit reverse sweep over the same packed weights (transposed
// packs), reusing the forward activations stashed per tile, so the forward runs once.
//
// Execution model: 256-thread workgroups (4 waves), persistent over tiles of 32 points, 2 workgroups
// per CU.  See mlp_prims.h for the LDS "activation image" and the weight-pack layout.
#include "mlp_prims.h"
#include "vqn_neus_desc.h"
#include <stdlib.h>

using namespace eng;

// In-kernel phase stamps (cdna_hip_programming.md section 7): a DIAGNOSTIC build only (make stamps -> lib/libvqnerf_hip_stamps.so,
// -DVQN_STAMPS); the shipped library contains none of this.  Wave 0 of every workgroup accumulates shader-clock cycles per
// phase and adds them to g_stamps at exit: [0] tile set-up (points, posenc), [1] GEMM main loops (operand waits included),
// [2] epilogues, [3] barrier waits, [4] everything else, [5] total, [6] workgroups.
#ifdef VQN_STAMPS
__device__ unsigned long long g_stamps[8];
#define VQN_STAMP_DECL unsigned long long st_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long st_prev = __builtin_amdgcn_s_memtime(); const unsigned long long st_begin = st_prev;
#define VQN_STAMP(i) { const unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_[i] += st_now - st_prev; st_prev = st_now; }
#define VQN_STAMP_FLUSH if (threadIdx.x == 0) { st_[5] = __builtin_amdgcn_s_memtime() - st_begin; for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&g_stamps[i_], st_[i_]); atomicAdd(&g_stamps[6], 1ull); }
extern "C" int vqn_debug_read_stamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return -3;
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}
#else
#define VQN_STAMP_DECL
#define VQN_STAMP(i)
#define VQN_STAMP_FLUSH
#endif

// 16-phase stamps of the two-image kernel (same DIAGNOSTIC build): [0] set-up [1] forward K loops [2] forward epilogues [3] forward
// barrier waits [4] sdf row + feature layer + sdf out [5] G_pre [6] reverse K loops (+ wTE) [7] reverse epilogues [8] reverse barrier
// waits [9] embedding chain rule [10] colour set-up [11] colour layers [12] colour out / turn-around [14] total [15] workgroups.
#ifdef VQN_STAMPS
__device__ unsigned long long g_stamps2[16];
#define FS_DECL unsigned long long fs_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long fs_prev = __builtin_amdgcn_s_memtime(); const unsigned long long fs_begin = fs_prev;
#define FS(i) { const unsigned long long fs_now = __builtin_amdgcn_s_memtime(); fs_[i] += fs_now - fs_prev; fs_prev = fs_now; }
#define FS_FLUSH if (threadIdx.x == 0) { fs_[14] = __builtin_amdgcn_s_memtime() - fs_begin; for (int i_ = 0; i_ < 15; ++i_) atomicAdd(&g_stamps2[i_], fs_[i_]); atomicAdd(&g_stamps2[15], 1ull); }
extern "C" int vqn_debug_read_stamps2(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps2), sizeof(unsigned long long) * 16) != hipSuccess) return -3;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps2), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}
#else
#define FS_DECL
#define FS(i)
#define FS_FLUSH
#endif

namespace {

constexpr int E0 = 0;        // LDS rows [0,8): embedding / colour-net extras / d sdf / d embedding
constexpr int E_ROWS = 8;

struct Smalls {              // per-tile scalars, after the activation rows
  float pts[96], dirs[96], part[512], grad[96];
};

template <bool FINE>
__global__ __launch_bounds__(256, 2) void neus_points_kernel(
    const SdfDesc sd, const ColDesc cd, const f32x4* __restrict__ wsdf, const f32x4* __restrict__ wcol,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ zv,
    const float* __restrict__ pts_direct, const float* __restrict__ dirs_direct, const long P, const int S,
    f32x4* __restrict__ scratch, float* __restrict__ out_sdf, float* __restrict__ out_grad,
    float* __restrict__ out_rgb) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  const int MT = sd.max_tiles;
  const int X0 = E_ROWS, Y0 = E_ROWS + 4 * MT;
  Smalls* sm = reinterpret_cast<Smalls*>(lds + (size_t)(E_ROWS + 8 * MT) * 64);
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: let the compiler know
  const int n_lin = sd.n_lin;
  const int emb_tiles = (sd.emb_feats + 31) >> 5;
  const long n_tiles = (P + 31) >> 5;
  f32x4* save = FINE ? scratch + (size_t)blockIdx.x * (size_t)(n_lin - 1) * 4 * MT * 64 : nullptr;
  const int feat_slot = (n_lin - 2) * 4 * MT;

  VQN_STAMP_DECL
  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p0 = tile << 5;
    VQN_STAMP(4)
    // ---------------- points of this tile ----------------
    if (tid < 32) {
      long pt = p0 + tid;
      if (pt >= P) pt = P - 1;
      float x, y, z, dx = 0.f, dy = 0.f, dz = 0.f;
      if (pts_direct != nullptr) {
        x = pts_direct[pt * 3 + 0]; y = pts_direct[pt * 3 + 1]; z = pts_direct[pt * 3 + 2];
        if (FINE) { dx = dirs_direct[pt * 3 + 0]; dy = dirs_direct[pt * 3 + 1]; dz = dirs_direct[pt * 3 + 2]; }
      } else {
        const long ray = pt / S;
        const float t = zv[pt];
        dx = rays_d[ray * 3 + 0]; dy = rays_d[ray * 3 + 1]; dz = rays_d[ray * 3 + 2];
        // o + d * z with separate mul/add roundings, as the reference's broadcasted expression
        x = rays_o[ray * 3 + 0] + __fmul_rn(dx, t);
        y = rays_o[ray * 3 + 1] + __fmul_rn(dy, t);
        z = rays_o[ray * 3 + 2] + __fmul_rn(dz, t);
      }
      sm->pts[tid * 3 + 0] = x; sm->pts[tid * 3 + 1] = y; sm->pts[tid * 3 + 2] = z;
      sm->dirs[tid * 3 + 0] = dx; sm->dirs[tid * 3 + 1] = dy; sm->dirs[tid * 3 + 2] = dz;
    }
    __syncthreads();
    const float xs = sm->pts[p * 3 + 0] * sd.scale, ys = sm->pts[p * 3 + 1] * sd.scale, zs = sm->pts[p * 3 + 2] * sd.scale;
    // ---------------- positional encoding -> E rows ----------------
    for (int r = wave; r < sd.emb_rows; r += 4) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = row_feat(r, h, j);
        v[j] = f < sd.emb_feats ? posenc_feat(f, xs, ys, zs) : 0.f;
      }
      lds[(E0 + r) * 64 + lane] = v;
    }
    __syncthreads();

    VQN_STAMP(0)
    // ---------------- SDF hidden layers ----------------
    int cur = X0, oth = Y0;
    f32x4 pre[4];                                  // first weight fragments of the next layer's first tile of this wave
    {
      const f32x4* wp0 = wsdf + sd.layers[0].w_off + (size_t)wave * sd.emb_rows * 64 + lane;
#pragma unroll
      for (int i = 0; i < 4; ++i) pre[i] = wp0[i * 64];
    }
    for (int l = 0; l < n_lin - 1; ++l) {
      const LayerDesc L = sd.layers[l];
      const f32x4* next_wp = nullptr;              // layer l + 1 has K rows = this layer's outputs (+ embedding at the skip)
      if (l + 1 < n_lin - 1)
        next_wp = wsdf + sd.layers[l + 1].w_off + (size_t)wave * (4 * L.n_out_tiles + ((l + 1 == sd.skip) ? sd.emb_rows : 0)) * 64 + lane;
      const KSegs ks = (l == 0) ? KSegs{E0, sd.emb_rows, 0, 0}
                                : KSegs{cur, 4 * sd.layers[l - 1].n_out_tiles, E0, (l == sd.skip) ? sd.emb_rows : 0};
      const int dst = (l == 0) ? X0 : oth;
      const bool do_save = FINE && (l < n_lin - 2);
      const f32x4* bp = wsdf + L.b_off;
      f32x4* sv = save + (size_t)l * 4 * MT * 64;
      gemm_tiles_chain(lds, ks, wsdf + L.w_off, L.n_out_tiles, wave, lane, pre, next_wp,
                 [&](int ot, f32x16& acc) { VQN_STAMP(2) init_bias(bp, ot, lane, acc); },
                 [&](int ot, const f32x16& acc) {
                   VQN_STAMP(1)
#pragma unroll
                   for (int rq = 0; rq < 4; ++rq) {
                     f32x4 v = acc_quad(acc, rq);
#pragma unroll
                     for (int j = 0; j < 4; ++j) v[j] = act_fwd<ACT_SOFTPLUS100>(v[j]);
                     lds[(dst + ot * 4 + rq) * 64 + lane] = v;
                     if (do_save) st_stream(sv + (ot * 4 + rq) * 64 + lane, v);
                   }
                 });
      VQN_STAMP(2)
      __syncthreads();
      VQN_STAMP(3)
      if (l == 0) { cur = X0; oth = Y0; } else { const int t = cur; cur = oth; oth = t; }
    }
    const int hid_rows = 4 * sd.layers[n_lin - 2].n_out_tiles;

    // ---------------- last layer: sdf row (VALU dot) [+ feature rows -> stash] ----------------
    rowdot<1>(lds, cur, hid_rows, wsdf + sd.last_w_off, sm->part, wave, lane);
    if (FINE && sd.layers[n_lin - 1].n_out_tiles > 0) {
      const LayerDesc L = sd.layers[n_lin - 1];
      const f32x4* bp = wsdf + L.b_off;
      f32x4* sv = save + (size_t)feat_slot * 64;
      gemm_tiles(lds, KSegs{cur, hid_rows, 0, 0}, wsdf + L.w_off, L.n_out_tiles, wave, lane,
                 [&](int ot, f32x16& acc) { init_bias(bp, ot, lane, acc); },
                 [&](int ot, const f32x16& acc) {
#pragma unroll
                   for (int rq = 0; rq < 4; ++rq) st_stream(sv + (ot * 4 + rq) * 64 + lane, acc_quad(acc, rq));
                 });
    }
    __syncthreads();
    if (tid < 32 && p0 + tid < P) {
      const float s = ((sm->part[tid] + sm->part[32 + tid]) + (sm->part[64 + tid] + sm->part[96 + tid])) +
                      (sd.last_b_off > 0 ? wsdf[sd.last_b_off][0] : sd.last_bias);
      out_sdf[p0 + tid] = s / sd.scale;
    }
    if (!FINE) { __syncthreads(); continue; }

    // ---------------- reverse sweep: d sdf / d x ----------------
    // G_pre(last hidden) = w_sdf_row (.) act'(h), in place
    for (int r0 = wave; r0 < hid_rows; r0 += 32) {            // 8 rows per pass: all fetches first (one L2 round trip per pass)
      f32x4 v[8], wv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int r = min(r0 + 4 * c, hid_rows - 1);
        v[c] = lds[(cur + r) * 64 + lane];
        wv[c] = wsdf[sd.last_w_off + r * 2 + h];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (r0 + 4 * c < hid_rows) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[c][j] = wv[c][j] * act_bwd_from_out<ACT_SOFTPLUS100>(v[c][j]);
          lds[(cur + r0 + 4 * c) * 64 + lane] = v[c];
        }
    }
    __syncthreads();
    for (int l = n_lin - 2; l >= 1; --l) {
      const LayerDesc L = sd.layers[l];
      const KSegs ks{cur, 4 * L.n_out_tiles, 0, 0};
      const f32x4* sv = save + (size_t)(l - 1) * 4 * MT * 64;
      const int dst = oth;
      f32x4 hv[4];                                   // stashed forward activations of this tile: requested BEFORE the K loop so that
                                                     // their (L2 / Infinity-Cache) latency hides under the MFMAs, not behind them
      gemm_tiles(lds, ks, wsdf + L.wT_off, sd.layers[l - 1].n_out_tiles, wave, lane,
                 [&](int ot, f32x16& acc) {
#pragma unroll
                   for (int rq = 0; rq < 4; ++rq) hv[rq] = ld_stream(sv + (ot * 4 + rq) * 64 + lane);
                   init_zero(acc);
                 },
                 [&](int ot, const f32x16& acc) {
#pragma unroll
                   for (int rq = 0; rq < 4; ++rq) {
                     f32x4 v = acc_quad(acc, rq);
#pragma unroll
                     for (int j = 0; j < 4; ++j) v[j] *= act_bwd_from_out<ACT_SOFTPLUS100>(hv[rq][j]);
                     lds[(dst + ot * 4 + rq) * 64 + lane] = v;
                   }
                 });
      if (l == sd.skip)
        gemm_tiles(lds, ks, wsdf + L.wTE_off, emb_tiles, wave, lane,
                   [&](int, f32x16& acc) { init_zero(acc); },
                   [&](int ot, const f32x16& acc) {
#pragma unroll
                     for (int rq = 0; rq < 4; ++rq) lds[(E0 + ot * 4 + rq) * 64 + lane] = acc_quad(acc, rq);
                   });
      __syncthreads();
      const int t = cur; cur = oth; oth = t;
    }
    {
      const LayerDesc L = sd.layers[0];
      const bool accumulate = sd.skip >= 1;
      gemm_tiles(lds, KSegs{cur, 4 * L.n_out_tiles, 0, 0}, wsdf + L.wTE_off, emb_tiles, wave, lane,
                 [&](int ot, f32x16& acc) {
                   if (accumulate) init_rows(lds + (E0 + ot * 4) * 64, lane, acc); else init_zero(acc);
                 },
                 [&](int ot, const f32x16& acc) {
#pragma unroll
                   for (int rq = 0; rq < 4; ++rq) lds[(E0 + ot * 4 + rq) * 64 + lane] = acc_quad(acc, rq);
                 });
    }
    __syncthreads();
    // chain through the embedding: thread (point pp, component c) sums its features in a fixed order
    if (tid < 96) {
      const int pp = tid & 31, c = tid >> 5;
      const float x0 = sm->pts[pp * 3 + 0] * sd.scale, x1 = sm->pts[pp * 3 + 1] * sd.scale, x2 = sm->pts[pp * 3 + 2] * sd.scale;
      const float* ldsf = reinterpret_cast<const float*>(lds);
      auto G = [&](int f) {
        const int t = f >> 5, fi = f & 31, hh = fi & 1, rr = fi >> 1;
        return ldsf[(((E0 + t * 4 + (rr >> 2)) * 64) + pp + 32 * hh) * 4 + (rr & 3)];
      };
      float g = G(c);
      int cc;
      for (int k = 0; k < sd.multires; ++k) {
        const int fs = 3 + 6 * k + c, fc = fs + 3;
        g = fmaf(G(fs), posenc_jac(fs, x0, x1, x2, &cc), g);
        g = fmaf(G(fc), posenc_jac(fc, x0, x1, x2, &cc), g);
      }
      sm->grad[pp * 3 + c] = g;
      if (p0 + pp < P) out_grad[(p0 + pp) * 3 + c] = g;
    }
    __syncthreads();
    if (cd.n_lin == 0) continue;          // SDFNetwork.gradient(): no colour net

    // ---------------- colour network ----------------
    {
      const float px = sm->pts[p * 3 + 0], py = sm->pts[p * 3 + 1], pz = sm->pts[p * 3 + 2];
      const float dx = sm->dirs[p * 3 + 0], dy = sm->dirs[p * 3 + 1], dz = sm->dirs[p * 3 + 2];
      for (int r = wave; r < cd.extra_rows; r += 4) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int f = row_feat(r, h, j);
          float val = 0.f;
          if (f < 3) val = f == 0 ? px : (f == 1 ? py : pz);
          else if (f < 3 + cd.n_view_feats) val = posenc_feat(f - 3, dx, dy, dz);
          else if (f < cd.extra_feats) val = sm->grad[p * 3 + (f - 3 - cd.n_view_feats)];
          v[j] = val;
        }
        lds[(E0 + r) * 64 + lane] = v;
      }
      const f32x4* sv = save + (size_t)feat_slot * 64;
      const int feat_rows = 4 * sd.layers[n_lin - 1].n_out_tiles;
      for (int r0 = wave; r0 < feat_rows; r0 += 32) {          // feature rows back from the stash: 8 fetches in flight per pass
        f32x4 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = ld_stream(sv + min(r0 + 4 * c, feat_rows - 1) * 64 + lane);
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (r0 + 4 * c < feat_rows) lds[(X0 + r0 + 4 * c) * 64 + lane] = v[c];
      }
      __syncthreads();
      cur = X0; oth = Y0;
      int in_rows = feat_rows;
      for (int l = 0; l < cd.n_lin - 1; ++l) {
        const LayerDesc L = cd.layers[l];
        const KSegs ks{cur, in_rows, E0, l == 0 ? cd.extra_rows : 0};
        const f32x4* bp = wcol + L.b_off;
        const int dst = oth;
        gemm_tiles(lds, ks, wcol + L.w_off, L.n_out_tiles, wave, lane,
                   [&](int ot, f32x16& acc) { init_bias(bp, ot, lane, acc); },
                   [&](int ot, const f32x16& acc) {
#pragma unroll
                     for (int rq = 0; rq < 4; ++rq) {
                       f32x4 v = acc_quad(acc, rq);
#pragma unroll
                       for (int j = 0; j < 4; ++j) v[j] = act_fwd<ACT_RELU>(v[j]);
                       lds[(dst + ot * 4 + rq) * 64 + lane] = v;
                     }
                   });
        __syncthreads();
        const int t = cur; cur = oth; oth = t;
        in_rows = 4 * L.n_out_tiles;
      }
      rowdot<3>(lds, cur, in_rows, wcol + cd.last_w_off, sm->part, wave, lane);
      __syncthreads();
      if (tid < 96) {
        const int pp = tid & 31, o = tid >> 5;
        float v = ((sm->part[(0 * 32 + pp) * 3 + o] + sm->part[(1 * 32 + pp) * 3 + o]) +
                   (sm->part[(2 * 32 + pp) * 3 + o] + sm->part[(3 * 32 + pp) * 3 + o])) +
                  (cd.last_b_off > 0 ? wcol[cd.last_b_off][o] : cd.last_bias[o]);
        if (cd.squeeze_out) v = 1.f / (1.f + expf(-v));
        if (p0 + pp < P) out_rgb[(p0 + pp) * 3 + o] = v;
      }
    }
    __syncthreads();
  }
  VQN_STAMP(4)
  VQN_STAMP_FLUSH
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-image form (the default for networks of >= 5 tiles that fit): one 512-thread workgroup per CU holds TWO 32-point
// images; wave w owns output tiles w, w + 8, ... and applies each weight fragment to both images (gemm_tiles2).  All eight
// waves do the same matrix work between two barriers and no second workgroup competes for the matrix pipe, so the waves
// of a layer finish together (the one-image form loses ~20 % to barrier skew between its two co-resident workgroups);
// barriers and the L2 weight stream per point halve.  The VALU phases split by image: waves 0-3 image 0, waves 4-7 image 1.
// Same per-point arithmetic as the one-image kernel except the two embedding-gradient GEMMs of the reverse sweep, which are
// split over K (wte_split_k: a different, still fixed, summation order for d sdf / d x).
struct Smalls2 {
  float pts[2][96], dirs[2][96], part[2][512], grad[2][96];
};

// d sdf / d embedding of one layer for both images: E rows (+)= W_E^T . G over the K rows [k_row0, k_row0 + ng).  The result
// has only emb_tiles (<= 2) output tiles, so as a plain gemm_tiles2 call it keeps 2 of the 8 waves busy for a whole K = 256 pass;
// here the waves split it as (tile, K slice) units, park their partial tiles in the free activation buffer `tmp_row0`
// (4 rows per unit and image, `tmp_rows` available: 8 units at d_hidden = 256) and sum them in a fixed slice order.  Ends with the E rows written and a barrier passed.
__device__ __forceinline__ void wte_split_k(f32x4* __restrict__ lds, const int IS, const int k_row0, const int ng,
                                            const f32x4* __restrict__ w, const int emb_tiles, const int tmp_row0, const int tmp_rows,
                                            const int e_row0, const bool accumulate, const int wave, const int lane) {
  const int KS = min(8, tmp_rows >> 2) / emb_tiles, gk = (ng + KS - 1) / KS;          // K slices (4 parking rows each), groups per slice
  const int t = wave % emb_tiles, kq = wave / emb_tiles;
  if (kq < KS) {
    f32x16 acc0, acc1;
    init_zero(acc0); init_zero(acc1);
    const int g0 = kq * gk, g1 = min(ng, g0 + gk);
    const f32x4* __restrict__ wp = w + (size_t)t * ng * 64 + lane;
    for (int g = g0; g < g1; g += 8) {
      f32x4 a[8], p[8], q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int gg = min(g + i, ng - 1), r = (k_row0 + gg) * 64 + lane;
        a[i] = wp[gg * 64]; p[i] = lds[r]; q[i] = lds[r + IS];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (g + i < g1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], p[i][j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], q[i][j], acc1, 0, 0, 0);
          }
        }
    }
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      lds[(tmp_row0 + wave * 4 + rq) * 64 + lane] = acc_quad(acc0, rq);
      lds[IS + (tmp_row0 + wave * 4 + rq) * 64 + lane] = acc_quad(acc1, rq);
    }
  }
  __syncthreads();
  for (int r = wave; r < 8 * emb_tiles; r += 8) {                   // (image, tile, row quad) units
    const int im = r / (4 * emb_tiles), er = r - im * 4 * emb_tiles, tt = er >> 2, rq = er & 3;
    f32x4* li = lds + (size_t)im * IS;
    f32x4 sum = li[(tmp_row0 + tt * 4 + rq) * 64 + lane];
    for (int k = 1; k < KS; ++k) sum += li[(tmp_row0 + (k * emb_tiles + tt) * 4 + rq) * 64 + lane];
    if (accumulate) sum += li[(e_row0 + er) * 64 + lane];
    li[(e_row0 + er) * 64 + lane] = sum;
  }
  __syncthreads();
}

// ---- training forward (TRAIN): the two-image fine kernel also leaves what the backward tile programs and the weight-gradient
// contraction read (geo/train_programs.py, prog_fwd's stores) in the tile format of csrc/tile_vm.hip, [point tile][feature tile][32
// features][32 points] f32: E (embedding), U_1..U_nL (hidden activations), OUTF ([sdf ; features], 257 rows), GH_0..GH_{nL-1} (adjoints of
// the reverse sweep), EXTR (colour-net extras), C_1..C_nC (colour activations).  A row quad of the activation image -- lane (p, h), component
// j = feature 2 (4 rq + j) + h of the tile -- goes out as four 256-byte stores (feature rows 8 rq + 2 j and 8 rq + 2 j + 1 are adjacent).
struct TrainOut {
  float* E; float* OUTF; float* EXTR;
  float* U[VQN_MAX_SDF_LAYERS]; float* GH[VQN_MAX_SDF_LAYERS]; float* C[VQN_MAX_COL_LAYERS];
  int e_tiles, outf_tiles, extr_tiles;
};

__device__ __forceinline__ void tfmt_store_quad(float* __restrict__ T, const long ptile, const int n_ft, const int ft, const int rq,
                                                const int lane, const f32x4 v) {
  float* base = T + ((ptile * n_ft + ft) * 32 + 8 * rq) * 32 + lane;
#pragma unroll
  for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v[j], base + 64 * j);     // written once, read by later launches: past the L2-resident packs
}

template <bool FINE, bool TRAIN = false>
__global__ __launch_bounds__(512, 1) void neus_points2_kernel(
    const SdfDesc sd, const ColDesc cd, const f32x4* __restrict__ wsdf, const f32x4* __restrict__ wcol,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ zv,
    const float* __restrict__ pts_direct, const float* __restrict__ dirs_direct, const long P, const int S,
    f32x4* __restrict__ scratch, float* __restrict__ out_sdf, float* __restrict__ out_grad,
    float* __restrict__ out_rgb, const TrainOut to) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  const int MT = sd.max_tiles;
  const int IMG = E_ROWS + 8 * MT, IS = IMG * 64;
  const int X0 = E_ROWS, Y0 = E_ROWS + 4 * MT;
  Smalls2* sm = reinterpret_cast<Smalls2*>(lds + (size_t)2 * IS);
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = wave >> 2, w4 = wave & 3;
  f32x4* ldsi = lds + (size_t)img * IS;
  const int n_lin = sd.n_lin;
  const int emb_tiles = (sd.emb_feats + 31) >> 5;
  const long n_tiles = (P + 31) >> 5, n_pairs = (n_tiles + 1) >> 1;
  const size_t per_img = (size_t)(n_lin - 1) * 4 * MT * 64;
#ifdef VQN_DIAG_STASH_ROT     // timing only (make diag FLAG=VQN_DIAG_STASH_ROT=4): every workgroup walks ROT stash regions in turn, so ROT x 128 MB of
                              // stash are in flight instead of 128 MB -- is the stash fast because it sits in the 256 MiB Infinity Cache?
  f32x4* save0 = nullptr;
#else
  f32x4* save0 = FINE ? scratch + (size_t)blockIdx.x * 2 * per_img : nullptr;
#endif
  const int feat_slot = (n_lin - 2) * 4 * MT;
  f32x4 pre[4];
  f32x4 nopre[4];

  FS_DECL
  for (long pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
#ifdef VQN_DIAG_STASH_ROT
    if (FINE) save0 = scratch + ((size_t)((pair / gridDim.x) % VQN_DIAG_STASH_ROT) * gridDim.x + blockIdx.x) * 2 * per_img;
#endif
    FS(12)
    // ---------------- points of both tiles ----------------
    if (tid < 64) {
      const int im = tid >> 5, t = tid & 31;
      long pt = ((2 * pair + im) << 5) + t;
      if (pt >= P) pt = P - 1;
      float x, y, z, dx = 0.f, dy = 0.f, dz = 0.f;
      if (pts_direct != nullptr) {
        x = pts_direct[pt * 3 + 0]; y = pts_direct[pt * 3 + 1]; z = pts_direct[pt * 3 + 2];
        if (FINE) { dx = dirs_direct[pt * 3 + 0]; dy = dirs_direct[pt * 3 + 1]; dz = dirs_direct[pt * 3 + 2]; }
      } else {
        const long ray = pt / S;
        const float tt = zv[pt];
        dx = rays_d[ray * 3 + 0]; dy = rays_d[ray * 3 + 1]; dz = rays_d[ray * 3 + 2];
        x = rays_o[ray * 3 + 0] + __fmul_rn(dx, tt);
        y = rays_o[ray * 3 + 1] + __fmul_rn(dy, tt);
        z = rays_o[ray * 3 + 2] + __fmul_rn(dz, tt);
      }
      sm->pts[im][t * 3 + 0] = x; sm->pts[im][t * 3 + 1] = y; sm->pts[im][t * 3 + 2] = z;
      sm->dirs[im][t * 3 + 0] = dx; sm->dirs[im][t * 3 + 1] = dy; sm->dirs[im][t * 3 + 2] = dz;
    }
    __syncthreads();
    const float xs = sm->pts[img][p * 3 + 0] * sd.scale, ys = sm->pts[img][p * 3 + 1] * sd.scale, zs = sm->pts[img][p * 3 + 2] * sd.scale;
    const long ptile_w = 2 * pair + img;                   // (TRAIN) the point tile of this wave's image; stores are skipped for a phantom tile
    for (int r = w4; r < (TRAIN ? 4 * to.e_tiles : sd.emb_rows); r += 4) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = row_feat(r, h, j);
        v[j] = (r < sd.emb_rows && f < sd.emb_feats) ? posenc_feat(f, xs, ys, zs) : 0.f;
      }
      if (r < sd.emb_rows) ldsi[(E0 + r) * 64 + lane] = v;
      if (TRAIN && ptile_w < n_tiles) tfmt_store_quad(to.E, ptile_w, to.e_tiles, r >> 2, r & 3, lane, v);
    }
    __syncthreads();

    FS(0)
    // ---------------- SDF hidden layers ----------------
    int cur = X0, oth = Y0;
    if (wave < sd.layers[0].n_out_tiles) {
      const f32x4* wp0 = wsdf + sd.layers[0].w_off + (size_t)wave * sd.emb_rows * 64 + lane;
#pragma unroll
      for (int i = 0; i < 4; ++i) pre[i] = wp0[min(i, sd.emb_rows - 1) * 64];
    }
    for (int l = 0; l < n_lin - 1; ++l) {
      const LayerDesc L = sd.layers[l];
      const f32x4* next_wp = nullptr;
      if (l + 1 < n_lin - 1 && wave < sd.layers[l + 1].n_out_tiles)
        next_wp = wsdf + sd.layers[l + 1].w_off + (size_t)wave * (4 * L.n_out_tiles + ((l + 1 == sd.skip) ? sd.emb_rows : 0)) * 64 + lane;
      const KSegs ks = (l == 0) ? KSegs{E0, sd.emb_rows, 0, 0}
                                : KSegs{cur, 4 * sd.layers[l - 1].n_out_tiles, E0, (l == sd.skip) ? sd.emb_rows : 0};
      const int dst = (l == 0) ? X0 : oth;
      const bool do_save = FINE && (l < n_lin - 2);
      const f32x4* bp = wsdf + L.b_off;
      float* const t_u = TRAIN ? to.U[l + 1] : nullptr;          // (one scalar load per layer: no dynamic index inside the epilogue)
      auto bias_init = [&](int ot, int im, f32x16& acc) { if (im == 0) { FS(2) } init_bias(bp, ot, lane, acc); };
      auto epi_rq = [&](int ot, int im, int rq, const f32x16& acc) {
        if (im == 0 && rq == 0) { FS(1) }
        f32x4* li = lds + (size_t)im * IS;
        f32x4* sv = save0 + (size_t)im * per_img + (size_t)l * 4 * MT * 64;
        f32x4 v = {acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_fwd<ACT_SOFTPLUS100>(v[j]);
        li[(dst + ot * 4 + rq) * 64 + lane] = v;
        if (do_save) st_stream(sv + (ot * 4 + rq) * 64 + lane, v);
        if (TRAIN && 2 * pair + im < n_tiles) tfmt_store_quad(t_u, 2 * pair + im, L.n_out_tiles, ot, rq, lane, v);
      };
      gemm_tiles2<8>(lds, IS, ks, wsdf + L.w_off, L.n_out_tiles, wave, lane, pre, true, next_wp, bias_init, epi_rq);
      FS(2)
      __syncthreads();
      FS(3)
      if (l == 0) { cur = X0; oth = Y0; } else { const int t = cur; cur = oth; oth = t; }
    }
    const int hid_rows = 4 * sd.layers[n_lin - 2].n_out_tiles;

    // ---------------- last layer: sdf row (VALU dot, per image) [+ feature rows -> stash] ----------------
    rowdot<1>(ldsi, cur, hid_rows, wsdf + sd.last_w_off, sm->part[img], w4, lane);
    if (FINE && sd.layers[n_lin - 1].n_out_tiles > 0) {
      const LayerDesc L = sd.layers[n_lin - 1];
      const f32x4* bp = wsdf + L.b_off;
      gemm_tiles2<8>(lds, IS, KSegs{cur, hid_rows, 0, 0}, wsdf + L.w_off, L.n_out_tiles, wave, lane, nopre, false, nullptr,
                     [&](int ot, int, f32x16& acc) { init_bias(bp, ot, lane, acc); },
                     [&](int ot, int im, int rq, const f32x16& acc) {
                       f32x4* sv = save0 + (size_t)im * per_img + (size_t)feat_slot * 64;
                       const f32x4 v = {acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]};
                       st_stream(sv + (ot * 4 + rq) * 64 + lane, v);
                       if (TRAIN && 2 * pair + im < n_tiles) {              // OUTF = [sdf ; features]: feature f of this GEMM is row f + 1
                         float* base = to.OUTF + (2 * pair + im) * (long)to.outf_tiles * 1024;
#pragma unroll
                         for (int j = 0; j < 4; ++j) {
                           const int f = 32 * ot + 2 * (4 * rq + j) + h + 1;
                           if (f < 32 * to.outf_tiles) __builtin_nontemporal_store(v[j], base + f * 32 + p);
                         }
                       }
                     });
    }
    __syncthreads();
    if (tid < 64) {
      const int im = tid >> 5, t = tid & 31;
      const long pt = ((2 * pair + im) << 5) + t;
      if (pt < P) {
        const float* pr = sm->part[im];
        const float s = ((pr[t] + pr[32 + t]) + (pr[64 + t] + pr[96 + t])) + (sd.last_b_off > 0 ? wsdf[sd.last_b_off][0] : sd.last_bias);
        out_sdf[pt] = s / sd.scale;
      }
      if (TRAIN && 2 * pair + im < n_tiles) {                        // row 0 of OUTF (the raw sdf output) and the zero tail beyond row F - 1
        float* base = to.OUTF + (2 * pair + im) * (long)to.outf_tiles * 1024;
        const float* pr = sm->part[im];
        base[t] = ((pr[t] + pr[32 + t]) + (pr[64 + t] + pr[96 + t])) + (sd.last_b_off > 0 ? wsdf[sd.last_b_off][0] : sd.last_bias);
        for (int f = 32 * sd.layers[n_lin - 1].n_out_tiles + 1; f < 32 * to.outf_tiles; ++f) base[f * 32 + t] = 0.f;
      }
    }
    if (!FINE) { __syncthreads(); FS(4) continue; }
    FS(4)

    // ---------------- reverse sweep: d sdf / d x ----------------
    float* const t_gh_top = TRAIN ? to.GH[n_lin - 2] : nullptr;
    for (int r0 = w4; r0 < hid_rows; r0 += 32) {
      f32x4 v[8], wv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int r = min(r0 + 4 * c, hid_rows - 1);
        v[c] = ldsi[(cur + r) * 64 + lane];
        wv[c] = wsdf[sd.last_w_off + r * 2 + h];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (r0 + 4 * c < hid_rows) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[c][j] = wv[c][j] * act_bwd_from_out<ACT_SOFTPLUS100>(v[c][j]);
          ldsi[(cur + r0 + 4 * c) * 64 + lane] = v[c];
          if (TRAIN && ptile_w < n_tiles)
            tfmt_store_quad(t_gh_top, ptile_w, sd.layers[n_lin - 2].n_out_tiles, (r0 + 4 * c) >> 2, (r0 + 4 * c) & 3, lane, v[c]);
        }
    }
    __syncthreads();
    FS(5)
    for (int l = n_lin - 2; l >= 1; --l) {
      const LayerDesc L = sd.layers[l];
      const KSegs ks{cur, 4 * L.n_out_tiles, 0, 0};
      const int dst = oth;
      if (l == sd.skip)                                 // embedding part of the skip layer first: `oth` is still free for the partials
        wte_split_k(lds, IS, cur, 4 * L.n_out_tiles, wsdf + L.wTE_off, emb_tiles, oth, 4 * MT, E0, false, wave, lane);
      f32x4 hv[2][4];
      float* const t_gh = TRAIN ? to.GH[l - 1] : nullptr;
      const int gh_tiles = sd.layers[l - 1].n_out_tiles;
      gemm_tiles2<8>(lds, IS, ks, wsdf + L.wT_off, sd.layers[l - 1].n_out_tiles, wave, lane, nopre, false, nullptr,
                     [&](int ot, int im, f32x16& acc) {
                       if (im == 0) { FS(7) }
                       const f32x4* sv = save0 + (size_t)im * per_img + (size_t)(l - 1) * 4 * MT * 64;
#pragma unroll
                       for (int rq = 0; rq < 4; ++rq) hv[im][rq] = ld_stream(sv + (ot * 4 + rq) * 64 + lane);
                       init_zero(acc);
                     },
                     [&](int ot, int im, int rq, const f32x16& acc) {
                       if (im == 0 && rq == 0) { FS(6) }
                       f32x4* li = lds + (size_t)im * IS;
                       f32x4 v = {acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]};
#pragma unroll
                       for (int j = 0; j < 4; ++j) v[j] *= act_bwd_from_out<ACT_SOFTPLUS100>(hv[im][rq][j]);
                       li[(dst + ot * 4 + rq) * 64 + lane] = v;
                       if (TRAIN && 2 * pair + im < n_tiles) tfmt_store_quad(t_gh, 2 * pair + im, gh_tiles, ot, rq, lane, v);
                     });
      FS(7)
      __syncthreads();
      FS(8)
      const int t = cur; cur = oth; oth = t;
    }
    wte_split_k(lds, IS, cur, 4 * sd.layers[0].n_out_tiles, wsdf + sd.layers[0].wTE_off, emb_tiles, oth, 4 * MT, E0, sd.skip >= 1, wave, lane);
    FS(6)
    if (tid < 192) {
      const int im = tid / 96, r = tid - 96 * im, pp = r & 31, c = r >> 5;
      const float x0 = sm->pts[im][pp * 3 + 0] * sd.scale, x1 = sm->pts[im][pp * 3 + 1] * sd.scale, x2 = sm->pts[im][pp * 3 + 2] * sd.scale;
      const float* ldsf = reinterpret_cast<const float*>(lds + (size_t)im * IS);
      auto G = [&](int f) {
        const int t = f >> 5, fi = f & 31, hh = fi & 1, rr = fi >> 1;
        return ldsf[(((E0 + t * 4 + (rr >> 2)) * 64) + pp + 32 * hh) * 4 + (rr & 3)];
      };
      float g = G(c);
      int cc;
      for (int k = 0; k < sd.multires; ++k) {
        const int fs = 3 + 6 * k + c, fc = fs + 3;
        g = fmaf(G(fs), posenc_jac(fs, x0, x1, x2, &cc), g);
        g = fmaf(G(fc), posenc_jac(fc, x0, x1, x2, &cc), g);
      }
      sm->grad[im][pp * 3 + c] = g;
      const long pt = ((2 * pair + im) << 5) + pp;
      if (pt < P) out_grad[pt * 3 + c] = g;
    }
    __syncthreads();
    FS(9)
    if (cd.n_lin == 0) continue;

    // ---------------- colour network ----------------
    {
      const float px = sm->pts[img][p * 3 + 0], py = sm->pts[img][p * 3 + 1], pz = sm->pts[img][p * 3 + 2];
      const float dx = sm->dirs[img][p * 3 + 0], dy = sm->dirs[img][p * 3 + 1], dz = sm->dirs[img][p * 3 + 2];
      for (int r = w4; r < (TRAIN ? 4 * to.extr_tiles : cd.extra_rows); r += 4) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int f = row_feat(r, h, j);
          float val = 0.f;
          if (r >= cd.extra_rows) val = 0.f;
          else if (f < 3) val = f == 0 ? px : (f == 1 ? py : pz);
          else if (f < 3 + cd.n_view_feats) val = posenc_feat(f - 3, dx, dy, dz);
          else if (f < cd.extra_feats) val = sm->grad[img][p * 3 + (f - 3 - cd.n_view_feats)];
          v[j] = val;
        }
        if (r < cd.extra_rows) ldsi[(E0 + r) * 64 + lane] = v;
        if (TRAIN && ptile_w < n_tiles) tfmt_store_quad(to.EXTR, ptile_w, to.extr_tiles, r >> 2, r & 3, lane, v);
      }
      const f32x4* sv = save0 + (size_t)img * per_img + (size_t)feat_slot * 64;
      const int feat_rows = 4 * sd.layers[n_lin - 1].n_out_tiles;
      for (int r0 = w4; r0 < feat_rows; r0 += 32) {
        f32x4 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = ld_stream(sv + min(r0 + 4 * c, feat_rows - 1) * 64 + lane);
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (r0 + 4 * c < feat_rows) ldsi[(X0 + r0 + 4 * c) * 64 + lane] = v[c];
      }
      __syncthreads();
      FS(10)
      cur = X0; oth = Y0;
      int in_rows = feat_rows;
      for (int l = 0; l < cd.n_lin - 1; ++l) {
        const LayerDesc L = cd.layers[l];
        const KSegs ks{cur, in_rows, E0, l == 0 ? cd.extra_rows : 0};
        const f32x4* bp = wcol + L.b_off;
        const int dst = oth;
        float* const t_c = TRAIN ? to.C[l + 1] : nullptr;
        gemm_tiles2<8>(lds, IS, ks, wcol + L.w_off, L.n_out_tiles, wave, lane, nopre, false, nullptr,
                       [&](int ot, int, f32x16& acc) { init_bias(bp, ot, lane, acc); },
                       [&](int ot, int im, int rq, const f32x16& acc) {
                         f32x4* li = lds + (size_t)im * IS;
                         f32x4 v = {acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]};
#pragma unroll
                         for (int j = 0; j < 4; ++j) v[j] = act_fwd<ACT_RELU>(v[j]);
                         li[(dst + ot * 4 + rq) * 64 + lane] = v;
                         if (TRAIN && 2 * pair + im < n_tiles) tfmt_store_quad(t_c, 2 * pair + im, L.n_out_tiles, ot, rq, lane, v);
                       });
        __syncthreads();
        const int t = cur; cur = oth; oth = t;
        in_rows = 4 * L.n_out_tiles;
      }
      FS(11)
      rowdot<3>(ldsi, cur, in_rows, wcol + cd.last_w_off, sm->part[img], w4, lane);
      __syncthreads();
      if (tid < 192) {
        const int im = tid / 96, r = tid - 96 * im, pp = r & 31, o = r >> 5;
        const float* pr = sm->part[im];
        float v = ((pr[(0 * 32 + pp) * 3 + o] + pr[(1 * 32 + pp) * 3 + o]) + (pr[(2 * 32 + pp) * 3 + o] + pr[(3 * 32 + pp) * 3 + o])) +
                  (cd.last_b_off > 0 ? wcol[cd.last_b_off][o] : cd.last_bias[o]);
        if (cd.squeeze_out) v = 1.f / (1.f + expf(-v));
        const long pt = ((2 * pair + im) << 5) + pp;
        if (pt < P) out_rgb[pt * 3 + o] = v;
      }
    }
    __syncthreads();
  }
  FS(12)
  FS_FLUSH
}

int check_sdf_desc(const SdfDesc& d) {
  if (d.n_lin < 2 || d.n_lin > VQN_MAX_SDF_LAYERS) return 1;
  if (d.max_tiles < 1 || d.max_tiles > 16) return 2;
  if (d.emb_feats < 3 || d.emb_feats > 64 || d.emb_rows < 1 || d.emb_rows > 8) return 3;
  if (d.skip >= d.n_lin - 1 || d.skip == 0) return 4;
  for (int l = 0; l < d.n_lin; ++l)
    if (d.layers[l].n_out_tiles < 0 || d.layers[l].n_out_tiles > d.max_tiles) return 5;
  if (!(d.scale > 0.f)) return 6;
  return 0;
}

size_t lds_bytes(int MT) { return (size_t)(E_ROWS + 8 * MT) * 1024 + sizeof(Smalls); }
size_t lds_bytes2(int MT) { return (size_t)2 * (E_ROWS + 8 * MT) * 1024 + sizeof(Smalls2); }

// two 32-point images per workgroup for networks wide enough to give eight waves a tile each (and small enough to fit),
// unless VQN_NEUS_TILE32 is set
bool use_two_images(int MT) {
  static const bool forced32 = getenv("VQN_NEUS_TILE32") != nullptr;
  return !forced32 && MT >= 5 && lds_bytes2(MT) <= 160 * 1024;
}

}  // namespace

extern "C" int vqn_neus_sdf_points(const int32_t* sdf_desc, const float* wbuf_sdf, const float* rays_o,
                                   const float* rays_d, const float* z, const float* pts, int64_t P, int S,
                                   float* out_sdf, void* stream) {
  VQN_CHECK_ARG(sdf_desc && wbuf_sdf && out_sdf, "sdf_desc, wbuf_sdf, out_sdf must be non-null");
  VQN_CHECK_ARG(P >= 0, "P >= 0");
  if (P == 0) return VQN_OK;
  VQN_CHECK_ARG(pts != nullptr || (rays_o && rays_d && z && S > 0), "either pts or (rays_o, rays_d, z, S) required");
  SdfDesc sd;
  memcpy(&sd, sdf_desc, sizeof(SdfDesc));
  VQN_CHECK_SHAPE(check_sdf_desc(sd) == 0, "invalid SDF network descriptor");
  ColDesc cd;
  memset(&cd, 0, sizeof(cd));
  const long n_tiles = (P + 31) / 32;
  if (use_two_images(sd.max_tiles)) {
    const size_t lds2 = lds_bytes2(sd.max_tiles);
    VQN_HIP(hipFuncSetAttribute((const void*)neus_points2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    long grid = (long)vqn_num_cus();
    if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
    hipLaunchKernelGGL(neus_points2_kernel<false>, dim3((unsigned)grid), dim3(512), lds2, (hipStream_t)stream, sd, cd,
                       reinterpret_cast<const f32x4*>(wbuf_sdf), (const f32x4*)nullptr, rays_o, rays_d, z, pts,
                       (const float*)nullptr, (long)P, S, (f32x4*)nullptr, out_sdf, (float*)nullptr, (float*)nullptr, TrainOut{});
    VQN_LAUNCH_CHECK();
    return VQN_OK;
  }
  const size_t lds = lds_bytes(sd.max_tiles);
  VQN_CHECK_SHAPE(lds <= 160 * 1024, "network too wide for LDS");
  if (lds > 64 * 1024)
    VQN_HIP(hipFuncSetAttribute((const void*)neus_points_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long grid = (long)vqn_num_cus() * 2;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(neus_points_kernel<false>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, sd, cd,
                     reinterpret_cast<const f32x4*>(wbuf_sdf), (const f32x4*)nullptr, rays_o, rays_d, z, pts,
                     (const float*)nullptr, (long)P, S, (f32x4*)nullptr, out_sdf, (float*)nullptr, (float*)nullptr);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

// Training forward: vqn_neus_fine_points at explicit (pts, dirs) that ALSO writes the saved tensors of the training engine (see
// TrainOut above).  tensors: device pointers in the order [E, OUTF, EXTR, U_1..U_nL, GH_0..GH_{nL-1}, C_1..C_nC] with nL = n_lin - 1
// SDF hidden layers and nC = col n_lin - 1 colour hidden layers; each [ceil(P/32)][tiles][32][32] f32 with tiles = ceil(width / 32)
// (e_tiles / outf_tiles / extr_tiles given, the others from the descriptors).  Needs the two-image form (networks of >= 5 tiles that fit).
extern "C" int vqn_neus_train_fwd(const int32_t* sdf_desc, const float* wbuf_sdf, const int32_t* col_desc, const float* wbuf_col,
                                  const float* pts, const float* dirs, int64_t P, void* scratch, int64_t scratch_bytes,
                                  float* const* tensors, int n_tensors, int e_tiles, int outf_tiles, int extr_tiles, float* out_sdf,
                                  float* out_n, float* out_rgb, void* stream) {
  VQN_CHECK_ARG(sdf_desc && wbuf_sdf && col_desc && wbuf_col && pts && dirs && scratch && tensors && out_sdf && out_n && out_rgb, "null pointer");
  VQN_CHECK_ARG(P >= 1, "P >= 1");
  SdfDesc sd;
  ColDesc cd;
  memcpy(&sd, sdf_desc, sizeof(SdfDesc));
  memcpy(&cd, col_desc, sizeof(ColDesc));
  VQN_CHECK_SHAPE(check_sdf_desc(sd) == 0, "invalid SDF network descriptor");
  VQN_CHECK_SHAPE(cd.n_lin >= 2 && cd.n_lin <= VQN_MAX_COL_LAYERS && cd.d_out == 3 && sd.layers[sd.n_lin - 1].n_out_tiles >= 1, "colour net");
  VQN_CHECK_SHAPE(use_two_images(sd.max_tiles), "the training forward kernel is the two-image form only");
  const int nL = sd.n_lin - 1, nC = cd.n_lin - 1;
  VQN_CHECK_ARG(n_tensors == 3 + 2 * nL + nC, "tensors: [E, OUTF, EXTR, U_1..U_nL, GH_0..GH_{nL-1}, C_1..C_nC]");
  VQN_CHECK_SHAPE(e_tiles * 4 >= sd.emb_rows && e_tiles <= 2 && extr_tiles * 4 >= cd.extra_rows && extr_tiles <= 2 &&
                  outf_tiles >= sd.layers[sd.n_lin - 1].n_out_tiles && 32 * outf_tiles >= 32 * sd.layers[sd.n_lin - 1].n_out_tiles + 1, "tile counts");
  TrainOut to;
  memset(&to, 0, sizeof(to));
  for (int i = 0; i < n_tensors; ++i) VQN_CHECK_ARG(tensors[i] != nullptr, "null tensor pointer");
  to.E = tensors[0]; to.OUTF = tensors[1]; to.EXTR = tensors[2];
  for (int l = 1; l <= nL; ++l) to.U[l] = tensors[3 + (l - 1)];
  for (int l = 0; l < nL; ++l) to.GH[l] = tensors[3 + nL + l];
  for (int l = 1; l <= nC; ++l) to.C[l] = tensors[3 + 2 * nL + (l - 1)];
  to.e_tiles = e_tiles; to.outf_tiles = outf_tiles; to.extr_tiles = extr_tiles;
  const long n_tiles = (P + 31) / 32;
  const int64_t per_wg = (int64_t)(sd.n_lin - 1) * 4 * sd.max_tiles * 1024;
  const size_t lds2 = lds_bytes2(sd.max_tiles);
  VQN_HIP(hipFuncSetAttribute((const void*)neus_points2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  long grid = (long)vqn_num_cus();
  if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
  if ((int64_t)grid * 2 * per_wg > scratch_bytes) grid = (long)(scratch_bytes / (2 * per_wg));
  VQN_CHECK_ARG(grid >= 1, "scratch too small (see vqn_neus_fine_scratch_bytes)");
  hipLaunchKernelGGL((neus_points2_kernel<true, true>), dim3((unsigned)grid), dim3(512), lds2, (hipStream_t)stream, sd, cd,
                     reinterpret_cast<const f32x4*>(wbuf_sdf), reinterpret_cast<const f32x4*>(wbuf_col), (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, pts, dirs, (long)P, 1, reinterpret_cast<f32x4*>(scratch), out_sdf, out_n,
                     out_rgb, to);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int64_t vqn_neus_fine_scratch_bytes(const int32_t* sdf_desc) {
  if (!sdf_desc) return -1;
  SdfDesc sd;
  memcpy(&sd, sdf_desc, sizeof(SdfDesc));
  // (descriptors of every engine: the x3 packs count 3 rows per 16 embedding features, up to 12)
  const int emb_rows = sd.emb_rows;
  if (emb_rows >= 1 && emb_rows <= 12) sd.emb_rows = 1;
  if (check_sdf_desc(sd) != 0) return -2;
#ifdef VQN_DIAG_STASH_ROT
  return (int64_t)VQN_DIAG_STASH_ROT * vqn_num_cus() * 2 * (int64_t)(sd.n_lin - 1) * 4 * sd.max_tiles * 1024;
#else
  return (int64_t)vqn_num_cus() * 2 * (int64_t)(sd.n_lin - 1) * 4 * sd.max_tiles * 1024;
#endif
}

extern "C" int vqn_neus_fine_points(const int32_t* sdf_desc, const float* wbuf_sdf, const int32_t* col_desc,
                                    const float* wbuf_col, const float* rays_o, const float* rays_d, const float* z,
                                    const float* pts, const float* dirs, int64_t P, int S, void* scratch,
                                    int64_t scratch_bytes, float* out_sdf, float* out_grad, float* out_rgb,
                                    void* stream) {
  VQN_CHECK_ARG(sdf_desc && wbuf_sdf && col_desc && wbuf_col, "descriptors and weight packs must be non-null");
  VQN_CHECK_ARG(out_sdf && out_grad && scratch, "out_sdf, out_grad and scratch must be non-null");
  VQN_CHECK_ARG(P >= 0, "P >= 0");
  if (P == 0) return VQN_OK;
  VQN_CHECK_ARG((pts != nullptr && dirs != nullptr) || (rays_o && rays_d && z && S > 0),
                "either (pts, dirs) or (rays_o, rays_d, z, S) required");
  SdfDesc sd;
  ColDesc cd;
  memcpy(&sd, sdf_desc, sizeof(SdfDesc));
  memcpy(&cd, col_desc, sizeof(ColDesc));
  VQN_CHECK_SHAPE(check_sdf_desc(sd) == 0, "invalid SDF network descriptor");
  if (cd.n_lin != 0) {
    VQN_CHECK_ARG(out_rgb != nullptr, "out_rgb must be non-null when a colour net is given");
    VQN_CHECK_SHAPE(sd.layers[sd.n_lin - 1].n_out_tiles >= 1, "SDF network has no feature outputs (d_out == 1)");
    VQN_CHECK_SHAPE(cd.n_lin >= 2 && cd.n_lin <= VQN_MAX_COL_LAYERS && cd.d_out == 3, "colour net: 2..8 layers, d_out == 3");
    VQN_CHECK_SHAPE(cd.extra_feats >= 3 && cd.extra_feats <= 64 && cd.extra_rows >= 1 && cd.extra_rows <= 8, "colour net extras");
    for (int l = 0; l < cd.n_lin - 1; ++l)
      VQN_CHECK_SHAPE(cd.layers[l].n_out_tiles >= 1 && cd.layers[l].n_out_tiles <= sd.max_tiles, "colour layer wider than max_tiles");
  }
  const long n_tiles = (P + 31) / 32;
  const int64_t per_wg = (int64_t)(sd.n_lin - 1) * 4 * sd.max_tiles * 1024;
  if (use_two_images(sd.max_tiles)) {
    const size_t lds2 = lds_bytes2(sd.max_tiles);
    VQN_HIP(hipFuncSetAttribute((const void*)neus_points2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    long grid = (long)vqn_num_cus();
    if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
#ifdef VQN_DIAG_STASH_ROT
    VQN_CHECK_ARG((int64_t)VQN_DIAG_STASH_ROT * grid * 2 * per_wg <= scratch_bytes, "diag build: scratch must hold ROT regions per workgroup");
#else
    if ((int64_t)grid * 2 * per_wg > scratch_bytes) grid = (long)(scratch_bytes / (2 * per_wg));
#endif
    VQN_CHECK_ARG(grid >= 1, "scratch too small (see vqn_neus_fine_scratch_bytes)");
    hipLaunchKernelGGL(neus_points2_kernel<true>, dim3((unsigned)grid), dim3(512), lds2, (hipStream_t)stream, sd, cd,
                       reinterpret_cast<const f32x4*>(wbuf_sdf), reinterpret_cast<const f32x4*>(wbuf_col), rays_o, rays_d,
                       z, pts, dirs, (long)P, S, reinterpret_cast<f32x4*>(scratch), out_sdf, out_grad, out_rgb, TrainOut{});
    VQN_LAUNCH_CHECK();
    return VQN_OK;
  }
  const size_t lds = lds_bytes(sd.max_tiles);
  VQN_CHECK_SHAPE(lds <= 160 * 1024, "network too wide for LDS");
  if (lds > 64 * 1024)
    VQN_HIP(hipFuncSetAttribute((const void*)neus_points_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long grid = (long)vqn_num_cus() * 2;
  if (grid > n_tiles) grid = n_tiles;
  if ((int64_t)grid * per_wg > scratch_bytes) grid = (long)(scratch_bytes / per_wg);
  VQN_CHECK_ARG(grid >= 1, "scratch too small (see vqn_neus_fine_scratch_bytes)");
  hipLaunchKernelGGL(neus_points_kernel<true>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, sd, cd,
                     reinterpret_cast<const f32x4*>(wbuf_sdf), reinterpret_cast<const f32x4*>(wbuf_col), rays_o, rays_d,
                     z, pts, dirs, (long)P, S, reinterpret_cast<f32x4*>(scratch), out_sdf, out_grad, out_rgb);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
