// Split-precision twins of the fused NeuS kernels (neus_mlp.hip): the same per-tile program -- positional encoding, SDF
// hidden layers, sdf row, [reverse sweep for d sdf / d x, colour network] -- on the f16 hi/lo engine of mlp_prims_f16s.h
// (3 f16 MFMAs per product, f32 accumulate; weights streamed through a register ring that runs ahead across layers).
// Same reference ops as neus_mlp.hip (geo/NeuS-ours2/models/fields.py:72-107, :147-172, embedder.py:16-34).  Opt-in:
// results agree with the f32 kernels to ~1e-6 relative, not bitwise.  Descriptors are SdfDesc / ColDesc; packs come from
// SdfPackPlan(mode='f16s') / ColPackPlan(matrix_mode='f16s').
#include "mlp_prims_f16s.h"
#include "vqn_neus_desc.h"
#include <stdlib.h>

using namespace eng;

// The activation stash uses default-policy accesses here: on this engine the streaming (nt) form of the f32 kernels measured 2.5 %
// slower (the kernel leans on the L2 for its weight stream either way).
#define st_stream(p, ...) (*(p) = (__VA_ARGS__))
#define ld_stream(p) (*(p))

// In-kernel phase stamps, DIAGNOSTIC build only (make -C csrc stamps; see neus_mlp.hip): wave 0 of every workgroup adds
// shader-clock cycles per phase to g_stamps16: [0] set-up (points, posenc) [1] forward K loops [2] forward epilogues
// [3] forward barrier waits [4] sdf row + feature layer + sdf out [5] G_pre [6] reverse K loops [7] reverse epilogues
// [8] reverse barrier waits [9] embedding chain rule [10] colour set-up [11] colour layers [12] colour out [14] total [15] workgroups.
#ifdef VQN_STAMPS
__device__ unsigned long long g_stamps16[16];
#define FS_DECL unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_prev = __builtin_amdgcn_s_memtime(); const unsigned long long st_begin = st_prev;
#define FS(i) { const unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_[i] += st_now - st_prev; st_prev = st_now; }
#define FS_FLUSH if (threadIdx.x == 0) { st_[14] = __builtin_amdgcn_s_memtime() - st_begin; for (int i_ = 0; i_ < 15; ++i_) atomicAdd(&g_stamps16[i_], st_[i_]); atomicAdd(&g_stamps16[15], 1ull); }
extern "C" int vqn_debug_read_stamps16(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps16), sizeof(unsigned long long) * 16) != hipSuccess) return -3;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps16), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}
#else
#define FS_DECL
#define FS(i)
#define FS_FLUSH
#endif

namespace {

constexpr int E0 = 0;        // LDS rows [0,8): embedding / colour-net extras / d sdf / d embedding (3 steps = 6 rows used)
constexpr int E_ROWS = 8;
constexpr int MAX_CALLS = 40;
constexpr int RING = 2;

struct Smalls {
  float pts[96], dirs[96], part[512], grad[96];
  int tab[MAX_CALLS * 4];    // GEMM calls of one tile in program order: {float4 offset, 0 = SDF pack / 1 = colour pack, K rows, out tiles}
  int n_calls;
};

template <bool FINE>
__global__ __launch_bounds__(256, 2) void neus_points_f16s_kernel(
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
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_lin = sd.n_lin;
  const int emb_tiles = (sd.emb_feats + 31) >> 5;
  const long n_tiles = (P + 31) >> 5;
  f32x4* save = FINE ? scratch + (size_t)blockIdx.x * (size_t)(n_lin - 1) * 4 * MT * 64 : nullptr;
  const int feat_slot = (n_lin - 2) * 4 * MT;
  const bool has_col = FINE && cd.n_lin != 0;

  // ---------------- the tile program's GEMM calls, in order (the weight stream follows this table) ----------------
  if (tid == 0) {
    int n = 0;
    auto add = [&](int off, int which, int krows, int tiles) {
      sm->tab[4 * n] = off; sm->tab[4 * n + 1] = which; sm->tab[4 * n + 2] = krows; sm->tab[4 * n + 3] = tiles; ++n;
    };
    for (int l = 0; l < n_lin - 1; ++l)
      add(sd.layers[l].w_off, 0, l == 0 ? sd.emb_rows : 4 * sd.layers[l - 1].n_out_tiles + (l == sd.skip ? sd.emb_rows : 0),
          sd.layers[l].n_out_tiles);
    if (FINE) {
      const int hid_rows = 4 * sd.layers[n_lin - 2].n_out_tiles;
      if (sd.layers[n_lin - 1].n_out_tiles > 0) add(sd.layers[n_lin - 1].w_off, 0, hid_rows, sd.layers[n_lin - 1].n_out_tiles);
      for (int l = n_lin - 2; l >= 1; --l) {
        add(sd.layers[l].wT_off, 0, 4 * sd.layers[l].n_out_tiles, sd.layers[l - 1].n_out_tiles);
        if (l == sd.skip) add(sd.layers[l].wTE_off, 0, 4 * sd.layers[l].n_out_tiles, emb_tiles);
      }
      add(sd.layers[0].wTE_off, 0, 4 * sd.layers[0].n_out_tiles, emb_tiles);
      if (has_col) {
        int in_rows = 4 * sd.layers[n_lin - 1].n_out_tiles;
        for (int l = 0; l < cd.n_lin - 1; ++l) {
          add(cd.layers[l].w_off, 1, in_rows + (l == 0 ? cd.extra_rows : 0), cd.layers[l].n_out_tiles);
          in_rows = 4 * cd.layers[l].n_out_tiles;
        }
      }
    }
    sm->n_calls = n;
  }
  __syncthreads();
  const int n_calls = __builtin_amdgcn_readfirstlane(sm->n_calls);
  // next call (after `idx`, wrapping into the next tile) in which this wave owns a tile
  auto next_stream = [&](int idx, const f32x4*& nwp, int& nnb) {
    nwp = wsdf; nnb = 1;
    for (int k = 1; k <= n_calls; ++k) {
      const int m = (idx + k) % n_calls;
      const int tiles = __builtin_amdgcn_readfirstlane(sm->tab[4 * m + 3]);
      if (wave < tiles) {
        const int off = __builtin_amdgcn_readfirstlane(sm->tab[4 * m]), which = __builtin_amdgcn_readfirstlane(sm->tab[4 * m + 1]);
        nnb = (__builtin_amdgcn_readfirstlane(sm->tab[4 * m + 2]) + 7) >> 3;
        nwp = (which ? wcol : wsdf) + off + (size_t)wave * nnb * 512 + lane;
        return;
      }
    }
  };
  f32x4 ring[RING][8];
  {
    const f32x4* wp0; int nb0;
    next_stream(n_calls - 1, wp0, nb0);
    ring_prime<RING>(ring, wp0, nb0);
  }
  int call = 0;
  auto G = [&](const f32x4* wbase, const KSegs ks, const int tiles, auto init, auto epi) {
    const f32x4* nwp; int nnb;
    next_stream(call, nwp, nnb);
    ++call;
    gemm_tiles_f16s_ring<4, RING>(lds, ks, wbase, tiles, wave, lane, ring, nwp, nnb, init, epi);
  };

  FS_DECL
  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p0 = tile << 5;
    call = 0;
    FS(12)
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
        x = rays_o[ray * 3 + 0] + __fmul_rn(dx, t);
        y = rays_o[ray * 3 + 1] + __fmul_rn(dy, t);
        z = rays_o[ray * 3 + 2] + __fmul_rn(dz, t);
      }
      sm->pts[tid * 3 + 0] = x; sm->pts[tid * 3 + 1] = y; sm->pts[tid * 3 + 2] = z;
      sm->dirs[tid * 3 + 0] = dx; sm->dirs[tid * 3 + 1] = dy; sm->dirs[tid * 3 + 2] = dz;
    }
    __syncthreads();
    const float xs = sm->pts[p * 3 + 0] * sd.scale, ys = sm->pts[p * 3 + 1] * sd.scale, zs = sm->pts[p * 3 + 2] * sd.scale;
    // ---------------- positional encoding -> E rows (split image) ----------------
    for (int sl = wave; sl < (sd.emb_rows >> 1); sl += 4) {
      float x[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int f = step_feat(sl, h, jj);
        x[jj] = f < sd.emb_feats ? posenc_feat(f, xs, ys, zs) : 0.f;
      }
      f32x4 hi, lo;
      split8(x, hi, lo);
      lds[(E0 + 2 * sl) * 64 + lane] = hi;
      lds[(E0 + 2 * sl + 1) * 64 + lane] = lo;
    }
    __syncthreads();

    FS(0)
    // ---------------- SDF hidden layers ----------------
    int cur = X0, oth = Y0;
    for (int l = 0; l < n_lin - 1; ++l) {
      const LayerDesc L = sd.layers[l];
      const KSegs ks = (l == 0) ? KSegs{E0, sd.emb_rows, 0, 0}
                                : KSegs{cur, 4 * sd.layers[l - 1].n_out_tiles, E0, (l == sd.skip) ? sd.emb_rows : 0};
      const int dst = (l == 0) ? X0 : oth;
      const bool do_save = FINE && (l < n_lin - 2);
      const f32x4* bp = wsdf + L.b_off;
      f32x4* sv = save + (size_t)l * 4 * MT * 64;
      G(wsdf + L.w_off, ks, L.n_out_tiles,
        [&](int ot, f32x16& acc) { FS(2) init_bias_f16s(bp, ot, lane, acc); },
        [&](int ot, const f32x16& acc1, const f32x16& acc2) {
          FS(1)
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = act_fwd<ACT_SOFTPLUS100>(fmaf(acc2[i], LO_INV, acc1[i]));
          store_tile_f16s(lds, dst + ot * 4, lane, v);
          if (do_save) {                            // what the reverse sweep needs of this layer: act'(x) = 1 - exp(-100 h)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              st_stream(sv + (ot * 4 + q) * 64 + lane, (f32x4){act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q]), act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q + 1]),
                                                               act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q + 2]), act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q + 3])});
          }
        });
      FS(2)
      __syncthreads();
      FS(3)
      if (l == 0) { cur = X0; oth = Y0; } else { const int t = cur; cur = oth; oth = t; }
    }
    const int hid_rows = 4 * sd.layers[n_lin - 2].n_out_tiles;

    // ---------------- last layer: sdf row (VALU dot) [+ feature rows -> stash] ----------------
    rowdot_f16s<1>(lds, cur, hid_rows, wsdf + sd.last_w_off, sm->part, wave, lane);
    if (FINE && sd.layers[n_lin - 1].n_out_tiles > 0) {
      const LayerDesc L = sd.layers[n_lin - 1];
      const f32x4* bp = wsdf + L.b_off;
      f32x4* sv = save + (size_t)feat_slot * 64;
      G(wsdf + L.w_off, KSegs{cur, hid_rows, 0, 0}, L.n_out_tiles,
        [&](int ot, f32x16& acc) { init_bias_f16s(bp, ot, lane, acc); },
        [&](int ot, const f32x16& acc1, const f32x16& acc2) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            st_stream(sv + (ot * 4 + q) * 64 + lane, (f32x4){fmaf(acc2[4 * q], LO_INV, acc1[4 * q]), fmaf(acc2[4 * q + 1], LO_INV, acc1[4 * q + 1]),
                                                             fmaf(acc2[4 * q + 2], LO_INV, acc1[4 * q + 2]), fmaf(acc2[4 * q + 3], LO_INV, acc1[4 * q + 3])});
        });
    }
    __syncthreads();
    if (tid < 32 && p0 + tid < P) {
      const float s = ((sm->part[tid] + sm->part[32 + tid]) + (sm->part[64 + tid] + sm->part[96 + tid])) +
                      (sd.last_b_off > 0 ? wsdf[sd.last_b_off][0] : sd.last_bias);
      out_sdf[p0 + tid] = s / sd.scale;
    }
    if (!FINE) { __syncthreads(); FS(4) continue; }
    FS(4)

    // ---------------- reverse sweep: d sdf / d x ----------------
    // G_pre(last hidden) = w_sdf_row (.) act'(h), in place
    {
      const int ns = hid_rows >> 1;
      const f32x4* wimg = wsdf + sd.last_w_off;                   // [1][ns][2][8] f32
      for (int q0 = wave; q0 < ns; q0 += 16) {                    // 4 steps per pass, all fetches first
        f32x4 bh[4], bl[4], w0[4], w1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int q = min(q0 + 4 * c, ns - 1);
          bh[c] = lds[(cur + 2 * q) * 64 + lane];
          bl[c] = lds[(cur + 2 * q + 1) * 64 + lane];
          w0[c] = wimg[(q * 2 + h) * 2];
          w1[c] = wimg[(q * 2 + h) * 2 + 1];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (q0 + 4 * c < ns) {
            float x[8];
            join8(bh[c], bl[c], x);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              x[i] = w0[c][i] * act_bwd_from_out<ACT_SOFTPLUS100>(x[i]);
              x[4 + i] = w1[c][i] * act_bwd_from_out<ACT_SOFTPLUS100>(x[4 + i]);
            }
            f32x4 hi, lo;
            split8(x, hi, lo);
            lds[(cur + 2 * (q0 + 4 * c)) * 64 + lane] = hi;
            lds[(cur + 2 * (q0 + 4 * c) + 1) * 64 + lane] = lo;
          }
      }
    }
    __syncthreads();
    FS(5)
    for (int l = n_lin - 2; l >= 1; --l) {
      const LayerDesc L = sd.layers[l];
      const KSegs ks{cur, 4 * L.n_out_tiles, 0, 0};
      const f32x4* sv = save + (size_t)(l - 1) * 4 * MT * 64;
      const int dst = oth;
      f32x4 hv[4];                                   // stashed act' of this tile: requested right after the drain, lands under the K loop
      G(wsdf + L.wT_off, ks, sd.layers[l - 1].n_out_tiles,
        [&](int ot, f32x16& acc) {
          FS(7)
#pragma unroll
          for (int q = 0; q < 4; ++q) hv[q] = ld_stream(sv + (ot * 4 + q) * 64 + lane);
          init_zero(acc);
        },
        [&](int ot, const f32x16& acc1, const f32x16& acc2) {
          FS(6)
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = fmaf(acc2[i], LO_INV, acc1[i]) * hv[i >> 2][i & 3];
          store_tile_f16s(lds, dst + ot * 4, lane, v);
        });
      if (l == sd.skip)
        G(wsdf + L.wTE_off, ks, emb_tiles,
          [&](int, f32x16& acc) { init_zero(acc); },
          [&](int ot, const f32x16& acc1, const f32x16& acc2) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaf(acc2[i], LO_INV, acc1[i]);
            store_tile_f16s(lds, E0 + ot * 4, lane, v);
          });
      FS(7)
      __syncthreads();
      FS(8)
      const int t = cur; cur = oth; oth = t;
    }
    {
      const LayerDesc L = sd.layers[0];
      const bool accumulate = sd.skip >= 1;
      G(wsdf + L.wTE_off, KSegs{cur, 4 * L.n_out_tiles, 0, 0}, emb_tiles,
        [&](int ot, f32x16& acc) {
          if (accumulate) {
            float v[16];
            load_tile_f16s(lds, E0 + ot * 4, lane, v);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = v[i];
          } else init_zero(acc);
        },
        [&](int ot, const f32x16& acc1, const f32x16& acc2) {
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = fmaf(acc2[i], LO_INV, acc1[i]);
          store_tile_f16s(lds, E0 + ot * 4, lane, v);
        });
    }
    __syncthreads();
    FS(6)
    // chain through the embedding: thread (point pp, component c) sums its features in a fixed order
    if (tid < 96) {
      const int pp = tid & 31, c = tid >> 5;
      const float x0 = sm->pts[pp * 3 + 0] * sd.scale, x1 = sm->pts[pp * 3 + 1] * sd.scale, x2 = sm->pts[pp * 3 + 2] * sd.scale;
      float g = lds_feat_f16s(lds, E0, c, pp);
      int cc;
      for (int k = 0; k < sd.multires; ++k) {
        const int fs = 3 + 6 * k + c, fc = fs + 3;
        g = fmaf(lds_feat_f16s(lds, E0, fs, pp), posenc_jac(fs, x0, x1, x2, &cc), g);
        g = fmaf(lds_feat_f16s(lds, E0, fc, pp), posenc_jac(fc, x0, x1, x2, &cc), g);
      }
      sm->grad[pp * 3 + c] = g;
      if (p0 + pp < P) out_grad[(p0 + pp) * 3 + c] = g;
    }
    __syncthreads();
    FS(9)
    if (cd.n_lin == 0) continue;          // SDFNetwork.gradient(): no colour net

    // ---------------- colour network ----------------
    {
      const float px = sm->pts[p * 3 + 0], py = sm->pts[p * 3 + 1], pz = sm->pts[p * 3 + 2];
      const float dx = sm->dirs[p * 3 + 0], dy = sm->dirs[p * 3 + 1], dz = sm->dirs[p * 3 + 2];
      for (int sl = wave; sl < (cd.extra_rows >> 1); sl += 4) {
        float x[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int f = step_feat(sl, h, jj);
          float val = 0.f;
          if (f < 3) val = f == 0 ? px : (f == 1 ? py : pz);
          else if (f < 3 + cd.n_view_feats) val = posenc_feat(f - 3, dx, dy, dz);
          else if (f < cd.extra_feats) val = sm->grad[p * 3 + (f - 3 - cd.n_view_feats)];
          x[jj] = val;
        }
        f32x4 hi, lo;
        split8(x, hi, lo);
        lds[(E0 + 2 * sl) * 64 + lane] = hi;
        lds[(E0 + 2 * sl + 1) * 64 + lane] = lo;
      }
      const f32x4* sv = save + (size_t)feat_slot * 64;
      const int feat_tiles = sd.layers[n_lin - 1].n_out_tiles;
      for (int t = wave; t < feat_tiles; t += 4) {            // feature tiles back from the stash (register order) -> split rows
        f32x4 q4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) q4[q] = ld_stream(sv + (t * 4 + q) * 64 + lane);
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = q4[i >> 2][i & 3];
        store_tile_f16s(lds, X0 + 4 * t, lane, v);
      }
      __syncthreads();
      FS(10)
      cur = X0; oth = Y0;
      int in_rows = 4 * feat_tiles;
      for (int l = 0; l < cd.n_lin - 1; ++l) {
        const LayerDesc L = cd.layers[l];
        const KSegs ks{cur, in_rows, E0, l == 0 ? cd.extra_rows : 0};
        const f32x4* bp = wcol + L.b_off;
        const int dst = oth;
        G(wcol + L.w_off, ks, L.n_out_tiles,
          [&](int ot, f32x16& acc) { init_bias_f16s(bp, ot, lane, acc); },
          [&](int ot, const f32x16& acc1, const f32x16& acc2) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = act_fwd<ACT_RELU>(fmaf(acc2[i], LO_INV, acc1[i]));
            store_tile_f16s(lds, dst + ot * 4, lane, v);
          });
        __syncthreads();
        const int t = cur; cur = oth; oth = t;
        in_rows = 4 * L.n_out_tiles;
      }
      FS(11)
      rowdot_f16s<3>(lds, cur, in_rows, wcol + cd.last_w_off, sm->part, wave, lane);
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
  FS(12)
  FS_FLUSH
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-image form (the default): one 512-thread workgroup per CU holds TWO 32-point images and every wave applies its weight
// fragments to both (gemm_tiles_f16s_ring2) -- half the L2 weight stream per point.  Wave w owns out tiles w, w + 8, ...;
// the VALU phases (row dots, G_pre, set-up) split by image: waves 0-3 image 0, waves 4-7 image 1.
struct Smalls2 {
  float pts[2][96], dirs[2][96], part[2][512], grad[2][96];
  int tab[MAX_CALLS * 4];
  int n_calls;
};

template <bool FINE>
__global__ __launch_bounds__(512, 1) void neus_points_f16s2_kernel(
    const SdfDesc sd, const ColDesc cd, const f32x4* __restrict__ wsdf, const f32x4* __restrict__ wcol,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ zv,
    const float* __restrict__ pts_direct, const float* __restrict__ dirs_direct, const long P, const int S,
    f32x4* __restrict__ scratch, float* __restrict__ out_sdf, float* __restrict__ out_grad,
    float* __restrict__ out_rgb) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  const int MT = sd.max_tiles;
  const int IMG = E_ROWS + 8 * MT, IS = IMG * 64;            // rows / float4 per image
  const int X0 = E_ROWS, Y0 = E_ROWS + 4 * MT;
  Smalls2* sm = reinterpret_cast<Smalls2*>(lds + (size_t)2 * IS);
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = wave >> 2, w4 = wave & 3;                  // VALU phases: this wave's image and its rank among the image's 4 waves
  f32x4* ldsi = lds + (size_t)img * IS;
  const int n_lin = sd.n_lin;
  const int emb_tiles = (sd.emb_feats + 31) >> 5;
  const long n_tiles = (P + 31) >> 5, n_pairs = (n_tiles + 1) >> 1;
  const size_t per_img = (size_t)(n_lin - 1) * 4 * MT * 64;
  f32x4* save0 = FINE ? scratch + (size_t)blockIdx.x * 2 * per_img : nullptr;
  const int feat_slot = (n_lin - 2) * 4 * MT;
  const bool has_col = FINE && cd.n_lin != 0;

  if (tid == 0) {
    int n = 0;
    auto add = [&](int off, int which, int krows, int tiles) {
      sm->tab[4 * n] = off; sm->tab[4 * n + 1] = which; sm->tab[4 * n + 2] = krows; sm->tab[4 * n + 3] = tiles; ++n;
    };
    for (int l = 0; l < n_lin - 1; ++l)
      add(sd.layers[l].w_off, 0, l == 0 ? sd.emb_rows : 4 * sd.layers[l - 1].n_out_tiles + (l == sd.skip ? sd.emb_rows : 0),
          sd.layers[l].n_out_tiles);
    if (FINE) {
      const int hid_rows = 4 * sd.layers[n_lin - 2].n_out_tiles;
      if (sd.layers[n_lin - 1].n_out_tiles > 0) add(sd.layers[n_lin - 1].w_off, 0, hid_rows, sd.layers[n_lin - 1].n_out_tiles);
      for (int l = n_lin - 2; l >= 1; --l) {
        add(sd.layers[l].wT_off, 0, 4 * sd.layers[l].n_out_tiles, sd.layers[l - 1].n_out_tiles);
        if (l == sd.skip) add(sd.layers[l].wTE_off, 0, 4 * sd.layers[l].n_out_tiles, emb_tiles);
      }
      add(sd.layers[0].wTE_off, 0, 4 * sd.layers[0].n_out_tiles, emb_tiles);
      if (has_col) {
        int in_rows = 4 * sd.layers[n_lin - 1].n_out_tiles;
        for (int l = 0; l < cd.n_lin - 1; ++l) {
          add(cd.layers[l].w_off, 1, in_rows + (l == 0 ? cd.extra_rows : 0), cd.layers[l].n_out_tiles);
          in_rows = 4 * cd.layers[l].n_out_tiles;
        }
      }
    }
    sm->n_calls = n;
  }
  __syncthreads();
  const int n_calls = __builtin_amdgcn_readfirstlane(sm->n_calls);
  auto next_stream = [&](int idx, const f32x4*& nwp, int& nnb) {
    nwp = wsdf; nnb = 1;
    for (int k = 1; k <= n_calls; ++k) {
      const int m = (idx + k) % n_calls;
      const int tiles = __builtin_amdgcn_readfirstlane(sm->tab[4 * m + 3]);
      if (wave < tiles) {
        const int off = __builtin_amdgcn_readfirstlane(sm->tab[4 * m]), which = __builtin_amdgcn_readfirstlane(sm->tab[4 * m + 1]);
        nnb = (__builtin_amdgcn_readfirstlane(sm->tab[4 * m + 2]) + 7) >> 3;
        nwp = (which ? wcol : wsdf) + off + (size_t)wave * nnb * 512 + lane;
        return;
      }
    }
  };
  f32x4 ring[RING][8];
  {
    const f32x4* wp0; int nb0;
    next_stream(n_calls - 1, wp0, nb0);
    ring_prime<RING>(ring, wp0, nb0);
  }
  int call = 0;
  auto G = [&](const f32x4* wbase, const KSegs ks, const int tiles, auto init, auto epi) {
    const f32x4* nwp; int nnb;
    next_stream(call, nwp, nnb);
    ++call;
    // activation fragments two steps ahead in the SDF-only kernel (one step: 3 % slower there), one step in the fine kernel, which is
    // at the 256-VGPR limit: 44 -> 19 spilled registers, 98.0 -> 96.7 ms per launch
    gemm_tiles_f16s_ring2<8, RING, FINE ? 1 : 2>(lds, IS, ks, wbase, tiles, wave, lane, ring, nwp, nnb, init, epi);
  };

  for (long pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
    const long p0 = (2 * pair + img) << 5;                     // this wave's image (VALU phases); image i of the pair: (2 pair + i) * 32
    call = 0;
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
    for (int sl = w4; sl < (sd.emb_rows >> 1); sl += 4) {
      float x[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int f = step_feat(sl, h, jj);
        x[jj] = f < sd.emb_feats ? posenc_feat(f, xs, ys, zs) : 0.f;
      }
      f32x4 hi, lo;
      split8(x, hi, lo);
      ldsi[(E0 + 2 * sl) * 64 + lane] = hi;
      ldsi[(E0 + 2 * sl + 1) * 64 + lane] = lo;
    }
    __syncthreads();

    // ---------------- SDF hidden layers ----------------
    int cur = X0, oth = Y0;
    for (int l = 0; l < n_lin - 1; ++l) {
      const LayerDesc L = sd.layers[l];
      const KSegs ks = (l == 0) ? KSegs{E0, sd.emb_rows, 0, 0}
                                : KSegs{cur, 4 * sd.layers[l - 1].n_out_tiles, E0, (l == sd.skip) ? sd.emb_rows : 0};
      const int dst = (l == 0) ? X0 : oth;
      const bool do_save = FINE && (l < n_lin - 2);
      const f32x4* bp = wsdf + L.b_off;
      G(wsdf + L.w_off, ks, L.n_out_tiles,
        [&](int ot, int, f32x16& acc) { init_bias_f16s(bp, ot, lane, acc); },
        [&](int ot, int im, const f32x16& acc1, const f32x16& acc2) {
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = act_fwd<ACT_SOFTPLUS100>(fmaf(acc2[i], LO_INV, acc1[i]));
          store_tile_f16s(lds + (size_t)im * IS, dst + ot * 4, lane, v);
          if (do_save) {
            f32x4* sv = save0 + (size_t)im * per_img + (size_t)l * 4 * MT * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              st_stream(sv + (ot * 4 + q) * 64 + lane, (f32x4){act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q]), act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q + 1]),
                                                               act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q + 2]), act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q + 3])});
          }
        });
      __syncthreads();
      if (l == 0) { cur = X0; oth = Y0; } else { const int t = cur; cur = oth; oth = t; }
    }
    const int hid_rows = 4 * sd.layers[n_lin - 2].n_out_tiles;

    // ---------------- last layer: sdf row (VALU dot, per image) [+ feature rows -> stash] ----------------
    rowdot_f16s<1>(ldsi, cur, hid_rows, wsdf + sd.last_w_off, sm->part[img], w4, lane);
    if (FINE && sd.layers[n_lin - 1].n_out_tiles > 0) {
      const LayerDesc L = sd.layers[n_lin - 1];
      const f32x4* bp = wsdf + L.b_off;
      G(wsdf + L.w_off, KSegs{cur, hid_rows, 0, 0}, L.n_out_tiles,
        [&](int ot, int, f32x16& acc) { init_bias_f16s(bp, ot, lane, acc); },
        [&](int ot, int im, const f32x16& acc1, const f32x16& acc2) {
          f32x4* sv = save0 + (size_t)im * per_img + (size_t)feat_slot * 64;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            st_stream(sv + (ot * 4 + q) * 64 + lane, (f32x4){fmaf(acc2[4 * q], LO_INV, acc1[4 * q]), fmaf(acc2[4 * q + 1], LO_INV, acc1[4 * q + 1]),
                                                             fmaf(acc2[4 * q + 2], LO_INV, acc1[4 * q + 2]), fmaf(acc2[4 * q + 3], LO_INV, acc1[4 * q + 3])});
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
    }
    if (!FINE) { __syncthreads(); continue; }

    // ---------------- reverse sweep: d sdf / d x ----------------
    {
      const int ns = hid_rows >> 1;
      const f32x4* wimg = wsdf + sd.last_w_off;
      for (int q0 = w4; q0 < ns; q0 += 16) {
        f32x4 bh[4], bl[4], w0[4], w1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int q = min(q0 + 4 * c, ns - 1);
          bh[c] = ldsi[(cur + 2 * q) * 64 + lane];
          bl[c] = ldsi[(cur + 2 * q + 1) * 64 + lane];
          w0[c] = wimg[(q * 2 + h) * 2];
          w1[c] = wimg[(q * 2 + h) * 2 + 1];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (q0 + 4 * c < ns) {
            float x[8];
            join8(bh[c], bl[c], x);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              x[i] = w0[c][i] * act_bwd_from_out<ACT_SOFTPLUS100>(x[i]);
              x[4 + i] = w1[c][i] * act_bwd_from_out<ACT_SOFTPLUS100>(x[4 + i]);
            }
            f32x4 hi, lo;
            split8(x, hi, lo);
            ldsi[(cur + 2 * (q0 + 4 * c)) * 64 + lane] = hi;
            ldsi[(cur + 2 * (q0 + 4 * c) + 1) * 64 + lane] = lo;
          }
      }
    }
    __syncthreads();
    for (int l = n_lin - 2; l >= 1; --l) {
      const LayerDesc L = sd.layers[l];
      const KSegs ks{cur, 4 * L.n_out_tiles, 0, 0};
      const int dst = oth;
      f32x4 hv[2][4];
      G(wsdf + L.wT_off, ks, sd.layers[l - 1].n_out_tiles,
        [&](int ot, int im, f32x16& acc) {
          const f32x4* sv = save0 + (size_t)im * per_img + (size_t)(l - 1) * 4 * MT * 64;
#pragma unroll
          for (int q = 0; q < 4; ++q) hv[im][q] = ld_stream(sv + (ot * 4 + q) * 64 + lane);
          init_zero(acc);
        },
        [&](int ot, int im, const f32x16& acc1, const f32x16& acc2) {
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = fmaf(acc2[i], LO_INV, acc1[i]) * hv[im][i >> 2][i & 3];
          store_tile_f16s(lds + (size_t)im * IS, dst + ot * 4, lane, v);
        });
      if (l == sd.skip)
        G(wsdf + L.wTE_off, ks, emb_tiles,
          [&](int, int, f32x16& acc) { init_zero(acc); },
          [&](int ot, int im, const f32x16& acc1, const f32x16& acc2) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaf(acc2[i], LO_INV, acc1[i]);
            store_tile_f16s(lds + (size_t)im * IS, E0 + ot * 4, lane, v);
          });
      __syncthreads();
      const int t = cur; cur = oth; oth = t;
    }
    {
      const LayerDesc L = sd.layers[0];
      const bool accumulate = sd.skip >= 1;
      G(wsdf + L.wTE_off, KSegs{cur, 4 * L.n_out_tiles, 0, 0}, emb_tiles,
        [&](int ot, int im, f32x16& acc) {
          if (accumulate) {
            float v[16];
            load_tile_f16s(lds + (size_t)im * IS, E0 + ot * 4, lane, v);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = v[i];
          } else init_zero(acc);
        },
        [&](int ot, int im, const f32x16& acc1, const f32x16& acc2) {
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = fmaf(acc2[i], LO_INV, acc1[i]);
          store_tile_f16s(lds + (size_t)im * IS, E0 + ot * 4, lane, v);
        });
    }
    __syncthreads();
    if (tid < 192) {
      const int im = tid / 96, r = tid - 96 * im, pp = r & 31, c = r >> 5;
      const f32x4* li = lds + (size_t)im * IS;
      const float x0 = sm->pts[im][pp * 3 + 0] * sd.scale, x1 = sm->pts[im][pp * 3 + 1] * sd.scale, x2 = sm->pts[im][pp * 3 + 2] * sd.scale;
      float g = lds_feat_f16s(li, E0, c, pp);
      int cc;
      for (int k = 0; k < sd.multires; ++k) {
        const int fs = 3 + 6 * k + c, fc = fs + 3;
        g = fmaf(lds_feat_f16s(li, E0, fs, pp), posenc_jac(fs, x0, x1, x2, &cc), g);
        g = fmaf(lds_feat_f16s(li, E0, fc, pp), posenc_jac(fc, x0, x1, x2, &cc), g);
      }
      sm->grad[im][pp * 3 + c] = g;
      const long pt = ((2 * pair + im) << 5) + pp;
      if (pt < P) out_grad[pt * 3 + c] = g;
    }
    __syncthreads();
    if (cd.n_lin == 0) continue;

    // ---------------- colour network ----------------
    {
      const float px = sm->pts[img][p * 3 + 0], py = sm->pts[img][p * 3 + 1], pz = sm->pts[img][p * 3 + 2];
      const float dx = sm->dirs[img][p * 3 + 0], dy = sm->dirs[img][p * 3 + 1], dz = sm->dirs[img][p * 3 + 2];
      for (int sl = w4; sl < (cd.extra_rows >> 1); sl += 4) {
        float x[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int f = step_feat(sl, h, jj);
          float val = 0.f;
          if (f < 3) val = f == 0 ? px : (f == 1 ? py : pz);
          else if (f < 3 + cd.n_view_feats) val = posenc_feat(f - 3, dx, dy, dz);
          else if (f < cd.extra_feats) val = sm->grad[img][p * 3 + (f - 3 - cd.n_view_feats)];
          x[jj] = val;
        }
        f32x4 hi, lo;
        split8(x, hi, lo);
        ldsi[(E0 + 2 * sl) * 64 + lane] = hi;
        ldsi[(E0 + 2 * sl + 1) * 64 + lane] = lo;
      }
      const f32x4* sv = save0 + (size_t)img * per_img + (size_t)feat_slot * 64;
      const int feat_tiles = sd.layers[n_lin - 1].n_out_tiles;
      for (int t = w4; t < feat_tiles; t += 4) {
        f32x4 q4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) q4[q] = ld_stream(sv + (t * 4 + q) * 64 + lane);
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = q4[i >> 2][i & 3];
        store_tile_f16s(ldsi, X0 + 4 * t, lane, v);
      }
      __syncthreads();
      cur = X0; oth = Y0;
      int in_rows = 4 * feat_tiles;
      for (int l = 0; l < cd.n_lin - 1; ++l) {
        const LayerDesc L = cd.layers[l];
        const KSegs ks{cur, in_rows, E0, l == 0 ? cd.extra_rows : 0};
        const f32x4* bp = wcol + L.b_off;
        const int dst = oth;
        G(wcol + L.w_off, ks, L.n_out_tiles,
          [&](int ot, int, f32x16& acc) { init_bias_f16s(bp, ot, lane, acc); },
          [&](int ot, int im, const f32x16& acc1, const f32x16& acc2) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = act_fwd<ACT_RELU>(fmaf(acc2[i], LO_INV, acc1[i]));
            store_tile_f16s(lds + (size_t)im * IS, dst + ot * 4, lane, v);
          });
        __syncthreads();
        const int t = cur; cur = oth; oth = t;
        in_rows = 4 * L.n_out_tiles;
      }
      rowdot_f16s<3>(ldsi, cur, in_rows, wcol + cd.last_w_off, sm->part[img], w4, lane);
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
}

int check_sdf_desc(const SdfDesc& d) {
  if (d.n_lin < 2 || d.n_lin > VQN_MAX_SDF_LAYERS) return 1;
  if (d.max_tiles < 1 || d.max_tiles > 16) return 2;
  if (d.emb_feats < 3 || d.emb_feats > 64 || d.emb_rows < 2 || d.emb_rows > 8 || (d.emb_rows & 1)) return 3;
  if (d.skip >= d.n_lin - 1 || d.skip == 0) return 4;
  for (int l = 0; l < d.n_lin; ++l)
    if (d.layers[l].n_out_tiles < 0 || d.layers[l].n_out_tiles > d.max_tiles) return 5;
  if (!(d.scale > 0.f)) return 6;
  if (3 * d.n_lin + 8 > MAX_CALLS) return 7;
  return 0;
}

size_t lds_bytes(int MT) { return (size_t)(E_ROWS + 8 * MT) * 1024 + sizeof(Smalls); }
size_t lds_bytes2(int MT) { return (size_t)2 * (E_ROWS + 8 * MT) * 1024 + sizeof(Smalls2); }

// two 32-point images per workgroup (half the L2 weight stream) unless VQN_F16S_TILE32 is set or the network is too wide for it
bool use_two_images(int MT) {
  static const bool forced32 = getenv("VQN_F16S_TILE32") != nullptr;
  return !forced32 && lds_bytes2(MT) <= 160 * 1024;
}

}  // namespace

extern "C" int vqn_neus_sdf_points_f16s(const int32_t* sdf_desc, const float* wbuf_sdf, const float* rays_o,
                                        const float* rays_d, const float* z, const float* pts, int64_t P, int S,
                                        float* out_sdf, void* stream) {
  VQN_CHECK_ARG(sdf_desc && wbuf_sdf && out_sdf, "sdf_desc, wbuf_sdf, out_sdf must be non-null");
  VQN_CHECK_ARG(P >= 0, "P >= 0");
  if (P == 0) return VQN_OK;
  VQN_CHECK_ARG(pts != nullptr || (rays_o && rays_d && z && S > 0), "either pts or (rays_o, rays_d, z, S) required");
  SdfDesc sd;
  memcpy(&sd, sdf_desc, sizeof(SdfDesc));
  VQN_CHECK_SHAPE(check_sdf_desc(sd) == 0, "invalid SDF network descriptor (split-precision packs have even row counts)");
  ColDesc cd;
  memset(&cd, 0, sizeof(cd));
  const long n_tiles = (P + 31) / 32;
  if (use_two_images(sd.max_tiles)) {
    const size_t lds2 = lds_bytes2(sd.max_tiles);
    VQN_HIP(hipFuncSetAttribute((const void*)neus_points_f16s2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    long grid = (long)vqn_num_cus();
    if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
    hipLaunchKernelGGL(neus_points_f16s2_kernel<false>, dim3((unsigned)grid), dim3(512), lds2, (hipStream_t)stream, sd, cd,
                       reinterpret_cast<const f32x4*>(wbuf_sdf), (const f32x4*)nullptr, rays_o, rays_d, z, pts,
                       (const float*)nullptr, (long)P, S, (f32x4*)nullptr, out_sdf, (float*)nullptr, (float*)nullptr);
    VQN_LAUNCH_CHECK();
    return VQN_OK;
  }
  const size_t lds = lds_bytes(sd.max_tiles);
  VQN_CHECK_SHAPE(lds <= 160 * 1024, "network too wide for LDS");
  if (lds > 64 * 1024)
    VQN_HIP(hipFuncSetAttribute((const void*)neus_points_f16s_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long grid = (long)vqn_num_cus() * 2;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(neus_points_f16s_kernel<false>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, sd, cd,
                     reinterpret_cast<const f32x4*>(wbuf_sdf), (const f32x4*)nullptr, rays_o, rays_d, z, pts,
                     (const float*)nullptr, (long)P, S, (f32x4*)nullptr, out_sdf, (float*)nullptr, (float*)nullptr);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_neus_fine_points_f16s(const int32_t* sdf_desc, const float* wbuf_sdf, const int32_t* col_desc,
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
  VQN_CHECK_SHAPE(check_sdf_desc(sd) == 0, "invalid SDF network descriptor (split-precision packs have even row counts)");
  if (cd.n_lin != 0) {
    VQN_CHECK_ARG(out_rgb != nullptr, "out_rgb must be non-null when a colour net is given");
    VQN_CHECK_SHAPE(sd.layers[sd.n_lin - 1].n_out_tiles >= 1, "SDF network has no feature outputs (d_out == 1)");
    VQN_CHECK_SHAPE(cd.n_lin >= 2 && cd.n_lin <= VQN_MAX_COL_LAYERS && cd.d_out == 3, "colour net: 2..8 layers, d_out == 3");
    VQN_CHECK_SHAPE(cd.extra_feats >= 3 && cd.extra_feats <= 64 && cd.extra_rows >= 2 && cd.extra_rows <= 8 && !(cd.extra_rows & 1),
                    "colour net extras");
    for (int l = 0; l < cd.n_lin - 1; ++l)
      VQN_CHECK_SHAPE(cd.layers[l].n_out_tiles >= 1 && cd.layers[l].n_out_tiles <= sd.max_tiles, "colour layer wider than max_tiles");
  }
  const long n_tiles = (P + 31) / 32;
  const int64_t per_wg = (int64_t)(sd.n_lin - 1) * 4 * sd.max_tiles * 1024;
  if (use_two_images(sd.max_tiles)) {
    const size_t lds2 = lds_bytes2(sd.max_tiles);
    VQN_HIP(hipFuncSetAttribute((const void*)neus_points_f16s2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    long grid = (long)vqn_num_cus();
    if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
    if ((int64_t)grid * 2 * per_wg > scratch_bytes) grid = (long)(scratch_bytes / (2 * per_wg));
    VQN_CHECK_ARG(grid >= 1, "scratch too small (see vqn_neus_fine_scratch_bytes)");
    hipLaunchKernelGGL(neus_points_f16s2_kernel<true>, dim3((unsigned)grid), dim3(512), lds2, (hipStream_t)stream, sd, cd,
                       reinterpret_cast<const f32x4*>(wbuf_sdf), reinterpret_cast<const f32x4*>(wbuf_col), rays_o, rays_d,
                       z, pts, dirs, (long)P, S, reinterpret_cast<f32x4*>(scratch), out_sdf, out_grad, out_rgb);
    VQN_LAUNCH_CHECK();
    return VQN_OK;
  }
  const size_t lds = lds_bytes(sd.max_tiles);
  VQN_CHECK_SHAPE(lds <= 160 * 1024, "network too wide for LDS");
  if (lds > 64 * 1024)
    VQN_HIP(hipFuncSetAttribute((const void*)neus_points_f16s_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long grid = (long)vqn_num_cus() * 2;
  if (grid > n_tiles) grid = n_tiles;
  if ((int64_t)grid * per_wg > scratch_bytes) grid = (long)(scratch_bytes / per_wg);
  VQN_CHECK_ARG(grid >= 1, "scratch too small (see vqn_neus_fine_scratch_bytes)");
  hipLaunchKernelGGL(neus_points_f16s_kernel<true>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, sd, cd,
                     reinterpret_cast<const f32x4*>(wbuf_sdf), reinterpret_cast<const f32x4*>(wbuf_col), rays_o, rays_d,
                     z, pts, dirs, (long)P, S, reinterpret_cast<f32x4*>(scratch), out_sdf, out_grad, out_rgb);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
