// Exact-split twins of the fused NeuS kernels (neus_mlp.hip): the same per-tile program -- positional encoding, SDF hidden
// layers, sdf row, [reverse sweep for d sdf / d x, colour network] -- on the bf16x3 engine of mlp_prims_x3.h (every f32 operand
// split exactly into three bf16 pieces, six bf16 MFMAs per product down to 2^-24, f32 accumulate; weights streamed through a
// register ring that runs ahead across layers).  Same reference ops as neus_mlp.hip (geo/NeuS-ours2/models/fields.py:72-107,
// :147-172, embedder.py:16-34).  f32-level results (not bitwise those of the f32 kernels: the sums associate differently).
// Descriptors are SdfDesc / ColDesc with x3 row counts; packs come from SdfPackPlan(mode='x3') / ColPackPlan(matrix_mode='x3').
//
// Shape of the kernel.  One 512-thread workgroup per CU holds TWO 32-point images and every wave applies its weight fragments to
// both (gemm_tiles_x3_ring2).  An x3 image needs 1.5x the LDS rows of an f32 image (6 rows per 32 features), so the layers run
// IN PLACE: one activation buffer X (6 * max_tiles rows) + the embedding / extras rows E per image = 60 KB at 256 features, two
// images = 120 KB of the CU's 160.  In place means: wave w owns output tile w of a layer (layers of at most 8 tiles = 256
// features), keeps the finished tile in registers as its six piece fragments, and writes it over the input only after a
// workgroup barrier has seen every wave out of its K loop -- two barriers per layer instead of one.
#include "mlp_prims_x3.h"
#include "vqn_neus_desc.h"
#include <stdlib.h>

using namespace eng;

// the reverse sweep's stashed act' rows are requested right after a tile's drain so that they land under the K loop; fetching them in
// the epilogue instead frees 32 registers on paper, but the fine kernel then spills MORE (50 vs 39 registers: -Rpass-analysis)
#define VQN_X3_HV_EARLY 1

namespace {

constexpr int E0 = 0;         // LDS rows [0, 12): embedding / colour-net extras / d sdf / d embedding (up to 4 steps = 64 features)
constexpr int E_ROWS = 12;
constexpr int X0 = E_ROWS;    // the activation buffer
constexpr int MAX_CALLS = 40;
constexpr int RING = 2;
constexpr int NW = 8;

struct Smalls2 {
  float pts[2][96], dirs[2][96], part[2][512], grad[2][96];
  int tab[MAX_CALLS * 4];     // GEMM calls of one tile pair in program order: {float4 offset, 0 = SDF pack / 1 = colour pack, K blocks, out tiles}
  int n_calls;
};

// ---- training forward (TRAIN): as neus_points2_kernel<FINE, TRAIN> of csrc/neus_mlp.hip -- the fine kernel also leaves what the backward
// (csrc/neus_train_bwd.hip) and the weight-gradient contraction read, in the tile format [point tile][feature tile][32 features][32 points]
// f32.  The values stored are the f32 ones the epilogues hold BEFORE the split into bf16 pieces.
struct TrainOut {
  float* E; float* OUTF; float* EXTR;
  float* U[VQN_MAX_SDF_LAYERS]; float* GH[VQN_MAX_SDF_LAYERS]; float* C[VQN_MAX_COL_LAYERS];
  int e_tiles, outf_tiles, extr_tiles;
};
// an accumulator tile: register i of lane (p, h) is feature (i & 3) + 8 (i >> 2) + 4 h of the tile
__device__ __forceinline__ void tfmt_store_acc(float* __restrict__ T, const long ptile, const int n_ft, const int ot, const int lane, const float (&v)[16]) {
#ifdef VQN_DIAG_RT_NO_ST        // timing only
  asm volatile("" ::"v"(v[0]), "v"(v[3]), "v"(v[7]));
  return;
#endif
  float* base = T + ((ptile * n_ft + ot) * 32 + 4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 16; ++i) __builtin_nontemporal_store(v[i], base + ((i & 3) + 8 * (i >> 2)) * 32);
}
// a K step of an image: slot jj of lane (p, h) is feature 16 sl + 8 (jj >> 2) + 4 h + (jj & 3)
__device__ __forceinline__ void tfmt_store_step(float* __restrict__ T, const long ptile, const int n_ft, const int sl, const int lane, const float (&x)[8]) {
#ifdef VQN_DIAG_RT_NO_ST        // timing only
  asm volatile("" ::"v"(x[0]), "v"(x[3]), "v"(x[7]));
  return;
#endif
  float* base = T + ((ptile * n_ft + (sl >> 1)) * 32 + 16 * (sl & 1) + 4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) __builtin_nontemporal_store(x[jj], base + (8 * (jj >> 2) + (jj & 3)) * 32);
}

template <bool FINE, int NACC, bool TRAIN = false>
__global__ __launch_bounds__(512, 1) void neus_points_x3_kernel(
    const SdfDesc sd, const ColDesc cd, const f32x4* __restrict__ wsdf, const f32x4* __restrict__ wcol,
    const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ zv,
    const float* __restrict__ pts_direct, const float* __restrict__ dirs_direct, const long P, const int S,
    f32x4* __restrict__ scratch, float* __restrict__ out_sdf, float* __restrict__ out_grad,
    float* __restrict__ out_rgb, const TrainOut to) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  const int MT = sd.max_tiles;
  const int IMG = E_ROWS + 6 * MT, IS = IMG * 64;            // rows / float4 per image
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
  auto blocks_of = [](int krows) { return (krows / 3 + 1) >> 1; };

  // ---------------- the tile program's GEMM calls, in order (the weight stream follows this table) ----------------
  if (tid == 0) {
    int n = 0;
    auto add = [&](int off, int which, int krows, int tiles) {
      sm->tab[4 * n] = off; sm->tab[4 * n + 1] = which; sm->tab[4 * n + 2] = blocks_of(krows); sm->tab[4 * n + 3] = tiles; ++n;
    };
    for (int l = 0; l < n_lin - 1; ++l)
      add(sd.layers[l].w_off, 0, l == 0 ? sd.emb_rows : 6 * sd.layers[l - 1].n_out_tiles + (l == sd.skip ? sd.emb_rows : 0),
          sd.layers[l].n_out_tiles);
    if (FINE) {
      const int hid_rows = 6 * sd.layers[n_lin - 2].n_out_tiles;
      if (sd.layers[n_lin - 1].n_out_tiles > 0) add(sd.layers[n_lin - 1].w_off, 0, hid_rows, sd.layers[n_lin - 1].n_out_tiles);
      for (int l = n_lin - 2; l >= 1; --l) {
        if (l == sd.skip) add(sd.layers[l].wTE_off, 0, 6 * sd.layers[l].n_out_tiles, emb_tiles);
        add(sd.layers[l].wT_off, 0, 6 * sd.layers[l].n_out_tiles, sd.layers[l - 1].n_out_tiles);
      }
      add(sd.layers[0].wTE_off, 0, 6 * sd.layers[0].n_out_tiles, emb_tiles);
      if (has_col) {
        int in_rows = 6 * sd.layers[n_lin - 1].n_out_tiles;
        for (int l = 0; l < cd.n_lin - 1; ++l) {
          add(cd.layers[l].w_off, 1, in_rows + (l == 0 ? cd.extra_rows : 0), cd.layers[l].n_out_tiles);
          in_rows = 6 * cd.layers[l].n_out_tiles;
        }
      }
    }
    sm->n_calls = n;
  }
  __syncthreads();
  const int n_calls = __builtin_amdgcn_readfirstlane(sm->n_calls);
  // next call (after `idx`, wrapping into the next tile pair) in which this wave owns a tile
  auto next_stream = [&](int idx, const f32x4*& nwp, int& nnb) {
    nwp = wsdf + lane; nnb = 1;
    for (int k = 1; k <= n_calls; ++k) {
      const int m = (idx + k) % n_calls;
      const int tiles = __builtin_amdgcn_readfirstlane(sm->tab[4 * m + 3]);
      if (wave < tiles) {
        const int off = __builtin_amdgcn_readfirstlane(sm->tab[4 * m]), which = __builtin_amdgcn_readfirstlane(sm->tab[4 * m + 1]);
        nnb = __builtin_amdgcn_readfirstlane(sm->tab[4 * m + 2]);
        nwp = (which ? wcol : wsdf) + off + (size_t)wave * nnb * 384 + lane;
        return;
      }
    }
  };
  f32x4 ring[RING][6];
  {
    const f32x4* wp0; int nb0;
    next_stream(n_calls - 1, wp0, nb0);
    ring_prime_x3<RING>(ring, wp0, nb0);
  }
  int call = 0;
  auto G = [&](const f32x4* wbase, const KSegs ks, const int tiles, auto init, auto epi) {
    const f32x4* nwp; int nnb;
    next_stream(call, nwp, nnb);
    ++call;
    gemm_tiles_x3_ring2<NW, RING, NACC>(lds, IS, ks, wbase, tiles, wave, lane, ring, nwp, nnb, init, epi);
  };
  // a layer whose output tile `wave` goes over its own input (the X buffer): finished tiles wait in `o` for the barrier
  // A layer IN PLACE (its output tile `wave` goes over its own input, the X buffer): the accumulators stay where the K loop left them,
  // a barrier sees every wave out of its K loop -- the input rows are dead --, then the epilogues write their tiles straight over the
  // input (store_tile_x3) and a second barrier publishes them; waves without a tile in this GEMM reach the same two barriers from
  // the branch below.  (Rounds 3's form parked the finished, already split tiles in 48 registers across the barrier: they ended in
  // scratch, and every layer re-read 18 x 16 B per lane behind the epilogue's global stores -- round 4, found with in-kernel stamps
  // on csrc/refl_train_x3.hip, where the same change took a third off the backward.)
  auto GC = [&](const f32x4* wbase, const KSegs ks, const int tiles, auto init, auto epi) {
    const f32x4* nwp; int nnb;
    next_stream(call, nwp, nnb);
    ++call;
    gemm_tiles_x3_ring2<NW, RING, NACC>(lds, IS, ks, wbase, tiles, wave, lane, ring, nwp, nnb, init,
                                        [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          if (decltype(im_c)::value == 0) __syncthreads(); epi(ot, im_c, acc); });
    if (wave >= tiles) __syncthreads();
    __syncthreads();
  };

  for (long pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
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
    // ---------------- positional encoding -> E rows (x3 image) ----------------
    const long ptile_w = 2 * pair + img;                    // (TRAIN) this wave's image; stores are skipped for a phantom tile
    for (int sl = w4; sl < (TRAIN ? 2 * to.e_tiles : sd.emb_rows / 3); sl += 4) {
      float x[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int f = step_feat(sl, h, jj);
        x[jj] = (sl < sd.emb_rows / 3 && f < sd.emb_feats) ? posenc_feat(f, xs, ys, zs) : 0.f;
      }
      if (TRAIN && ptile_w < n_tiles) tfmt_store_step(to.E, ptile_w, to.e_tiles, sl, lane, x);
      if (sl < sd.emb_rows / 3) {
        f32x4 q0, q1, q2;
        split3x8(x, q0, q1, q2);
        ldsi[(E0 + 3 * sl) * 64 + lane] = q0;
        ldsi[(E0 + 3 * sl + 1) * 64 + lane] = q1;
        ldsi[(E0 + 3 * sl + 2) * 64 + lane] = q2;
      }
    }
    __syncthreads();

    // ---------------- SDF hidden layers (in place) ----------------
    for (int l = 0; l < n_lin - 1; ++l) {
      const LayerDesc L = sd.layers[l];
      const KSegs ks = (l == 0) ? KSegs{E0, sd.emb_rows, 0, 0}
                                : KSegs{X0, 6 * sd.layers[l - 1].n_out_tiles, E0, (l == sd.skip) ? sd.emb_rows : 0};
      const bool do_save = FINE && (l < n_lin - 2);
      const f32x4* bp = wsdf + L.b_off;
      float* const t_u = TRAIN ? to.U[l + 1] : nullptr;
      GC(wsdf + L.w_off, ks, L.n_out_tiles,
        [&](int ot, auto, f32x16& acc) __attribute__((always_inline)) { init_bias_f16s(bp, ot, lane, acc); },
        [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = act_fwd<ACT_SOFTPLUS100>(acc[i]);
          if (TRAIN && 2 * pair + im < n_tiles) tfmt_store_acc(t_u, 2 * pair + im, L.n_out_tiles, ot, lane, v);
          store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
          if (do_save) {                            // what the reverse sweep needs of this layer: act'(x) = 1 - exp(-100 h)
            f32x4* sv = save0 + (size_t)im * per_img + (size_t)l * 4 * MT * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              st_stream(sv + (ot * 4 + q) * 64 + lane, (f32x4){act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q]), act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q + 1]),
                                                               act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q + 2]), act_bwd_from_out<ACT_SOFTPLUS100>(v[4 * q + 3])});
          }
        });
    }
    const int hid_rows = 6 * sd.layers[n_lin - 2].n_out_tiles;

    // ---------------- last layer: sdf row (VALU dot, per image) [+ feature rows -> stash] ----------------
    rowdot_x3<1>(ldsi, X0, hid_rows, wsdf + sd.last_w_off, sm->part[img], w4, lane);
    if (FINE && sd.layers[n_lin - 1].n_out_tiles > 0) {
      const LayerDesc L = sd.layers[n_lin - 1];
      const f32x4* bp = wsdf + L.b_off;
      G(wsdf + L.w_off, KSegs{X0, hid_rows, 0, 0}, L.n_out_tiles,
        [&](int ot, auto, f32x16& acc) __attribute__((always_inline)) { init_bias_f16s(bp, ot, lane, acc); },
        [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          f32x4* sv = save0 + (size_t)im * per_img + (size_t)feat_slot * 64;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            st_stream(sv + (ot * 4 + q) * 64 + lane, (f32x4){acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
          if (TRAIN && 2 * pair + im < n_tiles) {            // OUTF = [sdf ; features]: feature f of this GEMM is row f + 1
            float* base = to.OUTF + (2 * pair + im) * (long)to.outf_tiles * 1024;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int f = 32 * ot + (i & 3) + 8 * (i >> 2) + 4 * h + 1;
              if (f < 32 * to.outf_tiles) __builtin_nontemporal_store(acc[i], base + f * 32 + p);
            }
          }
        });
    }
    __syncthreads();
    if (tid < 64) {
      const int im = tid >> 5, t = tid & 31;
      const long pt = ((2 * pair + im) << 5) + t;
      const float* pr = sm->part[im];
      const float s = ((pr[t] + pr[32 + t]) + (pr[64 + t] + pr[96 + t])) + (sd.last_b_off > 0 ? wsdf[sd.last_b_off][0] : sd.last_bias);
      if (pt < P) out_sdf[pt] = s / sd.scale;
      if (TRAIN && 2 * pair + im < n_tiles) {                 // row 0 of OUTF (the raw sdf output) and the zero tail beyond row F - 1
        float* base = to.OUTF + (2 * pair + im) * (long)to.outf_tiles * 1024;
        base[t] = s;
        for (int f = 32 * sd.layers[n_lin - 1].n_out_tiles + 1; f < 32 * to.outf_tiles; ++f) base[f * 32 + t] = 0.f;
      }
    }
    if (!FINE) { __syncthreads(); continue; }

    // ---------------- reverse sweep: d sdf / d x ----------------
    // G_pre(last hidden) = w_sdf_row (.) act'(h), in place (a lane rewrites exactly what it read)
    {
      const int ns = hid_rows / 3;
      const f32x4* wimg = wsdf + sd.last_w_off;                   // [1][ns][2][8] f32
      for (int q0 = w4; q0 < ns; q0 += 16) {                      // 4 steps per pass, all fetches first
        f32x4 b0[4], b1[4], b2[4], w0[4], w1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int q = min(q0 + 4 * c, ns - 1);
          b0[c] = ldsi[(X0 + 3 * q) * 64 + lane];
          b1[c] = ldsi[(X0 + 3 * q + 1) * 64 + lane];
          b2[c] = ldsi[(X0 + 3 * q + 2) * 64 + lane];
          w0[c] = wimg[(q * 2 + h) * 2];
          w1[c] = wimg[(q * 2 + h) * 2 + 1];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (q0 + 4 * c < ns) {
            float x[8];
            join3x8(b0[c], b1[c], b2[c], x);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              x[i] = w0[c][i] * act_bwd_from_out<ACT_SOFTPLUS100>(x[i]);
              x[4 + i] = w1[c][i] * act_bwd_from_out<ACT_SOFTPLUS100>(x[4 + i]);
            }
            f32x4 q0_, q1_, q2_;
            split3x8(x, q0_, q1_, q2_);
            const int q = q0 + 4 * c;
            if (TRAIN && ptile_w < n_tiles) tfmt_store_step(to.GH[n_lin - 2], ptile_w, sd.layers[n_lin - 2].n_out_tiles, q, lane, x);
            ldsi[(X0 + 3 * q) * 64 + lane] = q0_;
            ldsi[(X0 + 3 * q + 1) * 64 + lane] = q1_;
            ldsi[(X0 + 3 * q + 2) * 64 + lane] = q2_;
          }
      }
    }
    __syncthreads();
    for (int l = n_lin - 2; l >= 1; --l) {
      const LayerDesc L = sd.layers[l];
      const KSegs ks{X0, 6 * L.n_out_tiles, 0, 0};
      if (l == sd.skip)                              // the embedding's share first: it reads X and writes E (nobody reads E now)
        G(wsdf + L.wTE_off, ks, emb_tiles,
          [&](int, auto, f32x16& acc) __attribute__((always_inline)) { init_zero(acc); },
          [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = acc[i];
            store_tile_x3(lds + (size_t)im * IS, E0 + ot * 6, lane, v);
          });
      const int tiles = sd.layers[l - 1].n_out_tiles;
#ifdef VQN_X3_HV_EARLY       // stashed act' of the tile requested right after the drain (lands under the K loop) at the price of 32 registers
      f32x4 hv[2][4];
      GC(wsdf + L.wT_off, ks, tiles,
        [&](int ot, auto im_c, f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          const f32x4* sv = save0 + (size_t)im * per_img + (size_t)(l - 1) * 4 * MT * 64;
#pragma unroll
          for (int q = 0; q < 4; ++q) hv[im][q] = ld_stream(sv + (ot * 4 + q) * 64 + lane);
          init_zero(acc);
        },
        [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = acc[i] * hv[im][i >> 2][i & 3];
          if (TRAIN && 2 * pair + im < n_tiles) tfmt_store_acc(to.GH[l - 1], 2 * pair + im, tiles, ot, lane, v);
          store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
        });
#else                        // ... or in the epilogue itself: the kernel sits at the 256-register limit, and the other wave of the SIMD covers the wait
      GC(wsdf + L.wT_off, ks, tiles,
        [&](int, auto, f32x16& acc) __attribute__((always_inline)) { init_zero(acc); },
        [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          const f32x4* sv = save0 + (size_t)im * per_img + (size_t)(l - 1) * 4 * MT * 64;
          f32x4 hv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) hv[q] = ld_stream(sv + (ot * 4 + q) * 64 + lane);
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = acc[i] * hv[i >> 2][i & 3];
          if (TRAIN && 2 * pair + im < n_tiles) tfmt_store_acc(to.GH[l - 1], 2 * pair + im, tiles, ot, lane, v);
          store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
        });
#endif
    }
    {
      const LayerDesc L = sd.layers[0];
      const bool accumulate = sd.skip >= 1;
      G(wsdf + L.wTE_off, KSegs{X0, 6 * L.n_out_tiles, 0, 0}, emb_tiles,
        [&](int ot, auto im_c, f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          if (accumulate) {
            float v[16];
            load_tile_x3(lds + (size_t)im * IS, E0 + ot * 6, lane, v);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = v[i];
          } else init_zero(acc);
        },
        [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = acc[i];
          store_tile_x3(lds + (size_t)im * IS, E0 + ot * 6, lane, v);
        });
    }
    __syncthreads();
    // chain through the embedding: thread (image, point pp, component c) sums its features in a fixed order
    if (tid < 192) {
      const int im = tid / 96, r = tid - 96 * im, pp = r & 31, c = r >> 5;
      const f32x4* li = lds + (size_t)im * IS;
      const float x0 = sm->pts[im][pp * 3 + 0] * sd.scale, x1 = sm->pts[im][pp * 3 + 1] * sd.scale, x2 = sm->pts[im][pp * 3 + 2] * sd.scale;
      float g = lds_feat_x3(li, E0, c, pp);
      int cc;
      for (int k = 0; k < sd.multires; ++k) {
        const int fs = 3 + 6 * k + c, fc = fs + 3;
        g = fmaf(lds_feat_x3(li, E0, fs, pp), posenc_jac(fs, x0, x1, x2, &cc), g);
        g = fmaf(lds_feat_x3(li, E0, fc, pp), posenc_jac(fc, x0, x1, x2, &cc), g);
      }
      sm->grad[im][pp * 3 + c] = g;
      const long pt = ((2 * pair + im) << 5) + pp;
      if (pt < P) out_grad[pt * 3 + c] = g;
    }
    __syncthreads();
    if (cd.n_lin == 0) continue;          // SDFNetwork.gradient(): no colour net

    // ---------------- colour network (in place) ----------------
    {
      const float px = sm->pts[img][p * 3 + 0], py = sm->pts[img][p * 3 + 1], pz = sm->pts[img][p * 3 + 2];
      const float dx = sm->dirs[img][p * 3 + 0], dy = sm->dirs[img][p * 3 + 1], dz = sm->dirs[img][p * 3 + 2];
      for (int sl = w4; sl < (TRAIN ? 2 * to.extr_tiles : cd.extra_rows / 3); sl += 4) {
        float x[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int f = step_feat(sl, h, jj);
          float val = 0.f;
          if (sl >= cd.extra_rows / 3) val = 0.f;
          else if (f < 3) val = f == 0 ? px : (f == 1 ? py : pz);
          else if (f < 3 + cd.n_view_feats) val = posenc_feat(f - 3, dx, dy, dz);
          else if (f < cd.extra_feats) val = sm->grad[img][p * 3 + (f - 3 - cd.n_view_feats)];
          x[jj] = val;
        }
        if (TRAIN && ptile_w < n_tiles) tfmt_store_step(to.EXTR, ptile_w, to.extr_tiles, sl, lane, x);
        if (sl < cd.extra_rows / 3) {
          f32x4 q0, q1, q2;
          split3x8(x, q0, q1, q2);
          ldsi[(E0 + 3 * sl) * 64 + lane] = q0;
          ldsi[(E0 + 3 * sl + 1) * 64 + lane] = q1;
          ldsi[(E0 + 3 * sl + 2) * 64 + lane] = q2;
        }
      }
      const f32x4* sv = save0 + (size_t)img * per_img + (size_t)feat_slot * 64;
      const int feat_tiles = sd.layers[n_lin - 1].n_out_tiles;
      for (int t = w4; t < feat_tiles; t += 4) {            // feature tiles back from the stash (register order) -> piece rows
        f32x4 q4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) q4[q] = ld_stream(sv + (t * 4 + q) * 64 + lane);
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = q4[i >> 2][i & 3];
        store_tile_x3(ldsi, X0 + 6 * t, lane, v);
      }
      __syncthreads();
      int in_rows = 6 * feat_tiles;
      for (int l = 0; l < cd.n_lin - 1; ++l) {
        const LayerDesc L = cd.layers[l];
        const KSegs ks{X0, in_rows, E0, l == 0 ? cd.extra_rows : 0};
        const f32x4* bp = wcol + L.b_off;
        float* const t_c = TRAIN ? to.C[l + 1] : nullptr;
        GC(wcol + L.w_off, ks, L.n_out_tiles,
          [&](int ot, auto, f32x16& acc) __attribute__((always_inline)) { init_bias_f16s(bp, ot, lane, acc); },
          [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = act_fwd<ACT_RELU>(acc[i]);
            if (TRAIN && 2 * pair + im < n_tiles) tfmt_store_acc(t_c, 2 * pair + im, L.n_out_tiles, ot, lane, v);
            store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
          });
        in_rows = 6 * L.n_out_tiles;
      }
      rowdot_x3<3>(ldsi, X0, in_rows, wcol + cd.last_w_off, sm->part[img], w4, lane);
      __syncthreads();
      if (tid < 192) {
        const int im = tid / 96, r = tid - 96 * im, pp = r & 31, oc = r >> 5;
        const float* pr = sm->part[im];
        float v = ((pr[(0 * 32 + pp) * 3 + oc] + pr[(1 * 32 + pp) * 3 + oc]) + (pr[(2 * 32 + pp) * 3 + oc] + pr[(3 * 32 + pp) * 3 + oc])) +
                  (cd.last_b_off > 0 ? wcol[cd.last_b_off][oc] : cd.last_bias[oc]);
        if (cd.squeeze_out) v = 1.f / (1.f + expf(-v));
        const long pt = ((2 * pair + im) << 5) + pp;
        if (pt < P) out_rgb[pt * 3 + oc] = v;
      }
    }
    __syncthreads();
  }
}

int check_sdf_desc_x3(const SdfDesc& d) {
  if (d.n_lin < 2 || d.n_lin > VQN_MAX_SDF_LAYERS) return 1;
  if (d.max_tiles < 1 || d.max_tiles > NW) return 2;                       // in-place layers: one output tile per wave
  if (d.emb_feats < 3 || d.emb_feats > 64 || d.emb_rows != x3_rows(d.emb_feats)) return 3;
  if (d.skip >= d.n_lin - 1 || d.skip == 0) return 4;
  for (int l = 0; l < d.n_lin; ++l)
    if (d.layers[l].n_out_tiles < 0 || d.layers[l].n_out_tiles > d.max_tiles) return 5;
  if (!(d.scale > 0.f)) return 6;
  if (3 * d.n_lin + 8 > MAX_CALLS) return 7;
  return 0;
}

size_t lds_bytes_x3(int MT) { return (size_t)2 * (E_ROWS + 6 * MT) * 1024 + sizeof(Smalls2); }

// accumulators per image: 1 (all six terms of a step into one) or 2 (a0 w0 apart from the five smaller terms); VQN_X3_NACC overrides
int x3_nacc() {
  // default 2: measured against the float64 evaluation of the networks (2048 points, full nets) the one-accumulator form sits at
  // 1.4x the f32 kernels' max sdf error, the two-accumulator form at 1.14x (gradients and colour below the f32 kernels') for 4 % time
  static const int n = [] { const char* e = getenv("VQN_X3_NACC"); return (e && e[0] == '1') ? 1 : 2; }();
  return n;
}

}  // namespace

extern "C" int vqn_neus_sdf_points_x3(const int32_t* sdf_desc, const float* wbuf_sdf, const float* rays_o,
                                      const float* rays_d, const float* z, const float* pts, int64_t P, int S,
                                      float* out_sdf, void* stream) {
  VQN_CHECK_ARG(sdf_desc && wbuf_sdf && out_sdf, "sdf_desc, wbuf_sdf, out_sdf must be non-null");
  VQN_CHECK_ARG(P >= 0, "P >= 0");
  if (P == 0) return VQN_OK;
  VQN_CHECK_ARG(pts != nullptr || (rays_o && rays_d && z && S > 0), "either pts or (rays_o, rays_d, z, S) required");
  SdfDesc sd;
  memcpy(&sd, sdf_desc, sizeof(SdfDesc));
  VQN_CHECK_SHAPE(check_sdf_desc_x3(sd) == 0, "invalid SDF network descriptor for the x3 engine (x3 packs: 3 rows per 16 features; layers of at most 256 outputs)");
  ColDesc cd;
  memset(&cd, 0, sizeof(cd));
  const long n_tiles = (P + 31) / 32;
  const size_t lds2 = lds_bytes_x3(sd.max_tiles);
  VQN_CHECK_SHAPE(lds2 <= 160 * 1024, "network too wide for LDS");
  auto kern = x3_nacc() == 2 ? neus_points_x3_kernel<false, 2> : neus_points_x3_kernel<false, 1>;
  VQN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  long grid = (long)vqn_num_cus();
  if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds2, (hipStream_t)stream, sd, cd,
                     reinterpret_cast<const f32x4*>(wbuf_sdf), (const f32x4*)nullptr, rays_o, rays_d, z, pts,
                     (const float*)nullptr, (long)P, S, (f32x4*)nullptr, out_sdf, (float*)nullptr, (float*)nullptr, TrainOut{});
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_neus_fine_points_x3(const int32_t* sdf_desc, const float* wbuf_sdf, const int32_t* col_desc,
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
  VQN_CHECK_SHAPE(check_sdf_desc_x3(sd) == 0, "invalid SDF network descriptor for the x3 engine (x3 packs: 3 rows per 16 features; layers of at most 256 outputs)");
  if (cd.n_lin != 0) {
    VQN_CHECK_ARG(out_rgb != nullptr, "out_rgb must be non-null when a colour net is given");
    VQN_CHECK_SHAPE(sd.layers[sd.n_lin - 1].n_out_tiles >= 1, "SDF network has no feature outputs (d_out == 1)");
    VQN_CHECK_SHAPE(cd.n_lin >= 2 && cd.n_lin <= VQN_MAX_COL_LAYERS && cd.d_out == 3, "colour net: 2..8 layers, d_out == 3");
    VQN_CHECK_SHAPE(cd.extra_feats >= 3 && cd.extra_feats <= 64 && cd.extra_rows == x3_rows(cd.extra_feats), "colour net extras");
    for (int l = 0; l < cd.n_lin - 1; ++l)
      VQN_CHECK_SHAPE(cd.layers[l].n_out_tiles >= 1 && cd.layers[l].n_out_tiles <= sd.max_tiles, "colour layer wider than max_tiles");
  }
  const long n_tiles = (P + 31) / 32;
  const int64_t per_wg = (int64_t)(sd.n_lin - 1) * 4 * sd.max_tiles * 1024;
  const size_t lds2 = lds_bytes_x3(sd.max_tiles);
  VQN_CHECK_SHAPE(lds2 <= 160 * 1024, "network too wide for LDS");
  auto kern = x3_nacc() == 2 ? neus_points_x3_kernel<true, 2> : neus_points_x3_kernel<true, 1>;
  VQN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  long grid = (long)vqn_num_cus();
  if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
  if ((int64_t)grid * 2 * per_wg > scratch_bytes) grid = (long)(scratch_bytes / (2 * per_wg));
  VQN_CHECK_ARG(grid >= 1, "scratch too small (see vqn_neus_fine_scratch_bytes)");
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds2, (hipStream_t)stream, sd, cd,
                     reinterpret_cast<const f32x4*>(wbuf_sdf), reinterpret_cast<const f32x4*>(wbuf_col), rays_o, rays_d,
                     z, pts, dirs, (long)P, S, reinterpret_cast<f32x4*>(scratch), out_sdf, out_grad, out_rgb, TrainOut{});
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

// Training forward on the exact-split engine: vqn_neus_train_fwd with x3 packs and descriptors (vqn_neus_pack_create(..., f16s = 2)).
// Same outputs and saved tensors (f32 values, taken before the split into pieces); layers of at most 256 outputs.
extern "C" int vqn_neus_train_fwd_x3(const int32_t* sdf_desc, const float* wbuf_sdf, const int32_t* col_desc, const float* wbuf_col,
                                     const float* pts, const float* dirs, int64_t P, void* scratch, int64_t scratch_bytes,
                                     float* const* tensors, int n_tensors, int e_tiles, int outf_tiles, int extr_tiles, float* out_sdf,
                                     float* out_n, float* out_rgb, void* stream) {
  VQN_CHECK_ARG(sdf_desc && wbuf_sdf && col_desc && wbuf_col && pts && dirs && scratch && tensors && out_sdf && out_n && out_rgb, "null pointer");
  VQN_CHECK_ARG(P >= 1, "P >= 1");
  SdfDesc sd;
  ColDesc cd;
  memcpy(&sd, sdf_desc, sizeof(SdfDesc));
  memcpy(&cd, col_desc, sizeof(ColDesc));
  VQN_CHECK_SHAPE(check_sdf_desc_x3(sd) == 0, "invalid SDF network descriptor for the x3 engine");
  VQN_CHECK_SHAPE(cd.n_lin >= 2 && cd.n_lin <= VQN_MAX_COL_LAYERS && cd.d_out == 3 && sd.layers[sd.n_lin - 1].n_out_tiles >= 1, "colour net");
  VQN_CHECK_SHAPE(cd.extra_feats >= 3 && cd.extra_feats <= 64 && cd.extra_rows == x3_rows(cd.extra_feats), "colour net extras");
  for (int l = 0; l < cd.n_lin - 1; ++l)
    VQN_CHECK_SHAPE(cd.layers[l].n_out_tiles >= 1 && cd.layers[l].n_out_tiles <= sd.max_tiles, "colour layer wider than max_tiles");
  const int nL = sd.n_lin - 1, nC = cd.n_lin - 1;
  VQN_CHECK_ARG(n_tensors == 3 + 2 * nL + nC, "tensors: [E, OUTF, EXTR, U_1..U_nL, GH_0..GH_{nL-1}, C_1..C_nC]");
  VQN_CHECK_SHAPE(2 * e_tiles * 3 >= sd.emb_rows && e_tiles <= 2 && 2 * extr_tiles * 3 >= cd.extra_rows && extr_tiles <= 2 &&
                  32 * outf_tiles >= 32 * sd.layers[sd.n_lin - 1].n_out_tiles + 1, "tile counts");
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
  const size_t lds2 = lds_bytes_x3(sd.max_tiles);
  VQN_CHECK_SHAPE(lds2 <= 160 * 1024, "network too wide for LDS");
  auto kern = x3_nacc() == 2 ? neus_points_x3_kernel<true, 2, true> : neus_points_x3_kernel<true, 1, true>;
  VQN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  long grid = (long)vqn_num_cus();
  if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
  if ((int64_t)grid * 2 * per_wg > scratch_bytes) grid = (long)(scratch_bytes / (2 * per_wg));
  VQN_CHECK_ARG(grid >= 1, "scratch too small (see vqn_neus_fine_scratch_bytes)");
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds2, (hipStream_t)stream, sd, cd, reinterpret_cast<const f32x4*>(wbuf_sdf),
                     reinterpret_cast<const f32x4*>(wbuf_col), (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, pts, dirs,
                     (long)P, 1, reinterpret_cast<f32x4*>(scratch), out_sdf, out_n, out_rgb, to);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
