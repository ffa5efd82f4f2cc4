// vqn_neus_train_bwd on the exact-split engine (csrc/mlp_prims_x3.h): the same backward pass as csrc/neus_train_bwd.hip -- colour
// backward, tangent pass, reverse sweep with the second-order sources, one launch -- with every layer GEMM as six bf16 MFMAs per
// product over bf16 piece triples (products exact to 2^-24, f32 accumulation), the layers in place in one activation buffer (one
// output tile per wave, finished tiles parked in registers across the barrier), weights streamed through the engine's ring along
// a table of the pass's GEMM calls.  What it stands for in the reference: loss.backward() through renderer.py:216-227 under
// exp_runner.py:153-168.  Adjoints in and saved tensors in / out are the f32 ones of the f32 kernel (tile format, same contract).
//
// Also here: vqn_pack_x3_gather, the weight pack of this kernel in one launch (gather from the flat parameter vector + the exact
// three-way split of every gathered value into bf16 pieces, geo/packing.py: split_pack_x3).
#include "mlp_prims_x3.h"
#include "vqnerf_hip.h"
#include <stdlib.h>

using namespace eng;

namespace {

constexpr int E0 = 0;
constexpr int E_ROWS = 12;
constexpr int X0 = E_ROWS;
constexpr int RING = 2;
constexpr int NW = 8;
constexpr int MAX_CALLS = 40;
constexpr int TB_MAX_L = 12;

struct TrainBwdDesc {        // as csrc/neus_train_bwd.hip; here emb_rows counts x3 rows, offT / offB / offCB / offBtop / offCBfeat index the
  int nL, nC, skip, emb_rows, emb_feats, e_tiles, max_tiles, feat_tiles;     // piece pack and offWrow / offCBnrm the f32 image buffer (float4 units)
  int outf_tiles, squeeze, offBtop, offWrow, offCBfeat, offCBnrm;
  float scale, inv_scale;
  int ts[TB_MAX_L], tc[TB_MAX_L], offT[TB_MAX_L], offB[TB_MAX_L], offCB[TB_MAX_L];
};
constexpr int TB_DESC_INTS = 16 + 5 * TB_MAX_L;
static_assert(sizeof(TrainBwdDesc) == TB_DESC_INTS * 4, "descriptor layout");

struct TrainBwdPtrs {
  const float* X; const float* G_RGB; const float* RGB; const float* G_N; const float* G_SDF;
  const float* U[TB_MAX_L]; const float* GH[TB_MAX_L]; const float* C[TB_MAX_L];
  float* DC[TB_MAX_L]; float* UD[TB_MAX_L]; float* AB[TB_MAX_L];
  float* GOUTF; float* ED;
};

struct SmallsB {
  float pts[2][96], dout[2][96], gn[2][96], v[2][96], gs[2][32], part[2][512];
  int tab[MAX_CALLS * 3];      // GEMM calls of one tile pair in program order: {float4 offset, K blocks, out tiles}
  int n_calls;
};

// accumulator tile <-> tile format: register i of lane (p, h) is feature (i & 3) + 8 (i >> 2) + 4 h of the feature tile
__device__ __forceinline__ void tf_store_acc(float* __restrict__ T, const long ptile, const int n_ft, const int ot, const int lane, const float (&v)[16]) {
#ifdef VQN_DIAG_RT_NO_ST        // timing only
  asm volatile("" ::"v"(v[0]), "v"(v[3]), "v"(v[7]));
  return;
#endif
  float* base = T + ((ptile * n_ft + ot) * 32 + 4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 16; ++i) __builtin_nontemporal_store(v[i], base + ((i & 3) + 8 * (i >> 2)) * 32);
}
__device__ __forceinline__ void tf_load_acc(const float* __restrict__ T, const long ptile, const int n_ft, const int ot, const int lane, float (&v)[16]) {
  const float* base = T + ((ptile * n_ft + ot) * 32 + 4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = base[((i & 3) + 8 * (i >> 2)) * 32];
}
// a K step of an image: slot jj of lane (p, h) is feature 16 sl + 8 (jj >> 2) + 4 h + (jj & 3)
__device__ __forceinline__ void tf_store_step(float* __restrict__ T, const long ptile, const int n_ft, const int sl, const int lane, const float (&x)[8]) {
#ifdef VQN_DIAG_RT_NO_ST        // timing only
  asm volatile("" ::"v"(x[0]), "v"(x[3]), "v"(x[7]));
  return;
#endif
  float* base = T + ((ptile * n_ft + (sl >> 1)) * 32 + 16 * (sl & 1) + 4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) __builtin_nontemporal_store(x[jj], base + (8 * (jj >> 2) + (jj & 3)) * 32);
}

template <int NACC>
__global__ __launch_bounds__(512, 1) void neus_train_bwd_x3_kernel(const TrainBwdDesc bd, const f32x4* __restrict__ wx, const f32x4* __restrict__ wf,
                                                                   const TrainBwdPtrs tp, const long P, f32x4* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  const int MT = bd.max_tiles;
  const int IMG = E_ROWS + 6 * MT, IS = IMG * 64;
  SmallsB* sm = reinterpret_cast<SmallsB*>(lds + (size_t)2 * IS);
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = wave >> 2, w4 = wave & 3;
  f32x4* ldsi = lds + (size_t)img * IS;
  const int nL = bd.nL, nC = bd.nC;
  const long n_tiles = (P + 31) >> 5, n_pairs = (n_tiles + 1) >> 1;
  const size_t per_img = (size_t)(nL + 1) * 4 * MT * 64;           // stash: S_0..S_{nL-1}, g_feat (accumulator-order quads)
  f32x4* save0 = scratch + (size_t)blockIdx.x * 2 * per_img;
  const int feat_slot = nL * 4 * MT;
  auto blocks_of = [](int krows) { return (krows / 3 + 1) >> 1; };

  // ---------------- the pass's GEMM calls, in order (the weight stream follows this table) ----------------
  if (tid == 0) {
    int n = 0;
    auto add = [&](int off, int krows, int tiles) { sm->tab[3 * n] = off; sm->tab[3 * n + 1] = blocks_of(krows); sm->tab[3 * n + 2] = tiles; ++n; };
    for (int l = nC; l >= 1; --l) add(bd.offCB[l], l == nC ? 3 : 6 * bd.tc[l], bd.tc[l - 1]);
    add(bd.offCBfeat, 6 * bd.tc[0], bd.feat_tiles);
    for (int l = 0; l < nL; ++l) add(bd.offT[l], l == 0 ? bd.emb_rows : 6 * bd.ts[l - 1] + (l == bd.skip ? bd.emb_rows : 0), bd.ts[l]);
    add(bd.offBtop, 6 * bd.feat_tiles, bd.ts[nL - 1]);
    for (int l = nL - 1; l >= 1; --l) add(bd.offB[l], 6 * bd.ts[l], bd.ts[l - 1]);
    sm->n_calls = n;
  }
  __syncthreads();
  const int n_calls = __builtin_amdgcn_readfirstlane(sm->n_calls);
  auto next_stream = [&](int idx, const f32x4*& nwp, int& nnb) {
    nwp = wx + lane; nnb = 1;
    for (int k = 1; k <= n_calls; ++k) {
      const int m = (idx + k) % n_calls;
      const int tiles = __builtin_amdgcn_readfirstlane(sm->tab[3 * m + 2]);
      if (wave < tiles) {
        const int off = __builtin_amdgcn_readfirstlane(sm->tab[3 * m]);
        nnb = __builtin_amdgcn_readfirstlane(sm->tab[3 * m + 1]);
        nwp = wx + off + (size_t)wave * nnb * 384 + lane;
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
  // epilogue operands of this wave's tile, both images, requested right before the K loop.  (The f32 kernel's refinement -- the NEXT GEMM's
  // operands requested in place from inside the current epilogue -- measured slower here: 6.80 -> 7.19 ms, 39 -> 99 spilled VGPRs: the
  // operands would live across commit() and the ring refills, and this engine has no registers to spare.  Requesting them half way through
  // the K loop instead (a hook inside gemm_tiles_x3_ring2): 6.7 -> 7.0 ms, 39 -> 91 spilled VGPRs -- the same verdict.)
  float au[2][16], ab[2][16];

  for (long pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
    call = 0;
    const long ptile_w = 2 * pair + img;
    const bool live_w = ptile_w < n_tiles;
    // ---------------- points and incoming adjoints of both tiles ----------------
    if (tid < 64) {
      const int im = tid >> 5, t = tid & 31;
      const long pt = ((2 * pair + im) << 5) + t;
      const bool valid = pt < P;
      const long q = valid ? pt : P - 1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        sm->pts[im][t * 3 + c] = tp.X[q * 3 + c];
        float d = 0.f;
        if (valid) {
          d = tp.G_RGB[pt * 3 + c];
          if (tp.RGB != nullptr) { const float y = tp.RGB[pt * 3 + c]; d = d * y * (1.0f - y); }
        }
        sm->dout[im][t * 3 + c] = d;
        sm->gn[im][t * 3 + c] = (valid && tp.G_N != nullptr) ? tp.G_N[pt * 3 + c] : 0.f;
      }
      sm->gs[im][t] = (valid && tp.G_SDF != nullptr) ? tp.G_SDF[pt] * bd.inv_scale : 0.f;
    }
    __syncthreads();
    if (w4 < 2) {                                            // delta_nC: one K step in the E rows; DC_nC is one feature tile = two steps
      float x[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int f = step_feat(w4, h, jj);
        x[jj] = f < 3 ? sm->dout[img][p * 3 + f] : 0.f;
      }
      if (live_w) tf_store_step(tp.DC[nC], ptile_w, 1, w4, lane, x);
      if (w4 == 0) {
        f32x4 q0, q1, q2;
        split3x8(x, q0, q1, q2);
        ldsi[(E0 + 0) * 64 + lane] = q0; ldsi[(E0 + 1) * 64 + lane] = q1; ldsi[(E0 + 2) * 64 + lane] = q2;
      }
    }
    __syncthreads();

    // ---------------- colour network backward (in place) ----------------
    for (int l = nC; l >= 1; --l) {
      const KSegs ks = (l == nC) ? KSegs{E0, 3, 0, 0} : KSegs{X0, 6 * bd.tc[l], 0, 0};
      const int n_ot = bd.tc[l - 1];
      const float* const t_c = tp.C[l];
      float* const t_dc = tp.DC[l - 1];
      GC(wx + bd.offCB[l], ks, n_ot,
        [&](int ot, auto im_c, f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          if (2 * pair + im < n_tiles) tf_load_acc(t_c, 2 * pair + im, n_ot, ot, lane, au[im]);
          else {
#pragma unroll
            for (int i = 0; i < 16; ++i) au[im][i] = 0.f;
          }
          init_zero(acc);
        },
        [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = acc[i] * act_bwd_from_out<ACT_RELU>(au[im][i]);
          if (2 * pair + im < n_tiles) tf_store_acc(t_dc, 2 * pair + im, n_ot, ot, lane, v);
          store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
        });
    }
    {
      const int in_rows = 6 * bd.tc[0];
      rowdot_x3<3>(ldsi, X0, in_rows, wf + bd.offCBnrm, sm->part[img], w4, lane);
      G(wx + bd.offCBfeat, KSegs{X0, in_rows, 0, 0}, bd.feat_tiles,
        [&](int, auto, f32x16& acc) __attribute__((always_inline)) { init_zero(acc); },
        [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          f32x4* sv = save0 + (size_t)im * per_img + (size_t)feat_slot * 64;
#pragma unroll
          for (int q = 0; q < 4; ++q) st_stream(sv + (ot * 4 + q) * 64 + lane, (f32x4){acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
          if (2 * pair + im < n_tiles) {                     // GOUTF = [0 ; d features]: feature f is row f + 1
            float* base = tp.GOUTF + (2 * pair + im) * (long)bd.outf_tiles * 1024;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int f = 32 * ot + (i & 3) + 8 * (i >> 2) + 4 * h + 1;
              if (f < 32 * bd.outf_tiles) __builtin_nontemporal_store(acc[i], base + f * 32 + p);
            }
          }
        });
      __syncthreads();
      if (tid < 192) {
        const int im = tid / 96, r = tid - 96 * im, pp = r & 31, c = r >> 5;
        const float* pr = sm->part[im];
        const float g = (pr[(0 * 32 + pp) * 3 + c] + pr[(1 * 32 + pp) * 3 + c]) + (pr[(2 * 32 + pp) * 3 + c] + pr[(3 * 32 + pp) * 3 + c]);
        sm->v[im][pp * 3 + c] = sm->gn[im][pp * 3 + c] + g;
      } else if (tid < 256) {
        const int im = (tid - 192) >> 5, t = tid & 31;
        if (2 * pair + im < n_tiles) {
          float* base = tp.GOUTF + (2 * pair + im) * (long)bd.outf_tiles * 1024;
          base[t] = sm->gs[im][t];                                        // row 0: d loss / d sdf / scale -- the contraction GOUTF x u_L then
                                                                          // yields row 0 of the last layer's gradient (and its bias) by itself
          for (int f = 32 * bd.feat_tiles + 1; f < 32 * bd.outf_tiles; ++f) base[f * 32 + t] = 0.f;
        }
      }
      __syncthreads();
    }

    // ---------------- tangent pass ----------------
    {
      const float x0 = sm->pts[img][p * 3 + 0] * bd.scale, x1 = sm->pts[img][p * 3 + 1] * bd.scale, x2 = sm->pts[img][p * 3 + 2] * bd.scale;
      const float v0 = sm->v[img][p * 3 + 0], v1 = sm->v[img][p * 3 + 1], v2 = sm->v[img][p * 3 + 2];
      for (int sl = w4; sl < 2 * bd.e_tiles; sl += 4) {
        float x[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int f = step_feat(sl, h, jj);
          float val = 0.f;
          if (sl < bd.emb_rows / 3 && f < bd.emb_feats) {
            int c;
            const float jac = posenc_jac(f, x0, x1, x2, &c);
            val = jac * (c == 0 ? v0 : (c == 1 ? v1 : v2));
          }
          x[jj] = val;
        }
        if (live_w) tf_store_step(tp.ED, ptile_w, bd.e_tiles, sl, lane, x);
        if (sl < bd.emb_rows / 3) {
          f32x4 q0, q1, q2;
          split3x8(x, q0, q1, q2);
          ldsi[(E0 + 3 * sl) * 64 + lane] = q0; ldsi[(E0 + 3 * sl + 1) * 64 + lane] = q1; ldsi[(E0 + 3 * sl + 2) * 64 + lane] = q2;
        }
      }
    }
    __syncthreads();
    for (int l = 0; l < nL; ++l) {
      const KSegs ks = (l == 0) ? KSegs{E0, bd.emb_rows, 0, 0} : KSegs{X0, 6 * bd.ts[l - 1], E0, (l == bd.skip) ? bd.emb_rows : 0};
      const int n_ot = bd.ts[l];
      const float* const t_u = tp.U[l + 1];
      const float* const t_gh = tp.GH[l];
      float* const t_ud = tp.UD[l + 1];
      GC(wx + bd.offT[l], ks, n_ot,
        [&](int ot, auto im_c, f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          if (2 * pair + im < n_tiles) {
            tf_load_acc(t_u, 2 * pair + im, n_ot, ot, lane, au[im]);
            tf_load_acc(t_gh, 2 * pair + im, n_ot, ot, lane, ab[im]);
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) { au[im][i] = 0.f; ab[im][i] = 0.f; }
          }
          init_zero(acc);
        },
        [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          f32x4* sv = save0 + (size_t)im * per_img + (size_t)l * 4 * MT * 64;
          float v[16], s[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float a = acc[i];
            const float e = fast_exp(-100.f * au[im][i]);
            v[i] = a * (1.f - e);
            s[i] = ab[im][i] * a * (100.f * e);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) st_stream(sv + (ot * 4 + q) * 64 + lane, (f32x4){s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]});
          if (2 * pair + im < n_tiles) tf_store_acc(t_ud, 2 * pair + im, n_ot, ot, lane, v);
          store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
        });
    }

    // ---------------- reverse sweep ----------------
    {                                                       // g_feat back from the stash (accumulator order) -> piece rows of X
      const f32x4* sv = save0 + (size_t)img * per_img + (size_t)feat_slot * 64;
      for (int t = w4; t < bd.feat_tiles; t += 4) {
        f32x4 q4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) q4[q] = ld_stream(sv + (t * 4 + q) * 64 + lane);
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = q4[i >> 2][i & 3];
        store_tile_x3(ldsi, X0 + 6 * t, lane, v);
      }
    }
    __syncthreads();
    for (int l = nL; l >= 1; --l) {
      const bool top = l == nL;
      const KSegs ks = top ? KSegs{X0, 6 * bd.feat_tiles, 0, 0} : KSegs{X0, 6 * bd.ts[l], 0, 0};
      const int n_ot = bd.ts[l - 1];
      const float* const t_u = tp.U[l];
      float* const t_ab = tp.AB[l - 1];
      const f32x4* const wrow = wf + bd.offWrow;
      GC(wx + (top ? bd.offBtop : bd.offB[l]), ks, n_ot,
        [&](int ot, auto im_c, f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          const f32x4* sv = save0 + (size_t)im * per_img + (size_t)(l - 1) * 4 * MT * 64;
          if (2 * pair + im < n_tiles) tf_load_acc(t_u, 2 * pair + im, n_ot, ot, lane, au[im]);
          else {
#pragma unroll
            for (int i = 0; i < 16; ++i) au[im][i] = 0.f;
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 t = ld_stream(sv + (ot * 4 + q) * 64 + lane);
            ab[im][4 * q] = t[0]; ab[im][4 * q + 1] = t[1]; ab[im][4 * q + 2] = t[2]; ab[im][4 * q + 3] = t[3];
          }
          if (top) {                                        // the sdf row of the last layer as the rank-1 start value
            init_bias_f16s(wrow, ot, lane, acc);
            const float gs = sm->gs[im][p];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= gs;
          } else init_zero(acc);
        },
        [&](int ot, auto im_c, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int im = decltype(im_c)::value; (void)im;
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = acc[i] * act_bwd_from_out<ACT_SOFTPLUS100>(au[im][i]) + ab[im][i];
          if (2 * pair + im < n_tiles) tf_store_acc(t_ab, 2 * pair + im, n_ot, ot, lane, v);
          store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
        });
    }
  }
}

size_t lds_bytes_b(int MT) { return (size_t)2 * (E_ROWS + 6 * MT) * 1024 + sizeof(SmallsB); }

int load_desc(const int32_t* desc, TrainBwdDesc& bd) {
  memcpy(&bd, desc, sizeof(TrainBwdDesc));
  if (bd.nL < 2 || bd.nL >= TB_MAX_L || bd.nC < 1 || bd.nC >= TB_MAX_L) return 1;
  if (bd.max_tiles < 1 || bd.max_tiles > NW || lds_bytes_b(bd.max_tiles) > 160 * 1024) return 2;
  if (bd.emb_feats < 3 || bd.emb_feats > 64 || bd.emb_rows != x3_rows(bd.emb_feats) || bd.emb_rows > E_ROWS || 2 * bd.e_tiles * 3 < bd.emb_rows || bd.e_tiles > 2) return 3;
  if (bd.skip == 0 || bd.skip >= bd.nL) return 4;
  if (bd.feat_tiles < 1 || bd.feat_tiles > bd.max_tiles || 32 * bd.outf_tiles < 32 * bd.feat_tiles + 1) return 5;
  for (int l = 0; l < bd.nL; ++l) if (bd.ts[l] < 1 || bd.ts[l] > bd.max_tiles) return 6;
  for (int l = 0; l < bd.nC; ++l) if (bd.tc[l] < 1 || bd.tc[l] > bd.max_tiles) return 7;
  if (!(bd.scale > 0.f)) return 8;
  if (2 * bd.nL + bd.nC + 2 > MAX_CALLS) return 9;
  return 0;
}

int bwd_nacc() {
  static const int n = [] { const char* e = getenv("VQN_X3_BWD_NACC"); return (e && e[0] == '2') ? 2 : 1; }();
  return n;
}

// out[t][q][lane][k] (bf16) = piece q of flat[gidx[t][lane][k]]: t over (tile, K step) of every matrix, 64 lanes x 8 slots per step
// (blocks beyond the piece pack's: the plain f32 gather out_f[i] = flat[fidx[i]] of the thin images -- biases in accumulator order, row-dot
//  images -- that every caller of this pack needs from the same flat vector: one launch for both)
__global__ __launch_bounds__(256) void pack_x3_gather_kernel(const float* __restrict__ flat, const int32_t* __restrict__ gidx, const long n_steps,
                                                             unsigned short* __restrict__ out, const int32_t* __restrict__ fidx, const long n_f,
                                                             float* __restrict__ out_f) {
  const long pack_blocks = (n_steps + 3) / 4;
  if ((long)blockIdx.x >= pack_blocks) {
    const long i = ((long)blockIdx.x - pack_blocks) * 256 + threadIdx.x;
    if (i < n_f) out_f[i] = flat[fidx[i]];
    return;
  }
  const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= n_steps) return;
  const int32_t* ip = gidx + (t * 64 + lane) * 8;
  unsigned short pc[3][8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float x = flat[ip[k]];
    const float p0 = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float r1 = x - p0;
    const float p1 = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float p2 = __uint_as_float(__float_as_uint(r1 - p1) & 0xffff0000u);
    pc[0][k] = (unsigned short)(__float_as_uint(p0) >> 16);
    pc[1][k] = (unsigned short)(__float_as_uint(p1) >> 16);
    pc[2][k] = (unsigned short)(__float_as_uint(p2) >> 16);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    unsigned short* op = out + ((t * 3 + q) * 64 + lane) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) op[k] = pc[q][k];
  }
}

}  // namespace

extern "C" int vqn_pack_x3_gather2(const float* flat, const int32_t* gidx, int64_t n_steps, void* out, const int32_t* fidx, int64_t n_f32,
                                   float* out_f32, void* stream) {
  VQN_CHECK_ARG(flat && gidx && out, "null pointer");
  VQN_CHECK_ARG(n_steps >= 1, "n_steps >= 1");
  VQN_CHECK_ARG(n_f32 >= 0 && (n_f32 == 0 || (fidx && out_f32)), "fidx / out_f32 with n_f32 > 0");
  const long blocks = (n_steps + 3) / 4 + (n_f32 + 255) / 256;
  hipLaunchKernelGGL(pack_x3_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, flat, gidx, (long)n_steps,
                     reinterpret_cast<unsigned short*>(out), fidx, (long)n_f32, out_f32);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_pack_x3_gather(const float* flat, const int32_t* gidx, int64_t n_steps, void* out, void* stream) {
  return vqn_pack_x3_gather2(flat, gidx, n_steps, out, nullptr, 0, nullptr, stream);
}

extern "C" int64_t vqn_neus_train_bwd_x3_scratch_bytes(const int32_t* desc) {
  if (!desc) return -1;
  TrainBwdDesc bd;
  if (load_desc(desc, bd) != 0) return -1;
  return (int64_t)vqn_num_cus() * 2 * (bd.nL + 1) * 4 * bd.max_tiles * 1024;
}

extern "C" int vqn_neus_train_bwd_x3(const int32_t* desc, const void* wbuf_pieces, const float* wbuf_f32, const float* pts, const float* g_rgb,
                                     const float* rgb, const float* g_n, const float* g_sdf, int64_t P, void* scratch, int64_t scratch_bytes,
                                     const float* const* saved, int n_saved, float* const* outs, int n_outs, void* stream) {
  VQN_CHECK_ARG(desc && wbuf_pieces && wbuf_f32 && pts && g_rgb && scratch && saved && outs, "null pointer");
  VQN_CHECK_ARG(P >= 1, "P >= 1");
  TrainBwdDesc bd;
  VQN_CHECK_SHAPE(load_desc(desc, bd) == 0, "invalid backward descriptor for the x3 engine (layers of at most 256 outputs)");
  const int nL = bd.nL, nC = bd.nC;
  VQN_CHECK_ARG(n_saved == 2 * nL + nC, "saved: [U_1..U_nL, GH_0..GH_{nL-1}, C_1..C_nC]");
  VQN_CHECK_ARG(n_outs == (nC + 1) + 2 + 2 * nL, "outs: [DC_0..DC_nC, GOUTF, ED, UD_1..UD_nL, AB_0..AB_{nL-1}]");
  VQN_CHECK_ARG((rgb != nullptr) == (bd.squeeze != 0), "rgb: the forward's colours when the colour net ends in a sigmoid, else NULL");
  for (int i = 0; i < n_saved; ++i) VQN_CHECK_ARG(saved[i] != nullptr, "null saved tensor");
  for (int i = 0; i < n_outs; ++i) VQN_CHECK_ARG(outs[i] != nullptr, "null output tensor");
  TrainBwdPtrs tp;
  memset(&tp, 0, sizeof(tp));
  tp.X = pts; tp.G_RGB = g_rgb; tp.RGB = rgb; tp.G_N = g_n; tp.G_SDF = g_sdf;
  for (int l = 1; l <= nL; ++l) tp.U[l] = saved[l - 1];
  for (int l = 0; l < nL; ++l) tp.GH[l] = saved[nL + l];
  for (int l = 1; l <= nC; ++l) tp.C[l] = saved[2 * nL + l - 1];
  for (int l = 0; l <= nC; ++l) tp.DC[l] = outs[l];
  tp.GOUTF = outs[nC + 1];
  tp.ED = outs[nC + 2];
  for (int l = 1; l <= nL; ++l) tp.UD[l] = outs[nC + 3 + l - 1];
  for (int l = 0; l < nL; ++l) tp.AB[l] = outs[nC + 3 + nL + l];
  const long n_tiles = (P + 31) / 32;
  const int64_t per_wg = (int64_t)2 * (nL + 1) * 4 * bd.max_tiles * 1024;
  const size_t lds = lds_bytes_b(bd.max_tiles);
  auto kern = bwd_nacc() == 2 ? neus_train_bwd_x3_kernel<2> : neus_train_bwd_x3_kernel<1>;
  VQN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long grid = (long)vqn_num_cus();
  if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
  if ((int64_t)grid * per_wg > scratch_bytes) grid = (long)(scratch_bytes / per_wg);
  VQN_CHECK_ARG(grid >= 1, "scratch too small (see vqn_neus_train_bwd_x3_scratch_bytes)");
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, bd, reinterpret_cast<const f32x4*>(wbuf_pieces),
                     reinterpret_cast<const f32x4*>(wbuf_f32), tp, (long)P, reinterpret_cast<f32x4*>(scratch));
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
