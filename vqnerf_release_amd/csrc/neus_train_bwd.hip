// Backward of the fused NeuS core for gfx950: colour-network backward + SDF-network backward (with the second-order terms of
// the normals, which the colour net and the eikonal loss read) as ONE launch of the two-image engine of csrc/neus_mlp.hip --
// 512-thread workgroups, two 32-point activation images in LDS, the eight waves a tile each per layer and image pair, weight
// fragments streamed from L2.  It is what loss.backward() does to
//     geo/NeuS-ours2/models/renderer.py:216-227  (sdf_network(pts), sdf_network.gradient(pts), color_network(...))
// under exp_runner.py:153-168, down to the per-point adjoints; the sums over points (the weight gradients) are the contraction
// kernels' (csrc/wgrad*.hip), which read what this kernel and the forward (vqn_neus_train_fwd) leave in the tile format
// [point tile][feature tile][32 features][32 points] f32.
//
// Per point, with u_l the SDF hidden activations, g^_l the adjoints of the forward's d sdf / d x sweep (GH_l), c_l the colour
// activations (all saved by the forward) and the incoming adjoints (d rgb, d n, d sdf):
//   colour backward   delta_nC = d rgb (* rgb (1 - rgb));  delta_{l-1} = (Wc_l^T delta_l) relu'(c_l)                    -> DC_l
//                     g_feat = Wc_0[:, feat]^T delta_0 (-> GOUTF rows 1..),  v = d n + Wc_0[:, normals]^T delta_0
//   tangent pass      e' = J_posenc(x) v (-> ED);  a'_l = W_l [u'_{l-1} (, e')];  u'_l = a'_l s'(u_l)                    -> UD_l
//                     S_l = g^_l a'_l s''/s'(u_l)            (second-order source, per-workgroup stash)
//   reverse sweep     ab_{L-1} = (W_L[1:]^T g_feat + W_L[0]^T d sdf / scale) s'(u_L) + S_{L-1};
//                     ab_{l-1} = (W_l[:, u]^T ab_l) s'(u_l) + S_{l-1}                                                    -> AB_l
// (the same statement as the interpreted programs prog_cbwd / prog_sbwd of geo/train_programs.py, which remain the path of
// networks this kernel does not take: fewer than five or more than eight feature tiles.  csrc/neus_train_bwd_x3.hip is the same pass on the
// exact-split engine, the trainers' default.)
#include "mlp_prims.h"
#include "vqnerf_hip.h"
#include <stdlib.h>

using namespace eng;

namespace {

constexpr int E0 = 0;
constexpr int E_ROWS = 8;
constexpr int TB_MAX_L = 12;

struct TrainBwdDesc {        // int32 words, filled by the host (geo/train_programs.py: NeusTrainEngine._bwd_static)
  int nL, nC, skip, emb_rows, emb_feats, e_tiles, max_tiles, feat_tiles;
  int outf_tiles, squeeze, offBtop, offWrow, offCBfeat, offCBnrm;
  float scale, inv_scale;
  int ts[TB_MAX_L];          // feature tiles of SDF hidden layer l's output
  int tc[TB_MAX_L];          // feature tiles of colour hidden layer l's output
  int offT[TB_MAX_L];        // W_l (K = [u_{l-1} (, e)]), float4 units into the pack
  int offB[TB_MAX_L];        // W_l[:, :out_{l-1}]^T, l = 1..nL-1
  int offCB[TB_MAX_L];       // Wc_l^T, l = 1..nC (l = nC: K = one row)
};
constexpr int TB_DESC_INTS = 16 + 5 * TB_MAX_L;
static_assert(sizeof(TrainBwdDesc) == TB_DESC_INTS * 4, "descriptor layout");

struct TrainBwdPtrs {
  const float* X; const float* G_RGB; const float* RGB; const float* G_N; const float* G_SDF;
  const float* U[TB_MAX_L];    // U[l], l = 1..nL: output of SDF layer l - 1
  const float* GH[TB_MAX_L];   // GH[l], l = 0..nL-1
  const float* C[TB_MAX_L];    // C[l], l = 1..nC: output of colour layer l - 1
  float* DC[TB_MAX_L];         // DC[l], l = 0..nC
  float* UD[TB_MAX_L];         // UD[l], l = 1..nL
  float* AB[TB_MAX_L];         // AB[l], l = 0..nL-1
  float* GOUTF; float* ED;
};

struct SmallsB {
  float pts[2][96], dout[2][96], gn[2][96], v[2][96], gs[2][32], part[2][512];
};

// a row quad of an activation image <-> the tile format: lane (p, h), component j = feature 8 rq + 2 j + h of the feature tile
__device__ __forceinline__ void tf_store(float* __restrict__ T, const long ptile, const int n_ft, const int ft, const int rq, const int lane,
                                         const f32x4 v) {
  float* base = T + ((ptile * n_ft + ft) * 32 + 8 * rq) * 32 + lane;
#pragma unroll
  for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v[j], base + 64 * j);
}
__device__ __forceinline__ f32x4 tf_load(const float* __restrict__ T, const long ptile, const int n_ft, const int ft, const int rq,
                                         const int lane) {
  const float* base = T + ((ptile * n_ft + ft) * 32 + 8 * rq) * 32 + lane;
  return (f32x4){base[0], base[64], base[128], base[192]};
}

__global__ __launch_bounds__(512, 1) void neus_train_bwd2_kernel(const TrainBwdDesc bd, const f32x4* __restrict__ wb, const TrainBwdPtrs tp,
                                                                 const long P, f32x4* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  const int MT = bd.max_tiles;
  const int IMG = E_ROWS + 8 * MT, IS = IMG * 64;
  const int X0 = E_ROWS, Y0 = E_ROWS + 4 * MT;
  SmallsB* sm = reinterpret_cast<SmallsB*>(lds + (size_t)2 * IS);
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = wave >> 2, w4 = wave & 3;
  f32x4* ldsi = lds + (size_t)img * IS;
  const int nL = bd.nL, nC = bd.nC;
  const long n_tiles = (P + 31) >> 5, n_pairs = (n_tiles + 1) >> 1;
  const size_t per_img = (size_t)(nL + 1) * 4 * MT * 64;           // stash: S_0..S_{nL-1}, g_feat
  f32x4* save0 = scratch + (size_t)blockIdx.x * 2 * per_img;
  const int feat_slot = nL * 4 * MT;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 nopre[4];

  // Epilogue operands (saved activations from HBM, second-order sources from the stash) of a wave's tile, both images.  Every GEMM
  // here has at most eight output tiles: a wave owns tile `wave` or none.  A row quad of the NEXT GEMM's operands is requested into
  // the same registers right after the current epilogue has used them, so the HBM round trip runs under the rest of the epilogue,
  // the barrier and the weight prologue (vmcnt retires in order: a request made at a GEMM's own start would hold up its first
  // weight wait for the whole HBM latency) at no cost in registers.
  struct Aux { const float* a; const float* b; const f32x4* s; int tiles; };
  f32x4 ca[2][4], cb[2][4];

  for (long pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
    const long ptile_w = 2 * pair + img;
    const bool live_w = ptile_w < n_tiles;
    auto fetch_q = [&](const Aux& x, int im, int rq) {
      if (wave >= x.tiles) return;
      const bool live = 2 * pair + im < n_tiles;
      ca[im][rq] = (x.a != nullptr && live) ? tf_load(x.a, 2 * pair + im, x.tiles, wave, rq, lane) : zero4;
      if (x.b != nullptr) cb[im][rq] = live ? tf_load(x.b, 2 * pair + im, x.tiles, wave, rq, lane) : zero4;
      else if (x.s != nullptr) cb[im][rq] = ld_stream(x.s + (size_t)im * per_img + (wave * 4 + rq) * 64 + lane);
    };
    auto fetch = [&](const Aux& x) {
#pragma unroll
      for (int im = 0; im < 2; ++im) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) fetch_q(x, im, rq);
      }
    };
    fetch(Aux{tp.C[nC], nullptr, nullptr, bd.tc[nC - 1]});
    // ---------------- points and incoming adjoints of both tiles (zero for points past P: everything below is linear in them) -----------
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
          if (tp.RGB != nullptr) { const float o = tp.RGB[pt * 3 + c]; d = d * o * (1.0f - o); }      // sigmoid output (fields.py:171)
        }
        sm->dout[im][t * 3 + c] = d;
        sm->gn[im][t * 3 + c] = (valid && tp.G_N != nullptr) ? tp.G_N[pt * 3 + c] : 0.f;
      }
      sm->gs[im][t] = (valid && tp.G_SDF != nullptr) ? tp.G_SDF[pt] * bd.inv_scale : 0.f;
    }
    __syncthreads();
    {                                                      // delta_nC: one feature tile, rows 0..3 of the E region (K of the next GEMM: row 0)
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = row_feat(w4, h, j);
        v[j] = f < 3 ? sm->dout[img][p * 3 + f] : 0.f;
      }
      ldsi[(E0 + w4) * 64 + lane] = v;
      if (live_w) tf_store(tp.DC[nC], ptile_w, 1, 0, w4, lane, v);
    }
    __syncthreads();

    // ---------------- colour network backward ----------------
    int cur = X0, oth = Y0;
    for (int l = nC; l >= 1; --l) {
      const KSegs ks = (l == nC) ? KSegs{E0, 1, 0, 0} : KSegs{cur, 4 * bd.tc[l], 0, 0};
      const int dst = (l == nC) ? X0 : oth;
      const int n_ot = bd.tc[l - 1];
      float* const t_dc = tp.DC[l - 1];
      const Aux nxt = (l > 1) ? Aux{tp.C[l - 1], nullptr, nullptr, bd.tc[l - 2]} : Aux{tp.U[1], tp.GH[0], nullptr, bd.ts[0]};
      gemm_tiles2<8>(lds, IS, ks, wb + bd.offCB[l], n_ot, wave, lane, nopre, false, nullptr,
                     [&](int, int, f32x16& acc) { init_zero(acc); },
                     [&](int ot, int im, int rq, const f32x16& acc) {
                       f32x4* li = lds + (size_t)im * IS;
                       f32x4 v = {acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]};
#pragma unroll
                       for (int j = 0; j < 4; ++j) v[j] *= act_bwd_from_out<ACT_RELU>(ca[im][rq][j]);
                       fetch_q(nxt, im, rq);
                       li[(dst + ot * 4 + rq) * 64 + lane] = v;
                       if (2 * pair + im < n_tiles) tf_store(t_dc, 2 * pair + im, n_ot, ot, rq, lane, v);
                     });
      if (wave >= n_ot) fetch(nxt);                        // (a wave without a tile in this GEMM may own one in the next)
      __syncthreads();
      if (l == nC) { cur = X0; oth = Y0; } else { const int t = cur; cur = oth; oth = t; }
    }
    // adjoints of the colour net's inputs: the normals (three row dots) and the features (-> GOUTF rows 1.., and the stash for the reverse sweep)
    {
      const int in_rows = 4 * bd.tc[0];
      rowdot<3>(ldsi, cur, in_rows, wb + bd.offCBnrm, sm->part[img], w4, lane);
      gemm_tiles2<8>(lds, IS, KSegs{cur, in_rows, 0, 0}, wb + bd.offCBfeat, bd.feat_tiles, wave, lane, nopre, false, nullptr,
                     [&](int, int, f32x16& acc) { init_zero(acc); },
                     [&](int ot, int im, int rq, const f32x16& acc) {
                       f32x4* sv = save0 + (size_t)im * per_img + (size_t)feat_slot * 64;
                       const f32x4 v = {acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]};
                       st_stream(sv + (ot * 4 + rq) * 64 + lane, v);
                       if (2 * pair + im < n_tiles) {                     // GOUTF = [d sdf-output (0 here) ; d features]: feature f is row f + 1
                         float* base = tp.GOUTF + (2 * pair + im) * (long)bd.outf_tiles * 1024;
#pragma unroll
                         for (int j = 0; j < 4; ++j) {
                           const int f = 32 * ot + 2 * (4 * rq + j) + h + 1;
                           if (f < 32 * bd.outf_tiles) __builtin_nontemporal_store(v[j], base + f * 32 + p);
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
        if (2 * pair + im < n_tiles) {                                    // row 0 and the tail of GOUTF past the last feature
          float* base = tp.GOUTF + (2 * pair + im) * (long)bd.outf_tiles * 1024;
          base[t] = sm->gs[im][t];                                        // row 0: d loss / d sdf / scale -- the contraction GOUTF x u_L then
                                                                          // yields row 0 of the last layer's gradient (and its bias) by itself
          for (int f = 32 * bd.feat_tiles + 1; f < 32 * bd.outf_tiles; ++f) base[f * 32 + t] = 0.f;
        }
      }
      __syncthreads();
    }

    // ---------------- tangent pass: e' = J_posenc(x) v, then the forward layers without bias ----------------
    {
      const float x0 = sm->pts[img][p * 3 + 0] * bd.scale, x1 = sm->pts[img][p * 3 + 1] * bd.scale, x2 = sm->pts[img][p * 3 + 2] * bd.scale;
      const float v0 = sm->v[img][p * 3 + 0], v1 = sm->v[img][p * 3 + 1], v2 = sm->v[img][p * 3 + 2];
      for (int r = w4; r < 4 * bd.e_tiles; r += 4) {
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int f = row_feat(r, h, j);
          float val = 0.f;
          if (r < bd.emb_rows && f < bd.emb_feats) {
            int c;
            const float jac = posenc_jac(f, x0, x1, x2, &c);
            val = jac * (c == 0 ? v0 : (c == 1 ? v1 : v2));
          }
          y[j] = val;
        }
        if (r < bd.emb_rows) ldsi[(E0 + r) * 64 + lane] = y;
        if (live_w) tf_store(tp.ED, ptile_w, bd.e_tiles, r >> 2, r & 3, lane, y);
      }
    }
    __syncthreads();
    cur = X0; oth = Y0;
    for (int l = 0; l < nL; ++l) {
      const KSegs ks = (l == 0) ? KSegs{E0, bd.emb_rows, 0, 0} : KSegs{cur, 4 * bd.ts[l - 1], E0, (l == bd.skip) ? bd.emb_rows : 0};
      const int dst = (l == 0) ? X0 : oth;
      const int n_ot = bd.ts[l];
      float* const t_ud = tp.UD[l + 1];
      // next: the following tangent layer, or the top of the reverse sweep (u_L and S_{L-1}: this very GEMM's stash stores when
      // l = nL - 1 -- same wave, same addresses, issued after them)
      const Aux nxt = (l + 1 < nL) ? Aux{tp.U[l + 2], tp.GH[l + 1], nullptr, bd.ts[l + 1]}
                                   : Aux{tp.U[nL], nullptr, save0 + (size_t)(nL - 1) * 4 * MT * 64, bd.ts[nL - 1]};
      gemm_tiles2<8>(lds, IS, ks, wb + bd.offT[l], n_ot, wave, lane, nopre, false, nullptr,
                     [&](int, int, f32x16& acc) { init_zero(acc); },
                     [&](int ot, int im, int rq, const f32x16& acc) {
                       f32x4* li = lds + (size_t)im * IS;
                       f32x4* sv = save0 + (size_t)im * per_img + (size_t)l * 4 * MT * 64;
                       f32x4 v, s;
#pragma unroll
                       for (int j = 0; j < 4; ++j) {
                         const float a = acc[4 * rq + j];
                         const float e = fast_exp(-100.f * ca[im][rq][j]);           // softplus(beta = 100): s' = 1 - e, s''/s' = 100 e
                         v[j] = a * (1.f - e);
                         s[j] = cb[im][rq][j] * a * (100.f * e);
                       }
                       li[(dst + ot * 4 + rq) * 64 + lane] = v;
                       st_stream(sv + (ot * 4 + rq) * 64 + lane, s);
                       fetch_q(nxt, im, rq);                 // (l = nL - 1: reads back the stash quad stored just above -- same wave, in order)
                       if (2 * pair + im < n_tiles) tf_store(t_ud, 2 * pair + im, n_ot, ot, rq, lane, v);
                     });
      if (wave >= n_ot) fetch(nxt);
      __syncthreads();
      if (l == 0) { cur = X0; oth = Y0; } else { const int t = cur; cur = oth; oth = t; }
    }

    // ---------------- reverse sweep ----------------
    {                                                       // g_feat back from the stash into the free buffer
      const f32x4* sv = save0 + (size_t)img * per_img + (size_t)feat_slot * 64;
      const int feat_rows = 4 * bd.feat_tiles;
      for (int r0 = w4; r0 < feat_rows; r0 += 32) {
        f32x4 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = ld_stream(sv + min(r0 + 4 * c, feat_rows - 1) * 64 + lane);
#pragma unroll
        for (int c = 0; c < 8; ++c)
          if (r0 + 4 * c < feat_rows) ldsi[(oth + r0 + 4 * c) * 64 + lane] = v[c];
      }
    }
    __syncthreads();
    for (int l = nL; l >= 1; --l) {
      // l = nL: ab_{nL-1} from the last layer (features by GEMM, the sdf row as the rank-1 start value); else ab_{l-1} from ab_l
      const bool top = l == nL;
      const KSegs ks = top ? KSegs{oth, 4 * bd.feat_tiles, 0, 0} : KSegs{cur, 4 * bd.ts[l], 0, 0};
      const int dst = top ? cur : oth;
      const int n_ot = bd.ts[l - 1];
      float* const t_ab = tp.AB[l - 1];
      const f32x4* const wrow = wb + bd.offWrow;
      const Aux nxt = (l > 1) ? Aux{tp.U[l - 1], nullptr, save0 + (size_t)(l - 2) * 4 * MT * 64, bd.ts[l - 2]} : Aux{nullptr, nullptr, nullptr, 0};
      gemm_tiles2<8>(lds, IS, ks, wb + (top ? bd.offBtop : bd.offB[l]), n_ot, wave, lane, nopre, false, nullptr,
                     [&](int ot, int im, f32x16& acc) {
                       if (top) {
                         const float gs = sm->gs[im][p];
#pragma unroll
                         for (int rq = 0; rq < 4; ++rq) {
                           const f32x4 w = wrow[(ot * 4 + rq) * 2 + h];
#pragma unroll
                           for (int j = 0; j < 4; ++j) acc[4 * rq + j] = w[j] * gs;
                         }
                       } else init_zero(acc);
                     },
                     [&](int ot, int im, int rq, const f32x16& acc) {
                       f32x4* li = lds + (size_t)im * IS;
                       f32x4 v;
#pragma unroll
                       for (int j = 0; j < 4; ++j)
                         v[j] = acc[4 * rq + j] * act_bwd_from_out<ACT_SOFTPLUS100>(ca[im][rq][j]) + cb[im][rq][j];
                       fetch_q(nxt, im, rq);
                       li[(dst + ot * 4 + rq) * 64 + lane] = v;
                       if (2 * pair + im < n_tiles) tf_store(t_ab, 2 * pair + im, n_ot, ot, rq, lane, v);
                     });
      if (wave >= n_ot) fetch(nxt);
      __syncthreads();
      if (!top) { const int t = cur; cur = oth; oth = t; }
    }
  }
}

size_t lds_bytes_b(int MT) { return (size_t)2 * (E_ROWS + 8 * MT) * 1024 + sizeof(SmallsB); }

int load_desc(const int32_t* desc, TrainBwdDesc& bd) {
  memcpy(&bd, desc, sizeof(TrainBwdDesc));
  if (bd.nL < 2 || bd.nL >= TB_MAX_L || bd.nC < 1 || bd.nC >= TB_MAX_L) return 1;
  if (bd.max_tiles < 5 || bd.max_tiles > 8 || lds_bytes_b(bd.max_tiles) > 160 * 1024) return 2;      // (<= 8: a wave owns at most one tile per GEMM)
  if (bd.emb_rows < 1 || bd.emb_rows > E_ROWS || bd.e_tiles * 4 < bd.emb_rows || bd.e_tiles > 2 || bd.emb_feats < 3 || bd.emb_feats > 8 * bd.emb_rows) return 3;
  if (bd.skip == 0 || bd.skip >= bd.nL) return 4;
  if (bd.feat_tiles < 1 || bd.feat_tiles > bd.max_tiles || 32 * bd.outf_tiles < 32 * bd.feat_tiles + 1) return 5;
  for (int l = 0; l < bd.nL; ++l) if (bd.ts[l] < 1 || bd.ts[l] > bd.max_tiles) return 6;
  for (int l = 0; l < bd.nC; ++l) if (bd.tc[l] < 1 || bd.tc[l] > bd.max_tiles) return 7;
  if (!(bd.scale > 0.f)) return 8;
  return 0;
}

}  // namespace

extern "C" int64_t vqn_neus_train_bwd_scratch_bytes(const int32_t* desc) {
  if (!desc) return -1;
  TrainBwdDesc bd;
  if (load_desc(desc, bd) != 0) return -1;
  return (int64_t)vqn_num_cus() * 2 * (bd.nL + 1) * 4 * bd.max_tiles * 1024;
}

extern "C" int vqn_neus_train_bwd(const int32_t* desc, const float* wbuf, const float* pts, const float* g_rgb, const float* rgb,
                                  const float* g_n, const float* g_sdf, int64_t P, void* scratch, int64_t scratch_bytes,
                                  const float* const* saved, int n_saved, float* const* outs, int n_outs, void* stream) {
  VQN_CHECK_ARG(desc && wbuf && pts && g_rgb && scratch && saved && outs, "null pointer");
  VQN_CHECK_ARG(P >= 1, "P >= 1");
  TrainBwdDesc bd;
  VQN_CHECK_SHAPE(load_desc(desc, bd) == 0, "invalid backward descriptor");
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
  VQN_HIP(hipFuncSetAttribute((const void*)neus_train_bwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  long grid = (long)vqn_num_cus();
  if (grid > (n_tiles + 1) / 2) grid = (n_tiles + 1) / 2;
  if ((int64_t)grid * per_wg > scratch_bytes) grid = (long)(scratch_bytes / per_wg);
  VQN_CHECK_ARG(grid >= 1, "scratch too small (see vqn_neus_train_bwd_scratch_bytes)");
  hipLaunchKernelGGL(neus_train_bwd2_kernel, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, bd,
                     reinterpret_cast<const f32x4*>(wbuf), tp, (long)P, reinterpret_cast<f32x4*>(scratch));
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
