// Training passes of the reflectance Dense stacks on the exact-split engine (csrc/mlp_prims_x3.h) -- round 4.
//
// What they stand for in the reference: `tape.gradient` through `_pred_enc_at` / `_pred_{diff,spec,rough}_at`
// (decomp/nerfvq_nfr3/nerfactor/models/vq_nfr.py:771-828 over networks/{embedder,mlp,seq}.py) under train_nfr.py:562-576 -- the
// forward with everything the backward needs kept, and the backward down to the per-point adjoints of every Dense layer (the
// weight gradients themselves are the contractions of csrc/wgrad*.hip over the tensors these kernels leave).  Rounds 1-3 ran these
// passes as interpreted tile programs on the f32-input MFMA (csrc/tile_vm.hip: 10 of the 19 ms of a 262,144-point step).
//
// One stack shape, two uses (ReflDesc):
//   [positional encoding -> encoder layers (one skip-concat of the encoding) -> z]   optional (n_enc = 0: the input is z rows)
//   -> up to three heads on z: Dense(w0) relu -> Dense(w1) relu -> Dense(c <= 3) sigmoid over [y1 ; z]  (mlp.Network(skip_at = [1]))
// A = encoder + continuous heads of vq_nfr / nfr_unit, B = the VQ heads on the quantised rows.
//
// Kernel shape = the NeuS exact-split kernels': one 512-thread workgroup per CU holds TWO 32-point images as bf16 piece triples,
// layers run IN PLACE (finished tiles wait in registers for the barrier), weights stream through the register ring along a table of the
// pass's GEMM calls.  Layers of more than four output tiles: wave w owns tile w for both images (gemm_tiles_x3_ring2); layers of at
// most four (the 128-wide ones): the waves split by image (gemm_tile_x3_ring1) so that all eight have a tile.  The 1..3-output last
// layer of a head never touches the matrix pipe: forward row dots, backward rank-c updates on the vector ALU.
// Saved tensors and adjoints are f32 in the tile format [point tile][feature tile][32 features][32 points] the contraction reads.
#include "mlp_prims_x3.h"
#include "vqnerf_hip.h"
#include <stdlib.h>
#include <type_traits>

using namespace eng;
template <int I> using IC = std::integral_constant<int, I>;

namespace {

constexpr int E0 = 0;
constexpr int E_ROWS = 12;
constexpr int X0 = E_ROWS;
constexpr int RING = 2;
constexpr int NW = 8;
constexpr int RT_MAX_L = 8;
constexpr int RT_MAX_H = 3;
constexpr int MAX_CALLS = RT_MAX_L + 2 * RT_MAX_H;

// In-kernel phase timing of the backward (diagnostic build -DVQN_RT_STAMPS only; scripts/debug/refl_stamps.py): wave 0 of every
// workgroup accumulates shader-clock cycles per phase.
#ifdef VQN_RT_STAMPS
__device__ unsigned long long g_rt_stamps[8 * 16];      // [wave][phase]
#define RT_STAMP_DECL unsigned long long st_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_prev = __builtin_amdgcn_s_memtime(); const unsigned long long st_begin = st_prev;
#define RT_STAMP(i) { const unsigned long long st_now = __builtin_amdgcn_s_memtime(); st_[i] += st_now - st_prev; st_prev = st_now; }
#define RT_STAMP_FLUSH if ((threadIdx.x & 63) == 0) { st_[9] = __builtin_amdgcn_s_memtime() - st_begin; unsigned long long* g_ = g_rt_stamps + 16 * (threadIdx.x >> 6); for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_[i_], st_[i_]); atomicAdd(&g_[10], 1ull); }
#else
#define RT_STAMP_DECL
#define RT_STAMP(i) {}
#define RT_STAMP_FLUSH
#endif

// Round 5 -- a SECOND z input for the heads (stage 3, ref_nfr.py:148-152,203-213: the diffuse / roughness heads read [z_xyz ; z_ref], 512
// wide): `zx_tiles` = z_tiles switches it on.  The rows of the second input (z_xyz: the frozen stage-2 encoder's output, no adjoint
// wanted) stand in a second image region X1 for the whole unit; a head's first layer is ONE GEMM over the two K segments [X0: z ; X1: zx]
// (the pack holds the 2 Z input rows of the Dense kernel in that order), its last layer's row dots take a third share from X1.  The
// backward kernel is unchanged: d / d z comes from the z rows of the kernels only, the weight gradients of the zx rows are two more
// contractions against the tile-format copy of zx this kernel leaves.  One image per workgroup only (two regions = 108 KB of LDS).
struct ReflDesc {
  int n_enc, skip, emb_rows, emb_feats, e_tiles, max_tiles, n_heads, z_tiles;
  int z_feats, zx_tiles, offW2zx[RT_MAX_H], rsv[3];
  int te[RT_MAX_L], act[RT_MAX_L], offW[RT_MAX_L], offBias[RT_MAX_L], offWb[RT_MAX_L];
  int t0[RT_MAX_H], t1[RT_MAX_H], c[RT_MAX_H], offW0[RT_MAX_H], offW1[RT_MAX_H], offB0[RT_MAX_H], offB1[RT_MAX_H], offW2y[RT_MAX_H],
      offW2z[RT_MAX_H], offB2[RT_MAX_H], offW1b[RT_MAX_H], offW0b[RT_MAX_H], offA2y[RT_MAX_H], offA2z[RT_MAX_H];
};
constexpr int RT_DESC_INTS = 16 + 5 * RT_MAX_L + 14 * RT_MAX_H;
static_assert(sizeof(ReflDesc) == RT_DESC_INTS * 4, "descriptor layout");

struct ReflFwdPtrs {
  const float* X; const float* ZR;          // points [N, 3] (n_enc > 0) | input rows [N, z_feats] (n_enc = 0)
  float* E; float* Y[RT_MAX_L];             // saved: encoding, every encoder layer's output (the last one is z)
  float* ZT;                                // z in the tile format: Y[n_enc - 1] with an encoder, else the copy this kernel writes
  float* ZROWS;                             // z as rows [N, z_feats] (with an encoder; may be NULL)
  float* H0[RT_MAX_H]; float* H1[RT_MAX_H]; float* OUT[RT_MAX_H];
  int save;                                 // 0: inference -- nothing is kept for a backward (only z's tile-format copy, which heads 2 and 3 re-read)
  const float* ZX; float* ZXT;              // second head input (zx_tiles > 0): rows [N, z_feats] and their tile-format copy (may be NULL)
};

struct ReflBwdPtrs {
  const float* G_OUT[RT_MAX_H]; const float* OUT[RT_MAX_H]; const float* H0[RT_MAX_H]; const float* H1[RT_MAX_H];
  float* D2[RT_MAX_H]; float* D1[RT_MAX_H]; float* D0[RT_MAX_H];
  const float* G_Z[4]; int n_gz;            // rows [N, z_feats]: adjoints of z from outside this launch's heads, summed in order
  int run_heads, run_enc;                   // which part of the stack this launch walks (both: the whole backward in one launch)
  int d2_shared, d2_row0[RT_MAX_H];         // shared: head k writes rows d2_row0[k].. of ONE delta_2 tile (nothing else); else rows 0.. + zeros
  float* GZ_ROWS;                           // without the encoder part: d / d z rows [gridDim.y][N, z_feats] (one slice per head when split)
  const float* Y[RT_MAX_L]; float* D[RT_MAX_L];
};

struct SmallsR {
  float pts[2][96], part[2 * 4 * 32 * 3], h2z[RT_MAX_H][2][96], d2[2][96];
  int tab[MAX_CALLS * 4];                   // GEMM calls of one unit in program order: {float4 offset, K blocks, out tiles, 0}
  int n_calls;
};

// accumulator tile <-> tile format: register i of lane (p, h) is feature (i & 3) + 8 (i >> 2) + 4 h of the feature tile
__device__ __forceinline__ void tf_store_acc(float* __restrict__ T, const long ptile, const int n_ft, const int ot, const int lane, const float (&v)[16]) {
#ifdef VQN_DIAG_RT_NO_ST        // timing only
  asm volatile("" ::"v"(v[0]), "v"(v[5]), "v"(v[10]), "v"(v[15]));
  return;
#endif
  float* base = T + ((ptile * n_ft + ot) * 32 + 4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 16; ++i) __builtin_nontemporal_store(v[i], base + ((i & 3) + 8 * (i >> 2)) * 32);
}
__device__ __forceinline__ void tf_load_acc(const float* __restrict__ T, const long ptile, const int n_ft, const int ot, const int lane, float (&v)[16]) {
#ifdef VQN_DIAG_RT_NO_LD        // timing only
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 0.5f + 0.001f * (float)(lane + i);
  return;
#endif
  // (streaming `nt` loads here -- so that the saved tensors, 14 KB per point passing through once, leave the weight packs in L2 --
  //  measured no better: backward 3.18 -> 3.26 ms per 262,144 points)
  const float* base = T + ((ptile * n_ft + ot) * 32 + 4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = base[((i & 3) + 8 * (i >> 2)) * 32];
}
// a K step of an image: slot jj of lane (p, h) is feature 16 sl + 8 (jj >> 2) + 4 h + (jj & 3)
__device__ __forceinline__ void tf_store_step(float* __restrict__ T, const long ptile, const int n_ft, const int sl, const int lane, const float (&x)[8]) {
  float* base = T + ((ptile * n_ft + (sl >> 1)) * 32 + 16 * (sl & 1) + 4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) __builtin_nontemporal_store(x[jj], base + (8 * (jj >> 2) + (jj & 3)) * 32);
}

__device__ __forceinline__ void act_apply(const int act, const f32x16& acc, float (&v)[16]) {
  if (act == ACT_RELU) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fmaxf(acc[i], 0.f);
  } else if (act == ACT_SIGMOID) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = act_fwd<ACT_SIGMOID>(acc[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = acc[i];
  }
}
// v *= act'(y) with the derivative taken from the layer's OUTPUT y
__device__ __forceinline__ void dact_mul(const int act, const float (&y)[16], float (&v)[16]) {
  if (act == ACT_RELU) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = y[i] > 0.f ? v[i] : 0.f;
  } else if (act == ACT_SIGMOID) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] *= y[i] * (1.f - y[i]);
  }
}

// shared pieces of the two kernels --------------------------------------------------------------------------------------------
// NIMG = 2: a workgroup holds two 32-point images (a "unit" = a tile pair), every weight fragment serves both; layers of at most four
// output tiles split the waves by image.  NIMG = 1 (small batches: fewer tile pairs than CUs): one image per workgroup, wave w owns
// output tile w -- twice the workgroups, half the matrix time per layer.  gridDim.y > 1 (small batches again): one HEAD per workgroup
// row, the encoder (forward) evaluated by each of them -- the reference batch of 2048 points is 64 tiles, a quarter of the chip.
// (-DVQN_REFL_OCC2 A/B: the second workgroup of a CU starts half a layer late, so that its K loops fall under the first one's epilogues
//  instead of beside its K loops -- partners that start together stay in lockstep)
#if defined(VQN_REFL_OCC2) && defined(VQN_OCC2_DELAY)
#define RT_OCC2_PHASE_DELAY()                                                                                \
  if (NIMG == 1 && blockIdx.x >= gridDim.x / 2)                                                              \
    for (int i_ = 0; i_ < VQN_OCC2_DELAY; ++i_) __builtin_amdgcn_s_sleep(127)
#else
#define RT_OCC2_PHASE_DELAY() (void)0
#endif
#define REFL_PROLOGUE()                                                                                      \
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];                                                \
  const int MT = rd.max_tiles;                                                                               \
  const int IMG = E_ROWS + 6 * MT + 6 * rd.zx_tiles, IS = IMG * 64;                                          \
  const int X1 = E_ROWS + 6 * MT;                       /* rows of the second head input (zx_tiles > 0) */   \
  SmallsR* sm = reinterpret_cast<SmallsR*>(lds + (size_t)NIMG * IS);                                         \
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;                                \
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                 \
  constexpr int NWI = NIMG == 1 ? 8 : 4;                /* waves per image in the vector-ALU phases */        \
  const int img = NIMG == 1 ? 0 : (wave >> 2), w4 = NIMG == 1 ? wave : (wave & 3);                           \
  f32x4* ldsi = lds + (size_t)img * IS;                                                                      \
  float* part_i = sm->part + img * (4 * 32 * 3);        /* (NIMG = 1: all 8 x 32 x 3 floats are image 0's) */ \
  const long n_tiles = (P + 31) >> 5, n_units = NIMG == 1 ? n_tiles : ((n_tiles + 1) >> 1);                  \
  const int k_lo = gridDim.y > 1 ? (int)blockIdx.y : 0, k_hi = gridDim.y > 1 ? k_lo + 1 : rd.n_heads;       \
  auto blocks_of = [](int krows) { return (krows / 3 + 1) >> 1; };                                           \
  auto ptile_of = [&](long unit, int im) { return NIMG == 1 ? unit : 2 * unit + im; };                       \
  (void)ldsi; (void)h; (void)p; (void)blocks_of; (void)part_i; (void)k_hi; (void)NWI; (void)X1;              \
  RT_OCC2_PHASE_DELAY();

// sum of the NWI per-wave partial row dots of (point pp, output c)
#define REFL_PART_SUM(pr, pp, nc, c)                                                                         \
  (NIMG == 1 ? (((pr)[(0 * 32 + (pp)) * (nc) + (c)] + (pr)[(1 * 32 + (pp)) * (nc) + (c)]) + ((pr)[(2 * 32 + (pp)) * (nc) + (c)] + (pr)[(3 * 32 + (pp)) * (nc) + (c)])) + \
               (((pr)[(4 * 32 + (pp)) * (nc) + (c)] + (pr)[(5 * 32 + (pp)) * (nc) + (c)]) + ((pr)[(6 * 32 + (pp)) * (nc) + (c)] + (pr)[(7 * 32 + (pp)) * (nc) + (c)])) \
             : (((pr)[(0 * 32 + (pp)) * (nc) + (c)] + (pr)[(1 * 32 + (pp)) * (nc) + (c)]) + ((pr)[(2 * 32 + (pp)) * (nc) + (c)] + (pr)[(3 * 32 + (pp)) * (nc) + (c)])))

// the wave's first tile of the next GEMM call (after `idx`, wrapping into the next unit) in which it owns one
#define REFL_STREAM()                                                                                        \
  const int n_calls = __builtin_amdgcn_readfirstlane(sm->n_calls);                                           \
  auto next_stream = [&](int idx, const f32x4*& nwp, int& nnb) {                                             \
    nwp = wx + lane; nnb = 1;                                                                                \
    for (int k = 1; k <= n_calls; ++k) {                                                                     \
      const int m = (idx + k) % n_calls;                                                                     \
      const int tiles = __builtin_amdgcn_readfirstlane(sm->tab[4 * m + 2]);                                  \
      const int mine = (NIMG == 2 && tiles <= 4) ? w4 : wave;                                                \
      if (mine < tiles) {                                                                                    \
        const int off = __builtin_amdgcn_readfirstlane(sm->tab[4 * m]);                                      \
        nnb = __builtin_amdgcn_readfirstlane(sm->tab[4 * m + 1]);                                            \
        nwp = wx + off + (size_t)mine * nnb * 384 + lane;                                                    \
        return;                                                                                              \
      }                                                                                                      \
    }                                                                                                        \
  };                                                                                                         \
  f32x4 ring[RING][6];                                                                                       \
  if (n_calls > 0) {                                                                                         \
    const f32x4* wp0; int nb0;                                                                               \
    next_stream(n_calls - 1, wp0, nb0);                                                                      \
    ring_prime_x3<RING>(ring, wp0, nb0);                                                                     \
  }                                                                                                          \
  int call = 0;                                                                                              \
  /* init(ot, im, slot, acc) / epi(ot, im, slot, acc) once per (tile, image) the wave owns; `slot` (a compile-time constant    \
     after inlining: 0 in the one-tile forms, the image in the two-image form) indexes the per-image REGISTER arrays (o, au) -- a  \
     run-time index would send them to scratch */                                                                               \
  auto Gx = [&](const int off, const KSegs ks, const int tiles, auto commit_c, auto init, auto epi) __attribute__((always_inline)) { \
    constexpr bool COMMIT = decltype(commit_c)::value != 0;                                                  \
    const f32x4* nwp; int nnb;                                                                               \
    next_stream(call, nwp, nnb);                                                                             \
    ++call;                                                                                                  \
    /* COMMIT (a layer IN PLACE): the wave's accumulators stay where the K loop left them, a barrier sees every wave out of its K   \
       loop -- the input rows are dead --, then the epilogues write their tiles straight over the input (store_tile_x3) and a second  \
       barrier publishes them.  Waves without a tile in this GEMM arrive at the same two barriers from the else branches.  (Parking  \
       the finished, already split tiles in registers across the barrier instead -- 48 of them -- ended in scratch: every commit     \
       re-read 18 x 16 B per lane behind the epilogue's global stores, a third of the backward's time.) */                          \
    if constexpr (NIMG == 1) {                                                                               \
      if (wave < tiles)                                                                                      \
        gemm_tile_x3_ring1<RING>(lds, ks, wx + off, wave, lane, ring, nwp, nnb,                              \
                                 [&](f32x16& acc) __attribute__((always_inline)) { init(wave, 0, IC<0>{}, acc); },                 \
                                 [&](const f32x16& acc) __attribute__((always_inline)) { if (COMMIT) __syncthreads(); epi(wave, 0, IC<0>{}, acc); }); \
      else if (COMMIT) __syncthreads();                                                                      \
    } else {                                                                                                 \
      if (tiles <= 4) {                                                                                      \
        if (w4 < tiles)                                                                                      \
          gemm_tile_x3_ring1<RING>(ldsi, ks, wx + off, w4, lane, ring, nwp, nnb,                             \
                                   [&](f32x16& acc) __attribute__((always_inline)) { init(w4, img, IC<0>{}, acc); },               \
                                   [&](const f32x16& acc) __attribute__((always_inline)) { if (COMMIT) __syncthreads(); epi(w4, img, IC<0>{}, acc); }); \
        else if (COMMIT) __syncthreads();                                                                    \
      } else {                                                                                               \
        gemm_tiles_x3_ring2<NW, RING, 1>(lds, IS, ks, wx + off, tiles, wave, lane, ring, nwp, nnb,           \
                                         [&](int ot, int im, f32x16& acc) __attribute__((always_inline)) { if (im == 0) init(ot, 0, IC<0>{}, acc); else init(ot, 1, IC<1>{}, acc); }, \
                                         [&](int ot, int im, const f32x16& acc) __attribute__((always_inline)) {                                       \
                                           if (im == 0) { if (COMMIT) __syncthreads(); epi(ot, 0, IC<0>{}, acc); } else epi(ot, 1, IC<1>{}, acc); });    \
        if (COMMIT && wave >= tiles) __syncthreads();                                                        \
      }                                                                                                      \
    }                                                                                                        \
    if (COMMIT) __syncthreads();                                                                             \
  };                                                                                                         \
  auto G = [&](const int off, const KSegs ks, const int tiles, auto init, auto epi) __attribute__((always_inline)) { Gx(off, ks, tiles, IC<0>{}, init, epi); };  \
  auto GC = [&](const int off, const KSegs ks, const int tiles, auto init, auto epi) __attribute__((always_inline)) { Gx(off, ks, tiles, IC<1>{}, init, epi); }; (void)G; \
  /* fn(ot, im, slot) for every (tile, image) this wave owns in a `tiles`-tile tensor (the GEMMs' ownership) */ \
  auto owned = [&](const int tiles, auto fn) __attribute__((always_inline)) {                                                               \
    if constexpr (NIMG == 1) { if (wave < tiles) fn(wave, 0, IC<0>{}); }                                           \
    else {                                                                                                   \
      if (tiles <= 4) { if (w4 < tiles) fn(w4, img, IC<0>{}); }                                                    \
      else if (wave < tiles) { fn(wave, 0, IC<0>{}); fn(wave, 1, IC<1>{}); }                                             \
    }                                                                                                        \
  };

// ================================================================ forward ================================================================
// A/B of VERDICT r3 #2(a), -DVQN_REFL_OCC2: the one-image form compiled for FOUR waves per SIMD (128 registers) and launched as two
// workgroups per CU, so that one workgroup's epilogue runs under the other's K loop (profiles/r04_refl_occ2_ab.txt has the outcome)
#ifdef VQN_REFL_OCC2
#define RT_WAVES_EU(N) ((N) == 1 ? 4 : 1)
#define RT_WGS_PER_CU(N) ((N) == 1 ? 2 : 1)
#else
#define RT_WAVES_EU(N) 1
#define RT_WGS_PER_CU(N) 1
#endif

template <int NIMG>
__global__ __launch_bounds__(512, RT_WAVES_EU(NIMG)) void refl_train_fwd_x3_kernel(const ReflDesc rd, const f32x4* __restrict__ wx, const f32x4* __restrict__ wf,
                                                                   const ReflFwdPtrs tp, const long P) {
  REFL_PROLOGUE();
  const int nE = rd.n_enc, ZT = rd.z_tiles;
  const bool wr_shared = blockIdx.y == 0;             // (heads split over blockIdx.y: the encoder's tensors are written by row 0 only)
  if (tid == 0) {
    int n = 0;
    auto add = [&](int off, int krows, int tiles) { sm->tab[4 * n] = off; sm->tab[4 * n + 1] = blocks_of(krows); sm->tab[4 * n + 2] = tiles; sm->tab[4 * n + 3] = 0; ++n; };
    for (int l = 0; l < nE; ++l) add(rd.offW[l], l == 0 ? rd.emb_rows : 6 * rd.te[l - 1] + (l == rd.skip ? rd.emb_rows : 0), rd.te[l]);
    for (int k = k_lo; k < k_hi; ++k) { add(rd.offW0[k], 6 * ZT + 6 * rd.zx_tiles, rd.t0[k]); add(rd.offW1[k], 6 * rd.t0[k], rd.t1[k]); }
    sm->n_calls = n;
  }
  __syncthreads();
  REFL_STREAM();
  const int ZXT = rd.zx_tiles;

  for (long unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    call = 0;
    const long ptile_w = ptile_of(unit, img);
    const bool live_w = ptile_w < n_tiles;
    if (nE > 0) {
      // ---------------- points -> positional encoding -> E rows ----------------
      if (tid < 32 * NIMG) {
        const int im = tid >> 5, t = tid & 31;
        long pt = (ptile_of(unit, im) << 5) + t;
        if (pt >= P) pt = P - 1;
        sm->pts[im][t * 3 + 0] = tp.X[pt * 3 + 0]; sm->pts[im][t * 3 + 1] = tp.X[pt * 3 + 1]; sm->pts[im][t * 3 + 2] = tp.X[pt * 3 + 2];
      }
      __syncthreads();
      const float xs = sm->pts[img][p * 3 + 0], ys = sm->pts[img][p * 3 + 1], zs = sm->pts[img][p * 3 + 2];
      for (int sl = w4; sl < 2 * rd.e_tiles; sl += NWI) {
        float x[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int f = step_feat(sl, h, jj);
          x[jj] = (sl < rd.emb_rows / 3 && f < rd.emb_feats) ? posenc_feat(f, xs, ys, zs) : 0.f;
        }
        if (live_w && wr_shared && tp.save) tf_store_step(tp.E, ptile_w, rd.e_tiles, sl, lane, x);
        if (sl < rd.emb_rows / 3) {
          f32x4 q0, q1, q2;
          split3x8(x, q0, q1, q2);
          ldsi[(E0 + 3 * sl) * 64 + lane] = q0; ldsi[(E0 + 3 * sl + 1) * 64 + lane] = q1; ldsi[(E0 + 3 * sl + 2) * 64 + lane] = q2;
        }
      }
      __syncthreads();
      // ---------------- encoder layers (in place) ----------------
      for (int l = 0; l < nE; ++l) {
        const KSegs ks = (l == 0) ? KSegs{E0, rd.emb_rows, 0, 0} : KSegs{X0, 6 * rd.te[l - 1], E0, (l == rd.skip) ? rd.emb_rows : 0};
        const int n_ot = rd.te[l], act = rd.act[l];
        const f32x4* bp = wf + rd.offBias[l];
        float* const t_y = tp.Y[l];
        const bool top = l == nE - 1;
        GC(rd.offW[l], ks, n_ot,
          [&](int ot, int, auto, f32x16& acc) __attribute__((always_inline)) { init_bias_f16s(bp, ot, lane, acc); },
          [&](int ot, int im, auto slc_, const f32x16& acc) __attribute__((always_inline)) {
            constexpr int sl_ = decltype(slc_)::value; (void)sl_;
            float v[16];
            act_apply(act, acc, v);
            const long ptile = ptile_of(unit, im);
            if (ptile < n_tiles && wr_shared) {
              if (tp.save || top) tf_store_acc(t_y, ptile, n_ot, ot, lane, v);
              if (top && tp.ZROWS != nullptr) {
                const long pt = (ptile << 5) + p;
                if (pt < P) {
                  float* row = tp.ZROWS + pt * (long)rd.z_feats + 32 * ot + 4 * h;
#pragma unroll
                  for (int q = 0; q < 4; ++q)
                    if (32 * ot + 8 * q + 4 * h < rd.z_feats)
                      *reinterpret_cast<f32x4*>(row + 8 * q) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                }
              }
            }
            store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
          });
      }
    } else {
      // ---------------- input rows -> X (piece triples) + their tile-format copy ----------------
      const long pt = (ptile_w << 5) + p;
      const bool valid = pt < P;
      const float* row = tp.ZR + (valid ? pt : P - 1) * (long)rd.z_feats;
      for (int sl = w4; sl < 2 * ZT; sl += NWI) {
        float x[8];
        const int f0 = 16 * sl + 4 * h;
        const f32x4 a = (f0 < rd.z_feats) ? *reinterpret_cast<const f32x4*>(row + f0) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 b = (f0 + 8 < rd.z_feats) ? *reinterpret_cast<const f32x4*>(row + f0 + 8) : (f32x4){0.f, 0.f, 0.f, 0.f};
        x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
        if (!valid) {
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = 0.f;
        }
        if (live_w && wr_shared) tf_store_step(tp.ZT, ptile_w, ZT, sl, lane, x);
        f32x4 q0, q1, q2;
        split3x8(x, q0, q1, q2);
        ldsi[(X0 + 3 * sl) * 64 + lane] = q0; ldsi[(X0 + 3 * sl + 1) * 64 + lane] = q1; ldsi[(X0 + 3 * sl + 2) * 64 + lane] = q2;
      }
      __syncthreads();
    }

    // ---------------- heads ----------------
    if (ZXT > 0) {
      // the second input's rows -> region X1 (piece triples) + their tile-format copy (the contraction's operand for the zx rows of
      // the first / last Dense kernels); nothing writes X1 until the next unit
      const long pt = (ptile_w << 5) + p;
      const bool valid = pt < P;
      const float* row = tp.ZX + (valid ? pt : P - 1) * (long)rd.z_feats;
      for (int sl = w4; sl < 2 * ZXT; sl += NWI) {
        float x[8];
        const int f0 = 16 * sl + 4 * h;
        const f32x4 a = (f0 < rd.z_feats) ? *reinterpret_cast<const f32x4*>(row + f0) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 b = (f0 + 8 < rd.z_feats) ? *reinterpret_cast<const f32x4*>(row + f0 + 8) : (f32x4){0.f, 0.f, 0.f, 0.f};
        x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
        if (!valid) {
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = 0.f;
        }
        if (live_w && wr_shared && tp.ZXT != nullptr) tf_store_step(tp.ZXT, ptile_w, ZXT, sl, lane, x);
        f32x4 q0, q1, q2;
        split3x8(x, q0, q1, q2);
        ldsi[(X1 + 3 * sl) * 64 + lane] = q0; ldsi[(X1 + 3 * sl + 1) * 64 + lane] = q1; ldsi[(X1 + 3 * sl + 2) * 64 + lane] = q2;
      }
      __syncthreads();
    }
    // the z share of every head's last layer while z is in the buffer: part -> h2z[k][img][p * 3 + c]
    for (int k = k_lo; k < k_hi; ++k) {
      if (rd.c[k] == 1) rowdot_x3<1, NWI>(ldsi, X0, 6 * ZT, wf + rd.offW2z[k], part_i, w4, lane);
      else rowdot_x3<3, NWI>(ldsi, X0, 6 * ZT, wf + rd.offW2z[k], part_i, w4, lane);
      __syncthreads();
      const int nc = rd.c[k] == 1 ? 1 : 3;
      if (tid < 32 * NIMG * nc) {
        const int im = tid / (32 * nc), r = tid - 32 * nc * im, pp = r % 32, c = r / 32;
        const float* pr = sm->part + im * (4 * 32 * 3);
        sm->h2z[k][im][pp * 3 + c] = REFL_PART_SUM(pr, pp, nc, c);
      }
      __syncthreads();
      if (ZXT > 0) {                                   // + the zx share (the rows of the last Dense kernel that face the second input)
        if (rd.c[k] == 1) rowdot_x3<1, NWI>(ldsi, X1, 6 * ZXT, wf + rd.offW2zx[k], part_i, w4, lane);
        else rowdot_x3<3, NWI>(ldsi, X1, 6 * ZXT, wf + rd.offW2zx[k], part_i, w4, lane);
        __syncthreads();
        if (tid < 32 * NIMG * nc) {
          const int im = tid / (32 * nc), r = tid - 32 * nc * im, pp = r % 32, c = r / 32;
          const float* pr = sm->part + im * (4 * 32 * 3);
          sm->h2z[k][im][pp * 3 + c] += REFL_PART_SUM(pr, pp, nc, c);
        }
        __syncthreads();
      }
    }
    for (int k = k_lo; k < k_hi; ++k) {
      if (k > k_lo) {                                    // z back into the buffer (the previous head ran over it)
        owned(ZT, [&](int ot, int im, auto) __attribute__((always_inline)) {
          float v[16];
          const long ptile = ptile_of(unit, im);
          if (ptile < n_tiles) tf_load_acc(tp.ZT, ptile, ZT, ot, lane, v);
          else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = 0.f;
          }
          store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
        });
        __syncthreads();
      }
      {
        const int n_ot = rd.t0[k];
        const f32x4* bp = wf + rd.offB0[k];
        float* const t_y = tp.H0[k];
        GC(rd.offW0[k], KSegs{X0, 6 * ZT, X1, 6 * ZXT}, n_ot,
          [&](int ot, int, auto, f32x16& acc) __attribute__((always_inline)) { init_bias_f16s(bp, ot, lane, acc); },
          [&](int ot, int im, auto slc_, const f32x16& acc) __attribute__((always_inline)) {
            constexpr int sl_ = decltype(slc_)::value; (void)sl_;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaxf(acc[i], 0.f);
            if (tp.save && ptile_of(unit, im) < n_tiles) tf_store_acc(t_y, ptile_of(unit, im), n_ot, ot, lane, v);
            store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
          });
      }
      {
        const int n_ot = rd.t1[k];
        const f32x4* bp = wf + rd.offB1[k];
        float* const t_y = tp.H1[k];
        GC(rd.offW1[k], KSegs{X0, 6 * rd.t0[k], 0, 0}, n_ot,
          [&](int ot, int, auto, f32x16& acc) __attribute__((always_inline)) { init_bias_f16s(bp, ot, lane, acc); },
          [&](int ot, int im, auto slc_, const f32x16& acc) __attribute__((always_inline)) {
            constexpr int sl_ = decltype(slc_)::value; (void)sl_;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaxf(acc[i], 0.f);
            if (tp.save && ptile_of(unit, im) < n_tiles) tf_store_acc(t_y, ptile_of(unit, im), n_ot, ot, lane, v);
            store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
          });
      }
      // last layer: row dots over y1, + the z share, + bias -> sigmoid
      if (rd.c[k] == 1) rowdot_x3<1, NWI>(ldsi, X0, 6 * rd.t1[k], wf + rd.offW2y[k], part_i, w4, lane);
      else rowdot_x3<3, NWI>(ldsi, X0, 6 * rd.t1[k], wf + rd.offW2y[k], part_i, w4, lane);
      __syncthreads();
      {
        const int nc = rd.c[k] == 1 ? 1 : 3, cc = rd.c[k];
        if (tid < 32 * NIMG * nc) {
          const int im = tid / (32 * nc), r = tid - 32 * nc * im, pp = r % 32, c = r / 32;
          const float* pr = sm->part + im * (4 * 32 * 3);
          const float s = REFL_PART_SUM(pr, pp, nc, c) + sm->h2z[k][im][pp * 3 + c] + wf[rd.offB2[k]][c];
          const long pt = (ptile_of(unit, im) << 5) + pp;
          if (pt < P && c < cc) tp.OUT[k][pt * cc + c] = act_fwd<ACT_SIGMOID>(s);
        }
      }
      __syncthreads();
    }
  }
}

// ================================================================ backward ================================================================
template <int NIMG>
__global__ __launch_bounds__(512, RT_WAVES_EU(NIMG)) void refl_train_bwd_x3_kernel(const ReflDesc rd, const f32x4* __restrict__ wx, const f32x4* __restrict__ wf,
                                                                   const ReflBwdPtrs tp, const long P, f32x4* __restrict__ scratch) {
  REFL_PROLOGUE();
  RT_STAMP_DECL
  const int ZT = rd.z_tiles;
  const int nE = tp.run_enc ? rd.n_enc : 0;                    // (the encoder part is walked only when asked for)
  const bool heads = tp.run_heads != 0 && k_hi > k_lo;
  const size_t per_img = (size_t)4 * ZT * 64;                  // the d / d z accumulator between heads (accumulator-order quads)
  f32x4* save0 = scratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NIMG * per_img;
  if (tid == 0) {
    int n = 0;
    auto add = [&](int off, int krows, int tiles) { sm->tab[4 * n] = off; sm->tab[4 * n + 1] = blocks_of(krows); sm->tab[4 * n + 2] = tiles; sm->tab[4 * n + 3] = 0; ++n; };
    if (heads)
      for (int k = k_lo; k < k_hi; ++k) { add(rd.offW1b[k], 6 * rd.t1[k], rd.t0[k]); add(rd.offW0b[k], 6 * rd.t0[k], ZT); }
    for (int l = nE - 1; l >= 1; --l) add(rd.offWb[l], 6 * rd.te[l], rd.te[l - 1]);
    sm->n_calls = n;
  }
  __syncthreads();
  REFL_STREAM();
  float au[NIMG][16];
  float* const gz_rows = tp.GZ_ROWS != nullptr ? tp.GZ_ROWS + (size_t)blockIdx.y * P * rd.z_feats : nullptr;

  for (long unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    call = 0;
    RT_STAMP(0)
    // top of the stack, what happens to a finished d / d z tile: d_top = (gz + the incoming rows) act'(z) -> the encoder backward (when
    // this launch walks the encoder), or d / d z rows (when it does not)
    auto top_tile = [&](const int ot, const int im, auto slc_, float (&v)[16]) __attribute__((always_inline)) {
      constexpr int sl_ = decltype(slc_)::value; (void)sl_;
      const long ptile = ptile_of(unit, im);
      const long pt = (ptile << 5) + p;
      const bool valid = pt < P;
      if (valid && blockIdx.y == 0)                // (one workgroup row per head: the incoming rows go into ONE of the slices that are summed later)
        for (int j = 0; j < tp.n_gz; ++j) {
          const float* row = tp.G_Z[j] + pt * (long)rd.z_feats + 32 * ot + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (32 * ot + 8 * q + 4 * h < rd.z_feats) {
              const f32x4 g = *reinterpret_cast<const f32x4*>(row + 8 * q);
              v[4 * q] += g[0]; v[4 * q + 1] += g[1]; v[4 * q + 2] += g[2]; v[4 * q + 3] += g[3];
            }
        }
      if (!valid) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0.f;
      }
      if (nE > 0) {
        if (ptile < n_tiles) {
          float y[16];
          tf_load_acc(tp.Y[nE - 1], ptile, ZT, ot, lane, y);
          dact_mul(rd.act[nE - 1], y, v);
          tf_store_acc(tp.D[nE - 1], ptile, ZT, ot, lane, v);
        }
        store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
      } else if (valid) {
        float* row = gz_rows + pt * (long)rd.z_feats + 32 * ot + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (32 * ot + 8 * q + 4 * h < rd.z_feats)
            *reinterpret_cast<f32x4*>(row + 8 * q) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      }
    };

    if (heads)
    for (int k = k_lo; k < k_hi; ++k) {
      const int cc = rd.c[k], T0 = rd.t0[k], T1 = rd.t1[k];
      // ---------------- delta_2 = g_out out (1 - out)  [c values per point] ----------------
      if (tid < 32 * NIMG) {
        const int im = tid >> 5, t = tid & 31;
        const long ptile = ptile_of(unit, im), pt = (ptile << 5) + t;
        float d[3] = {0.f, 0.f, 0.f};
        if (pt < P)
          for (int c = 0; c < cc; ++c) {
            const float y = tp.OUT[k][pt * cc + c];
            d[c] = tp.G_OUT[k][pt * cc + c] * y * (1.f - y);
          }
        sm->d2[im][t * 3 + 0] = d[0]; sm->d2[im][t * 3 + 1] = d[1]; sm->d2[im][t * 3 + 2] = d[2];
        if (ptile < n_tiles) {
          float* base = tp.D2[k] + ptile * 1024 + t;
          if (tp.d2_shared) {
            for (int f = 0; f < cc; ++f) base[(tp.d2_row0[k] + f) * 32] = d[f];
          } else {
            for (int f = 0; f < 32; ++f) base[f * 32] = f < cc ? d[f < 3 ? f : 0] : 0.f;
          }
        }
      }
      __syncthreads();
      RT_STAMP(1)
      float dd[NIMG][3];
#pragma unroll
      for (int im = 0; im < NIMG; ++im) { dd[im][0] = sm->d2[im][p * 3 + 0]; dd[im][1] = sm->d2[im][p * 3 + 1]; dd[im][2] = sm->d2[im][p * 3 + 2]; }
      // acc += sum_c img_c[tile] d2[slot][c]  (the last layer's transpose as rank-c updates on the vector ALU)
      auto rank_c = [&](const f32x4* imgs, const int tiles, const int ot, const int im, f32x16& acc) __attribute__((always_inline)) {
        for (int c = 0; c < cc; ++c) {
          f32x16 wv;
          init_bias_f16s(imgs + (size_t)c * tiles * 8, ot, lane, wv);
          float d;
          if constexpr (NIMG == 1) d = c == 0 ? dd[0][0] : (c == 1 ? dd[0][1] : dd[0][2]);
          else d = im == 0 ? (c == 0 ? dd[0][0] : (c == 1 ? dd[0][1] : dd[0][2])) : (c == 0 ? dd[NIMG - 1][0] : (c == 1 ? dd[NIMG - 1][1] : dd[NIMG - 1][2]));
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = fmaf(wv[i], d, acc[i]);
        }
      };
      // ---------------- delta_1 = (W2[:w1] delta_2) relu'(y1) -> X ----------------
      owned(T1, [&](int ot, int im, auto slc_) __attribute__((always_inline)) {
        constexpr int sl_ = decltype(slc_)::value; (void)sl_;
        f32x16 acc;
        init_zero(acc);
        rank_c(wf + rd.offA2y[k], T1, ot, im, acc);
        float v[16], y[16];
        const long ptile = ptile_of(unit, im);
        if (ptile < n_tiles) tf_load_acc(tp.H1[k], ptile, T1, ot, lane, y);
        else {
#pragma unroll
          for (int i = 0; i < 16; ++i) y[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = y[i] > 0.f ? acc[i] : 0.f;
        if (ptile < n_tiles) tf_store_acc(tp.D1[k], ptile, T1, ot, lane, v);
        store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
      });
      __syncthreads();
      RT_STAMP(2)
      // ---------------- delta_0 = (W1 delta_1) relu'(y0) ----------------
      {
        const float* const t_y = tp.H0[k];
        float* const t_d = tp.D0[k];
        GC(rd.offW1b[k], KSegs{X0, 6 * T1, 0, 0}, T0,
          [&](int ot, int im, auto slc_, f32x16& acc) __attribute__((always_inline)) {
            constexpr int sl_ = decltype(slc_)::value; (void)sl_;
            if (sl_ == 0) RT_STAMP(3)
            if (ptile_of(unit, im) < n_tiles) tf_load_acc(t_y, ptile_of(unit, im), T0, ot, lane, au[sl_]);
            else {
#pragma unroll
              for (int i = 0; i < 16; ++i) au[sl_][i] = 0.f;
            }
            init_zero(acc);
            RT_STAMP(4)
          },
          [&](int ot, int im, auto slc_, const f32x16& acc) __attribute__((always_inline)) {
            constexpr int sl_ = decltype(slc_)::value; (void)sl_;
            if (sl_ == 0) RT_STAMP(5)
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = au[sl_][i] > 0.f ? acc[i] : 0.f;
            if (ptile_of(unit, im) < n_tiles) tf_store_acc(t_d, ptile_of(unit, im), T0, ot, lane, v);
            store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
            RT_STAMP(6)
          });
        RT_STAMP(7)
      }
      // ---------------- d / d z += W0 delta_0 + W2[w1:] delta_2 ----------------
      {
        const bool first = k == k_lo, last = k == k_hi - 1;
        GC(rd.offW0b[k], KSegs{X0, 6 * T0, 0, 0}, ZT,
          [&](int ot, int im, auto slc_, f32x16& acc) __attribute__((always_inline)) {
            constexpr int sl_ = decltype(slc_)::value; (void)sl_;
            if (sl_ == 0) RT_STAMP(3)
            if (first) init_zero(acc);
            else {
              const f32x4* sv = save0 + (size_t)im * per_img;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 t = ld_stream(sv + (ot * 4 + q) * 64 + lane);
                acc[4 * q] = t[0]; acc[4 * q + 1] = t[1]; acc[4 * q + 2] = t[2]; acc[4 * q + 3] = t[3];
              }
            }
            rank_c(wf + rd.offA2z[k], ZT, ot, im, acc);
            RT_STAMP(4)
          },
          [&](int ot, int im, auto slc_, const f32x16& acc) __attribute__((always_inline)) {
            constexpr int sl_ = decltype(slc_)::value; (void)sl_;
            if (sl_ == 0) RT_STAMP(5)
            if (!last) {
              f32x4* sv = save0 + (size_t)im * per_img;
#pragma unroll
              for (int q = 0; q < 4; ++q) st_stream(sv + (ot * 4 + q) * 64 + lane, (f32x4){acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
            } else {
              float v[16];
#pragma unroll
              for (int i = 0; i < 16; ++i) v[i] = acc[i];
              top_tile(ot, im, slc_, v);
            }
            RT_STAMP(6)
          });
        RT_STAMP(7)
      }
    }
    if (!heads && nE > 0) {                                    // the encoder part alone: d_top from the incoming rows
      owned(ZT, [&](int ot, int im, auto slc_) __attribute__((always_inline)) {
        constexpr int sl_ = decltype(slc_)::value; (void)sl_;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = 0.f;
        top_tile(ot, im, slc_, v);
      });
      __syncthreads();
    }
    // ---------------- encoder layers, top down: delta_{l-1} = (W_l[y part] delta_l) act'_{l-1}(y_{l-1}) ----------------
    for (int l = nE - 1; l >= 1; --l) {
      const int n_ot = rd.te[l - 1], act = rd.act[l - 1];
      const float* const t_y = tp.Y[l - 1];
      float* const t_d = tp.D[l - 1];
      GC(rd.offWb[l], KSegs{X0, 6 * rd.te[l], 0, 0}, n_ot,
        [&](int ot, int im, auto slc_, f32x16& acc) __attribute__((always_inline)) {
            constexpr int sl_ = decltype(slc_)::value; (void)sl_;
          if (sl_ == 0) RT_STAMP(3)
          if (ptile_of(unit, im) < n_tiles) tf_load_acc(t_y, ptile_of(unit, im), n_ot, ot, lane, au[sl_]);
          else {
#pragma unroll
            for (int i = 0; i < 16; ++i) au[sl_][i] = 0.f;
          }
          init_zero(acc);
          RT_STAMP(4)
        },
        [&](int ot, int im, auto slc_, const f32x16& acc) __attribute__((always_inline)) {
            constexpr int sl_ = decltype(slc_)::value; (void)sl_;
          if (sl_ == 0) RT_STAMP(5)
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = acc[i];
          dact_mul(act, au[sl_], v);
          if (ptile_of(unit, im) < n_tiles) tf_store_acc(t_d, ptile_of(unit, im), n_ot, ot, lane, v);
          store_tile_x3(lds + (size_t)im * IS, X0 + 6 * ot, lane, v);
          RT_STAMP(6)
        });
      RT_STAMP(7)
    }
  }
  RT_STAMP(0)
  RT_STAMP_FLUSH
}

size_t lds_bytes_r(int MT, int nimg, int zx_tiles = 0) { return (size_t)nimg * (E_ROWS + 6 * MT + 6 * zx_tiles) * 1024 + sizeof(SmallsR); }

int load_desc_r(const int32_t* desc, ReflDesc& rd) {
  memcpy(&rd, desc, sizeof(ReflDesc));
  if (rd.n_enc < 0 || rd.n_enc > RT_MAX_L || rd.n_heads < 0 || rd.n_heads > RT_MAX_H || (rd.n_enc == 0 && rd.n_heads == 0)) return 1;
  if (rd.max_tiles < 1 || rd.max_tiles > NW || lds_bytes_r(rd.max_tiles, 2) > 160 * 1024) return 2;
  if (rd.z_tiles < 1 || rd.z_tiles > rd.max_tiles || rd.z_feats < 1 || rd.z_feats > 32 * rd.z_tiles || (rd.z_feats & 3)) return 3;
  if (rd.n_enc > 0) {
    if (rd.emb_feats < 3 || rd.emb_feats > 64 || rd.emb_rows != x3_rows(rd.emb_feats) || rd.emb_rows > E_ROWS || rd.e_tiles < 1 || rd.e_tiles > 2 ||
        2 * rd.e_tiles * 3 < rd.emb_rows) return 4;
    if (rd.skip < 0 || rd.skip >= rd.n_enc) return 5;
    for (int l = 0; l < rd.n_enc; ++l)
      if (rd.te[l] < 1 || rd.te[l] > rd.max_tiles || rd.act[l] < 0 || rd.act[l] > 3 || rd.act[l] == ACT_SOFTPLUS100) return 6;
    if (rd.te[rd.n_enc - 1] != rd.z_tiles) return 7;
  }
  for (int k = 0; k < rd.n_heads; ++k)
    if (rd.t0[k] < 1 || rd.t0[k] > rd.max_tiles || rd.t1[k] < 1 || rd.t1[k] > rd.max_tiles || rd.c[k] < 1 || rd.c[k] > 3) return 8;
  // a second head input: as wide as z, heads present, and one image (two regions) must fit the CU's LDS
  if (rd.zx_tiles != 0 && (rd.zx_tiles != rd.z_tiles || rd.n_heads < 1 || lds_bytes_r(rd.max_tiles, 1, rd.zx_tiles) > 160 * 1024)) return 9;
  return 0;
}

// one image per workgroup when the two-image form would leave more than half of the CUs without a tile pair (VQN_REFL_NIMG = 1 | 2
// forces a form: tests and A/B timings)
int images_per_wg(long n_tiles, const ReflDesc& rd) {
  if (rd.zx_tiles > 0) return 1;                                 // (two image regions per point tile: one image per workgroup)
  const char* e = getenv("VQN_REFL_NIMG");                       // (read per call: the tests switch it)
  if (e && (e[0] == '1' || e[0] == '2')) return e[0] - '0';
  return (n_tiles + 1) / 2 * 2 <= (long)vqn_num_cus() ? 1 : 2;
}

}  // namespace

extern "C" int vqn_refl_train_desc_ints(void) { return RT_DESC_INTS; }

extern "C" int vqn_refl_train_fwd_x3(const int32_t* desc, const void* wbuf_pieces, const float* wbuf_f32, const float* pts, const float* z_rows,
                                     int64_t P, float* const* saved, int n_saved, float* z_rows_out, float* const* head_out, int split_heads,
                                     int save_tensors, void* stream) {
  return vqn_refl_train_fwd_x3_zx(desc, wbuf_pieces, wbuf_f32, pts, z_rows, nullptr, P, saved, n_saved, z_rows_out, nullptr, head_out, split_heads,
                                  save_tensors, stream);
}

extern "C" int vqn_refl_train_fwd_x3_zx(const int32_t* desc, const void* wbuf_pieces, const float* wbuf_f32, const float* pts, const float* z_rows,
                                        const float* zx_rows, int64_t P, float* const* saved, int n_saved, float* z_rows_out, float* zx_tiles_out,
                                        float* const* head_out, int split_heads, int save_tensors, void* stream) {
  VQN_CHECK_ARG(desc && wbuf_pieces && wbuf_f32 && saved, "null pointer");
  VQN_CHECK_ARG(P >= 1, "P >= 1");
  ReflDesc rd;
  VQN_CHECK_SHAPE(load_desc_r(desc, rd) == 0, "invalid reflectance stack descriptor (layers of at most 256 outputs, at most 8 encoder layers, 3 heads)");
  const int nE = rd.n_enc, nH = rd.n_heads;
  VQN_CHECK_ARG((nE > 0) ? (pts != nullptr) : (z_rows != nullptr), "pts with an encoder, z_rows without");
  // saved: with an encoder [E, Y_0..Y_{nE-1}], without [ZT]; then per head [H0, H1]
  VQN_CHECK_ARG(n_saved == (nE > 0 ? 1 + nE : 1) + 2 * nH, "saved: [E, Y_0..] | [ZT], then [H0_k, H1_k] per head");
  VQN_CHECK_ARG(nH == 0 || head_out != nullptr, "head_out");
  // (save_tensors = 0, inference: only z's tile-format tensor -- Y_{n_enc-1} / ZT -- is written and must be given)
  for (int i = 0; i < n_saved; ++i)
    VQN_CHECK_ARG(saved[i] != nullptr || (!save_tensors && i != (nE > 0 ? nE : 0)), "null saved tensor");
  VQN_CHECK_ARG((rd.zx_tiles > 0) == (zx_rows != nullptr), "zx_rows exactly when the descriptor has a second head input");
  VQN_CHECK_SHAPE(zx_rows == nullptr || ((uintptr_t)zx_rows & 15) == 0, "zx_rows must be 16-byte aligned");
  ReflFwdPtrs tp;
  memset(&tp, 0, sizeof(tp));
  tp.X = pts; tp.ZR = z_rows; tp.ZROWS = z_rows_out; tp.save = save_tensors != 0;
  tp.ZX = zx_rows; tp.ZXT = zx_tiles_out;
  int s = 0;
  if (nE > 0) {
    tp.E = saved[s++];
    for (int l = 0; l < nE; ++l) tp.Y[l] = saved[s++];
    tp.ZT = tp.Y[nE - 1];
  } else tp.ZT = saved[s++];
  for (int k = 0; k < nH; ++k) {
    tp.H0[k] = saved[s++]; tp.H1[k] = saved[s++];
    VQN_CHECK_ARG(head_out[k] != nullptr, "null head output");
    tp.OUT[k] = head_out[k];
  }
  const long n_tiles = (P + 31) / 32;
  const int nimg = images_per_wg(n_tiles, rd);
  const size_t lds = lds_bytes_r(rd.max_tiles, nimg, rd.zx_tiles);
  auto kern = nimg == 1 ? refl_train_fwd_x3_kernel<1> : refl_train_fwd_x3_kernel<2>;
  VQN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long units = nimg == 1 ? n_tiles : (n_tiles + 1) / 2;
  long grid = (long)vqn_num_cus() * RT_WGS_PER_CU(nimg);
  if (grid > units) grid = units;
  const unsigned gy = (split_heads && nH > 1) ? (unsigned)nH : 1u;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, gy), dim3(512), lds, (hipStream_t)stream, rd, reinterpret_cast<const f32x4*>(wbuf_pieces),
                     reinterpret_cast<const f32x4*>(wbuf_f32), tp, (long)P);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int64_t vqn_refl_train_bwd_x3_scratch_bytes(const int32_t* desc) {
  if (!desc) return -1;
  ReflDesc rd;
  if (load_desc_r(desc, rd) != 0) return -1;
  return (int64_t)vqn_num_cus() * RT_MAX_H * 2 * 4 * rd.z_tiles * 1024;      // (covers two one-image workgroups per CU as well)
}

extern "C" int vqn_refl_train_bwd_x3(const int32_t* desc, const void* wbuf_pieces, const float* wbuf_f32, int64_t P, const float* const* g_out,
                                     const float* const* head_out, const float* const* g_z_rows, int n_gz, const float* const* saved, int n_saved,
                                     float* const* outs, int n_outs, float* gz_rows_out, const int32_t* d2_row0, int run_heads, int run_enc,
                                     int split_heads, void* scratch, int64_t scratch_bytes, void* stream) {
  VQN_CHECK_ARG(desc && wbuf_pieces && wbuf_f32 && saved && outs && scratch, "null pointer");
  VQN_CHECK_ARG(P >= 1, "P >= 1");
  ReflDesc rd;
  VQN_CHECK_SHAPE(load_desc_r(desc, rd) == 0, "invalid reflectance stack descriptor (layers of at most 256 outputs, at most 8 encoder layers, 3 heads)");
  const int nE = rd.n_enc, nH = rd.n_heads;
  run_heads = run_heads && nH > 0;
  run_enc = run_enc && nE > 0;
  VQN_CHECK_ARG(run_heads || run_enc, "nothing to run");
  // saved: [Y_0..Y_{nE-1}], then per head [H0, H1]; outs: [D_0..D_{nE-1}], then per head [D0, D1, D2]
  VQN_CHECK_ARG(n_saved == nE + 2 * nH, "saved: [Y_0..Y_{nE-1}], then [H0_k, H1_k] per head");
  VQN_CHECK_ARG(n_outs == nE + 3 * nH, "outs: [D_0..D_{nE-1}], then [D0_k, D1_k, D2_k] per head");
  VQN_CHECK_ARG(!run_heads || (g_out && head_out), "g_out / head_out");
  VQN_CHECK_ARG(n_gz >= 0 && n_gz <= 4 && (n_gz == 0 || g_z_rows != nullptr), "at most four incoming z-adjoint row tensors");
  VQN_CHECK_ARG(run_enc || gz_rows_out != nullptr, "gz_rows_out when the encoder part is not walked");
  VQN_CHECK_ARG(run_heads || n_gz > 0, "the encoder part alone needs incoming z adjoints");
  const bool split = split_heads && run_heads && nH > 1;
  VQN_CHECK_ARG(!(split && run_enc), "heads split over workgroup rows cannot continue into the encoder in the same launch");
  for (int i = 0; i < n_saved; ++i) VQN_CHECK_ARG(saved[i] != nullptr, "null saved tensor");
  for (int i = 0; i < n_outs; ++i) VQN_CHECK_ARG(outs[i] != nullptr, "null output tensor");
  ReflBwdPtrs tp;
  memset(&tp, 0, sizeof(tp));
  tp.n_gz = n_gz; tp.run_heads = run_heads; tp.run_enc = run_enc; tp.GZ_ROWS = gz_rows_out;
  tp.d2_shared = d2_row0 != nullptr;
  for (int k = 0; k < nH && d2_row0 != nullptr; ++k) {
    VQN_CHECK_ARG(d2_row0[k] >= 0 && d2_row0[k] + rd.c[k] <= 32, "d2_row0 outside the tile");
    tp.d2_row0[k] = d2_row0[k];
  }
  for (int j = 0; j < n_gz; ++j) { VQN_CHECK_ARG(g_z_rows[j] != nullptr, "null z adjoint"); tp.G_Z[j] = g_z_rows[j]; }
  for (int l = 0; l < nE; ++l) { tp.Y[l] = saved[l]; tp.D[l] = outs[l]; }
  for (int k = 0; k < nH; ++k) {
    if (run_heads) { VQN_CHECK_ARG(g_out[k] && head_out[k], "null head adjoint / output"); tp.G_OUT[k] = g_out[k]; tp.OUT[k] = head_out[k]; }
    tp.H0[k] = saved[nE + 2 * k]; tp.H1[k] = saved[nE + 2 * k + 1];
    tp.D0[k] = outs[nE + 3 * k]; tp.D1[k] = outs[nE + 3 * k + 1]; tp.D2[k] = outs[nE + 3 * k + 2];
  }
  const long n_tiles = (P + 31) / 32;
  const int nimg = images_per_wg(n_tiles, rd);
  const unsigned gy = split ? (unsigned)nH : 1u;
  const int64_t per_wg = (int64_t)nimg * 4 * rd.z_tiles * 1024;
  const size_t lds = lds_bytes_r(rd.max_tiles, nimg, rd.zx_tiles);
  auto kern = nimg == 1 ? refl_train_bwd_x3_kernel<1> : refl_train_bwd_x3_kernel<2>;
  VQN_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long units = nimg == 1 ? n_tiles : (n_tiles + 1) / 2;
  long grid = (long)vqn_num_cus() * RT_WGS_PER_CU(nimg);
  if (grid > units) grid = units;
  if ((int64_t)grid * gy * per_wg > scratch_bytes) grid = (long)(scratch_bytes / (per_wg * gy));
  VQN_CHECK_ARG(grid >= 1, "scratch too small (see vqn_refl_train_bwd_x3_scratch_bytes)");
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, gy), dim3(512), lds, (hipStream_t)stream, rd, reinterpret_cast<const f32x4*>(wbuf_pieces),
                     reinterpret_cast<const f32x4*>(wbuf_f32), tp, (long)P, reinterpret_cast<f32x4*>(scratch));
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

#ifdef VQN_RT_STAMPS
extern "C" int vqn_debug_read_rt_stamps(unsigned long long* out, int reset) {
  VQN_HIP(hipDeviceSynchronize());
  VQN_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rt_stamps), sizeof(unsigned long long) * 128));
  if (reset) {
    unsigned long long z[128] = {0};
    VQN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rt_stamps), z, sizeof(z)));
  }
  return VQN_OK;
}
#endif
