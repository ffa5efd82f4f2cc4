// Device primitives of the fused small-MLP engine (gfx950, wave64, fp32 MFMA).  Not a public header.
//
// Data layout ("activation image").  A workgroup (4 waves) owns a tile of 32 points.  Activations
// live in LDS as rows of 64 float4 (1 KB):  row (t, rq), lane (p = lane & 31, h = lane >> 5),
// component j  holds feature  32 t + 2 (4 rq + j) + h  of point p.  That is exactly
//   * the B operand of v_mfma_f32_32x32x2_f32 for K-step r = 4 rq + j (lane (p,h) supplies
//     B[k = h][n = p]), fetched for 4 consecutive K-steps by ONE conflict-free ds_read_b128, and
//   * the accumulator layout of the same instruction when the 32 output rows of a tile are
//     ordered by  phi(i) = 2 (i & 3) + 8 (i >> 3) + ((i >> 2) & 1)  (done by the host-side weight
//     packer), so an output tile is written back with four ds_write_b128 and no shuffles.
// Weights are packed on the host as A fragments: pack[out_tile][k_group][lane][j] =
//   W[out = 32 ot + phi(lane & 31)][in = feature(k_group, j, lane >> 5)]   (zero where padded),
// so each lane fetches its A operands for 4 MFMAs with one coalesced global_load_dwordx4 (the
// wave reads 1 KB contiguous); the packs stay L2-resident (3 MB for the NeuS nets).
#pragma once
#include "common.h"

namespace eng {

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SOFTPLUS100 = 2, ACT_SIGMOID = 3 };

__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

template <int ACT>
__device__ __forceinline__ float act_fwd(float x) {
  if (ACT == ACT_RELU) return fmaxf(x, 0.f);
  if (ACT == ACT_SOFTPLUS100) {           // nn.Softplus(beta=100, threshold=20)  (fields.py:70)
    const float t = 100.f * x;
    return t > 20.f ? x : 0.01f * fast_log(1.f + fast_exp(t));
  }
  if (ACT == ACT_SIGMOID) return fast_rcp(1.f + fast_exp(-x));
  return x;
}

// derivative of the activation expressed through its OUTPUT h
template <int ACT>
__device__ __forceinline__ float act_bwd_from_out(float h) {
  if (ACT == ACT_RELU) return h > 0.f ? 1.f : 0.f;
  if (ACT == ACT_SOFTPLUS100) return 1.f - fast_exp(-100.f * h);   // sigmoid(100 x) = 1 - exp(-100 softplus(x))
  if (ACT == ACT_SIGMOID) return h * (1.f - h);
  return 1.f;
}

struct KSegs {   // the K dimension of a layer = up to two runs of consecutive LDS rows
  int rowA, nA, rowB, nB;
};

// out[32 x 32-point tile `ot`] = sum over K rows;  wave w owns tiles w, w+4, ...
template <int NW = 4, class Init, class Epi>
__device__ __forceinline__ void gemm_tiles(const f32x4* __restrict__ lds, const KSegs ks,
                                           const f32x4* __restrict__ w, const int n_out_tiles, const int wave,
                                           const int lane, Init init, Epi epi) {
  const int ng = ks.nA + ks.nB;
  for (int ot = wave; ot < n_out_tiles; ot += NW) {
    f32x16 acc;
    const f32x4* __restrict__ wp = w + (size_t)ot * ng * 64 + lane;
    auto brow = [&](int g) { return ((g < ks.nA) ? (ks.rowA + g) : (ks.rowB + (g - ks.nA))) * 64 + lane; };
    // The weight fragments (global, L2 latency) run four K groups ahead in two register buffers; the activation rows (LDS,
    // short latency) ONE group ahead in a float4 pair: fetching them four groups ahead as well costs 24 more registers and
    // measured 3 % slower in the training programs.  Operand fetches are UNCONDITIONAL (group index clamped to the last one; a
    // clamped fetch is never multiplied) and the vector-memory counter is drained once before the loop: every path through
    // the loop then carries the same number of outstanding loads, and the compiler's s_waitcnt before a block's MFMAs is the
    // exact distance to that block's operands.  With guarded fetches it falls back to vmcnt(0) lgkmcnt(0) after issuing the
    // NEXT block's fetches, i.e. no overlap of fetch and matrix work inside a wave at all.
    f32x4 a0[4], a1[4], bc, bn;
    auto browc = [&](int g) { return brow(min(g, ng - 1)); };
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] = wp[min(i, ng - 1) * 64];
    bc = lds[browc(0)];
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) only
    init(ot, acc);                             // after the drain: loads issued here (stashed activations ...) land under the K loop
    __builtin_amdgcn_s_setprio(1);
    auto block = [&](const f32x4 (&a)[4], int g0) {          // groups g0 .. g0 + 3 with fragments a
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bn = lds[browc(g0 + i + 1)];
        if (g0 + i < ng) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], bc[j], acc, 0, 0, 0);
        }
        bc = bn;
      }
    };
    for (int g = 0; g < ng; g += 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a1[i] = wp[min(g + 4 + i, ng - 1) * 64];
      block(a0, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) a0[i] = wp[min(g + 8 + i, ng - 1) * 64];
      block(a1, g + 4);
    }
    __builtin_amdgcn_s_setprio(0);
    epi(ot, acc);
  }
}

// gemm_tiles with the epilogue of a wave's tile A issued inside the K loop of its tile B (software pipelining over the wave's
// output tiles; a wave owns two at 256 outputs and 4 waves).  init(ot, slot, acc) / epi_rq(ot, slot, rq, acc): `slot` (0 / 1,
// compile-time at every call) tells the caller which of two sets of per-tile epilogue operands to use -- tile B's are fetched
// while tile A's are still needed.  Same arithmetic and order as gemm_tiles.
// aux(ot, slot): issue the loads of a tile's epilogue operands.  Tile B's are issued together with tile A's, a whole K loop before
// they are used (the vector-memory counter is in order: a load requested at the start of its own tile's K loop is waited for at that
// loop's first weight-fragment wait).  NB: weight-fragment buffers of four K groups each; fragments run NB - 1 blocks ahead of the
// multiplies (NB = 2: the scheme of gemm_tiles).
template <int NW = 4, int NB = 2, class Aux, class Init, class EpiRq>
__device__ __forceinline__ void gemm_tiles_sw(const f32x4* __restrict__ lds, const KSegs ks, const f32x4* __restrict__ w,
                                              const int n_out_tiles, const int wave, const int lane, Aux aux, Init init, EpiRq epi_rq) {
  const int ng = ks.nA + ks.nB;
  auto brow = [&](int g) { g = min(g, ng - 1); return ((g < ks.nA) ? (ks.rowA + g) : (ks.rowB + (g - ks.nA))) * 64 + lane; };
  for (int base = wave; base < n_out_tiles; base += 2 * NW) {
    f32x16 accA, accB;
    f32x4 a[NB][4], bc, bn;
    auto group = [&](const f32x4& af, f32x16& acc) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bc[j], acc, 0, 0, 0);
    };
    auto block = [&](const f32x4 (&af)[4], int g0, f32x16& acc) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bn = lds[brow(g0 + i + 1)];
        if (g0 + i < ng) group(af[i], acc);
        bc = bn;
      }
    };
    auto fill = [&](const f32x4* __restrict__ wp, f32x4 (&af)[4], int g0) {          // unconditional, clamped (see gemm_tiles)
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = wp[min(g0 + i, ng - 1) * 64];
    };
    // one round of NB blocks starting at group g: block b multiplies buffer b while the buffer block b - 1 used is refilled
    auto round = [&](const f32x4* __restrict__ wp, int g, int b_first, f32x16& acc) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b < b_first) continue;
        fill(wp, a[(b + NB - 1) % NB], g + 4 * (b + NB - 1));
        block(a[b], g + 4 * b, acc);
      }
    };
    // ---- tile A
    {
      const f32x4* __restrict__ wp = w + (size_t)base * ng * 64 + lane;
#pragma unroll
      for (int b = 0; b < NB - 1; ++b) fill(wp, a[b], 4 * b);
      bc = lds[brow(0)];
      __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) only (see gemm_tiles)
      aux(base, 0);
      if (base + NW < n_out_tiles) aux(base + NW, 1);
      init(base, 0, accA);
      __builtin_amdgcn_s_setprio(1);
      for (int g = 0; g < ng; g += 4 * NB) round(wp, g, 0, accA);
      __builtin_amdgcn_s_setprio(0);
    }
    const int otB = base + NW;
    if (otB >= n_out_tiles) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) epi_rq(base, 0, rq, accA);
      break;
    }
    // ---- tile B: its first four K groups carry tile A's epilogue quads
    {
      const f32x4* __restrict__ wp = w + (size_t)otB * ng * 64 + lane;
#pragma unroll
      for (int b = 0; b < NB - 1; ++b) fill(wp, a[b], 4 * b);
      bc = lds[brow(0)];
      __builtin_amdgcn_s_waitcnt(0x0F70);
      init(otB, 1, accB);
      __builtin_amdgcn_s_setprio(1);
      fill(wp, a[NB - 1], 4 * (NB - 1));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bn = lds[brow(i + 1)];
        if (i < ng) group(a[0][i], accB);
        bc = bn;
        epi_rq(base, 0, i, accA);
      }
      round(wp, 0, 1, accB);
      for (int g = 4 * NB; g < ng; g += 4 * NB) round(wp, g, 0, accB);
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) epi_rq(otB, 1, rq, accB);
    }
  }
}

// gemm_tiles for a chain of layers: `pre` holds the first four weight fragments of this wave's FIRST tile (loaded by the
// previous layer before its epilogue and barrier, so their L2 latency is not paid after the barrier); before the epilogue
// of its LAST tile the function refills `pre` from `next_wp` (the same fragments of the next layer; nullptr = none).
template <int NW = 4, class Init, class Epi>
__device__ __forceinline__ void gemm_tiles_chain(const f32x4* __restrict__ lds, const KSegs ks,
                                                 const f32x4* __restrict__ w, const int n_out_tiles, const int wave,
                                                 const int lane, f32x4 (&pre)[4], const f32x4* __restrict__ next_wp,
                                                 Init init, Epi epi) {
  const int ng = ks.nA + ks.nB;
  for (int ot = wave; ot < n_out_tiles; ot += NW) {
    f32x16 acc;
    const f32x4* __restrict__ wp = w + (size_t)ot * ng * 64 + lane;
    auto brow = [&](int g) { return ((g < ks.nA) ? (ks.rowA + g) : (ks.rowB + (g - ks.nA))) * 64 + lane; };
    const bool first = ot == wave;
    // (operand pipeline, unconditional clamped fetches and the one drain per tile: see gemm_tiles)
    f32x4 a0[4], a1[4], bc, bn;
    auto browc = [&](int g) { return brow(min(g, ng - 1)); };
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (first) a0[i] = pre[i]; else a0[i] = wp[min(i, ng - 1) * 64]; }
    bc = lds[browc(0)];
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) only
    init(ot, acc);                             // after the drain: loads issued here (stashed activations ...) land under the K loop
    __builtin_amdgcn_s_setprio(1);
    auto block = [&](const f32x4 (&a)[4], int g0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bn = lds[browc(g0 + i + 1)];
        if (g0 + i < ng) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], bc[j], acc, 0, 0, 0);
        }
        bc = bn;
      }
    };
    for (int g = 0; g < ng; g += 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a1[i] = wp[min(g + 4 + i, ng - 1) * 64];
      block(a0, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) a0[i] = wp[min(g + 8 + i, ng - 1) * 64];
      block(a1, g + 4);
    }
    __builtin_amdgcn_s_setprio(0);
    if (ot + NW >= n_out_tiles && next_wp != nullptr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) pre[i] = next_wp[i * 64];
    }
    epi(ot, acc);
  }
  if (wave >= n_out_tiles && next_wp != nullptr) {      // a wave without a tile in this layer still owes the next layer its fragments
#pragma unroll
    for (int i = 0; i < 4; ++i) pre[i] = next_wp[i * 64];
  }
}

// Two-image form of gemm_tiles / gemm_tiles_chain: the workgroup (8 waves) holds TWO 32-point images `img_stride` float4
// apart and a wave applies each weight fragment to both (8 MFMAs per fetched float4).  One workgroup per CU whose waves all
// do the same amount of matrix work between two barriers: no cross-workgroup interference on the matrix pipe, half the
// barriers and half the L2 weight stream per point.  init(ot, img, acc) runs once per image, epi_rq(ot, img, rq, acc) once per
// image and row quad.  `pre` (use_pre): the chain prefetch of gemm_tiles_chain.
// Operand pipeline: the weight fragments (L2 latency) keep two 4-group register buffers; the activation rows of both images
// (LDS latency, ~1/4 of one group's matrix time) are fetched ONE group ahead into two float4 pairs.  Fetching them four
// groups ahead like the weights cost 48 more registers per wave, which put the fine kernel at the 256-VGPR limit of two
// waves per SIMD with spills: 231.5 -> 224.5 ms per launch without them.
template <int NW = 8, class Init, class EpiRq>
__device__ __forceinline__ void gemm_tiles2(const f32x4* __restrict__ lds, const int img_stride, const KSegs ks,
                                            const f32x4* __restrict__ w, const int n_out_tiles, const int wave,
                                            const int lane, f32x4 (&pre)[4], const bool use_pre,
                                            const f32x4* __restrict__ next_wp, Init init, EpiRq epi_rq) {
  const int ng = ks.nA + ks.nB;
  for (int ot = wave; ot < n_out_tiles; ot += NW) {
    f32x16 acc0, acc1;
    const f32x4* __restrict__ wp = w + (size_t)ot * ng * 64 + lane;
    auto brow = [&](int g) { g = min(g, ng - 1); return ((g < ks.nA) ? (ks.rowA + g) : (ks.rowB + (g - ks.nA))) * 64 + lane; };
    f32x4 a0[4], a1[4], pc, qc, pn, qn;                       // p: image 0, q: image 1
    const bool first = use_pre && ot == wave;
    // (unconditional, clamped operand fetches + one drain per tile: see gemm_tiles)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (first) a0[i] = pre[i]; else a0[i] = wp[min(i, ng - 1) * 64];
    }
    { const int r = brow(0); pc = lds[r]; qc = lds[r + img_stride]; }
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0) only (see gemm_tiles)
    init(ot, 0, acc0);
    init(ot, 1, acc1);
    __builtin_amdgcn_s_setprio(1);
    auto block = [&](const f32x4 (&a)[4], int g0) {          // groups g0 .. g0 + 3 with fragments a
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        { const int r = brow(g0 + i + 1); pn = lds[r]; qn = lds[r + img_stride]; }
        if (g0 + i < ng) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], pc[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], qc[j], acc1, 0, 0, 0);
          }
        }
        pc = pn; qc = qn;
      }
    };
    if ((ng & 7) == 0) {
      // K a whole number of double blocks (the 256-wide layers): the last block is peeled and image 1's half of it is issued one
      // K group (4 MFMAs) at a time between image 0's epilogue quads, so that quarter of the workgroup's epilogue work runs in
      // the shadow of matrix instructions (all 8 waves reach their epilogues together; nothing else covers them).  +0.9 %.
      // Same accumulation order per image.
      for (int g = 0; g < ng - 8; g += 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[i] = wp[(g + 4 + i) * 64];
        block(a0, g);
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[i] = wp[(g + 8 + i) * 64];
        block(a1, g + 4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) a1[i] = wp[(ng - 4 + i) * 64];
      block(a0, ng - 8);
      if (use_pre && ot + NW >= n_out_tiles && next_wp != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[i] = next_wp[i * 64];
      }
      f32x4 qt[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        { const int r = brow(ng - 4 + i + 1); pn = lds[r]; qn = lds[r + img_stride]; }
        qt[i] = qc;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][j], pc[j], acc0, 0, 0, 0);
        pc = pn; qc = qn;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][j], qt[i][j], acc1, 0, 0, 0);
        epi_rq(ot, 0, i, acc0);
      }
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) epi_rq(ot, 1, rq, acc1);
      continue;
    }
    for (int g = 0; g < ng; g += 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a1[i] = wp[min(g + 4 + i, ng - 1) * 64];
      block(a0, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) a0[i] = wp[min(g + 8 + i, ng - 1) * 64];
      block(a1, g + 4);
    }
    if (use_pre && ot + NW >= n_out_tiles && next_wp != nullptr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) pre[i] = next_wp[i * 64];
    }
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) epi_rq(ot, 0, rq, acc0);
    __builtin_amdgcn_s_setprio(0);             // (image 0's epilogue still at raised priority: measured 0.5 % better)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) epi_rq(ot, 1, rq, acc1);
  }
  if (use_pre && wave >= n_out_tiles && next_wp != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) pre[i] = next_wp[i * 64];
  }
}

__device__ __forceinline__ f32x4 acc_quad(const f32x16& acc, int rq) {
  // static rq only (callers unroll)
  return (f32x4){acc[4 * rq + 0], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]};
}

// acc init from a packed bias: bpack[ot][h][16].  `ot` is wave-uniform (callers take `wave` through readfirstlane), so both
// halves are fetched at uniform addresses -- scalar loads: short latency, and nothing enters the vector-memory counter that
// paces the K loop -- and each lane keeps its half.
__device__ __forceinline__ void init_bias(const f32x4* __restrict__ bpack, int ot, int lane, f32x16& acc) {
  const f32x4* b = bpack + ot * 8;
  const bool hi = lane >= 32;
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const f32x4 v0 = b[rq], v1 = b[4 + rq];
    acc[4 * rq + 0] = hi ? v1[0] : v0[0]; acc[4 * rq + 1] = hi ? v1[1] : v0[1];
    acc[4 * rq + 2] = hi ? v1[2] : v0[2]; acc[4 * rq + 3] = hi ? v1[3] : v0[3];
  }
}

__device__ __forceinline__ void init_zero(f32x16& acc) {
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
}

__device__ __forceinline__ void init_rows(const f32x4* __restrict__ rows /* tile base, 4 rows */, int lane, f32x16& acc) {
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    f32x4 v = rows[rq * 64 + lane];
    acc[4 * rq + 0] = v[0]; acc[4 * rq + 1] = v[1]; acc[4 * rq + 2] = v[2]; acc[4 * rq + 3] = v[3];
  }
}

// positional encoding feature f of a 3-vector (embedder.py:16-34): [x, sin(2^k x), cos(2^k x)]_k
__device__ __forceinline__ float posenc_feat(int f, float x0, float x1, float x2) {
  if (f < 3) return f == 0 ? x0 : (f == 1 ? x1 : x2);
  const int g = f - 3, k = g / 6, m = g - 6 * k, c = m >= 3 ? m - 3 : m;
  const float xc = c == 0 ? x0 : (c == 1 ? x1 : x2);
  const float arg = xc * (float)(1 << k);
  return m < 3 ? sinf(arg) : cosf(arg);
}
// d posenc_feat(f) / d x_c(f); *comp receives c(f)
__device__ __forceinline__ float posenc_jac(int f, float x0, float x1, float x2, int* comp) {
  if (f < 3) { *comp = f; return 1.f; }
  const int g = f - 3, k = g / 6, m = g - 6 * k, c = m >= 3 ? m - 3 : m;
  const float xc = c == 0 ? x0 : (c == 1 ? x1 : x2);
  const float fr = (float)(1 << k), arg = xc * fr;
  *comp = c;
  return m < 3 ? fr * cosf(arg) : -fr * sinf(arg);
}

// feature index held by (row r of a segment, lane half h, component j)
__device__ __forceinline__ int row_feat(int r, int h, int j) { return 32 * (r >> 2) + 2 * (4 * (r & 3) + j) + h; }

// part[(wave*32 + p)*NOUT + o] = this wave's share of sum_f wimg[o][f] * act[f][p] over rows
// [row0, row0+n_rows); the caller adds the four partials in a fixed order (deterministic).
// The weight rows come from L2: they are fetched a chunk of rows at a time, all fetches of a chunk (unconditional, clamped)
// before its first use, so a chunk pays one round trip instead of one per row; the sum order is the plain row order.
template <int NOUT, int NW = 4>
__device__ __forceinline__ void rowdot(const f32x4* __restrict__ lds, int row0, int n_rows,
                                       const f32x4* __restrict__ wimg /* [NOUT][n_rows][2] float4 */, float* out_s,
                                       int wave, int lane) {
  constexpr int CH = NOUT == 1 ? 8 : (NOUT <= 2 ? 4 : 3);
  float s[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) s[o] = 0.f;
  const int h = lane >> 5;
  for (int r0 = wave; r0 < n_rows; r0 += NW * CH) {
    f32x4 b[CH], wv[CH][NOUT];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int r = min(r0 + NW * c, n_rows - 1);
      b[c] = lds[(row0 + r) * 64 + lane];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) wv[c][o] = wimg[(o * n_rows + r) * 2 + h];
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (r0 + NW * c < n_rows) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          s[o] = fmaf(b[c][0], wv[c][o][0], s[o]); s[o] = fmaf(b[c][1], wv[c][o][1], s[o]);
          s[o] = fmaf(b[c][2], wv[c][o][2], s[o]); s[o] = fmaf(b[c][3], wv[c][o][3], s[o]);
        }
      }
  }
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    s[o] += __shfl_xor(s[o], 32);
    if (h == 0) out_s[(wave * 32 + (lane & 31)) * NOUT + o] = s[o];
  }
}

}  // namespace eng
