// Split-precision variant of the fused small-MLP engine (gfx950): every f32 value x travels as two f16 numbers
//   hi = f16(x) (round toward zero),  lo = f16((x - hi) * 2^11)      =>  x = hi + lo * 2^-11  to ~2^-22 |x|
// and a product a*w is taken as  a_hi*w_hi + 2^-11 (a_hi*w_lo + a_lo*w_hi)  on v_mfma_f32_32x32x16_f16 (f32 accumulate;
// the dropped a_lo*w_lo term is 2^-22 relative).  Three f16 MFMAs (3 x 32 cycles) cover what sixteen f32 MFMAs
// (8 x 64 cycles) cover in mlp_prims.h -- 5.3x less matrix-pipe time -- at 2^-21 instead of 2^-24 relative accuracy
// per product.  NOT bit-compatible with the f32 engine: an opt-in mode (results within ~1e-6 relative of it).
//
// Data layout ("split activation image").  Same geometry as mlp_prims.h -- a workgroup owns 32 points, activations
// live in LDS as rows of 64 x 16 B (1 KB), four rows per 32 features -- but a 16-feature K-step `sl` of a segment
// is the row PAIR (2 sl, 2 sl + 1) = (hi, lo): lane (p = lane & 31, h = lane >> 5), half jj = 0..7 of either row holds
// feature  16 sl + 8 (jj >> 2) + 4 h + (jj & 3)  of point p.  That is
//   * the B operand of v_mfma_f32_32x32x16_f16 (lane (n, h) supplies B[k = 8 h + jj][n]) under a fixed permutation of
//     the 16 features of the step, the same on the weight side, and
//   * the accumulator layout of the 32x32 MFMA: register reg of lane (n, h) is output row
//     (reg & 3) + 8 (reg >> 2) + 4 h, so registers 8 s .. 8 s + 7 ARE half-slots jj = 0..7 of step s of the output tile:
//     an output tile is split and written back with four ds_write_b128, no shuffles, no row permutation.
// Weights: pack[out_tile][2 step + part][lane][8 halfs] = part(W[out = 32 ot + (lane & 31)][in = feature(step, lane >> 5, jj)]),
// the same 4 KB per 32x32 block as the f32 packs, so descriptors (offsets in float4 units) keep their meaning.
#pragma once
#include "mlp_prims.h"

namespace eng {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;

__device__ __forceinline__ f16x8 as_h8(const f32x4 v) { return __builtin_bit_cast(f16x8, v); }

__device__ __forceinline__ float pack_rtz(float a, float b) {
  return __builtin_bit_cast(float, __builtin_amdgcn_cvt_pkrtz(a, b));
}

// 8 values -> (hi, lo) fragments
__device__ __forceinline__ void split8(const float (&x)[8], f32x4& hi, f32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const auto hh = __builtin_amdgcn_cvt_pkrtz(x[2 * i], x[2 * i + 1]);
    const float r0 = (x[2 * i] - (float)hh[0]) * LO_SCALE, r1 = (x[2 * i + 1] - (float)hh[1]) * LO_SCALE;
    hi[i] = __builtin_bit_cast(float, hh);
    lo[i] = pack_rtz(r0, r1);
  }
}

__device__ __forceinline__ void join8(const f32x4 hi, const f32x4 lo, float (&x)[8]) {
  const f16x8 h = as_h8(hi), l = as_h8(lo);
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = fmaf((float)l[i], LO_INV, (float)h[i]);
}

// feature held by half-slot jj of lane half h in local K-step sl of a segment
__device__ __forceinline__ int step_feat(int sl, int h, int jj) { return 16 * sl + 8 * (jj >> 2) + 4 * h + (jj & 3); }

// GEMM of the engine: out[32 x 32-point tile `ot`] over the K row pairs of `ks` (row counts even); wave w owns tiles w, w + NW,
// ...; acc1 collects hi*hi (+ the init, e.g. the bias), acc2 the two cross terms (scaled 2^11).
//
// The weight stream is decoupled from the layer structure.  On this engine a K = 256 tile is only
// 16 x 96 matrix-pipe cycles, less than two L2 round trips, so weight fragments must be in flight long before they are
// used: `A` is a ring of R 4-step blocks (R = 2: 128 features of K) that lives across tiles, layers and barriers.  After a
// block is consumed its slot is refilled with the block R positions further down the wave's stream: the same tile,
// then the wave's next tile of this layer, then (`next_wp`, `next_nb`) the wave's first tile of the next GEMM layer --
// issued before this layer's epilogue and barrier.  Everything that touches a memory counter is unconditional and the
// same on every path (eight global loads per block, two LDS reads per step), so the compiler's waitcnt bookkeeping stays
// exact across the loop: only the MFMAs of a padding block are skipped, by a scalar branch.  That needs
//   * packs padded with zero rows to whole blocks per tile (8 rows; a padded step multiplies a clamped, finite activation
//     row by zeros), and
//   * tiles padded to whole groups of R blocks in the ring's slot numbering, so block q of any tile always sits in slot
//     q % R (all register indices static); the refill of a padding position is a dummy re-load of the tile's last block.
// On entry the ring holds the first R blocks of this call's first tile (ring_prime once per kernel, then the chain keeps
// itself primed: the stream wraps to the next point tile).  `wave` must be wave-uniform (readfirstlane).
template <int R>
__device__ __forceinline__ void ring_prime(f32x4 (&A)[R][8], const f32x4* __restrict__ wp, const int nb) {
#pragma unroll
  for (int u = 0; u < R; ++u)
#pragma unroll
    for (int i = 0; i < 8; ++i) A[u][i] = wp[(min(u, nb - 1) * 8 + i) * 64];
}

template <int NW = 4, int R = 2, class Init, class Epi>
__device__ __forceinline__ void gemm_tiles_f16s_ring(const f32x4* __restrict__ lds, const KSegs ks,
                                                     const f32x4* __restrict__ w, const int n_out_tiles, const int wave,
                                                     const int lane, f32x4 (&A)[R][8],
                                                     const f32x4* __restrict__ next_wp, const int next_nb, Init init, Epi epi) {
  const int nr = ks.nA + ks.nB, ns = nr >> 1, nb = (nr + 7) >> 3, nbp = ((nb + R - 1) / R) * R;
  auto bstep = [&](int st) {                                  // LDS index of the hi row of K-step st (clamped to the last one)
    const int r = 2 * min(st, ns - 1);
    return ((r < ks.nA) ? (ks.rowA + r) : (ks.rowB + (r - ks.nA))) * 64 + lane;
  };
  for (int ot = wave; ot < n_out_tiles; ot += NW) {
    const f32x4* __restrict__ wp = w + (size_t)ot * nb * 512 + lane;
    const bool more = ot + NW < n_out_tiles;
    const f32x4* __restrict__ nwp = more ? wp + (size_t)NW * nb * 512 : next_wp;
    const int nnb = more ? nb : next_nb;
    f32x16 acc1, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
    f32x4 Bh[4], Bl[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int a = bstep(j); Bh[j] = lds[a]; Bl[j] = lds[a + 64]; }
    // drain the vector-memory counter once per tile (the ring was filled at least a whole epilogue + barrier ago: nothing
    // to wait for in practice).  With nothing pending on loop entry the compiler's in-loop waits are the exact distances
    // of the ring (vmcnt(8 R - 1 - 2 j ...)) instead of the vmcnt(0) it falls back to when entry and back-edge disagree.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    init(ot, acc1);                            // after the drain: vector loads issued here land under the K loop
    __builtin_amdgcn_s_setprio(1);
    for (int bi = 0; bi < nbp; bi += R) {
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int blk = bi + u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int a = bstep(blk * 4 + j + 2);
          Bh[(j + 2) & 3] = lds[a];
          Bl[(j + 2) & 3] = lds[a + 64];
          if (blk < nb) {
#ifdef VQN_DIAG_NO_MFMA      // timing only: the operand loads stay (their registers are "used"), the matrix work goes
            asm volatile("" ::"v"(A[u][2 * j]), "v"(A[u][2 * j + 1]), "v"(Bh[j]), "v"(Bl[j]));
#else
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A[u][2 * j]), as_h8(Bh[j]), acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A[u][2 * j]), as_h8(Bl[j]), acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A[u][2 * j + 1]), as_h8(Bh[j]), acc2, 0, 0, 0);
#endif
          }
        }
        // refill slot u: own block blk + R, else block u of the wave's next tile, else (padding position) a dummy re-load
        const int pos = blk + R;
        const bool own = pos < nb, nxt = pos >= nbp;
        const f32x4* __restrict__ src = nxt ? nwp : wp;
        const int sb = own ? pos : (nxt ? min(u, nnb - 1) : nb - 1);
#ifdef VQN_DIAG_W_L1         // timing only: every weight fragment from the same L1-resident 8 KB
#pragma unroll
        for (int i = 0; i < 8; ++i) A[u][i] = w[i * 64 + lane];
#else
#pragma unroll
        for (int i = 0; i < 8; ++i) A[u][i] = src[(sb * 8 + i) * 64];
#endif
      }
    }
    __builtin_amdgcn_s_setprio(0);
    epi(ot, acc1, acc2);
  }
}

// Two-image form: the workgroup holds TWO 32-point images (`img_stride` float4 apart) and a wave applies each weight
// fragment to both (6 MFMAs per step), so the L2 weight stream per point halves -- on this engine that stream, not the
// matrix pipe, is what binds with one image per fragment (measured: 15 TB/s of L2 reads, the L2's limit).  init / epi
// are called once per image: init(ot, img, acc), epi(ot, img, acc1, acc2).
template <int NW = 8, int R = 2, int BD = 2 /* activation fragments fetched BD steps ahead (1 or 2) */, class Init, class Epi>
__device__ __forceinline__ void gemm_tiles_f16s_ring2(const f32x4* __restrict__ lds, const int img_stride, const KSegs ks,
                                                      const f32x4* __restrict__ w, const int n_out_tiles, const int wave,
                                                      const int lane, f32x4 (&A)[R][8],
                                                      const f32x4* __restrict__ next_wp, const int next_nb, Init init, Epi epi) {
  const int nr = ks.nA + ks.nB, ns = nr >> 1, nb = (nr + 7) >> 3, nbp = ((nb + R - 1) / R) * R;
  auto bstep = [&](int st) {
    const int r = 2 * min(st, ns - 1);
    return ((r < ks.nA) ? (ks.rowA + r) : (ks.rowB + (r - ks.nA))) * 64 + lane;
  };
  for (int ot = wave; ot < n_out_tiles; ot += NW) {
    const f32x4* __restrict__ wp = w + (size_t)ot * nb * 512 + lane;
    const bool more = ot + NW < n_out_tiles;
    const f32x4* __restrict__ nwp = more ? wp + (size_t)NW * nb * 512 : next_wp;
    const int nnb = more ? nb : next_nb;
    f32x16 p1, p2, q1, q2;                     // image 0: p1 (hi*hi + init), p2 (cross terms);  image 1: q1, q2
#pragma unroll
    for (int i = 0; i < 16; ++i) { p2[i] = 0.f; q2[i] = 0.f; }
    // activation fragments BD steps ahead (BD = 1 frees 32 registers; 3 % slower where registers are not the limit)
    f32x4 Bh0[4], Bl0[4], Bh1[4], Bl1[4];
#pragma unroll
    for (int j = 0; j < BD; ++j) {
      const int a = bstep(j);
      Bh0[j] = lds[a]; Bl0[j] = lds[a + 64]; Bh1[j] = lds[a + img_stride]; Bl1[j] = lds[a + img_stride + 64];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);        // one vmcnt(0) drain per tile: exact in-loop waits (see gemm_tiles_f16s_ring)
    init(ot, 0, p1);
    init(ot, 1, q1);
    __builtin_amdgcn_s_setprio(1);           // (different priorities for the two waves of a SIMD: measured, no effect)
    for (int bi = 0; bi < nbp; bi += R) {
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int blk = bi + u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int a = bstep(blk * 4 + j + BD);
          Bh0[(j + BD) & 3] = lds[a];
          Bl0[(j + BD) & 3] = lds[a + 64];
          Bh1[(j + BD) & 3] = lds[a + img_stride];
          Bl1[(j + BD) & 3] = lds[a + img_stride + 64];
          if (blk < nb) {
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A[u][2 * j]), as_h8(Bh0[j]), p1, 0, 0, 0);
            q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A[u][2 * j]), as_h8(Bh1[j]), q1, 0, 0, 0);
            p2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A[u][2 * j]), as_h8(Bl0[j]), p2, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A[u][2 * j]), as_h8(Bl1[j]), q2, 0, 0, 0);
            p2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A[u][2 * j + 1]), as_h8(Bh0[j]), p2, 0, 0, 0);
            q2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h8(A[u][2 * j + 1]), as_h8(Bh1[j]), q2, 0, 0, 0);
          }
        }
        const int pos = blk + R;
        const bool own = pos < nb, nxt = pos >= nbp;
        const f32x4* __restrict__ src = nxt ? nwp : wp;
        const int sb = own ? pos : (nxt ? min(u, nnb - 1) : nb - 1);
#ifdef VQN_DIAG_W_L1
#pragma unroll
        for (int i = 0; i < 8; ++i) A[u][i] = w[i * 64 + lane];
#else
#pragma unroll
        for (int i = 0; i < 8; ++i) A[u][i] = src[(sb * 8 + i) * 64];
#endif
      }
    }
    __builtin_amdgcn_s_setprio(0);
    epi(ot, 0, p1, p2);
    epi(ot, 1, q1, q2);
  }
}

// an output tile (16 values per lane in accumulator-register order) -> the four split rows of tile base row `row0`
__device__ __forceinline__ void store_tile_f16s(f32x4* __restrict__ lds, const int row0, const int lane, const float (&v)[16]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float x[8] = {v[8 * s], v[8 * s + 1], v[8 * s + 2], v[8 * s + 3], v[8 * s + 4], v[8 * s + 5], v[8 * s + 6], v[8 * s + 7]};
    f32x4 hi, lo;
    split8(x, hi, lo);
    lds[(row0 + 2 * s) * 64 + lane] = hi;
    lds[(row0 + 2 * s + 1) * 64 + lane] = lo;
  }
}

// the four split rows of a tile -> 16 register-order values (e.g. as an accumulator init)
__device__ __forceinline__ void load_tile_f16s(const f32x4* __restrict__ lds, const int row0, const int lane, float (&v)[16]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float x[8];
    join8(lds[(row0 + 2 * s) * 64 + lane], lds[(row0 + 2 * s + 1) * 64 + lane], x);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[8 * s + i] = x[i];
  }
}

// value of feature f (region starting at row0) of point pp, read half by half
__device__ __forceinline__ float lds_feat_f16s(const f32x4* __restrict__ lds, const int row0, const int f, const int pp) {
  const int sl = f >> 4, fi = f & 15, hh = (fi >> 2) & 1, jj = 4 * (fi >> 3) + (fi & 3);
  const _Float16* hi = reinterpret_cast<const _Float16*>(lds + (row0 + 2 * sl) * 64 + pp + 32 * hh);
  const _Float16* lo = reinterpret_cast<const _Float16*>(lds + (row0 + 2 * sl + 1) * 64 + pp + 32 * hh);
  return fmaf((float)lo[jj], LO_INV, (float)hi[jj]);
}

// part[(wave*32 + p)*NOUT + o] = this wave's share of sum_f wimg[o][f] * act[f][p] over the row pairs of [row0, row0+n_rows);
// wimg is the f32 image [NOUT][n_rows/2][2][8]; weight fetches of a chunk of steps are issued together (one L2 round trip).
template <int NOUT, int NW = 4>
__device__ __forceinline__ void rowdot_f16s(const f32x4* __restrict__ lds, int row0, int n_rows,
                                            const f32x4* __restrict__ wimg, float* out_s, int wave, int lane) {
  constexpr int CH = NOUT == 1 ? 4 : 2;
  const int ns = n_rows >> 1, h = lane >> 5;
  float s[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) s[o] = 0.f;
  for (int q0 = wave; q0 < ns; q0 += NW * CH) {
    f32x4 bh[CH], bl[CH], wv[CH][NOUT][2];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int q = min(q0 + NW * c, ns - 1);
      bh[c] = lds[(row0 + 2 * q) * 64 + lane];
      bl[c] = lds[(row0 + 2 * q + 1) * 64 + lane];
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        wv[c][o][0] = wimg[((o * ns + q) * 2 + h) * 2];
        wv[c][o][1] = wimg[((o * ns + q) * 2 + h) * 2 + 1];
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (q0 + NW * c < ns) {
        float x[8];
        join8(bh[c], bl[c], x);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          s[o] = fmaf(x[0], wv[c][o][0][0], s[o]); s[o] = fmaf(x[1], wv[c][o][0][1], s[o]);
          s[o] = fmaf(x[2], wv[c][o][0][2], s[o]); s[o] = fmaf(x[3], wv[c][o][0][3], s[o]);
          s[o] = fmaf(x[4], wv[c][o][1][0], s[o]); s[o] = fmaf(x[5], wv[c][o][1][1], s[o]);
          s[o] = fmaf(x[6], wv[c][o][1][2], s[o]); s[o] = fmaf(x[7], wv[c][o][1][3], s[o]);
        }
      }
  }
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    s[o] += __shfl_xor(s[o], 32);
    if (h == 0) out_s[(wave * 32 + (lane & 31)) * NOUT + o] = s[o];
  }
}

// acc init from a packed bias: bpack[ot][h][16], register order
__device__ __forceinline__ void init_bias_f16s(const f32x4* __restrict__ bpack, int ot, int lane, f32x16& acc) {
  const f32x4* b = bpack + ot * 8;               // ot is wave-uniform: both halves by scalar loads, each lane keeps its own
  const bool hi = lane >= 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v0 = b[q], v1 = b[4 + q];
    acc[4 * q + 0] = hi ? v1[0] : v0[0]; acc[4 * q + 1] = hi ? v1[1] : v0[1];
    acc[4 * q + 2] = hi ? v1[2] : v0[2]; acc[4 * q + 3] = hi ? v1[3] : v0[3];
  }
}

}  // namespace eng
