// Exact-split variant of the fused small-MLP engine (gfx950): every f32 value x travels as THREE bf16 numbers
//   p0 = bf16(x),  p1 = bf16(x - p0),  p2 = bf16(x - p0 - p1)        =>  x = p0 + p1 + p2  EXACTLY
// (8 + 8 + 8 significant bits; weights: truncation -- the high half of an f32 word IS its bf16 --, activations: round to nearest, see
// split3x8; the two subtractions are exact in f32 either way), and
// a product a*w keeps the six cross terms down to 2^-24 of |a||w|
//     a w = a0 w0 + (a0 w1 + a1 w0) + (a0 w2 + a1 w1 + a2 w0)                 [dropped: a1 w2 + a2 w1 + a2 w2 <= 2^-23 |a w|]
// i.e. what an f32 multiply rounds away, on v_mfma_f32_32x32x16_bf16 with f32 accumulation.  Six bf16 MFMAs (6 x 32 cycles)
// cover the K = 16 that takes eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles) in mlp_prims.h: 2.7x less matrix-pipe time at f32-level
// products.  Unlike the f16 pair engine (mlp_prims_f16s.h: 2^-21 per product, |w| < 6e4) the pieces have the f32 exponent range:
// no scaling, no range caveat.  NOT bit-compatible with the f32 engine (different association of the sums).
//
// Data layout ("x3 activation image").  Same geometry as the other engines -- a workgroup owns 32-point images, rows of
// 64 x 16 B (1 KB) in LDS -- but a 16-feature K step `sl` of a segment is the row TRIPLE (3 sl, 3 sl + 1, 3 sl + 2) = (p0, p1, p2):
// lane (p = lane & 31, h = lane >> 5), bf16 slot jj = 0..7 of each row holds feature 16 sl + 8 (jj >> 2) + 4 h + (jj & 3) of point p
// (`step_feat` of mlp_prims_f16s.h: the B operand of the 32x32x16 MFMA under a fixed permutation of the step's 16 features, the
// same on the weight side, and at the same time the accumulator layout -- registers 8 s .. 8 s + 7 of an output tile ARE slots
// 0..7 of its step s -- so an output tile is split and written back with six ds_write_b128, no shuffles).
// 32 features = 2 steps = 6 rows (1.5x the rows of the f32 image): a 256-wide activation is 48 KB per 32 points, which is why the
// kernels built on this engine run their layers IN PLACE (one activation buffer per image, see neus_mlp_x3.hip).
// Weights: pack[out_tile][step (padded to whole 2-step blocks)][piece][lane][8 bf16]
//        = piece(W[out = 32 ot + (lane & 31)][in = feature(step, lane >> 5, jj)]), 3 KB per tile and step.
#pragma once
#include "mlp_prims_f16s.h"      // step_feat, init_bias_f16s (accumulator-order bias images are shared with the f16 pair engine)
#include <type_traits>

namespace eng {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 as_b8(const f32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ f32x16 mma_x3(const f32x4 a, const f32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_b8(a), as_b8(b), c, 0, 0, 0);
}

// 8 values (slot order) -> the three piece rows.  Round 4: the ACTIVATION pieces are cut with round-to-nearest-even on the conversion
// instruction (v_cvt_pk_bf16_f32: two values per instruction; the residuals x - p0, x - p0 - p1 are still exact in f32 and the third
// piece still takes all that is left: |x - p0| <= 2^-9 |x| leaves <= 16 bits, |.. - p1| <= 2^-17 |x| leaves <= 8), not by truncation:
// nine instead of eleven vector instructions per pair of values, and -- the point -- residual pieces of BOTH signs, so that the three
// cross terms an x3 product drops (a1 w2 + a2 w1 + a2 w2) no longer all carry the sign of a w (with truncated pieces the sdf network's
// error against fp64 was almost pure bias: mean -5.9e-7 where its spread is 1e-7; tests/test_gpu_neus_x3.py prints both).
// Weight packs keep their truncated pieces (host-side packers, tests/test_abi.py); either cut is an exact three-term representation.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3x8(const float (&x)[8], f32x4& q0, f32x4& q1, f32x4& q2) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2_t e = {x[2 * i], x[2 * i + 1]};
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_t));
    q0[i] = __uint_as_float(u);
    const f32x2_t r = e - (f32x2_t){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
    const unsigned v = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    q1[i] = __uint_as_float(v);
    const f32x2_t t = r - (f32x2_t){__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)};
    q2[i] = __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_t)));
  }
}

// the three piece rows -> 8 values: (p2 + p1) + p0, every addition exact
__device__ __forceinline__ void join3x8(const f32x4 q0, const f32x4 q1, const f32x4 q2, float (&x)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned a = __float_as_uint(q0[i]), b = __float_as_uint(q1[i]), c = __float_as_uint(q2[i]);
    x[2 * i] = (__uint_as_float(c << 16) + __uint_as_float(b << 16)) + __uint_as_float(a << 16);
    x[2 * i + 1] = (__uint_as_float(c & 0xffff0000u) + __uint_as_float(b & 0xffff0000u)) + __uint_as_float(a & 0xffff0000u);
  }
}

// rows of a region holding `feats` features
__host__ __device__ __forceinline__ constexpr int x3_rows(int feats) { return 3 * ((feats + 15) / 16); }

// Weight ring: R blocks of 2 K steps (6 fragments = 6 KB per wave and block) that lives across tiles, layers and barriers, exactly
// as in mlp_prims_f16s.h (see the long comment there): unconditional clamped fetches, the refill of a consumed slot comes from the
// same tile, then the wave's next tile of this GEMM, then (`next_wp`, `next_nb`) the wave's first tile of its next GEMM call.
template <int R>
__device__ __forceinline__ void ring_prime_x3(f32x4 (&A)[R][6], const f32x4* __restrict__ wp, const int nb) {
#pragma unroll
  for (int u = 0; u < R; ++u)
#pragma unroll
    for (int i = 0; i < 6; ++i) A[u][i] = wp[(min(u, nb - 1) * 6 + i) * 64];
}

// Two-image GEMM of the engine: out[32 x 32-point tile `ot`] for BOTH images of the workgroup (`img_stride` float4 apart) over the K
// row triples of `ks` (row counts multiples of 3); wave w owns tiles w, w + NW, ...; every weight fragment is applied to both images
// (12 MFMAs per step).  ONE accumulator per image: the six terms of a step go in smallest first; the accumulate roundings are those
// of an f32 accumulation (6 per 16 products instead of 16).  init(ot, img, acc) / epi(ot, img, acc) are called once per image, img as std::integral_constant<int, 0 | 1>.
// `wave` must be wave-uniform (readfirstlane).
template <int NW = 8, int R = 2, int NACC = 1, class Init, class Epi>
__device__ __forceinline__ void gemm_tiles_x3_ring2(const f32x4* __restrict__ lds, const int img_stride, const KSegs ks,
                                                    const f32x4* __restrict__ w, const int n_out_tiles, const int wave,
                                                    const int lane, f32x4 (&A)[R][6],
                                                    const f32x4* __restrict__ next_wp, const int next_nb, Init init, Epi epi) {
  const int nr = ks.nA + ks.nB, ns = nr / 3, nb = (ns + 1) >> 1, nbp = ((nb + R - 1) / R) * R;
  auto bstep = [&](int st) {                                  // LDS index of the p0 row of K step st (clamped to the last one)
    const int r = 3 * min(st, ns - 1);
    return ((r < ks.nA) ? (ks.rowA + r) : (ks.rowB + (r - ks.nA))) * 64 + lane;
  };
  for (int ot = wave; ot < n_out_tiles; ot += NW) {
    const f32x4* __restrict__ wp = w + (size_t)ot * nb * 384 + lane;
    const bool more = ot + NW < n_out_tiles;
    const f32x4* __restrict__ nwp = more ? wp + (size_t)NW * nb * 384 : next_wp;
    const int nnb = more ? nb : next_nb;
    f32x16 acc0, acc1;
    f32x16 sm0, sm1;                           // NACC == 2: the five cross terms below a0 w0 collect here (they are <= 2^-7 of it)
    if (NACC == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { sm0[i] = 0.f; sm1[i] = 0.f; }
    }
    f32x4 B0[2][3], B1[2][3];                  // activation pieces of the current and the next step, image 0 / image 1
    {
      const int a = bstep(0);
#pragma unroll
      for (int q = 0; q < 3; ++q) { B0[0][q] = lds[a + 64 * q]; B1[0][q] = lds[a + img_stride + 64 * q]; }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);        // one vmcnt(0) drain per tile: exact in-loop waits (see gemm_tiles_f16s_ring)
    // (the image index travels as a TYPE: callers that keep per-image register arrays -- epilogue operands, parked tiles -- index them
    //  with a compile-time constant; a run-time index, which is what a lambda that is not inlined twice sees, sends such arrays to scratch)
    init(ot, std::integral_constant<int, 0>{}, acc0);
    init(ot, std::integral_constant<int, 1>{}, acc1);
    __builtin_amdgcn_s_setprio(1);
    for (int bi = 0; bi < nbp; bi += R) {
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int blk = bi + u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int a = bstep(blk * 2 + j + 1);
#pragma unroll
          for (int q = 0; q < 3; ++q) { B0[(j + 1) & 1][q] = lds[a + 64 * q]; B1[(j + 1) & 1][q] = lds[a + img_stride + 64 * q]; }
          if (blk < nb) {
            const f32x4 w0 = A[u][3 * j], w1 = A[u][3 * j + 1], w2 = A[u][3 * j + 2];
#ifdef VQN_DIAG_NO_MFMA      // timing only
            asm volatile("" ::"v"(w0), "v"(w1), "v"(w2), "v"(B0[j][0]), "v"(B0[j][1]), "v"(B0[j][2]), "v"(B1[j][0]), "v"(B1[j][1]), "v"(B1[j][2]));
#else
            if (NACC == 2) {
              sm0 = mma_x3(w2, B0[j][0], sm0);  sm1 = mma_x3(w2, B1[j][0], sm1);          // smallest terms first
              sm0 = mma_x3(w1, B0[j][1], sm0);  sm1 = mma_x3(w1, B1[j][1], sm1);
              sm0 = mma_x3(w0, B0[j][2], sm0);  sm1 = mma_x3(w0, B1[j][2], sm1);
              sm0 = mma_x3(w1, B0[j][0], sm0);  sm1 = mma_x3(w1, B1[j][0], sm1);
              sm0 = mma_x3(w0, B0[j][1], sm0);  sm1 = mma_x3(w0, B1[j][1], sm1);
              acc0 = mma_x3(w0, B0[j][0], acc0);  acc1 = mma_x3(w0, B1[j][0], acc1);
            } else {
              acc0 = mma_x3(w2, B0[j][0], acc0);  acc1 = mma_x3(w2, B1[j][0], acc1);      // smallest terms first
              acc0 = mma_x3(w1, B0[j][1], acc0);  acc1 = mma_x3(w1, B1[j][1], acc1);
              acc0 = mma_x3(w0, B0[j][2], acc0);  acc1 = mma_x3(w0, B1[j][2], acc1);
              acc0 = mma_x3(w1, B0[j][0], acc0);  acc1 = mma_x3(w1, B1[j][0], acc1);
              acc0 = mma_x3(w0, B0[j][1], acc0);  acc1 = mma_x3(w0, B1[j][1], acc1);
              acc0 = mma_x3(w0, B0[j][0], acc0);  acc1 = mma_x3(w0, B1[j][0], acc1);
            }
#endif
          }
        }
        // refill slot u: own block blk + R, else block u of the wave's next tile, else (padding position) a dummy re-load
        const int pos = blk + R;
        const bool own = pos < nb, nxt = pos >= nbp;
        const f32x4* __restrict__ src = nxt ? nwp : wp;
        const int sb = own ? pos : (nxt ? min(u, nnb - 1) : nb - 1);
#ifdef VQN_DIAG_W_L1         // timing only: every weight fragment from the same L1-resident 6 KB
#pragma unroll
        for (int i = 0; i < 6; ++i) A[u][i] = w[i * 64 + lane];
#else
#pragma unroll
        for (int i = 0; i < 6; ++i) A[u][i] = src[(sb * 6 + i) * 64];
#endif
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (NACC == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] += sm0[i]; acc1[i] += sm1[i]; }
    }
    epi(ot, std::integral_constant<int, 0>{}, acc0);
    epi(ot, std::integral_constant<int, 1>{}, acc1);
  }
}

// One-image form of the same GEMM for layers of at most NW / 2 output tiles (the 128-wide layers of the reflectance stacks): with one
// tile per wave such a layer would leave half of the eight waves without work, so the waves split by IMAGE instead -- waves 0..3
// take tile (wave & 3) of image 0, waves 4..7 the same tiles of image 1 (6 MFMAs per step; a weight fragment is fetched by two waves:
// these layers have K <= 256, a small share of the stream).  `lds_img`: the wave's own image; `ot` wave-uniform.  Ring handling as above
// (`next_wp` / `next_nb`: the wave's first tile of its next GEMM call).
template <int R = 2, class Init, class Epi>
__device__ __forceinline__ void gemm_tile_x3_ring1(const f32x4* __restrict__ lds_img, const KSegs ks, const f32x4* __restrict__ w,
                                                   const int ot, const int lane, f32x4 (&A)[R][6],
                                                   const f32x4* __restrict__ next_wp, const int next_nb, Init init, Epi epi) {
  const int nr = ks.nA + ks.nB, ns = nr / 3, nb = (ns + 1) >> 1, nbp = ((nb + R - 1) / R) * R;
  auto bstep = [&](int st) {
    const int r = 3 * min(st, ns - 1);
    return ((r < ks.nA) ? (ks.rowA + r) : (ks.rowB + (r - ks.nA))) * 64 + lane;
  };
  const f32x4* __restrict__ wp = w + (size_t)ot * nb * 384 + lane;
  f32x16 acc;
  f32x4 B[2][3];
  {
    const int a = bstep(0);
#pragma unroll
    for (int q = 0; q < 3; ++q) B[0][q] = lds_img[a + 64 * q];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  init(acc);
  __builtin_amdgcn_s_setprio(1);
  for (int bi = 0; bi < nbp; bi += R) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int blk = bi + u;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int a = bstep(blk * 2 + j + 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) B[(j + 1) & 1][q] = lds_img[a + 64 * q];
        if (blk < nb) {
          const f32x4 w0 = A[u][3 * j], w1 = A[u][3 * j + 1], w2 = A[u][3 * j + 2];
#ifdef VQN_DIAG_NO_MFMA      // timing only
          asm volatile("" ::"v"(w0), "v"(w1), "v"(w2), "v"(B[j][0]), "v"(B[j][1]), "v"(B[j][2]));
#else
          acc = mma_x3(w2, B[j][0], acc);                    // smallest terms first
          acc = mma_x3(w1, B[j][1], acc);
          acc = mma_x3(w0, B[j][2], acc);
          acc = mma_x3(w1, B[j][0], acc);
          acc = mma_x3(w0, B[j][1], acc);
          acc = mma_x3(w0, B[j][0], acc);
#endif
        }
      }
      const int pos = blk + R;
      const bool own = pos < nb, nxt = pos >= nbp;
      const f32x4* __restrict__ src = nxt ? next_wp : wp;
      const int sb = own ? pos : (nxt ? min(u, next_nb - 1) : nb - 1);
#ifdef VQN_DIAG_W_L1         // timing only: every weight fragment from the same L1-resident 6 KB
#pragma unroll
      for (int i = 0; i < 6; ++i) A[u][i] = w[i * 64 + lane];
      (void)src; (void)sb;
#else
#pragma unroll
      for (int i = 0; i < 6; ++i) A[u][i] = src[(sb * 6 + i) * 64];
#endif
    }
  }
  __builtin_amdgcn_s_setprio(0);
  epi(acc);
}

// an output tile (16 values per lane in accumulator-register order) -> its six piece fragments (2 steps x 3 pieces), in registers
__device__ __forceinline__ void split_tile_x3(const float (&v)[16], f32x4 (&o)[6]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float x[8] = {v[8 * s], v[8 * s + 1], v[8 * s + 2], v[8 * s + 3], v[8 * s + 4], v[8 * s + 5], v[8 * s + 6], v[8 * s + 7]};
    split3x8(x, o[3 * s], o[3 * s + 1], o[3 * s + 2]);
  }
}

// ... -> the six rows of tile base row `row0` (= region row0 + 6 * tile)
__device__ __forceinline__ void store_frags_x3(f32x4* __restrict__ lds, const int row0, const int lane, const f32x4 (&o)[6]) {
#pragma unroll
  for (int r = 0; r < 6; ++r) lds[(row0 + r) * 64 + lane] = o[r];
}

__device__ __forceinline__ void store_tile_x3(f32x4* __restrict__ lds, const int row0, const int lane, const float (&v)[16]) {
  f32x4 o[6];
  split_tile_x3(v, o);
  store_frags_x3(lds, row0, lane, o);
}

// the six rows of a tile -> 16 register-order values (e.g. as an accumulator init)
__device__ __forceinline__ void load_tile_x3(const f32x4* __restrict__ lds, const int row0, const int lane, float (&v)[16]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float x[8];
    join3x8(lds[(row0 + 3 * s) * 64 + lane], lds[(row0 + 3 * s + 1) * 64 + lane], lds[(row0 + 3 * s + 2) * 64 + lane], x);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[8 * s + i] = x[i];
  }
}

// value of feature f (region starting at row0) of point pp, read piece by piece
__device__ __forceinline__ float lds_feat_x3(const f32x4* __restrict__ lds, const int row0, const int f, const int pp) {
  const int sl = f >> 4, fi = f & 15, hh = (fi >> 2) & 1, jj = 4 * (fi >> 3) + (fi & 3);
  const unsigned short* r0 = reinterpret_cast<const unsigned short*>(lds + (row0 + 3 * sl) * 64 + pp + 32 * hh);
  const unsigned short* r1 = reinterpret_cast<const unsigned short*>(lds + (row0 + 3 * sl + 1) * 64 + pp + 32 * hh);
  const unsigned short* r2 = reinterpret_cast<const unsigned short*>(lds + (row0 + 3 * sl + 2) * 64 + pp + 32 * hh);
  return (__uint_as_float((unsigned)r2[jj] << 16) + __uint_as_float((unsigned)r1[jj] << 16)) + __uint_as_float((unsigned)r0[jj] << 16);
}

// part[(wave*32 + p)*NOUT + o] = this wave's share of sum_f wimg[o][f] * act[f][p] over the row triples of [row0, row0 + n_rows);
// wimg is the f32 image [NOUT][n_rows/3][2][8]; weight fetches of a chunk of steps are issued together (one L2 round trip).
template <int NOUT, int NW = 4>
__device__ __forceinline__ void rowdot_x3(const f32x4* __restrict__ lds, int row0, int n_rows,
                                          const f32x4* __restrict__ wimg, float* out_s, int wave, int lane) {
  constexpr int CH = NOUT == 1 ? 4 : 2;
  const int ns = n_rows / 3, h = lane >> 5;
  float s[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) s[o] = 0.f;
  for (int q0 = wave; q0 < ns; q0 += NW * CH) {
    f32x4 b0[CH], b1[CH], b2[CH], wv[CH][NOUT][2];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int q = min(q0 + NW * c, ns - 1);
      b0[c] = lds[(row0 + 3 * q) * 64 + lane];
      b1[c] = lds[(row0 + 3 * q + 1) * 64 + lane];
      b2[c] = lds[(row0 + 3 * q + 2) * 64 + lane];
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
        join3x8(b0[c], b1[c], b2[c], x);
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

}  // namespace eng
