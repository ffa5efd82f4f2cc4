// Weight-gradient contraction (see csrc/wgrad.hip) on the bf16 matrix pipe at f32 accuracy: every f32 operand is split EXACTLY into
// three bf16 pieces (8 + 8 + 8 significant bits, by truncation: x = p0 + p1 + p2 with no rounding anywhere), and a product keeps
// the six cross terms down to 2^-24 of |a||b|:
//     a b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)      [dropped: a1 b2 + a2 b1 + a2 b2 <= 2^-23 |a b|]
// i.e. what an f32 multiply rounds away, accumulated in f32 like the f32-input MFMA does.  Unlike an f16 split the pieces have the
// f32 exponent range: adjoints of 1e-9 need no scaling.  Six v_mfma_f32_32x32x16_bf16 (32 cycles each) cover the K = 16 that takes
// eight v_mfma_f32_32x32x2_f32 (64 cycles each): 2.7x less matrix-pipe time, which puts the 256 x 256 blocks of the NeuS nets at the
// HBM rate of their operand stream instead of the f32 pipe's.  Same workspace / reduction contract as vqn_wgrad_partials
// (deterministic: fixed order, no atomics).
#include "common.h"
#include "wgrad_batch.h"
#include <stdlib.h>
#include <type_traits>

int vqn_wgrad_partials_f32_internal(const float* A, int a_tiles, int a_t0, int a_nt, const float* B, int b_tiles, int b_t0, int b_nt,
                                    int64_t n_point_tiles, int n_split, float* ws, float* rowsum_ws, void* stream);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Pieces {
  u32x4 p0, p1, p2;          // bf16x8 each: elements j = 0..7 <-> points 8 kk + j of the step
};

// one pair of elements (slot i of the fragments) -- the unit in which the splits are dealt out between the MFMAs below.  Round 4: cut with
// round-to-nearest on v_cvt_pk_bf16_f32 (nine instead of eleven vector instructions per pair; residual pieces of both signs, so the
// dropped cross terms of a product do not all carry its sign -- see split3x8 in mlp_prims_x3.h); still x = p0 + p1 + p2 exactly.
typedef __bf16 bf16x2_w __attribute__((ext_vector_type(2)));
typedef float f32x2_w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(const float e0, const float e1, unsigned& q0, unsigned& q1, unsigned& q2) {
  const f32x2_w e = {e0, e1};
  q0 = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_w));
  const f32x2_w r = e - (f32x2_w){__uint_as_float(q0 << 16), __uint_as_float(q0 & 0xffff0000u)};
  q1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_w));
  const f32x2_w t = r - (f32x2_w){__uint_as_float(q1 << 16), __uint_as_float(q1 & 0xffff0000u)};
  q2 = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_w));
}

// eight consecutive f32 -> three exact bf16x8 pieces
__device__ __forceinline__ void split3(const f32x4 lo4, const f32x4 hi4, Pieces& o) {
  const float x[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned q0, q1, q2;
    split_pair(x[2 * i], x[2 * i + 1], q0, q1, q2);
    o.p0[i] = q0; o.p1[i] = q1; o.p2[i] = q2;
  }
}

__device__ __forceinline__ f32x16 mma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// wave w owns output tiles (w, w + 4) x (0..7); step q = (point tile, half u): 16 points
template <bool FULL>          // FULL: a_nt == b_nt == 8 -- no guards in the instruction stream
__global__ __launch_bounds__(256, 1) void wgrad_x3_kernel(const float* __restrict__ A, int a_tiles, int a_t0, int a_nt,
                                                          const float* __restrict__ B, int b_tiles, int b_t0, int b_nt,
                                                          long n_ptiles, float* __restrict__ ws, float* __restrict__ rowsum_ws) {
  constexpr int NOT = 2, BT = 8;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fi = lane & 31, kk = lane >> 5;
  f32x16 acc[NOT][BT];
#pragma unroll
  for (int a = 0; a < NOT; ++a)
#pragma unroll
    for (int b = 0; b < BT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  f32x4 rs[NOT];
#pragma unroll
  for (int a = 0; a < NOT; ++a) rs[a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const long my_tiles = (n_ptiles - blockIdx.x + gridDim.x - 1) / gridDim.x, n_q = 2 * my_tiles;
  // unconditional clamped fetches, one step ahead (two register buffers): see wgrad.hip
  auto fetch = [&](long q, f32x4 (&af)[NOT][2], f32x4 (&bf)[BT][2]) {
    long t = blockIdx.x + (q >> 1) * (long)gridDim.x;
    if (t >= n_ptiles) t = n_ptiles - 1;
    const int u = (int)(q & 1);
    const float* At = A + ((t * a_tiles + a_t0) * 32 + fi) * 32 + 8 * kk + 16 * u;
    const float* Bt = B + ((t * b_tiles + b_t0) * 32 + fi) * 32 + 8 * kk + 16 * u;
#pragma unroll
    for (int a = 0; a < NOT; ++a) {
      const float* p = At + (long)min(wave + 4 * a, a_nt - 1) * 1024;
      af[a][0] = *reinterpret_cast<const f32x4*>(p);
      af[a][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const float* p = Bt + (long)min(b, b_nt - 1) * 1024;
      bf[b][0] = *reinterpret_cast<const f32x4*>(p);
      bf[b][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
  };
  auto multiply = [&](const f32x4 (&af)[NOT][2], const f32x4 (&bf)[BT][2]) {
    Pieces pa[NOT];
#pragma unroll
    for (int a = 0; a < NOT; ++a) {
      rs[a] += af[a][0] + af[a][1];
      split3(af[a][0], af[a][1], pa[a]);
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (!FULL && b >= b_nt) continue;
      Pieces pb;
      split3(bf[b][0], bf[b][1], pb);
#pragma unroll
      for (int a = 0; a < NOT; ++a)
        if (FULL || wave + 4 * a < a_nt) {
          f32x16 c = acc[a][b];
          c = mma(pa[a].p2, pb.p0, c);            // smallest terms first
          c = mma(pa[a].p1, pb.p1, c);
          c = mma(pa[a].p0, pb.p2, c);
          c = mma(pa[a].p1, pb.p0, c);
          c = mma(pa[a].p0, pb.p1, c);
          c = mma(pa[a].p0, pb.p0, c);
          acc[a][b] = c;
        }
    }
  };
  f32x4 af[2][NOT][2], bf[2][BT][2];
  fetch(0, af[0], bf[0]);
  for (long q = 0; q < n_q; q += 2) {
    fetch(q + 1, af[1], bf[1]);
    multiply(af[0], bf[0]);
    fetch(q + 2, af[0], bf[0]);
    if (q + 1 < n_q) multiply(af[1], bf[1]);
  }
  const int cols = b_nt * 32;
  float* w = ws + (size_t)blockIdx.x * (size_t)(a_nt * 32) * cols;
#pragma unroll
  for (int a = 0; a < NOT; ++a) {
    const int ot = wave + 4 * a;
    if (ot >= a_nt) continue;
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b >= b_nt) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * kk;
        w[(size_t)(ot * 32 + row) * cols + b * 32 + fi] = acc[a][b][e];
      }
    }
    if (rowsum_ws != nullptr) {
      float r = (rs[a][0] + rs[a][1]) + (rs[a][2] + rs[a][3]);
      r += __shfl_xor(r, 32);
      if (kk == 0) rowsum_ws[(size_t)blockIdx.x * (a_nt * 32) + ot * 32 + fi] = r;
    }
  }
}

// The 256 x 256 form with the B operand cut into pieces ONCE per workgroup: wave w fetches and splits B tiles 2 w and 2 w + 1 of
// the step and parks the pieces in LDS (3 KB per tile, two step buffers), every wave then reads all eight tiles' pieces from there
// (24 ds_read_b128 per step and wave).  Without it each wave fetches all of B itself, half a 128-byte line per instruction, and the
// L2 -> L1 traffic (B four times over) binds the launch at 2.5x its matrix time.  One barrier per step.
// NOT = output tiles per wave: 2 (a_nt up to 8), or 1 for problems of a_nt <= 4 -- half the accumulators (128 registers), so that TWO
// workgroups share a CU and one's operand splits, LDS parking and barrier waits run under the other's MFMAs (round 4: the 128-wide
// layers of the reflectance stacks make most of their contractions such problems, and with one wave per SIMD they ran at 0.32 of the
// matrix pipe, overhead-bound per 16-point step).
template <bool FULL, int NOT_ = 2>          // FULL: a_nt == b_nt == 8 -- no guards in the instruction stream
__device__ __forceinline__ void wgrad_x3_lds_body(const float* __restrict__ A, int a_tiles, int a_t0, int a_nt,
                                                  const float* __restrict__ B, int b_tiles, int b_t0, int b_nt, long n_ptiles,
                                                  float* __restrict__ ws, float* __restrict__ rowsum_ws) {
  constexpr int NOT = NOT_, BT = 8;
  static_assert(!FULL || NOT_ == 2, "the 256 x 256 form has two output tiles per wave");
  const bool two = NOT == 2 && (FULL || (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + 4 < a_nt));      // this wave owns a second output tile
  __shared__ u32x4 pieces[2][BT][3][64];                       // 48 KB
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fi = lane & 31, kk = lane >> 5;
  f32x16 acc[NOT][BT];
#pragma unroll
  for (int a = 0; a < NOT; ++a)
#pragma unroll
    for (int b = 0; b < BT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  f32x4 rs[NOT];
#pragma unroll
  for (int a = 0; a < NOT; ++a) rs[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long my_tiles = (n_ptiles - blockIdx.x + gridDim.x - 1) / gridDim.x, n_q = 2 * my_tiles;
  auto fetch = [&](long q, f32x4 (&af)[NOT][2], f32x4 (&bf)[2][2]) {
    long t = blockIdx.x + (q >> 1) * (long)gridDim.x;
    if (t >= n_ptiles) t = n_ptiles - 1;
    const int u = (int)(q & 1);
    const float* At = A + ((t * a_tiles + a_t0) * 32 + fi) * 32 + 8 * kk + 16 * u;
    const float* Bt = B + ((t * b_tiles + b_t0) * 32 + fi) * 32 + 8 * kk + 16 * u;
#pragma unroll
    for (int a = 0; a < NOT; ++a) {
      const float* p = At + (long)(FULL ? wave + 4 * a : min(wave + 4 * a, a_nt - 1)) * 1024;
      af[a][0] = *reinterpret_cast<const f32x4*>(p);
      af[a][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* p = Bt + (long)(FULL ? 2 * wave + i : min(2 * wave + i, b_nt - 1)) * 1024;
      bf[i][0] = *reinterpret_cast<const f32x4*>(p);
      bf[i][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
  };
  auto park = [&](int buf, const f32x4 (&bf)[2][2]) {          // this wave's two B tiles -> pieces in LDS
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      Pieces pb;
      split3(bf[i][0], bf[i][1], pb);
      pieces[buf][2 * wave + i][0][lane] = pb.p0;
      pieces[buf][2 * wave + i][1][lane] = pb.p1;
      pieces[buf][2 * wave + i][2][lane] = pb.p2;
    }
  };
  // operand fetches run R - 1 steps ahead (a step multiplies for ~1.3 us; an HBM round trip under load is several times that)
  constexpr int R = 4;
  f32x4 af[R][NOT][2], bf[R][2][2];
#pragma unroll
  for (int i = 0; i < R - 1; ++i) fetch(i, af[i], bf[i]);
  park(0, bf[0]);
  __syncthreads();
  // FULL form: step q multiplies with the A pieces cut during step q - 1 and the B pieces parked during step q - 1 while it cuts
  // A (q + 1) and parks B (q + 1) -- one pair of A and one pair of B elements (22 vector instructions) per B tile, dealt out between
  // that tile's twelve MFMAs by the group barriers of its (small) scheduling region.  One wave per SIMD issues strictly in order: left
  // to itself the compiler emits the MFMA block and the splits one after the other.  Steps beyond n_q multiply zero pieces.
  Pieces pa[NOT];
  if (FULL) {
#pragma unroll
    for (int a = 0; a < NOT; ++a) {
      rs[a] += af[0][a][0] + af[0][a][1];
      split3(af[0][a][0], af[0][a][1], pa[a]);
    }
  }
  auto step_full = [&](long q, auto slot_c) {
    constexpr int slot = decltype(slot_c)::value, cur = slot & 1, nxt = (slot + 1) % R;
    fetch(q + R - 1, af[(slot + R - 1) % R], bf[(slot + R - 1) % R]);
    const bool live = q + 1 < n_q;                              // wave-uniform; applied as a select, not a branch
    Pieces pn[NOT], pk[2];
    u32x4 n0 = pieces[cur][0][0][lane], n1 = pieces[cur][0][1][lane], n2 = pieces[cur][0][2][lane];
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const u32x4 b0 = n0, b1 = n1, b2 = n2;
      if (b + 1 < BT) { n0 = pieces[cur][b + 1][0][lane]; n1 = pieces[cur][b + 1][1][lane]; n2 = pieces[cur][b + 1][2][lane]; }
      f32x16 c0 = acc[0][b], c1 = acc[1][b];
      c0 = mma(pa[0].p2, b0, c0); c1 = mma(pa[1].p2, b0, c1);
      c0 = mma(pa[0].p1, b1, c0); c1 = mma(pa[1].p1, b1, c1);
      c0 = mma(pa[0].p0, b2, c0); c1 = mma(pa[1].p0, b2, c1);
      c0 = mma(pa[0].p1, b0, c0); c1 = mma(pa[1].p1, b0, c1);
      c0 = mma(pa[0].p0, b1, c0); c1 = mma(pa[1].p0, b1, c1);
      c0 = mma(pa[0].p0, b0, c0); c1 = mma(pa[1].p0, b0, c1);
      acc[0][b] = c0; acc[1][b] = c1;
      {   // this tile's share of the next step's splits: pair i of A tile ta and of this wave's B tile tb
        constexpr int dummy = 0; (void)dummy;
        const int ta = b >> 2, i = b & 3;
        const f32x4 xa = af[nxt][ta][i >> 1];
        float e0 = xa[2 * (i & 1)], e1 = xa[2 * (i & 1) + 1];
        if (!live) { e0 = 0.f; e1 = 0.f; }
        rs[ta][2 * (i & 1)] += e0; rs[ta][2 * (i & 1) + 1] += e1;
        unsigned q0, q1, q2, k0, k1, k2;
        split_pair(e0, e1, q0, q1, q2);
        const f32x4 xb = bf[nxt][ta][i >> 1];
        split_pair(xb[2 * (i & 1)], xb[2 * (i & 1) + 1], k0, k1, k2);
        // (anchors: without them the optimiser sinks all sixteen pair splits below the tile loop, next to their first use)
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(k0), "+v"(k1), "+v"(k2));
        pn[ta].p0[i] = q0; pn[ta].p1[i] = q1; pn[ta].p2[i] = q2;
        pk[ta].p0[i] = k0; pk[ta].p1[i] = k1; pk[ta].p2[i] = k2;
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);         // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);         // two vector instructions of the splits
      }
      __builtin_amdgcn_sched_barrier(0);                         // regions of one tile: keeps the scheduler's problem small
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      pieces[cur ^ 1][2 * wave + i][0][lane] = pk[i].p0;
      pieces[cur ^ 1][2 * wave + i][1][lane] = pk[i].p1;
      pieces[cur ^ 1][2 * wave + i][2][lane] = pk[i].p2;
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NOT; ++a) pa[a] = pn[a];
  };
  auto step = [&](long q, auto slot_c) {                       // slot = q % R, compile-time at the call sites; LDS buffer q & 1
    constexpr int slot = decltype(slot_c)::value, cur = slot & 1;
    fetch(q + R - 1, af[(slot + R - 1) % R], bf[(slot + R - 1) % R]);
    if (q < n_q) {
      Pieces pa[NOT];
#pragma unroll
      for (int a = 0; a < NOT; ++a) {
        rs[a] += af[slot][a][0] + af[slot][a][1];
        split3(af[slot][a][0], af[slot][a][1], pa[a]);
      }
      u32x4 n0 = pieces[cur][0][0][lane], n1 = pieces[cur][0][1][lane], n2 = pieces[cur][0][2][lane];
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        if (!FULL && b >= b_nt) break;
        const u32x4 b0 = n0, b1 = n1, b2 = n2;
        if (b + 1 < BT) { n0 = pieces[cur][b + 1][0][lane]; n1 = pieces[cur][b + 1][1][lane]; n2 = pieces[cur][b + 1][2][lane]; }   // one tile ahead of the MFMAs
        // the wave's two output tiles alternate MFMA by MFMA (one wave per SIMD: nothing else covers a dependent chain)
        if constexpr (NOT == 2) {
          f32x16 c0 = acc[0][b], c1 = acc[NOT - 1][b];
          if (FULL || two) {
            c0 = mma(pa[0].p2, b0, c0); c1 = mma(pa[NOT - 1].p2, b0, c1);
            c0 = mma(pa[0].p1, b1, c0); c1 = mma(pa[NOT - 1].p1, b1, c1);
            c0 = mma(pa[0].p0, b2, c0); c1 = mma(pa[NOT - 1].p0, b2, c1);
            c0 = mma(pa[0].p1, b0, c0); c1 = mma(pa[NOT - 1].p1, b0, c1);
            c0 = mma(pa[0].p0, b1, c0); c1 = mma(pa[NOT - 1].p0, b1, c1);
            c0 = mma(pa[0].p0, b0, c0); c1 = mma(pa[NOT - 1].p0, b0, c1);
          } else {
            c0 = mma(pa[0].p2, b0, c0); c0 = mma(pa[0].p1, b1, c0); c0 = mma(pa[0].p0, b2, c0);
            c0 = mma(pa[0].p1, b0, c0); c0 = mma(pa[0].p0, b1, c0); c0 = mma(pa[0].p0, b0, c0);
          }
          acc[0][b] = c0; acc[NOT - 1][b] = c1;
        } else {
          f32x16 c0 = acc[0][b];
          c0 = mma(pa[0].p2, b0, c0); c0 = mma(pa[0].p1, b1, c0); c0 = mma(pa[0].p0, b2, c0);
          c0 = mma(pa[0].p1, b0, c0); c0 = mma(pa[0].p0, b1, c0); c0 = mma(pa[0].p0, b0, c0);
          acc[0][b] = c0;
        }
      }
    }
    park(cur ^ 1, bf[(slot + 1) % R]);
    __syncthreads();
  };
  if constexpr (FULL) {
    for (long q = 0; q < n_q; q += R) {
      step_full(q, std::integral_constant<int, 0>{});
      step_full(q + 1, std::integral_constant<int, 1>{});
      step_full(q + 2, std::integral_constant<int, 2>{});
      step_full(q + 3, std::integral_constant<int, 3>{});
    }
  } else
  for (long q = 0; q < n_q; q += R) {                          // n_q is even; R = 4: the tail steps beyond n_q only keep the barriers uniform
    step(q, std::integral_constant<int, 0>{});
    step(q + 1, std::integral_constant<int, 1>{});
    step(q + 2, std::integral_constant<int, 2>{});
    step(q + 3, std::integral_constant<int, 3>{});
  }
  const int cols = b_nt * 32;
  float* w = ws + (size_t)blockIdx.x * (size_t)(a_nt * 32) * cols;
#pragma unroll
  for (int a = 0; a < NOT; ++a) {
    const int ot = wave + 4 * a;
    if (ot >= a_nt) continue;
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b >= b_nt) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * kk;
        w[(size_t)(ot * 32 + row) * cols + b * 32 + fi] = acc[a][b][e];
      }
    }
    if (rowsum_ws != nullptr) {
      float r = (rs[a][0] + rs[a][1]) + (rs[a][2] + rs[a][3]);
      r += __shfl_xor(r, 32);
      if (kk == 0) rowsum_ws[(size_t)blockIdx.x * (a_nt * 32) + ot * 32 + fi] = r;
    }
  }
}

template <bool FULL>
__global__ __launch_bounds__(256, 1) void wgrad_x3_lds_kernel(const float* __restrict__ A, int a_tiles, int a_t0, int a_nt,
                                                              const float* __restrict__ B, int b_tiles, int b_t0, int b_nt, long n_ptiles,
                                                              float* __restrict__ ws, float* __restrict__ rowsum_ws) {
  wgrad_x3_lds_body<FULL>(A, a_tiles, a_t0, a_nt, B, b_tiles, b_t0, b_nt, n_ptiles, ws, rowsum_ws);
}

// many contractions over the same points in one launch (blockIdx.y: the problem)
template <bool FULL>
__global__ __launch_bounds__(256, 1) void wgrad_x3_lds_batched_kernel(const WgTable tab, long n_ptiles) {
  const WgProblem& P = tab.p[blockIdx.y];
  wgrad_x3_lds_body<FULL>(P.A, P.a_tiles, P.a_t0, P.a_nt, P.B, P.b_tiles, P.b_t0, P.b_nt, n_ptiles, P.ws, P.rs);
}
// ... of at most four A tiles: one output tile per wave, two workgroups per CU
__global__ __launch_bounds__(256, 2) void wgrad_x3_lds_batched_narrow_kernel(const WgTable tab, long n_ptiles) {
  const WgProblem& P = tab.p[blockIdx.y];
  wgrad_x3_lds_body<false, 1>(P.A, P.a_tiles, P.a_t0, P.a_nt, P.B, P.b_tiles, P.b_t0, P.b_nt, n_ptiles, P.ws, P.rs);
}

}  // namespace

extern "C" int vqn_wgrad_partials_x3(const float* A, int a_tiles, int a_t0, int a_nt, const float* B, int b_tiles, int b_t0,
                                     int b_nt, int64_t n_point_tiles, int n_split, float* ws, float* rowsum_ws, void* stream);

// The contractions of a whole backward pass (same points, hence the same split) in as few launches as they have kernel shapes:
// problem i is vqn_wgrad_partials (x3 == 0) / vqn_wgrad_partials_x3 (x3 != 0) of (A[i], a_tiles[i], ..., ws[i], rowsum_ws[i]) -- the
// same kernels, hence the same partial blocks bit for bit -- grouped by kernel variant, a launch per group of up to 24.
extern "C" int vqn_wgrad_partials_batched(int count, const float* const* A, const int32_t* a_tiles, const int32_t* a_t0, const int32_t* a_nt,
                                          const float* const* B, const int32_t* b_tiles, const int32_t* b_t0, const int32_t* b_nt,
                                          int64_t n_point_tiles, int n_split, float* const* ws, float* const* rowsum_ws, int x3, void* stream) {
  VQN_CHECK_ARG(count >= 0 && A && a_tiles && a_t0 && a_nt && B && b_tiles && b_t0 && b_nt && ws && rowsum_ws, "null pointer");
  VQN_CHECK_ARG(n_point_tiles >= 1 && n_split >= 1, "n_point_tiles >= 1, n_split >= 1");
  long grid = n_split;
  if (grid > n_point_tiles) grid = n_point_tiles;
  static const long small_from = [] { const char* e = getenv("VQN_WGRAD_X3_SMALL_TILES"); return (e && e[0]) ? atol(e) : 1024L; }();
  static const int no_lds = [] { const char* e = getenv("VQN_WGRAD_X3_NO_LDS"); return (e != nullptr && atoi(e) != 0) ? 1 : 0; }();
  // classes 0..2: the f32 kernels by shape; 3 / 4 / 5: the x3 LDS kernel, full / guarded / guarded with at most four A tiles
  static const int no_narrow = [] { const char* e = getenv("VQN_WGRAD_X3_NO_NARROW"); return (e != nullptr && atoi(e) != 0) ? 1 : 0; }();
  WgProblem cls[6][WG_MAX];
  int n_cls[6] = {0, 0, 0, 0, 0, 0};
  auto flush = [&](int c) -> int {
    if (n_cls[c] == 0) return VQN_OK;
    int rc = VQN_OK;
    if (c < 3) rc = vqn_wgrad_f32_batched_internal(cls[c], n_cls[c], c, (long)n_point_tiles, grid, stream);
    else {
      WgTable tab;
      memset(&tab, 0, sizeof(tab));
      for (int i = 0; i < n_cls[c]; ++i) tab.p[i] = cls[c][i];
      const dim3 g((unsigned)grid, (unsigned)n_cls[c]);
      if (c == 3) hipLaunchKernelGGL(wgrad_x3_lds_batched_kernel<true>, g, dim3(256), 0, (hipStream_t)stream, tab, (long)n_point_tiles);
      else if (c == 4) hipLaunchKernelGGL(wgrad_x3_lds_batched_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, tab, (long)n_point_tiles);
      else hipLaunchKernelGGL(wgrad_x3_lds_batched_narrow_kernel, g, dim3(256), 0, (hipStream_t)stream, tab, (long)n_point_tiles);
      VQN_LAUNCH_CHECK();
    }
    n_cls[c] = 0;
    return rc;
  };
  for (int i = 0; i < count; ++i) {
    VQN_CHECK_ARG(A[i] && B[i] && ws[i], "null pointer in problem");
    VQN_CHECK_SHAPE(a_nt[i] >= 1 && a_nt[i] <= 8 && b_nt[i] >= 1 && b_nt[i] <= 8, "1..8 feature tiles per operand and problem");
    VQN_CHECK_SHAPE(a_t0[i] >= 0 && a_t0[i] + a_nt[i] <= a_tiles[i] && b_t0[i] >= 0 && b_t0[i] + b_nt[i] <= b_tiles[i], "feature-tile range outside the tensor");
    VQN_CHECK_SHAPE(((uintptr_t)A[i] & 15) == 0 && ((uintptr_t)B[i] & 15) == 0, "operands must be 16-byte aligned");
    const bool f32 = !x3 || (a_nt[i] <= 4 && n_point_tiles < small_from);
    if (!f32 && no_lds) {                                   // (diagnostic switch: the non-LDS x3 kernels have no batched form)
      const int n = vqn_wgrad_partials_x3(A[i], a_tiles[i], a_t0[i], a_nt[i], B[i], b_tiles[i], b_t0[i], b_nt[i], n_point_tiles, n_split, ws[i],
                                          rowsum_ws[i], stream);
      if (n < 0) return n;
      continue;
    }
    const int c = f32 ? ((a_nt[i] <= 4 && b_nt[i] <= 4) ? 0 : (a_nt[i] <= 4 ? 1 : 2)) : ((a_nt[i] == 8 && b_nt[i] == 8) ? 3 : ((a_nt[i] <= 4 && !no_narrow) ? 5 : 4));
    cls[c][n_cls[c]++] = WgProblem{A[i], B[i], ws[i], rowsum_ws[i], a_tiles[i], a_t0[i], a_nt[i], b_tiles[i], b_t0[i], b_nt[i]};
    if (n_cls[c] == WG_MAX) { const int rc = flush(c); if (rc != VQN_OK) return rc; }
  }
  for (int c = 0; c < 6; ++c) { const int rc = flush(c); if (rc != VQN_OK) return rc; }
  return (int)grid;
}

extern "C" int vqn_wgrad_partials_x3(const float* A, int a_tiles, int a_t0, int a_nt, const float* B, int b_tiles, int b_t0,
                                     int b_nt, int64_t n_point_tiles, int n_split, float* ws, float* rowsum_ws, void* stream) {
  VQN_CHECK_ARG(A && B && ws, "null pointer");
  VQN_CHECK_ARG(n_point_tiles >= 1 && n_split >= 1, "n_point_tiles >= 1, n_split >= 1");
  VQN_CHECK_SHAPE(a_nt >= 1 && a_nt <= 8 && b_nt >= 1 && b_nt <= 8, "1..8 feature tiles per operand and call");
  VQN_CHECK_SHAPE(a_t0 >= 0 && a_t0 + a_nt <= a_tiles && b_t0 >= 0 && b_t0 + b_nt <= b_tiles, "feature-tile range outside the tensor");
  VQN_CHECK_SHAPE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "operands must be 16-byte aligned");
  // small blocks (<= 128 output features) at the reference batch (64 point tiles) are latency-, not matrix-bound: the f32 kernels with
  // their deeper rings -- no difference measured there (6.50 vs 6.52 ms per 2048-point step).  From 1024 point tiles on they take the
  // x3 kernel too: 262,144-point reflectance step 22.9 -> 22.0 ms (A / B / A on one box).  VQN_WGRAD_X3_SMALL_TILES=<n> moves the switch.
  static const long small_from = [] { const char* e = getenv("VQN_WGRAD_X3_SMALL_TILES"); return (e && e[0]) ? atol(e) : 1024L; }();
  if (a_nt <= 4 && n_point_tiles < small_from)
    return vqn_wgrad_partials_f32_internal(A, a_tiles, a_t0, a_nt, B, b_tiles, b_t0, b_nt, n_point_tiles, n_split, ws, rowsum_ws, stream);
  long grid = n_split;
  if (grid > n_point_tiles) grid = n_point_tiles;
  static const int no_lds = [] { const char* e = getenv("VQN_WGRAD_X3_NO_LDS"); return (e != nullptr && atoi(e) != 0) ? 1 : 0; }();
  if (a_nt == 8 && b_nt == 8 && !no_lds)
    hipLaunchKernelGGL(wgrad_x3_lds_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, A, a_tiles, a_t0, a_nt, B, b_tiles,
                       b_t0, b_nt, (long)n_point_tiles, ws, rowsum_ws);
  else if (!no_lds)
    hipLaunchKernelGGL(wgrad_x3_lds_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, A, a_tiles, a_t0, a_nt, B, b_tiles,
                       b_t0, b_nt, (long)n_point_tiles, ws, rowsum_ws);
  else if (a_nt == 8 && b_nt == 8)
    hipLaunchKernelGGL(wgrad_x3_kernel<true>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, A, a_tiles, a_t0, a_nt, B, b_tiles,
                       b_t0, b_nt, (long)n_point_tiles, ws, rowsum_ws);
  else
    hipLaunchKernelGGL(wgrad_x3_kernel<false>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, A, a_tiles, a_t0, a_nt, B, b_tiles,
                       b_t0, b_nt, (long)n_point_tiles, ws, rowsum_ws);
  VQN_LAUNCH_CHECK();
  return (int)grid;
}
