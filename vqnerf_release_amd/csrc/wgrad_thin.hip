// Thin weight-gradient contractions: out[r][f] = sum_p A[r][p] B[f][p] for at most EIGHT rows r of one feature tile of A -- the last
// layer of a reflectance head has 1..3 outputs (nfr_unit.py:110-129, vq_nfr.py:135-164), so its weight gradient is 1..3 rows against
// the 128 + 256 features of [y1 ; z].  On the matrix-pipe contraction kernels (csrc/wgrad_x3.hip) such a problem occupies one wave in
// four with 3 useful rows of 32 per MFMA and pays the same per-step barriers as a 256 x 256 block: twelve of them were a quarter of
// the 262,144-point reflectance step's contraction time.  Here it is what it is -- a stream over B at the HBM rate with 2 r FLOP per
// loaded float on the vector ALU: a thread owns one feature (its 32 points of a point tile are 128 contiguous bytes), the r rows of A
// wait in LDS (1 KB per point tile, broadcast reads), no MFMA, no cross-lane step, many workgroups per CU.
// f32 FMA chains in point order (a defined order: bit-reproducible).  Rows a_row0 .. a_row0 + a_rows - 1 of the A tile (the three heads
// of a family keep their 1..3 rows in ONE tile, so that z is streamed once for all of them).  Partial blocks per workgroup are
// TRANSPOSED, [n][32 b_nt features][8 rows], and the row sums [n][32] (bias gradient): vqn_wgrad_finalize sums them with src_rows =
// 32 b_nt, src_cols = 8 and picks a head's rows with its column range (col_first / cols_valid).
#include "common.h"
#include "vqnerf_hip.h"

namespace {

constexpr int TH_MAX = 24;
struct ThinProblem {
  const float* A; const float* B; float* ws; float* rs;
  int a_tiles, a_t0, a_row0, a_rows, b_tiles, b_t0, b_nt;
};
struct ThinTable { ThinProblem p[TH_MAX]; };

__global__ __launch_bounds__(256) void wgrad_thin_kernel(const ThinTable tab, const long n_ptiles) {
  const ThinProblem& P = tab.p[blockIdx.y];
  __shared__ __attribute__((aligned(16))) float dl[2][8][32];
  const int tid = threadIdx.x, f = tid & 31, bt = tid >> 5;
  const bool live = bt < P.b_nt;
  const int sr = tid >> 5, sp = tid & 31;                       // staging role: row sr, point sp of the A tile
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float rsum = 0.f;
  const long step = gridDim.x;
  long t = blockIdx.x;
  auto a_ptr = [&](long tt) { return P.A + ((tt * P.a_tiles + P.a_t0) * 32 + P.a_row0 + sr) * 32 + sp; };
  auto b_ptr = [&](long tt) { return reinterpret_cast<const f32x4*>(P.B + ((tt * P.b_tiles + P.b_t0 + (live ? bt : 0)) * 32 + f) * 32); };
  f32x4 row[8];
  float dnext = 0.f;
  if (t < n_ptiles) {
    dnext = sr < P.a_rows ? *a_ptr(t) : 0.f;
    const f32x4* bp = b_ptr(t);
#pragma unroll
    for (int q = 0; q < 8; ++q) row[q] = bp[q];
  }
  int buf = 0;
  for (; t < n_ptiles; t += step) {
    dl[buf][sr][sp] = dnext;
    __syncthreads();                                           // (one barrier per point tile: the other buffer is free by construction)
    f32x4 cur[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) cur[q] = row[q];
    const long tn = t + step;
    if (tn < n_ptiles) {                                        // next point tile's operands while this one is multiplied
      dnext = sr < P.a_rows ? *a_ptr(tn) : 0.f;
      const f32x4* bp = b_ptr(tn);
#pragma unroll
      for (int q = 0; q < 8; ++q) row[q] = bp[q];
    }
    if (tid < 8) {                                              // row sums of A (the bias gradient): thread r, points in order
      float s = 0.f;
#pragma unroll
      for (int pp = 0; pp < 32; ++pp) s += dl[buf][tid][pp];
      rsum += s;
    }
    if (live) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (r >= P.a_rows) break;
        float a = acc[r];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x4 d = *reinterpret_cast<const f32x4*>(&dl[buf][r][4 * q]);
          a = fmaf(cur[q][0], d[0], a); a = fmaf(cur[q][1], d[1], a); a = fmaf(cur[q][2], d[2], a); a = fmaf(cur[q][3], d[3], a);
        }
        acc[r] = a;
      }
    }
    buf ^= 1;
  }
  const int cols = P.b_nt * 32;
  if (live) {
    f32x4* w = reinterpret_cast<f32x4*>(P.ws + ((size_t)blockIdx.x * cols + bt * 32 + f) * 8);
    w[0] = (f32x4){acc[0], acc[1], acc[2], acc[3]};
    w[1] = (f32x4){acc[4], acc[5], acc[6], acc[7]};
  }
  if (P.rs != nullptr && tid < 32) P.rs[(size_t)blockIdx.x * 32 + tid] = tid < 8 ? rsum : 0.f;
}

}  // namespace

extern "C" int vqn_wgrad_thin_batched(int count, const float* const* A, const int32_t* a_tiles, const int32_t* a_t0, const int32_t* a_row0, const int32_t* a_rows,
                                      const float* const* B, const int32_t* b_tiles, const int32_t* b_t0, const int32_t* b_nt,
                                      int64_t n_point_tiles, int n_split, float* const* ws, float* const* rowsum_ws, void* stream) {
  VQN_CHECK_ARG(count >= 0 && A && a_tiles && a_t0 && a_row0 && a_rows && B && b_tiles && b_t0 && b_nt && ws && rowsum_ws, "null pointer");
  VQN_CHECK_ARG(n_point_tiles >= 1 && n_split >= 1, "n_point_tiles >= 1, n_split >= 1");
  long grid = n_split;
  if (grid > n_point_tiles) grid = n_point_tiles;
  for (int c0 = 0; c0 < count; c0 += TH_MAX) {
    ThinTable tab;
    memset(&tab, 0, sizeof(tab));
    const int n = count - c0 < TH_MAX ? count - c0 : TH_MAX;
    for (int k = 0; k < n; ++k) {
      const int i = c0 + k;
      VQN_CHECK_ARG(A[i] && B[i] && ws[i], "null pointer in problem");
      VQN_CHECK_SHAPE(a_rows[i] >= 1 && a_rows[i] <= 8 && a_row0[i] >= 0 && a_row0[i] + a_rows[i] <= 32 && b_nt[i] >= 1 && b_nt[i] <= 8,
                      "1..8 rows of one A tile, 1..8 feature tiles of B");
      VQN_CHECK_SHAPE(((uintptr_t)ws[i] & 15) == 0, "ws must be 16-byte aligned");
      VQN_CHECK_SHAPE(a_t0[i] >= 0 && a_t0[i] < a_tiles[i] && b_t0[i] >= 0 && b_t0[i] + b_nt[i] <= b_tiles[i], "feature-tile range outside the tensor");
      VQN_CHECK_SHAPE(((uintptr_t)B[i] & 15) == 0, "B must be 16-byte aligned");
      tab.p[k] = ThinProblem{A[i], B[i], ws[i], rowsum_ws[i], a_tiles[i], a_t0[i], a_row0[i], a_rows[i], b_tiles[i], b_t0[i], b_nt[i]};
    }
    hipLaunchKernelGGL(wgrad_thin_kernel, dim3((unsigned)grid, (unsigned)n), dim3(256), 0, (hipStream_t)stream, tab, (long)n_point_tiles);
    VQN_LAUNCH_CHECK();
  }
  return (int)grid;
}
