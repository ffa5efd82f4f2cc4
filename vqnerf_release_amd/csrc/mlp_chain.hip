// Descriptor-driven fused Dense-stack evaluator for gfx950: [posenc ->] layer -> layer -> ... with skip-concats,
// several heads per launch, activations resident in LDS (the "activation image" of mlp_prims.h), f32 MFMA.
// Replaces the Keras-Dense chains of the reflectance model, evaluated through `chunk_apply` in the reference:
//   decomp/nerfvq_nfr3/nerfactor/networks/embedder.py:23-47 + mlp.py:24-50 + seq.py:24-38
//   decomp/nerfvq_nfr3/nerfactor/models/vq_nfr.py:771-784   (_pred_enc_at: posenc -> fine_enc -> bottleneck -> z)
//   decomp/nerfvq_nfr3/nerfactor/models/vq_nfr.py:786-828   (_pred_diff_at/_pred_spec_at/_pred_rough_at: z -> heads)
//   decomp/nerfvq_nfr3/nerfactor/models/shape.py:169-179    (chunk_apply: disappears)
// The layer program (which LDS rows feed a layer, where its output goes, which outputs leave for HBM) is built
// on the host (vqnerf_release_amd/decomp/packing.py) -- the kernel is a small interpreter over it.
#include "mlp_prims.h"
#include "vqn_chain_desc.h"
#include "vqnerf_hip.h"
#include <math.h>

using namespace eng;

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

struct ChainSmalls {
  float part[8 * 32 * 4];
};

__device__ __forceinline__ float act_rt(int act, float x) {
  switch (act) {
    case ACT_RELU: return fmaxf(x, 0.f);
    case ACT_SIGMOID: return fast_rcp(1.f + fast_exp(-x));
    case ACT_SOFTPLUS100: return act_fwd<ACT_SOFTPLUS100>(x);
    default: return x;
  }
}

// One output tile through its activation into the LDS image.  The activation is a run-time field of the layer record, but it is
// resolved ONCE per tile here: with the switch inside the element loops every accumulator element went through its own chain of
// scalar compares and branches.
template <int ACT>
__device__ __forceinline__ void store_tile_act(f32x4* lds, int row0, int lane, const f32x16& acc) {
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    f32x4 v = acc_quad(acc, rq);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = act_fwd<ACT>(v[j]);
    lds[(row0 + rq) * 64 + lane] = v;
  }
}
__device__ __forceinline__ void store_tile(int act, f32x4* lds, int row0, int lane, const f32x16& acc) {
  switch (act) {
    case ACT_RELU: store_tile_act<ACT_RELU>(lds, row0, lane, acc); break;
    case ACT_SIGMOID: store_tile_act<ACT_SIGMOID>(lds, row0, lane, acc); break;
    case ACT_SOFTPLUS100: store_tile_act<ACT_SOFTPLUS100>(lds, row0, lane, acc); break;
    default: store_tile_act<ACT_NONE>(lds, row0, lane, acc); break;
  }
}

struct OutPtrs {
  float* p[VQN_CHAIN_MAX_OUTS];
  int ld[VQN_CHAIN_MAX_OUTS];
};

// the <= 4-output layers' weight images are tiny and the same for every point tile: one LDS copy per workgroup (their
// VALU dots would otherwise wait on an L2 round trip per row)
template <int NW>
__device__ __forceinline__ void chain_stage_smalls(const ChainDesc& d, const f32x4* __restrict__ wbuf, f32x4* smallw) {
  const int tid = threadIdx.x;
  for (int l = 0; l < d.n_layers; ++l)
    if (d.layers[l].kind == 1) {
      const int n4 = d.layers[l].n_out_tiles * (d.layers[l].kA_rows + d.layers[l].kB_rows) * 2;
      for (int i = tid; i < n4; i += NW * 64) smallw[d.layers[l].dst_row0 + i] = wbuf[d.layers[l].w_off + i];
    }
}

// One point tile through one layer program.  `resident`: the input image already stands in rows [in_row0, +in_rows) (written by the
// caller: the fused kernel below puts the straight-through rows there); an output slot whose pointer is null is not written.
template <int NW>
__device__ __forceinline__ void chain_tile(const ChainDesc& d, const f32x4* __restrict__ wbuf, const float* __restrict__ in, const long N,
                                           const OutPtrs& outs, f32x4* lds, ChainSmalls* sm, f32x4* smallw, const long tile,
                                           const bool resident, f32x4 (&pre)[4], int& pre_for) {
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: let the compiler know
  {
    const long p0 = tile << 5;
    const long pt = (p0 + p < N) ? p0 + p : N - 1;
    // ---------------- input image (also re-loadable later in the program: kind 2) ----------------
    auto load_input = [&](const int row0) {
      if (d.in_mode == 1) {                       // positional encoding of a 3-vector (embedder.py:23-47)
        const float x0 = in[pt * d.in_stride + 0], x1 = in[pt * d.in_stride + 1], x2 = in[pt * d.in_stride + 2];
        for (int r = wave; r < d.in_rows; r += NW) {
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int f = row_feat(r, h, j);
            v[j] = f < d.in_feats ? posenc_feat(f, x0, x1, x2) : 0.f;
          }
          lds[(row0 + r) * 64 + lane] = v;
        }
      } else {                                     // raw features [N, in_feats], row stride in_stride
        const float* xr = in + pt * (long)d.in_stride;
        const bool vec_ok = (d.in_stride & 3) == 0;
        for (int r = wave; r < d.in_rows; r += NW) {
          const int f0 = 32 * (r >> 2) + 8 * (r & 3) + 4 * h;
          f32x4 v;
          if (vec_ok && f0 + 3 < d.in_feats) v = *reinterpret_cast<const f32x4*>(xr + f0);
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (f0 + j < d.in_feats) ? xr[f0 + j] : 0.f;
          }
          // lanes (p,0) and (p,1) hold features f..f+3 and f+4..f+7; the image wants even / odd features
          const float s0 = h ? v[0] : v[1], s1 = h ? v[2] : v[3];
          const float r0 = __shfl_xor(s0, 32), r1 = __shfl_xor(s1, 32);
          f32x4 w;
          if (h == 0) { w[0] = v[0]; w[1] = v[2]; w[2] = r0; w[3] = r1; }
          else { w[0] = r0; w[1] = r1; w[2] = v[1]; w[3] = v[3]; }
          lds[(row0 + r) * 64 + lane] = w;
        }
      }
      __syncthreads();
    };
    if (!resident) load_input(d.in_row0);

    // ---------------- layer program ----------------
    for (int l = 0; l < d.n_layers; ++l) {
      const ChainLayer L = d.layers[l];
      const KSegs ks{L.kA_row0, L.kA_rows, L.kB_row0, L.kB_rows};
      if (L.kind == 2) {                            // input image again (it was not kept resident: LDS rows are the scarce resource)
        load_input(L.dst_row0);
      } else if (L.kind == 0) {
        const f32x4* bp = wbuf + L.b_off;
        const int act = L.act & 0xff, dst = L.dst_row0;
        // bit 8 of `act`: the output goes IN PLACE over one of the layer's own K segments (one output tile per wave at most):
        // every wave keeps its accumulators across a workgroup barrier that separates the last K read from the first write
        const bool late = (L.act & 0x100) != 0;
        f32x16 held;
        // weight fragments run one GEMM layer ahead: the first four of this wave's first tile of the NEXT GEMM layer in which
        // it owns a tile (program order, wrapping into the next point tile) are requested before this layer's epilogue + barrier
        int nl = -1;
        for (int k = 1; k <= d.n_layers && nl < 0; ++k) {
          const int m = (l + k) % d.n_layers;
          if (d.layers[m].kind == 0 && wave < d.layers[m].n_out_tiles && d.layers[m].kA_rows + d.layers[m].kB_rows >= 4) nl = m;
        }
        const f32x4* next_wp = nl < 0 ? nullptr
                                      : wbuf + d.layers[nl].w_off + (size_t)wave * (d.layers[nl].kA_rows + d.layers[nl].kB_rows) * 64 + lane;
        if (wave < L.n_out_tiles && pre_for != l) {             // nothing in flight for this layer yet (first layer of the launch)
          const int ng = L.kA_rows + L.kB_rows;
          const f32x4* wp = wbuf + L.w_off + (size_t)wave * ng * 64 + lane;
#pragma unroll
          for (int i = 0; i < 4; ++i) pre[i] = wp[min(i, ng - 1) * 64];
        }
        if (next_wp != nullptr) pre_for = nl;
        gemm_tiles_chain<NW>(lds, ks, wbuf + L.w_off, L.n_out_tiles, wave, lane, pre, next_wp,
                       [&](int ot, f32x16& acc) { init_bias(bp, ot, lane, acc); },
                       [&](int ot, const f32x16& acc) {
                         if (late) { held = acc; return; }
                         store_tile(act, lds, dst + ot * 4, lane, acc);
                       });
        if (late) {
          __syncthreads();
          if (wave < L.n_out_tiles) store_tile(act, lds, dst + wave * 4, lane, held);
        }
        __syncthreads();
        if (L.out_slot >= 0 && outs.p[L.out_slot] != nullptr) {      // image rows -> [N, out_feats] in HBM (16 B per lane)
          float* o = outs.p[L.out_slot];
          const int ld = outs.ld[L.out_slot];
          const bool vec_ok = (ld & 3) == 0;
          const int n_rows = (L.out_feats + 7) >> 3;
          for (int r = wave; r < n_rows; r += NW) {
            const f32x4 v = lds[(dst + r) * 64 + lane];
            const float s0 = h ? v[0] : v[2], s1 = h ? v[1] : v[3];
            const float r0 = __shfl_xor(s0, 32), r1 = __shfl_xor(s1, 32);
            f32x4 w;
            if (h == 0) { w[0] = v[0]; w[1] = r0; w[2] = v[1]; w[3] = r1; }
            else { w[0] = r0; w[1] = v[2]; w[2] = r1; w[3] = v[3]; }
            const int f0 = 32 * (r >> 2) + 8 * (r & 3) + 4 * h;
            if (p0 + p < N) {
              float* op = o + (p0 + p) * (long)ld + f0;
              if (vec_ok && f0 + 3 < L.out_feats) *reinterpret_cast<f32x4*>(op) = w;
              else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                  if (f0 + j < L.out_feats) op[j] = w[j];
              }
            }
          }
          __syncthreads();
        }
      } else {                                    // <= 4 outputs: VALU row-dots, fixed-order combine
        const int nout = L.n_out_tiles;
        const int n_rows = L.kA_rows + L.kB_rows;
        const f32x4* wimg = smallw + L.dst_row0;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = wave; r < n_rows; r += NW) {
          const int row = r < L.kA_rows ? L.kA_row0 + r : L.kB_row0 + (r - L.kA_rows);
          const f32x4 b = lds[row * 64 + lane];
#pragma unroll
          for (int o = 0; o < 4; ++o)
            if (o < nout) {
              const f32x4 wv = wimg[(o * n_rows + r) * 2 + h];
              s[o] = fmaf(b[0], wv[0], s[o]); s[o] = fmaf(b[1], wv[1], s[o]);
              s[o] = fmaf(b[2], wv[2], s[o]); s[o] = fmaf(b[3], wv[3], s[o]);
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          s[o] += __shfl_xor(s[o], 32);
          if (h == 0) sm->part[(wave * 32 + p) * 4 + o] = s[o];
        }
        __syncthreads();
        if (tid < 128) {
          const int pp = tid & 31, o = tid >> 5;
          if (o < nout && p0 + pp < N) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += sm->part[(w * 32 + pp) * 4 + o];
            // (constant indices only: a dynamically indexed member would push the whole layer record, and with it every
            // loop bound and row number, out of scalar registers into scratch)
            const float b4 = o == 0 ? L.bias4[0] : (o == 1 ? L.bias4[1] : (o == 2 ? L.bias4[2] : L.bias4[3]));
            v = act_rt(L.act, v + b4);
            outs.p[L.out_slot][(p0 + pp) * (long)outs.ld[L.out_slot] + o] = v;
          }
        }
        __syncthreads();
      }
    }
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void mlp_chain_kernel(const ChainDesc d,
                                                                             const f32x4* __restrict__ wbuf,
                                                                             const float* __restrict__ in, const long N,
                                                                             const OutPtrs outs) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  ChainSmalls* sm = reinterpret_cast<ChainSmalls*>(lds + (size_t)d.total_rows * 64);
  f32x4* smallw = lds + (size_t)d.total_rows * 64 + sizeof(ChainSmalls) / sizeof(f32x4);
  const long n_tiles = (N + 31) >> 5;
  chain_stage_smalls<NW>(d, wbuf, smallw);
  __syncthreads();
  f32x4 pre[4];                    // first weight fragments of this wave's next GEMM tile (see the GEMM layers)
  int pre_for = -1;                // layer they belong to
  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x)
    chain_tile<NW>(d, wbuf, in, N, outs, lds, sm, smallw, tile, false, pre, pre_for);
}

// ---- encoder + continuous heads -> VQ step -> VQ heads in ONE launch (vq_nfr.py:534-692, inference, K <= 16, z_dim = 256) --------
// Program A leaves z in LDS rows [z_row0, +32); waves 0 / 1 each take 16 of the tile's 32 points through EXACTLY the arithmetic of
// vq_assign_kernel<1, false, FUSE> (csrc/vq.hip; oracle/vq_strict.c): the A operand of the 16x16x4 MFMA chain is gathered from the
// image (feature f of point p: row f >> 3, lane p + 32 (f & 1), component (f & 7) >> 1), rows are l2-normalised in registers, the
// codebook comes as B fragments + |c|^2 from vqn_vq_codebook_frags (global, L2-resident: 16 KB), and the straight-through rows
// x^ + (q - x^) go back into the image where program B expects its input.  z and the quantised rows never leave the chip; indices,
// the commitment term and the code usage leave as in vqn_vq_quantize_rows.  Bit-identical outputs to the four-launch path (tested).
struct VqTail {
  const f32x4* frags;          // [KT][16][64] float4: frags[kt][t][l][e] = C[16 t + 4 (l >> 4) + e][16 kt + (l & 15)]
  const float* c2;             // [16 KT]
  int K;
  float eps;
  long long* idx;
  float* loss_part;            // [grid]
  float* counts;               // [K]
  float* ste_out;              // [N, 256] straight-through rows for HBM as well, or null
};

// (not inlined: the step is 1 % of a tile's time but holds a whole row per lane in registers; as a real call it has its own register
//  allocation and the layer programs' loops keep theirs)
template <int KT>
__device__ __noinline__ float vq_tail(f32x4* lds, const int z_row0, const int dst_row0, const VqTail vq, const long p0, const long N,
                                      int* hist) {
  float wave_loss = 0.f;
  const int lane = threadIdx.x & 63, col = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* ldsf = reinterpret_cast<float*>(lds);
  const bool active = wave < 2;
  const int pp = 16 * (wave & 1) + col;                            // this lane's point in the tile (rows of the MFMA: col)
  const bool rvalid = active && (p0 + pp) < N;
  f32x4 av[16];
  int kr = 0;
  if (active) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      // features 16 t + 4 q + e, e = 0..3: e & 1 picks the half-wave (lane pp / pp + 32), e >> 1 the component 2 (q & 1) + (e >> 1)
      const int row = z_row0 + (t >> 1) * 4 + 2 * (t & 1) + (q >> 1);
      const float* b = ldsf + ((size_t)row * 64 + pp) * 4 + 2 * (q & 1);
      const f32x2 lo = *reinterpret_cast<const f32x2*>(b), hi = *reinterpret_cast<const f32x2*>(b + 32 * 4);
      av[t] = (f32x4){lo[0], hi[0], lo[1], hi[1]};
    }
    // x^ = x / sqrt(max(sum x^2, eps)): the arithmetic of vq_assign_kernel's FUSE path
    float pr = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (!rvalid) av[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      pr = fmaf(av[i][0], av[i][0], pr); pr = fmaf(av[i][1], av[i][1], pr); pr = fmaf(av[i][2], av[i][2], pr); pr = fmaf(av[i][3], av[i][3], pr);
    }
    pr = pr + __shfl_xor(pr, 16);
    pr = pr + __shfl_xor(pr, 32);
    const float sc = 1.0f / sqrtf(fmaxf(pr, vq.eps));
#pragma unroll
    for (int i = 0; i < 16; ++i) av[i] = av[i] * sc;
    float p = 0.f;
    f32x4 acc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const f32x4 a = av[t];
      p = fmaf(a[0], a[0], p); p = fmaf(a[1], a[1], p); p = fmaf(a[2], a[2], p); p = fmaf(a[3], a[3], p);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const f32x4 b = vq.frags[(kt * 16 + t) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[kt], 0, 0, 0);
      }
      // (this step is 1 % of a tile's time: keep the compiler from hoisting all 16 KT fragment loads -- 64 KT registers, spilled)
      if (KT > 1 || (t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    p = p + __shfl_xor(p, 16);
    p = p + __shfl_xor(p, 32);
    float c2c[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) c2c[kt] = vq.c2[16 * kt + col];
    float best_v[4];
    int best_i[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x2 = __shfl(p, 4 * q + j);
      best_v[j] = INFINITY;
      best_i[j] = 0x7fffffff;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int code = 16 * kt + col;
        const float t1 = x2 - 2.0f * acc[kt][j];
        const float dv = t1 + c2c[kt];
        if (code < vq.K) {
          if (dv < best_v[j] || best_i[j] == 0x7fffffff) { best_v[j] = dv; best_i[j] = code; }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        const float ov = __shfl_xor(best_v[j], m);
        const int oi = __shfl_xor(best_i[j], m);
        const bool take = (oi != 0x7fffffff) && (best_i[j] == 0x7fffffff || ov < best_v[j] || (ov == best_v[j] && oi < best_i[j]));
        if (take) { best_v[j] = ov; best_i[j] = oi; }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int kj = __shfl(best_i[j], (col >> 2) * 16);
      if ((col & 3) == j) kr = kj;
    }
    const int kt_r = kr >> 4, kc_r = kr & 15;
    float lr = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const f32x4 cq = vq.frags[(kt_r * 16 + i) * 64 + 16 * q + kc_r];       // C[16 i + 4 q + e][kr], e = 0..3
      const f32x4 dq = cq - av[i];
      lr = fmaf(dq[0], dq[0], lr); lr = fmaf(dq[1], dq[1], lr); lr = fmaf(dq[2], dq[2], lr); lr = fmaf(dq[3], dq[3], lr);
      av[i] = av[i] + dq;                                            // the straight-through row
      if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    if (!rvalid) lr = 0.f;
    lr = lr + __shfl_xor(lr, 16);
    lr = lr + __shfl_xor(lr, 32);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) lr += __shfl_xor(lr, m);
    wave_loss += lr;
    if (q == 0 && rvalid) {
      vq.idx[p0 + pp] = (long long)kr;
      atomicAdd(&hist[kr], 1);
    }
    if (vq.ste_out != nullptr && rvalid) {
#pragma unroll
      for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(vq.ste_out + (size_t)(p0 + pp) * 256 + 16 * t + 4 * q) = av[t];
    }
  }
  __syncthreads();                                                   // every read of z is done: the rows may be overwritten
  if (active) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int row = dst_row0 + (t >> 1) * 4 + 2 * (t & 1) + (q >> 1);
      float* b = ldsf + ((size_t)row * 64 + pp) * 4 + 2 * (q & 1);
      *reinterpret_cast<f32x2*>(b) = (f32x2){av[t][0], av[t][2]};
      *reinterpret_cast<f32x2*>(b + 32 * 4) = (f32x2){av[t][1], av[t][3]};
    }
  }
  __syncthreads();
  return wave_loss;
}

template <int KT>
__global__ __launch_bounds__(256, 2) void mlp_chain_vq_kernel(const ChainDesc da, const f32x4* __restrict__ wa, const ChainDesc db,
                                                              const f32x4* __restrict__ wb, const float* __restrict__ in, const long N,
                                                              const OutPtrs oa, const OutPtrs ob, const int z_row0, const VqTail vq) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  const int rows = max(da.total_rows, db.total_rows);
  ChainSmalls* sm = reinterpret_cast<ChainSmalls*>(lds + (size_t)rows * 64);
  // ONE region for the <= 4-output layers' weight images, re-staged before each program of each tile (11 KB from L2): with a copy
  // per program the workgroup needs 90 KB of LDS and the CU holds one workgroup instead of two.  A program's first layer is a GEMM
  // layer, whose barrier orders the staging before the first use; the last reader of the previous image ended with a barrier.
  f32x4* smallw = lds + (size_t)rows * 64 + sizeof(ChainSmalls) / sizeof(f32x4);
  int* hist = reinterpret_cast<int*>(smallw + max(da.small_w4, db.small_w4));        // [16 KT] code usage of this workgroup, then 4 floats of wave sums
  const long n_tiles = (N + 31) >> 5;
  if (threadIdx.x < 16 * KT) hist[threadIdx.x] = 0;
  __syncthreads();
  f32x4 pre_a[4], pre_b[4];
  int pre_for_a = -1, pre_for_b = -1;
  float wave_loss = 0.f;
  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    chain_stage_smalls<4>(da, wa, smallw);
    chain_tile<4>(da, wa, in, N, oa, lds, sm, smallw, tile, false, pre_a, pre_for_a);
    chain_stage_smalls<4>(db, wb, smallw);
    wave_loss += vq_tail<KT>(lds, z_row0, db.in_row0, vq, tile << 5, N, hist);
    chain_tile<4>(db, wb, nullptr, N, ob, lds, sm, smallw, tile, true, pre_b, pre_for_b);
  }
  float* wsum = reinterpret_cast<float*>(hist + 16 * KT);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wsum[wave] = wave_loss;
  __syncthreads();
  if (threadIdx.x == 0) vq.loss_part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
  if ((int)threadIdx.x < vq.K && hist[threadIdx.x]) atomicAdd(&vq.counts[threadIdx.x], (float)hist[threadIdx.x]);
}

int check_desc(const ChainDesc& d) {
  if (d.n_layers < 1 || d.n_layers > VQN_CHAIN_MAX_LAYERS) return 1;
  if (d.n_waves != 4 && d.n_waves != 8) return 2;
  if (d.total_rows < 1 || d.small_w4 < 0 || (size_t)d.total_rows * 1024 + sizeof(ChainSmalls) + (size_t)d.small_w4 * 16 > 160 * 1024) return 3;
  if (d.in_rows < 1 || d.in_row0 < 0 || d.in_row0 + d.in_rows > d.total_rows) return 4;
  if (d.in_mode == 1 && (d.in_feats != 3 + 6 * d.n_freqs || d.n_freqs > 16)) return 5;
  if (d.in_feats < 1 || d.in_feats > 8 * d.in_rows || d.in_stride < 1) return 6;
  for (int l = 0; l < d.n_layers; ++l) {
    const ChainLayer& L = d.layers[l];
    if (L.kind == 2) {
      if (L.dst_row0 < 0 || L.dst_row0 + d.in_rows > d.total_rows) return 20;
      continue;
    }
    if (L.kA_rows < 0 || L.kB_rows < 0 || L.kA_rows + L.kB_rows < 1) return 10;
    if (L.kA_row0 < 0 || L.kA_row0 + L.kA_rows > d.total_rows) return 11;
    if (L.kB_rows > 0 && (L.kB_row0 < 0 || L.kB_row0 + L.kB_rows > d.total_rows)) return 12;
    if (L.out_slot >= VQN_CHAIN_MAX_OUTS) return 13;
    if (L.kind == 0) {
      if (L.n_out_tiles < 1 || L.dst_row0 < 0 || L.dst_row0 + 4 * L.n_out_tiles > d.total_rows) return 14;
      if (L.out_slot >= 0 && (L.out_feats < 1 || L.out_feats > 32 * L.n_out_tiles)) return 15;
      // a layer must not overwrite its own K rows -- unless it writes after a barrier (bit 8 of act), one tile per wave
      const int d0 = L.dst_row0, d1 = L.dst_row0 + 4 * L.n_out_tiles;
      if (L.act & 0x100) {
        if (L.n_out_tiles > d.n_waves) return 22;
      } else {
        if (d0 < L.kA_row0 + L.kA_rows && L.kA_row0 < d1) return 16;
        if (L.kB_rows > 0 && d0 < L.kB_row0 + L.kB_rows && L.kB_row0 < d1) return 17;
      }
    } else if (L.kind == 1) {
      if (L.n_out_tiles < 1 || L.n_out_tiles > 4 || L.out_slot < 0) return 18;
      if (L.dst_row0 < 0 || L.dst_row0 + L.n_out_tiles * (L.kA_rows + L.kB_rows) * 2 > d.small_w4) return 21;
    } else return 19;
  }
  return 0;
}

}  // namespace

int vqn_internal_finish_loss(const float* part, int n, float scale, float* loss, hipStream_t s);      // csrc/vq.hip

extern "C" int vqn_mlp_chain_vq_fwd(const int32_t* desc_a, const float* wbuf_a, const int32_t* desc_b, const float* wbuf_b,
                                    const float* in, int64_t N, float* const* outs_a, const int32_t* ld_a, float* const* outs_b,
                                    const int32_t* ld_b, const float* cb_frags, int K, float eps, float loss_scale, int64_t* idx,
                                    float* ste, float* loss, float* counts, float* ws, void* stream) {
  VQN_CHECK_ARG(desc_a && wbuf_a && desc_b && wbuf_b && outs_a && ld_a && outs_b && ld_b, "descriptors, packs and output tables must be non-null");
  VQN_CHECK_ARG(cb_frags && loss && counts && ws, "cb_frags, loss, counts and ws must be non-null");
  VQN_CHECK_ARG(N >= 0 && K >= 1, "N >= 0, K >= 1");
  VQN_CHECK_SHAPE(K <= 64, "K <= 64 (larger codebooks run the separate launches)");
  const int KT = K <= 16 ? 1 : (K <= 32 ? 2 : 4);
  hipStream_t s = (hipStream_t)stream;
  VQN_HIP(hipMemsetAsync(counts, 0, sizeof(float) * K, s));
  if (N == 0) {                                         /* mean over nothing: the reference yields NaN (0 / 0) */
    VQN_HIP(hipMemsetD32Async((hipDeviceptr_t)loss, 0x7fc00000, 1, s));     // quiet NaN, written on the device (capturable)
    return VQN_OK;
  }
  VQN_CHECK_ARG(in != nullptr && idx != nullptr, "in and idx must be non-null");
  VQN_CHECK_ARG(ste == nullptr || ((uintptr_t)ste & 15) == 0, "ste must be 16-byte aligned");
  ChainDesc da, db;
  memcpy(&da, desc_a, sizeof(ChainDesc));
  memcpy(&db, desc_b, sizeof(ChainDesc));
  int bad = check_desc(da);
  if (!bad) bad = check_desc(db) ? 100 + check_desc(db) : 0;
  if (bad) {
    vqn_set_error("vqn_mlp_chain_vq_fwd: unsupported shape: invalid chain descriptor (check %d)", bad);
    return VQN_ESHAPE;
  }
  VQN_CHECK_SHAPE(da.n_waves == 4 && db.n_waves == 4, "both programs must be 4-wave programs");
  int z_row0 = -1;
  for (int l = 0; l < da.n_layers; ++l)
    if (da.layers[l].kind == 0 && da.layers[l].out_slot == 0 && da.layers[l].out_feats == 256) z_row0 = da.layers[l].dst_row0;
  VQN_CHECK_SHAPE(z_row0 >= 0, "program A must leave a 256-feature z in LDS through output slot 0");
  VQN_CHECK_SHAPE(db.in_mode == 0 && db.in_feats == 256 && db.in_rows == 32, "program B must take a raw 256-feature input");
  for (int l = 0; l < db.n_layers; ++l) VQN_CHECK_SHAPE(db.layers[l].kind != 2, "program B must keep its input resident");
  VQN_CHECK_SHAPE(da.layers[0].kind == 0 && db.layers[0].kind == 0, "both programs must start with a GEMM layer");
  // z must survive program A to its end: no later layer of A may write over it
  {
    bool seen = false;
    for (int l = 0; l < da.n_layers; ++l) {
      const ChainLayer& L = da.layers[l];
      if (seen && L.kind == 0) {
        const int d0 = L.dst_row0, d1 = L.dst_row0 + 4 * L.n_out_tiles;
        VQN_CHECK_SHAPE(!(d0 < z_row0 + 32 && z_row0 < d1), "a layer of program A overwrites z before the VQ step");
      }
      if (L.kind == 2) VQN_CHECK_SHAPE(!seen || !(L.dst_row0 < z_row0 + 32 && z_row0 < L.dst_row0 + da.in_rows), "program A reloads its input over z");
      if (L.kind == 0 && L.out_slot == 0) seen = true;
    }
  }
  OutPtrs oa, ob;
  for (int i = 0; i < VQN_CHAIN_MAX_OUTS; ++i) { oa.p[i] = outs_a[i]; oa.ld[i] = ld_a[i]; ob.p[i] = outs_b[i]; ob.ld[i] = ld_b[i]; }
  for (int which = 0; which < 2; ++which) {
    const ChainDesc& d = which ? db : da;
    const OutPtrs& o = which ? ob : oa;
    for (int l = 0; l < d.n_layers; ++l) {
      const int sl = d.layers[l].out_slot;
      if (sl < 0) continue;
      if (which == 0 && sl == 0 && o.p[0] == nullptr) continue;                 // z stays on the chip
      VQN_CHECK_ARG(o.p[sl] != nullptr, "an output a descriptor writes is null");
      VQN_CHECK_ARG(o.ld[sl] >= (d.layers[l].kind == 0 ? d.layers[l].out_feats : d.layers[l].n_out_tiles),
                    "output leading dimension smaller than the layer's width");
      if (d.layers[l].kind == 0 && (o.ld[sl] & 3) == 0)
        VQN_CHECK_ARG(((uintptr_t)o.p[sl] & 15) == 0, "outputs with ld % 4 == 0 must be 16-byte aligned");
    }
  }
  const int rows = da.total_rows > db.total_rows ? da.total_rows : db.total_rows;
  const size_t lds = (size_t)rows * 1024 + sizeof(ChainSmalls) + (size_t)(da.small_w4 > db.small_w4 ? da.small_w4 : db.small_w4) * 16 + (size_t)KT * 16 * 4 + 4 * 4;
  VQN_CHECK_SHAPE(lds <= 160 * 1024, "the two programs do not fit in 160 KB of LDS");
  const long n_tiles = (N + 31) / 32;
  const void* kern = KT == 1 ? (const void*)mlp_chain_vq_kernel<1> : (KT == 2 ? (const void*)mlp_chain_vq_kernel<2> : (const void*)mlp_chain_vq_kernel<4>);
  if (lds > 64 * 1024) VQN_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int per_cu = (2 * lds <= 160 * 1024) ? 2 : 1;
  long grid = (long)vqn_num_cus() * per_cu;
  if (grid > n_tiles) grid = n_tiles;
  VQN_CHECK_SHAPE(grid + 1 <= VQN_QUANT_WS_FLOATS, "workspace too small for this device");
  VqTail vq;
  vq.frags = reinterpret_cast<const f32x4*>(cb_frags);
  vq.c2 = cb_frags + (size_t)KT * 16 * 64 * 4;
  vq.K = K; vq.eps = eps;
  vq.idx = reinterpret_cast<long long*>(idx);
  vq.loss_part = ws + 1;
  vq.counts = counts;
  vq.ste_out = ste;
#define VQN_FRONT(KT_)                                                                                                              \
  hipLaunchKernelGGL(mlp_chain_vq_kernel<KT_>, dim3((unsigned)grid), dim3(256), lds, s, da, reinterpret_cast<const f32x4*>(wbuf_a), db, \
                     reinterpret_cast<const f32x4*>(wbuf_b), in, (long)N, oa, ob, z_row0, vq)
  if (KT == 1) VQN_FRONT(1);
  else if (KT == 2) VQN_FRONT(2);
  else VQN_FRONT(4);
#undef VQN_FRONT
  VQN_LAUNCH_CHECK();
  return vqn_internal_finish_loss(ws + 1, (int)grid, loss_scale, loss, s);
}

extern "C" int vqn_mlp_chain_fwd(const int32_t* desc, const float* wbuf, const float* in, int64_t N, float* out0,
                                 int ld0, float* out1, int ld1, float* out2, int ld2, float* out3, int ld3,
                                 void* stream) {
  VQN_CHECK_ARG(desc && wbuf, "desc and wbuf must be non-null");
  VQN_CHECK_ARG(N >= 0, "N >= 0");
  if (N == 0) return VQN_OK;
  VQN_CHECK_ARG(in != nullptr, "in must be non-null");
  ChainDesc d;
  memcpy(&d, desc, sizeof(ChainDesc));
  const int bad = check_desc(d);
  if (bad) {
    vqn_set_error("vqn_mlp_chain_fwd: unsupported shape: invalid chain descriptor (check %d)", bad);
    return VQN_ESHAPE;
  }
  OutPtrs o;
  o.p[0] = out0; o.p[1] = out1; o.p[2] = out2; o.p[3] = out3;
  o.ld[0] = ld0; o.ld[1] = ld1; o.ld[2] = ld2; o.ld[3] = ld3;
  for (int l = 0; l < d.n_layers; ++l) {
    const int s = d.layers[l].out_slot;
    if (s >= 0) {
      VQN_CHECK_ARG(o.p[s] != nullptr, "an output the descriptor writes is null");
      VQN_CHECK_ARG(o.ld[s] >= (d.layers[l].kind == 0 ? d.layers[l].out_feats : d.layers[l].n_out_tiles),
                    "output leading dimension smaller than the layer's width");
      if (d.layers[l].kind == 0 && (o.ld[s] & 3) == 0)
        VQN_CHECK_ARG(((uintptr_t)o.p[s] & 15) == 0, "outputs with ld % 4 == 0 must be 16-byte aligned");
    }
  }
  if (d.in_mode == 0 && (d.in_stride & 3) == 0) VQN_CHECK_ARG(((uintptr_t)in & 15) == 0, "in must be 16-byte aligned");
  const size_t lds = (size_t)d.total_rows * 1024 + sizeof(ChainSmalls) + (size_t)d.small_w4 * 16;
  const long n_tiles = (N + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  if (d.n_waves == 4) {
    if (lds > 64 * 1024)
      VQN_HIP(hipFuncSetAttribute((const void*)mlp_chain_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = (2 * lds <= 160 * 1024) ? 2 : 1;
    long grid = (long)vqn_num_cus() * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(mlp_chain_kernel<4>, dim3((unsigned)grid), dim3(256), lds, s, d,
                       reinterpret_cast<const f32x4*>(wbuf), in, (long)N, o);
  } else {
    if (lds > 64 * 1024)
      VQN_HIP(hipFuncSetAttribute((const void*)mlp_chain_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long grid = (long)vqn_num_cus();
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(mlp_chain_kernel<8>, dim3((unsigned)grid), dim3(512), lds, s, d,
                       reinterpret_cast<const f32x4*>(wbuf), in, (long)N, o);
  }
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
