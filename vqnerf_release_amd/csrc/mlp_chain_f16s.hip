// Split-precision twin of mlp_chain.hip: the same descriptor-driven Dense-stack evaluator (same ChainDesc, same
// reference ops: embedder.py:23-47 + mlp.py:24-50 + seq.py:24-38, vq_nfr.py:771-828) on the f16 hi/lo engine of
// mlp_prims_f16s.h -- the "fp16 MFMA path" of BASELINE.json's batched-inference configuration.  Opt-in: results agree
// with the f32 kernel to ~1e-6 relative, not bitwise.  Row counts of every K segment are even (16 features per step);
// the host builds programs and packs for it with ChainBuilder(mode='f16s').
#include "mlp_prims_f16s.h"
#include <type_traits>
#include "vqn_chain_desc.h"

using namespace eng;

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

struct OutPtrs {
  float* p[VQN_CHAIN_MAX_OUTS];
  int ld[VQN_CHAIN_MAX_OUTS];
};

template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void mlp_chain_f16s_kernel(const ChainDesc d,
                                                                                  const f32x4* __restrict__ wbuf,
                                                                                  const float* __restrict__ in, const long N,
                                                                                  const OutPtrs outs) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  ChainSmalls* sm = reinterpret_cast<ChainSmalls*>(lds + (size_t)d.total_rows * 64);
  f32x4* smallw = lds + (size_t)d.total_rows * 64 + sizeof(ChainSmalls) / sizeof(f32x4);
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: let the compiler know
  const long n_tiles = (N + 31) >> 5;
  // the <= 4-output layers' weight images are tiny and the same for every point tile: one LDS copy per workgroup (their
  // VALU dots would otherwise wait on an L2 round trip per row)
  for (int l = 0; l < d.n_layers; ++l)
    if (d.layers[l].kind == 1) {
      const int n4 = d.layers[l].n_out_tiles * (d.layers[l].kA_rows + d.layers[l].kB_rows) * 2;
      for (int i = tid; i < n4; i += NW * 64) smallw[d.layers[l].dst_row0 + i] = wbuf[d.layers[l].w_off + i];
    }
  __syncthreads();

  // this wave's weight stream: the next GEMM layer (program order, wrapping to the next point tile) in which it owns a tile
  auto next_stream = [&](int l, const f32x4*& nwp, int& nnb) {
    nwp = wbuf; nnb = 1;                               // (no such layer: harmless dummy target)
    for (int k = 1; k <= d.n_layers; ++k) {
      const int m = (l + k) % d.n_layers;
      if (d.layers[m].kind == 0 && wave < d.layers[m].n_out_tiles) {
        nnb = (d.layers[m].kA_rows + d.layers[m].kB_rows + 7) >> 3;
        nwp = wbuf + d.layers[m].w_off + (size_t)wave * nnb * 512 + lane;
        return;
      }
    }
  };
  constexpr int RING = 2;
  f32x4 ring[RING][8];
  {
    const f32x4* wp0; int nb0;
    next_stream(d.n_layers - 1, wp0, nb0);            // = the first GEMM layer of the program in which this wave owns a tile
    ring_prime<RING>(ring, wp0, nb0);
  }

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p0 = tile << 5;
    const long pt = (p0 + p < N) ? p0 + p : N - 1;
    auto load_input = [&](const int row0) {
      const int n_steps = d.in_rows >> 1;
#ifdef VQN_DIAG_NO_IN
      if (tile != blockIdx.x) { __syncthreads(); return; }
#endif
      if (d.in_mode == 1) {                       // positional encoding of a 3-vector
        const float x0 = in[pt * d.in_stride + 0], x1 = in[pt * d.in_stride + 1], x2 = in[pt * d.in_stride + 2];
        for (int sl = wave; sl < n_steps; sl += NW) {
          float x[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const int f = step_feat(sl, h, jj);
            x[jj] = f < d.in_feats ? posenc_feat(f, x0, x1, x2) : 0.f;
          }
          f32x4 hi, lo;
          split8(x, hi, lo);
          lds[(row0 + 2 * sl) * 64 + lane] = hi;
          lds[(row0 + 2 * sl + 1) * 64 + lane] = lo;
        }
      } else {                                     // raw features [N, in_feats]: two runs of 4 consecutive floats per lane
        const float* xr = in + pt * (long)d.in_stride;
        const bool vec_ok = (d.in_stride & 3) == 0;
        for (int sl = wave; sl < n_steps; sl += NW) {
          float x[8];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int f0 = 16 * sl + 8 * q + 4 * h;
            if (vec_ok && f0 + 3 < d.in_feats) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(xr + f0);
              x[4 * q] = v[0]; x[4 * q + 1] = v[1]; x[4 * q + 2] = v[2]; x[4 * q + 3] = v[3];
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) x[4 * q + j] = (f0 + j < d.in_feats) ? xr[f0 + j] : 0.f;
            }
          }
          f32x4 hi, lo;
          split8(x, hi, lo);
          lds[(row0 + 2 * sl) * 64 + lane] = hi;
          lds[(row0 + 2 * sl + 1) * 64 + lane] = lo;
        }
      }
      __syncthreads();
    };
    load_input(d.in_row0);

    for (int l = 0; l < d.n_layers; ++l) {
      const ChainLayer L = d.layers[l];
      const KSegs ks{L.kA_row0, L.kA_rows, L.kB_row0, L.kB_rows};
      if (L.kind == 2) {
        load_input(L.dst_row0);
      } else if (L.kind == 0) {
        const f32x4* bp = wbuf + L.b_off;
        const int act = L.act, dst = L.dst_row0;
        float* o = L.out_slot >= 0 ? outs.p[L.out_slot] : nullptr;
        const int ld = L.out_slot >= 0 ? outs.ld[L.out_slot] : 0;
        const bool vec_ok = (ld & 3) == 0;
        const f32x4* nwp; int nnb;
        next_stream(l, nwp, nnb);
        gemm_tiles_f16s_ring<NW, RING>(lds, ks, wbuf + L.w_off, L.n_out_tiles, wave, lane, ring, nwp, nnb,
                            [&](int ot, f32x16& acc) { init_bias_f16s(bp, ot, lane, acc); },
                            [&](int ot, const f32x16& acc1, const f32x16& acc2) {
#pragma unroll
                              for (int s = 0; s < 2; ++s) {
                                float x[8];
                                // the activation is resolved once per half tile, not once per element (scalar branch chains)
                                auto fill = [&](auto act_c) {
#pragma unroll
                                  for (int jj = 0; jj < 8; ++jj) x[jj] = act_fwd<decltype(act_c)::value>(fmaf(acc2[8 * s + jj], LO_INV, acc1[8 * s + jj]));
                                };
                                switch (act) {
                                  case ACT_RELU: fill(std::integral_constant<int, ACT_RELU>{}); break;
                                  case ACT_SIGMOID: fill(std::integral_constant<int, ACT_SIGMOID>{}); break;
                                  case ACT_SOFTPLUS100: fill(std::integral_constant<int, ACT_SOFTPLUS100>{}); break;
                                  default: fill(std::integral_constant<int, ACT_NONE>{}); break;
                                }
                                f32x4 hi, lo;
                                split8(x, hi, lo);
                                lds[(dst + ot * 4 + 2 * s) * 64 + lane] = hi;
                                lds[(dst + ot * 4 + 2 * s + 1) * 64 + lane] = lo;
#ifdef VQN_DIAG_NO_OUT
                                if (o != nullptr && p0 + p < 0) {
#else
                                if (o != nullptr && p0 + p < N) {          // f32 values leave for HBM straight from the registers
#endif
#pragma unroll
                                  for (int q = 0; q < 2; ++q) {
                                    const int f0 = 32 * ot + 16 * s + 8 * q + 4 * h;
                                    float* op = o + (p0 + p) * (long)ld + f0;
                                    if (vec_ok && f0 + 3 < L.out_feats) *reinterpret_cast<f32x4*>(op) = (f32x4){x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
                                    else {
#pragma unroll
                                      for (int j = 0; j < 4; ++j)
                                        if (f0 + j < L.out_feats) op[j] = x[4 * q + j];
                                    }
                                  }
                                }
                              }
                            });
        __syncthreads();
      } else {                                    // <= 4 outputs: VALU dots over the re-joined values, fixed-order combine
        const int nout = L.n_out_tiles;
        const int n_steps = (L.kA_rows + L.kB_rows) >> 1, sA = L.kA_rows >> 1;
        const f32x4* wimg = smallw + L.dst_row0;        // [nout][n_steps][2][8] f32
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q = wave; q < n_steps; q += NW) {
          const int row = q < sA ? L.kA_row0 + 2 * q : L.kB_row0 + 2 * (q - sA);
          float x[8];
          join8(lds[row * 64 + lane], lds[(row + 1) * 64 + lane], x);
#pragma unroll
          for (int o = 0; o < 4; ++o)
            if (o < nout) {
              const f32x4 w0 = wimg[((o * n_steps + q) * 2 + h) * 2], w1 = wimg[((o * n_steps + q) * 2 + h) * 2 + 1];
              s[o] = fmaf(x[0], w0[0], s[o]); s[o] = fmaf(x[1], w0[1], s[o]);
              s[o] = fmaf(x[2], w0[2], s[o]); s[o] = fmaf(x[3], w0[3], s[o]);
              s[o] = fmaf(x[4], w1[0], s[o]); s[o] = fmaf(x[5], w1[1], s[o]);
              s[o] = fmaf(x[6], w1[2], s[o]); s[o] = fmaf(x[7], w1[3], s[o]);
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

int check_desc(const ChainDesc& d) {
  if (d.n_layers < 1 || d.n_layers > VQN_CHAIN_MAX_LAYERS) return 1;
  if (d.n_waves != 4 && d.n_waves != 8) return 2;
  if (d.total_rows < 1 || d.small_w4 < 0 || (size_t)d.total_rows * 1024 + sizeof(ChainSmalls) + (size_t)d.small_w4 * 16 > 160 * 1024) return 3;
  if (d.in_rows < 2 || (d.in_rows & 1) || d.in_row0 < 0 || d.in_row0 + d.in_rows > d.total_rows) return 4;
  if (d.in_mode == 1 && (d.in_feats != 3 + 6 * d.n_freqs || d.n_freqs > 16)) return 5;
  if (d.in_feats < 1 || d.in_feats > 8 * d.in_rows || d.in_stride < 1) return 6;
  for (int l = 0; l < d.n_layers; ++l) {
    const ChainLayer& L = d.layers[l];
    if (L.kind == 2) {
      if (L.dst_row0 < 0 || L.dst_row0 + d.in_rows > d.total_rows) return 20;
      continue;
    }
    if (L.kA_rows < 0 || L.kB_rows < 0 || L.kA_rows + L.kB_rows < 2 || (L.kA_rows & 1) || (L.kB_rows & 1)) return 10;
    if (L.kA_row0 < 0 || L.kA_row0 + L.kA_rows > d.total_rows) return 11;
    if (L.kB_rows > 0 && (L.kB_row0 < 0 || L.kB_row0 + L.kB_rows > d.total_rows)) return 12;
    if (L.out_slot >= VQN_CHAIN_MAX_OUTS) return 13;
    if (L.kind == 0) {
      if (L.n_out_tiles < 1 || L.dst_row0 < 0 || L.dst_row0 + 4 * L.n_out_tiles > d.total_rows) return 14;
      if (L.out_slot >= 0 && (L.out_feats < 1 || L.out_feats > 32 * L.n_out_tiles)) return 15;
      const int d0 = L.dst_row0, d1 = L.dst_row0 + 4 * L.n_out_tiles;
      if (d0 < L.kA_row0 + L.kA_rows && L.kA_row0 < d1) return 16;
      if (L.kB_rows > 0 && d0 < L.kB_row0 + L.kB_rows && L.kB_row0 < d1) return 17;
    } else if (L.kind == 1) {
      if (L.n_out_tiles < 1 || L.n_out_tiles > 4 || L.out_slot < 0) return 18;
      if (L.dst_row0 < 0 || L.dst_row0 + L.n_out_tiles * (L.kA_rows + L.kB_rows) * 2 > d.small_w4) return 21;
    } else return 19;
  }
  return 0;
}

}  // namespace

extern "C" int vqn_mlp_chain_fwd_f16s(const int32_t* desc, const float* wbuf, const float* in, int64_t N, float* out0,
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
    vqn_set_error("vqn_mlp_chain_fwd_f16s: unsupported shape: invalid chain descriptor (check %d)", bad);
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
      VQN_HIP(hipFuncSetAttribute((const void*)mlp_chain_f16s_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = (2 * lds <= 160 * 1024) ? 2 : 1;
    long grid = (long)vqn_num_cus() * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(mlp_chain_f16s_kernel<4>, dim3((unsigned)grid), dim3(256), lds, s, d,
                       reinterpret_cast<const f32x4*>(wbuf), in, (long)N, o);
  } else {
    if (lds > 64 * 1024)
      VQN_HIP(hipFuncSetAttribute((const void*)mlp_chain_f16s_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long grid = (long)vqn_num_cus();
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(mlp_chain_f16s_kernel<8>, dim3((unsigned)grid), dim3(512), lds, s, d,
                       reinterpret_cast<const f32x4*>(wbuf), in, (long)N, o);
  }
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
