// Tile-program interpreter for gfx950: runs a host-built list of ops over the LDS activation image of one 32-point
// tile (csrc/mlp_prims.h), persistent over tiles.  It is the training-path engine of the fused NeuS networks: the
// reference obtains these passes from autograd (loss.backward() through geo/NeuS-ours2/models/fields.py:72-107,
// 147-172, with create_graph=True at fields.py:100-106 for the eikonal term); here they are explicit programs
//   forward   : posenc -> SDF layers (saving every activation) -> reverse sweep for d sdf/dx (saving the
//               pre-activation adjoints) -> colour network
//   backward  : colour-network reverse sweep; tangent (JVP) pass of the SDF net along d loss/d normal; reverse
//               sweep carrying the first-order adjoints and the second-order source terms
// built by vqnerf_release_amd/geo/train_programs.py.  Weight gradients (contractions over points) are taken from the
// saved TFMT tensors by csrc/wgrad.hip.  Formats: include/vqn_vm_desc.h.
#include "mlp_prims.h"
#include <type_traits>
#include <vector>
#include "vqn_vm_desc.h"

#ifndef VQN_VM_NB
#define VQN_VM_NB 2          // weight-fragment buffers of the K loops (mlp_prims.h, gemm_tiles_sw)
#endif

using namespace eng;

// In-kernel phase stamps of the DIAGNOSTIC build (make stamps; never shipped): wave 0 of every workgroup accumulates shader-clock
// cycles: [0] op decode + non-GEMM ops, [1] GEMM tile set-up (epilogue-operand fetches issued, accumulators initialised),
// [2] K loops (operand waits included), [3] epilogues (activation derivative, stash stores, LDS write), [4] barrier waits,
// [5] total, [6] workgroups.
#ifdef VQN_STAMPS
__device__ unsigned long long g_stamps_vm[8];
#define VS_DECL unsigned long long vs_[6] = {0, 0, 0, 0, 0, 0}; unsigned long long vs_prev = __builtin_amdgcn_s_memtime(); const unsigned long long vs_begin = vs_prev;
#define VS(i) { const unsigned long long vs_now = __builtin_amdgcn_s_memtime(); vs_[i] += vs_now - vs_prev; vs_prev = vs_now; }
#define VS_FLUSH if (threadIdx.x == 0) { vs_[5] = __builtin_amdgcn_s_memtime() - vs_begin; for (int i_ = 0; i_ < 6; ++i_) atomicAdd(&g_stamps_vm[i_], vs_[i_]); atomicAdd(&g_stamps_vm[6], 1ull); }
extern "C" int vqn_debug_read_stamps_vm(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_vm), sizeof(unsigned long long) * 8) != hipSuccess) return -3;
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_vm), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}
#else
#define VS_DECL
#define VS(i)
#define VS_FLUSH
#endif

namespace {

struct VmTable {
  VmTensor t[VQN_VM_MAX_TENSORS];
};

__device__ __forceinline__ float vm_act(int act, float x) {
  switch (act) {
    case ACT_RELU: return fmaxf(x, 0.f);
    case ACT_SIGMOID: return fast_rcp(1.f + fast_exp(-x));
    case ACT_SOFTPLUS100: return act_fwd<ACT_SOFTPLUS100>(x);
    default: return x;
  }
}
// act'(pre-activation) expressed through the activation OUTPUT o
__device__ __forceinline__ float vm_dact(int act, float o) {
  switch (act) {
    case ACT_RELU: return o > 0.f ? 1.f : 0.f;
    case ACT_SIGMOID: return o * (1.f - o);
    case ACT_SOFTPLUS100: return 1.f - fast_exp(-100.f * o);
    default: return 1.f;
  }
}
// act''/act' through the output
__device__ __forceinline__ float vm_d2ratio(int act, float o) {
  switch (act) {
    case ACT_SIGMOID: return 1.f - 2.f * o;
    case ACT_SOFTPLUS100: return 100.f * fast_exp(-100.f * o);      // 100 (1 - act')
    default: return 0.f;
  }
}

__device__ __forceinline__ float i2f(int v) { return __int_as_float(v); }

// TFMT element (tile, feature-tile ot, feature-in-tile fi, point p)
__device__ __forceinline__ long tf_off(long tile, int tiles_f, int ot, int fi, int p) {
  return ((tile * tiles_f + ot) * 32 + fi) * 32 + p;
}

template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void tile_vm_kernel(const VmDesc* __restrict__ dp,
                                                                           const f32x4* __restrict__ wbuf,
                                                                           const VmTable tab, const long N) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: let the compiler know
  const long n_tiles = (N + 31) >> 5;
  const int n_ops = dp->n_ops;
  float* ldsf = reinterpret_cast<float*>(lds);
  VS_DECL

  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p0 = tile << 5;
    const long pt = (p0 + p < N) ? p0 + p : N - 1;
    const bool pvalid = p0 + p < N;
    for (int oi = 0; oi < n_ops; ++oi) {
      const VmOp* op = &dp->ops[oi];
      const int kind = op->kind;
      VS(0)
      if (kind == VM_GEMM) {
        const int n_out_tiles = op->p[0];
        const KSegs ks{op->p[1], op->p[2], op->p[3], op->p[4]};
        const int w_off = op->p[5], b_off = op->p[6], dst = op->p[7], epi = op->p[8], act = op->p[9];
        const int ia1 = op->p[10], ia2 = op->p[11], ist = op->p[12], ist2 = op->p[13], accum = op->p[14];
        const float* a1 = ia1 >= 0 ? tab.t[ia1].ptr : nullptr;
        const float* a2 = ia2 >= 0 ? tab.t[ia2].ptr : nullptr;
        float* st = ist >= 0 ? tab.t[ist].ptr : nullptr;
        float* st2 = ist2 >= 0 ? tab.t[ist2].ptr : nullptr;
#ifdef VQN_DIAG_VM_NO_ST                          // timing-only builds (make diag FLAG=...): results are wrong
        st = nullptr; st2 = nullptr;
#endif
        const int tf1 = ia1 >= 0 ? tab.t[ia1].ld : 0, tf2 = ia2 >= 0 ? tab.t[ia2].ld : 0;
        const int tfs = ist >= 0 ? tab.t[ist].ld : 0, tfs2 = ist2 >= 0 ? tab.t[ist2].ld : 0;
        // a tensor flagged per-workgroup (pad != 0) is a temporary of THIS program: written and read back by the same workgroup for
        // the same tile, so it is addressed by workgroup, not by tile -- [grid] images that stay in L2 / Infinity Cache instead of
        // [n_tiles] images streamed to HBM and back
        const long tl1 = (ia1 >= 0 && tab.t[ia1].pad) ? (long)blockIdx.x : tile, tl2 = (ia2 >= 0 && tab.t[ia2].pad) ? (long)blockIdx.x : tile;
        const long tls = (ist >= 0 && tab.t[ist].pad) ? (long)blockIdx.x : tile, tls2 = (ist2 >= 0 && tab.t[ist2].pad) ? (long)blockIdx.x : tile;
        const f32x4* bp = wbuf + b_off;
        // The op body is instantiated per (epilogue, activation) pair that the shipped programs use: with both as run-time values
        // every accumulator element went through a chain of scalar compares / branches (1,500 branches in the kernel, each
        // epilogue element 6-10 of them with waits inside); a pair not in the table runs the generic form (EPI = ACT = -1).
        auto run_gemm = [&](auto epi_c, auto act_c) {
          constexpr int EPI_C = decltype(epi_c)::value, ACT_C = decltype(act_c)::value;
          const int epi_ = EPI_C >= 0 ? EPI_C : epi, act_ = ACT_C >= 0 ? ACT_C : act;
          // epilogue operands of a tile, fetched before its K loop so that their latency hides under it; two sets: a wave's second
          // tile is set up while the epilogue of its first one is still being issued (gemm_tiles_sw)
          float r1[2][16], r2[2][16];
          auto g_aux = [&](int ot, auto slot) {
            constexpr int S = decltype(slot)::value;
            VS(2)
#ifdef VQN_DIAG_VM_NO_AUX
            for (int e = 0; e < 16; ++e) { r1[S][e] = 0.5f; r2[S][e] = 0.25f; }
#else
            if (epi_ != VM_EPI_ACT) {
#pragma unroll
              for (int e = 0; e < 16; ++e) r1[S][e] = a1[tf_off(tl1, tf1, ot, 8 * (e >> 2) + 2 * (e & 3) + h, p)];
              if (epi_ != VM_EPI_MUL_DACT) {
#pragma unroll
                for (int e = 0; e < 16; ++e) r2[S][e] = a2[tf_off(tl2, tf2, ot, 8 * (e >> 2) + 2 * (e & 3) + h, p)];
              }
            }
#endif
            VS(1)
          };
          auto g_init = [&](int ot, f32x16& acc) {
            VS(2)
            if (accum) init_rows(lds + (dst + ot * 4) * 64, lane, acc);
            else if (b_off >= 0) init_bias(bp, ot, lane, acc);
            else init_zero(acc);
            VS(1)
          };
          auto g_epi = [&](int ot, auto slot, int rq, const f32x16& acc) {
            constexpr int S = decltype(slot)::value;
            VS(2)
            f32x4 y, y2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float a = acc[4 * rq + j];
              float v, v2 = 0.f;
              if (epi_ == VM_EPI_ACT) v = vm_act(act_, a);
              else {
                const float o1 = r1[S][4 * rq + j];
                const float d1 = vm_dact(act_, o1);
                if (epi_ == VM_EPI_MUL_DACT) v = a * d1;
                else if (epi_ == VM_EPI_TANGENT) {
                  v = a * d1;
                  v2 = r2[S][4 * rq + j] * a * vm_d2ratio(act_, o1);
                } else v = a * d1 + r2[S][4 * rq + j];
              }
              y[j] = v;
              y2[j] = v2;
            }
            if (st) {
#pragma unroll
              for (int j = 0; j < 4; ++j) st[tf_off(tls, tfs, ot, 8 * rq + 2 * j + h, p)] = pvalid ? y[j] : 0.f;
            }
            if (epi_ == VM_EPI_TANGENT && st2) {
#pragma unroll
              for (int j = 0; j < 4; ++j) st2[tf_off(tls2, tfs2, ot, 8 * rq + 2 * j + h, p)] = pvalid ? y2[j] : 0.f;
            }
            if (dst >= 0) lds[(dst + ot * 4 + rq) * 64 + lane] = y;
            VS(3)
          };
          using S0 = std::integral_constant<int, 0>;
          using S1 = std::integral_constant<int, 1>;
          gemm_tiles_sw<NW, VQN_VM_NB>(lds, ks, wbuf + w_off, n_out_tiles, wave, lane,
                            [&](int ot, int slot) { if (slot == 0) g_aux(ot, S0{}); else g_aux(ot, S1{}); },
                            [&](int ot, int slot, f32x16& acc) { g_init(ot, acc); },
                            [&](int ot, int slot, int rq, const f32x16& acc) { if (slot == 0) g_epi(ot, S0{}, rq, acc); else g_epi(ot, S1{}, rq, acc); });
        };
#define VM_PAIR(E, A) case (E) * 8 + (A): run_gemm(std::integral_constant<int, (E)>{}, std::integral_constant<int, (A)>{}); break;
        switch (epi * 8 + act) {
          VM_PAIR(VM_EPI_ACT, ACT_NONE)
          VM_PAIR(VM_EPI_ACT, ACT_RELU)
          VM_PAIR(VM_EPI_ACT, ACT_SOFTPLUS100)
          VM_PAIR(VM_EPI_ACT, ACT_SIGMOID)
          VM_PAIR(VM_EPI_MUL_DACT, ACT_RELU)
          VM_PAIR(VM_EPI_MUL_DACT, ACT_SOFTPLUS100)
          VM_PAIR(VM_EPI_MUL_DACT, ACT_SIGMOID)
          VM_PAIR(VM_EPI_TANGENT, ACT_SOFTPLUS100)
          VM_PAIR(VM_EPI_BWD2, ACT_SOFTPLUS100)
          default: run_gemm(std::integral_constant<int, -1>{}, std::integral_constant<int, -1>{}); break;
        }
#undef VM_PAIR
        VS(2)
        __syncthreads();
        VS(4)
      } else if (kind == VM_LD_POSENC || kind == VM_LD_POSENC_JVP) {
        const bool jvp = kind == VM_LD_POSENC_JVP;
        const float* x = tab.t[op->p[0]].ptr;
        const int ldx = tab.t[op->p[0]].ld;
        const int dst = jvp ? op->p[2] : op->p[1], n_freqs = jvp ? op->p[3] : op->p[2], feats = jvp ? op->p[4] : op->p[3];
        const int ist = jvp ? op->p[5] : op->p[4];
        const float scale = i2f(jvp ? op->p[6] : op->p[5]);
        const float x0 = x[pt * ldx] * scale, x1 = x[pt * ldx + 1] * scale, x2 = x[pt * ldx + 2] * scale;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (jvp) {
          const float* v = tab.t[op->p[1]].ptr;
          const int ldv = tab.t[op->p[1]].ld;
          v0 = v[pt * ldv]; v1 = v[pt * ldv + 1]; v2 = v[pt * ldv + 2];
        }
        float* st = ist >= 0 ? tab.t[ist].ptr : nullptr;
        const int tfs = ist >= 0 ? tab.t[ist].ld : 0;
        const int rows = (feats + 7) >> 3;
        (void)n_freqs;
        for (int r = wave; r < rows; r += NW) {
          f32x4 y;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int f = row_feat(r, h, j);
            float val = 0.f;
            if (f < feats) {
              if (!jvp) val = posenc_feat(f, x0, x1, x2);
              else {
                int c;
                const float jac = posenc_jac(f, x0, x1, x2, &c);
                val = jac * (c == 0 ? v0 : (c == 1 ? v1 : v2));
              }
            }
            y[j] = val;
            if (st) st[tf_off(tile, tfs, r >> 2, 8 * (r & 3) + 2 * j + h, p)] = pvalid ? val : 0.f;
          }
          lds[(dst + r) * 64 + lane] = y;
        }
        __syncthreads();
      } else if (kind == VM_LD_T) {
        const float* t = tab.t[op->p[0]].ptr;
        const int tf = tab.t[op->p[0]].ld, dst = op->p[1], rows = op->p[2];
        for (int r = wave; r < rows; r += NW) {
          f32x4 y;
#pragma unroll
          for (int j = 0; j < 4; ++j) y[j] = t[tf_off(tile, tf, r >> 2, 8 * (r & 3) + 2 * j + h, p)];
          lds[(dst + r) * 64 + lane] = y;
        }
        __syncthreads();
      } else if (kind == VM_LD_VEC) {
        const float* t = tab.t[op->p[0]].ptr;
        const int ld = tab.t[op->p[0]].ld, dst = op->p[1], c = op->p[2], ist = op->p[4], f0 = op->p[5];
        const float scale = i2f(op->p[3]);
        float* st = ist >= 0 ? tab.t[ist].ptr : nullptr;
        const int tfs = ist >= 0 ? tab.t[ist].ld : 0;
        const int rows = (f0 + c + 7) >> 3;
        for (int r = wave; r < rows; r += NW) {
          f32x4 y;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int f = row_feat(r, h, j);
            const float val = (f >= f0 && f < f0 + c) ? t[pt * ld + (f - f0)] * scale : 0.f;
            y[j] = val;
            if (st) st[tf_off(tile, tfs, r >> 2, 8 * (r & 3) + 2 * j + h, p)] = pvalid ? val : 0.f;
          }
          lds[(dst + r) * 64 + lane] = y;
        }
        __syncthreads();
      } else if (kind == VM_LD_EXTRAS) {
        // colour-net extras [pts(3), posenc(view dir), normals(3)]  (fields.py:153-154)
        const float* x = tab.t[op->p[0]].ptr;
        const float* dr = tab.t[op->p[1]].ptr;
        const int ldx = tab.t[op->p[0]].ld, ldd = tab.t[op->p[1]].ld;
        const int inrm = op->p[2], dst = op->p[3], nvf = op->p[4], ist = op->p[5], feats = op->p[6];
        const int n_view = nvf > 0 ? 3 + 6 * nvf : 0;
        const float px = x[pt * ldx], py = x[pt * ldx + 1], pz = x[pt * ldx + 2];
        const float dx = dr[pt * ldd], dy = dr[pt * ldd + 1], dz = dr[pt * ldd + 2];
        float n0 = 0.f, n1 = 0.f, n2 = 0.f;
        if (inrm >= 0) {
          const float* nr = tab.t[inrm].ptr;
          const int ldn = tab.t[inrm].ld;
          n0 = nr[pt * ldn]; n1 = nr[pt * ldn + 1]; n2 = nr[pt * ldn + 2];
        }
        float* st = ist >= 0 ? tab.t[ist].ptr : nullptr;
        const int tfs = ist >= 0 ? tab.t[ist].ld : 0;
        const int rows = (feats + 7) >> 3;
        for (int r = wave; r < rows; r += NW) {
          f32x4 y;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int f = row_feat(r, h, j);
            float val = 0.f;
            if (f < 3) val = f == 0 ? px : (f == 1 ? py : pz);
            else if (f < 3 + n_view) val = posenc_feat(f - 3, dx, dy, dz);
            else if (f < feats) { const int c = f - 3 - n_view; val = c == 0 ? n0 : (c == 1 ? n1 : n2); }
            y[j] = val;
            if (st) st[tf_off(tile, tfs, r >> 2, 8 * (r & 3) + 2 * j + h, p)] = pvalid ? val : 0.f;
          }
          lds[(dst + r) * 64 + lane] = y;
        }
        __syncthreads();
      } else if (kind == VM_ST_VEC) {
        const int src = op->p[0], f0 = op->p[1], c = op->p[2], act = op->p[4];
        float* o = tab.t[op->p[3]].ptr;
        const int ld = tab.t[op->p[3]].ld;
        const float scale = i2f(op->p[5]);
        if (tid < 32 * c) {
          const int pp = tid & 31, k = tid >> 5, f = f0 + k;
          const int t = f >> 5, fi = f & 31, hh = fi & 1, rr = fi >> 1;
          const float v = ldsf[(((src + t * 4 + (rr >> 2)) * 64) + pp + 32 * hh) * 4 + (rr & 3)];
          if (p0 + pp < N) o[(p0 + pp) * ld + k] = vm_act(act, v) * scale;
        }
        __syncthreads();
      } else if (kind == VM_POSENC_VJP) {
        const int src = op->p[0], n_freqs = op->p[3];
        const float* x = tab.t[op->p[1]].ptr;
        const int ldx = tab.t[op->p[1]].ld;
        float* o = tab.t[op->p[2]].ptr;
        const int ldo = tab.t[op->p[2]].ld;
        const float scale = i2f(op->p[4]);
        if (tid < 96) {
          const int pp = tid & 31, c = tid >> 5;
          const long q = (p0 + pp < N) ? p0 + pp : N - 1;
          const float x0 = x[q * ldx] * scale, x1 = x[q * ldx + 1] * scale, x2 = x[q * ldx + 2] * scale;
          auto G = [&](int f) {
            const int t = f >> 5, fi = f & 31, hh = fi & 1, rr = fi >> 1;
            return ldsf[(((src + t * 4 + (rr >> 2)) * 64) + pp + 32 * hh) * 4 + (rr & 3)];
          };
          float g = G(c);
          int cc;
          for (int k = 0; k < n_freqs; ++k) {
            const int fs = 3 + 6 * k + c, fc = fs + 3;
            g = fmaf(G(fs), posenc_jac(fs, x0, x1, x2, &cc), g);
            g = fmaf(G(fc), posenc_jac(fc, x0, x1, x2, &cc), g);
          }
          if (p0 + pp < N) o[(p0 + pp) * ldo + c] = g;
        }
        __syncthreads();
      }
    }
  }
  VS(0)
  VS_FLUSH
}

// ---- row-major [N, F] <-> TFMT: one 32 x 32 transpose per (point tile, feature tile) through LDS ------------------
// (the reflectance trainer hands z, the head outputs and their adjoints between the tile programs, which want TFMT, and
// the VQ / shading kernels and torch, which want rows)
template <bool PACK>
__global__ __launch_bounds__(256) void tfmt_convert_kernel(const float* __restrict__ src, float* __restrict__ dst, long N, int F,
                                                           long ld, int tiles_f) {
  __shared__ float sq[32][33];
  const long tile = blockIdx.x;
  const int ft = blockIdx.y, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long tbase = ((tile * tiles_f + ft) * 32) * 32;
  if (PACK) {
#pragma unroll
    for (int r = ty; r < 32; r += 8) {                       // r: point in tile, tx: feature in tile
      const long pt = tile * 32 + r;
      const int f = ft * 32 + tx;
      sq[r][tx] = (pt < N && f < F) ? src[pt * ld + f] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) dst[tbase + r * 32 + tx] = sq[tx][r];        // r: feature, tx: point
  } else {
#pragma unroll
    for (int r = ty; r < 32; r += 8) sq[r][tx] = src[tbase + r * 32 + tx];        // r: feature, tx: point
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const long pt = tile * 32 + r;
      const int f = ft * 32 + tx;
      if (pt < N && f < F) dst[pt * ld + f] = sq[tx][r];
    }
  }
}

// [N, F] rows of incoming adjoints -> TFMT, multiplied on the way by the activation's derivative taken from the layer's SAVED TFMT
// output: D = g act'(y) (sigmoid: (g y) (1 - y), ReLU: g [y > 0], none: g), g == nullptr: zeros.  The top-layer delta of a backward
// program in one launch instead of unpack + three element-wise passes + pack.
__global__ __launch_bounds__(256) void tfmt_pack_delta_kernel(const float* __restrict__ g, const float* __restrict__ Y, float* __restrict__ dst, long N,
                                                              int F, long ld, int tiles_f, int act) {
  __shared__ float sq[32][33];
  const long tile = blockIdx.x;
  const int ft = blockIdx.y, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long tbase = ((tile * tiles_f + ft) * 32) * 32;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {                       // r: point in tile, tx: feature in tile
    const long pt = tile * 32 + r;
    const int f = ft * 32 + tx;
    sq[r][tx] = (g != nullptr && pt < N && f < F) ? g[pt * ld + f] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {                       // r: feature, tx: point
    const float gv = sq[tx][r], y = Y[tbase + r * 32 + tx];
    float v = gv;
    if (act == ACT_SIGMOID) v = __fmul_rn(__fmul_rn(gv, y), __fsub_rn(1.0f, y));
    else if (act == ACT_RELU) v = y > 0.f ? gv : 0.f;
    dst[tbase + r * 32 + tx] = (ft * 32 + r < F && tile * 32 + tx < N) ? v : 0.f;
  }
}

}  // namespace

extern "C" int vqn_tfmt_pack_delta(const float* g, int64_t N, int F, int64_t ldg, const float* y_tfmt, int act, float* t, int tiles_f,
                                   void* stream) {
  VQN_CHECK_ARG(N >= 0 && F >= 1 && (g == nullptr || ldg >= F) && tiles_f * 32 >= F, "N >= 0, 1 <= F <= ldg, tiles_f * 32 >= F");
  VQN_CHECK_ARG(act == ACT_NONE || act == ACT_RELU || act == ACT_SIGMOID, "act: 0 none, 1 relu, 3 sigmoid");
  if (N == 0) return VQN_OK;
  VQN_CHECK_ARG(y_tfmt && t, "null pointer");
  const long n_tiles = (N + 31) / 32;
  VQN_CHECK_SHAPE(tiles_f <= 65535, "at most 65535 feature tiles");
  hipLaunchKernelGGL(tfmt_pack_delta_kernel, dim3((unsigned)n_tiles, (unsigned)tiles_f), dim3(256), 0, (hipStream_t)stream, g, y_tfmt, t, (long)N,
                     F, (long)ldg, tiles_f, act);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_tfmt_pack(const float* x, int64_t N, int F, int64_t ldx, float* t, int tiles_f, void* stream) {
  VQN_CHECK_ARG(N >= 0 && F >= 1 && ldx >= F && tiles_f * 32 >= F, "N >= 0, 1 <= F <= ldx, tiles_f * 32 >= F");
  if (N == 0) return VQN_OK;
  VQN_CHECK_ARG(x && t, "null pointer");
  const long n_tiles = (N + 31) / 32;
  VQN_CHECK_SHAPE(tiles_f <= 65535, "at most 65535 feature tiles");
  hipLaunchKernelGGL((tfmt_convert_kernel<true>), dim3((unsigned)n_tiles, (unsigned)tiles_f), dim3(256), 0, (hipStream_t)stream, x, t,
                     (long)N, F, (long)ldx, tiles_f);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int vqn_tfmt_unpack(const float* t, int tiles_f, int64_t N, int F, float* x, int64_t ldx, void* stream) {
  VQN_CHECK_ARG(N >= 0 && F >= 1 && ldx >= F && tiles_f * 32 >= F, "N >= 0, 1 <= F <= ldx, tiles_f * 32 >= F");
  if (N == 0) return VQN_OK;
  VQN_CHECK_ARG(x && t, "null pointer");
  const long n_tiles = (N + 31) / 32;
  const int ft_used = (F + 31) / 32;
  hipLaunchKernelGGL((tfmt_convert_kernel<false>), dim3((unsigned)n_tiles, (unsigned)ft_used), dim3(256), 0, (hipStream_t)stream, t, x,
                     (long)N, F, (long)ldx, tiles_f);
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}

extern "C" int64_t vqn_tile_program_grid(const int32_t* desc_host, int64_t N) {
  if (desc_host == nullptr || N <= 0) return 0;
  const VmDesc* d = reinterpret_cast<const VmDesc*>(desc_host);
  const size_t lds = (size_t)d->total_rows * 1024;
  const long n_tiles = (N + 31) / 32;
  long grid = (long)vqn_num_cus() * ((d->n_waves == 4 && 2 * lds <= 160 * 1024) ? 2 : 1);
  return grid > n_tiles ? n_tiles : grid;
}

extern "C" int vqn_tile_program(const void* desc_dev, const int32_t* desc_host, const float* wbuf,
                                float* const* tensors, const int32_t* tensor_ld, int n_tensors, int64_t N,
                                void* stream) {
  VQN_CHECK_ARG(desc_dev && desc_host && wbuf && tensors && tensor_ld, "null pointer");
  VQN_CHECK_ARG(N >= 0 && n_tensors >= 0 && n_tensors <= VQN_VM_MAX_TENSORS, "N >= 0, n_tensors <= 96");
  if (N == 0) return VQN_OK;
  const VmDesc* d = reinterpret_cast<const VmDesc*>(desc_host);
  VQN_CHECK_SHAPE(d->n_ops >= 1 && d->n_ops <= VQN_VM_MAX_OPS, "1 <= n_ops <= 96");
  VQN_CHECK_SHAPE(d->n_waves == 4 || d->n_waves == 8, "n_waves must be 4 or 8");
  const size_t lds = (size_t)d->total_rows * 1024;
  VQN_CHECK_SHAPE(d->total_rows >= 1 && lds <= 160 * 1024, "program does not fit in 160 KB of LDS");
  // validate every op against the row budget and the tensor table before anything is launched
  const int32_t* const tensor_ld_arg = tensor_ld;         // negative entries flag per-workgroup tensors (vqn_tile_program_grid)
  std::vector<int32_t> ld_abs(n_tensors);
  for (int i = 0; i < n_tensors; ++i) ld_abs[i] = tensor_ld_arg[i] < 0 ? -tensor_ld_arg[i] : tensor_ld_arg[i];
  tensor_ld = ld_abs.data();                              // the checks below are about sizes
  for (int i = 0; i < d->n_ops; ++i) {
    const VmOp& op = d->ops[i];
    auto row_ok = [&](int r0, int n) { return r0 >= 0 && n >= 0 && r0 + n <= d->total_rows; };
    auto t_ok = [&](int t, bool optional) { return (optional && t < 0) || (t >= 0 && t < n_tensors && tensors[t] != nullptr); };
    bool ok = true;
    switch (op.kind) {
      case VM_GEMM:
        ok = op.p[0] >= 1 && row_ok(op.p[1], op.p[2]) && (op.p[4] == 0 || row_ok(op.p[3], op.p[4])) && op.p[2] + op.p[4] >= 1 &&
             (op.p[7] < 0 || row_ok(op.p[7], 4 * op.p[0])) && t_ok(op.p[10], true) && t_ok(op.p[11], true) &&
             t_ok(op.p[12], true) && t_ok(op.p[13], true) && op.p[8] >= 0 && op.p[8] <= 3 &&
             (op.p[8] == VM_EPI_ACT || op.p[10] >= 0) && ((op.p[8] != VM_EPI_TANGENT && op.p[8] != VM_EPI_BWD2) || op.p[11] >= 0) &&
             (!op.p[14] || op.p[7] >= 0);
        for (int k = 10; ok && k <= 13; ++k)
          if (op.p[k] >= 0) ok = tensor_ld[op.p[k]] >= op.p[0];
        break;
      case VM_LD_POSENC: ok = t_ok(op.p[0], false) && row_ok(op.p[1], (op.p[3] + 7) / 8) && op.p[3] == 3 + 6 * op.p[2] && t_ok(op.p[4], true); break;
      case VM_LD_POSENC_JVP: ok = t_ok(op.p[0], false) && t_ok(op.p[1], false) && row_ok(op.p[2], (op.p[4] + 7) / 8) && op.p[4] == 3 + 6 * op.p[3] && t_ok(op.p[5], true); break;
      case VM_LD_T: ok = t_ok(op.p[0], false) && row_ok(op.p[1], op.p[2]) && tensor_ld[op.p[0]] * 4 >= op.p[2]; break;
      case VM_LD_VEC: ok = t_ok(op.p[0], false) && op.p[2] >= 1 && op.p[2] <= 8 && op.p[5] >= 0 && row_ok(op.p[1], (op.p[5] + op.p[2] + 7) / 8) && t_ok(op.p[4], true) && tensor_ld[op.p[0]] >= op.p[2]; break;
      case VM_LD_EXTRAS: ok = t_ok(op.p[0], false) && t_ok(op.p[1], false) && t_ok(op.p[2], true) && row_ok(op.p[3], (op.p[6] + 7) / 8) && t_ok(op.p[5], true); break;
      case VM_ST_VEC: ok = row_ok(op.p[0], (op.p[1] + op.p[2] + 7) / 8) && op.p[2] >= 1 && op.p[2] <= 4 && t_ok(op.p[3], false) && tensor_ld[op.p[3]] >= op.p[2]; break;
      case VM_POSENC_VJP: ok = row_ok(op.p[0], (3 + 6 * op.p[3] + 7) / 8) && t_ok(op.p[1], false) && t_ok(op.p[2], false) && tensor_ld[op.p[2]] >= 3; break;
      default: ok = false;
    }
    if (!ok) {
      vqn_set_error("vqn_tile_program: unsupported shape: op %d (kind %d) is inconsistent with the row budget / tensor table", i, op.kind);
      return VQN_ESHAPE;
    }
  }
  tensor_ld = tensor_ld_arg;
  // per-workgroup tensors (negative ld) are understood by the GEMM op's aux / store operands only
  for (int i = 0; i < d->n_ops; ++i) {
    const VmOp& op = d->ops[i];
    const int nongemm[3] = {op.kind == VM_LD_T ? op.p[0] : -1, op.kind == VM_LD_VEC ? op.p[4] : -1,
                            op.kind == VM_LD_POSENC ? op.p[4] : (op.kind == VM_LD_POSENC_JVP ? op.p[5] : (op.kind == VM_LD_EXTRAS ? op.p[5] : -1))};
    for (int k = 0; k < 3; ++k)
      if (nongemm[k] >= 0 && nongemm[k] < n_tensors && tensor_ld[nongemm[k]] < 0) {
        vqn_set_error("vqn_tile_program: bad argument: op %d addresses a per-workgroup tensor (negative ld) outside a GEMM aux / store operand", i);
        return VQN_EARG;
      }
  }
  VmTable tab;
  memset(&tab, 0, sizeof(tab));
  for (int i = 0; i < n_tensors; ++i) { tab.t[i].ptr = tensors[i]; tab.t[i].ld = tensor_ld[i] < 0 ? -tensor_ld[i] : tensor_ld[i]; tab.t[i].pad = tensor_ld[i] < 0; }
  const long n_tiles = (N + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  const VmDesc* dd = reinterpret_cast<const VmDesc*>(desc_dev);
  if (d->n_waves == 4) {
    if (lds > 64 * 1024)
      VQN_HIP(hipFuncSetAttribute((const void*)tile_vm_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int per_cu = (2 * lds <= 160 * 1024) ? 2 : 1;
    long grid = (long)vqn_num_cus() * per_cu;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(tile_vm_kernel<4>, dim3((unsigned)grid), dim3(256), lds, s, dd, reinterpret_cast<const f32x4*>(wbuf), tab, (long)N);
  } else {
    if (lds > 64 * 1024)
      VQN_HIP(hipFuncSetAttribute((const void*)tile_vm_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long grid = (long)vqn_num_cus();
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(tile_vm_kernel<8>, dim3((unsigned)grid), dim3(512), lds, s, dd, reinterpret_cast<const f32x4*>(wbuf), tab, (long)N);
  }
  VQN_LAUNCH_CHECK();
  return VQN_OK;
}
