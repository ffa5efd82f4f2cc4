// The device half of the C pack builders (csrc/neus_pack.hip, csrc/chain_pack.hip): applies a host-built gather table
// (vqn_pack::Word, csrc/neus_pack_plan.h) to the caller's weight matrices.  Not a public header.
#pragma once
#include "common.h"
#include "neus_pack_plan.h"

namespace {

using vqn_pack::SKIP_SCALE;
using vqn_pack::Word;

struct PtrTable { const float* p[32]; };      // (kernel, bias) pairs of up to 16 layers

__global__ void pack_gather_kernel(const Word* __restrict__ words, int64_t n, PtrTable t, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Word w = words[i];
  if (w.src < 0) { out[i] = 0.0f; return; }
  const float* src = t.p[w.src & 0xff];
  const bool div = (w.src & SKIP_SCALE) != 0;
  auto get = [&](int32_t k) -> float {
    if (k < 0) return 0.0f;
    const float v = src[k];
    // [x, embedding] / sqrt(2) folded into the skip layer's matrix (fields.py:82), as a multiplication by the f32 reciprocal:
    // the arithmetic of `W / math.sqrt(2.0)` on a device tensor (what geo/packing.py does), so both packs agree bit for bit
    return div ? v * (1.0f / 1.41421356237309504880f) : v;
  };
  if (w.kind == 0) { out[i] = get(w.i0); return; }
  if (w.kind >= 3) {                                        // exact bf16x3 split by truncation (geo/packing.py: split3_exact)
    uint32_t b3 = 0;
    for (int e = 0; e < 2; ++e) {
      float r = get(e ? w.i1 : w.i0);
      uint32_t piece = 0;
      for (int k = 0; k <= w.kind - 3; ++k) {
        piece = __float_as_uint(r) & 0xffff0000u;
        r = r - __uint_as_float(piece);
      }
      b3 |= (piece >> 16) << (16 * e);
    }
    out[i] = __uint_as_float(b3);
    return;
  }
  uint32_t bits = 0;
  for (int e = 0; e < 2; ++e) {
    const float v = get(e ? w.i1 : w.i0);
    const _Float16 hi = (_Float16)v;                        // round to nearest even, as torch's .to(float16)
    const _Float16 h = w.kind == 1 ? hi : (_Float16)((v - (float)hi) * 2048.0f);
    uint16_t hb;
    __builtin_memcpy(&hb, &h, 2);
    bits |= (uint32_t)hb << (16 * e);
  }
  out[i] = __uint_as_float(bits);
}

}  // namespace
