// Weight packs + descriptors of the fused NeuS kernels, built in C from plain row-major matrices (C ABI: vqn_neus_pack_*).
//
// The kernels (csrc/neus_mlp.hip, csrc/neus_mlp_f16s.hip) consume the network weights as MFMA A-fragments in the order of
// eng::gemm_tiles (csrc/mlp_prims.h) and a flat int32 descriptor (include/vqn_neus_desc.h).  A pack is a pure GATHER of the
// effective weight matrices W_l [out_l, in_l] (weight-norm already applied: vqn_weight_norm_fwd, fields.py:65-66) and biases:
//
//     pack[out_tile][k_group][lane][j] = M[32 ot + phi(lane & 31)][col(k_group, j, lane >> 5)],   phi(i) = 2 (i & 3) + 8 (i >> 3) + ((i >> 2) & 1)
//
// so this file (i) lays the gather table out on the host once per network shape -- every f32 word of the pack names its source
// matrix and element -- and (ii) applies it with one kernel launch whenever the weights change (after every optimiser step in
// training).  It is the C statement of vqnerf_release_amd/geo/packing.py (SdfPackPlan / ColPackPlan), which the Python host
// uses; tests/test_abi.py holds the two to bit-identical packs and descriptors.  What the packs stand for in the reference:
// SDFNetwork / RenderingNetwork parameters, geo/NeuS-ours2/models/fields.py:9-172.
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "common.h"
#include "vqn_neus_desc.h"
#include "vqnerf_hip.h"

#include "pack_gather.h"

using namespace vqn_pack;

struct vqn_neus_pack {
  int f16s;                  // engine mode: 0 f32, 1 f16 pair, 2 bf16x3
  SdfShape sdf;
  ColShape col;
  bool has_col;
  int32_t sdf_desc[12 + 8 * VQN_MAX_SDF_LAYERS];
  int32_t col_desc[16 + 8 * VQN_MAX_COL_LAYERS];
  int64_t n_sdf, n_col;
  Word* d_words_sdf = nullptr;
  Word* d_words_col = nullptr;
  float* d_wbuf_sdf = nullptr;
  float* d_wbuf_col = nullptr;
};

extern "C" {

int vqn_neus_pack_create(const int32_t* sdf_dims, int sdf_n_lin, int sdf_skip, int multires, float scale, int col_mode, int col_d_hidden,
                         int col_n_layers, int multires_view, int squeeze_out, int f16s, vqn_neus_pack** out) {
  VQN_CHECK_ARG(out != nullptr, "out == NULL");
  *out = nullptr;
  VQN_CHECK_ARG(f16s >= 0 && f16s <= 2, "engine mode: 0 f32, 1 f16 pair (*_f16s entry points), 2 bf16x3 (*_x3 entry points)");
  vqn_neus_pack* p = new vqn_neus_pack();
  p->f16s = f16s;
  p->has_col = col_n_layers > 0;
  int max_tiles = 1;
  std::vector<Word> ws, wc;
  int rc = VQN_OK;
  if (p->has_col) {
    // the LDS buffers of the fused kernel hold the widest layer of BOTH nets (RenderingNetwork.max_tiles())
    max_tiles = tiles_of(col_d_hidden);
  }
  rc = sdf_shape(p->sdf, sdf_dims, sdf_n_lin, sdf_skip, multires, scale, max_tiles, p->f16s);
  if (rc == VQN_OK && p->has_col) {
    const int d_feature = p->sdf.feat_out;
    rc = col_shape(p->col, d_feature, col_mode, col_d_hidden, col_n_layers, 3, multires_view, squeeze_out, p->sdf.tiles[sdf_n_lin - 1], p->f16s);
  }
  if (rc != VQN_OK) { delete p; return rc; }
  sdf_plan(p->sdf, true, p->f16s, ws, p->sdf_desc);
  p->n_sdf = (int64_t)ws.size();
  if (p->has_col) col_plan(p->col, p->f16s, wc, p->col_desc);
  else memset(p->col_desc, 0, sizeof(p->col_desc));
  p->n_col = (int64_t)wc.size();
  auto fail = [&](hipError_t e) {
    vqn_set_error("vqn_neus_pack_create: HIP error %d (%s)", (int)e, hipGetErrorString(e));
    vqn_neus_pack_destroy(p);
    return VQN_EHIP;
  };
  hipError_t e;
  if ((e = hipMalloc(&p->d_words_sdf, ws.size() * sizeof(Word))) != hipSuccess) return fail(e);
  if ((e = hipMalloc(&p->d_wbuf_sdf, ws.size() * sizeof(float))) != hipSuccess) return fail(e);
  if ((e = hipMemcpy(p->d_words_sdf, ws.data(), ws.size() * sizeof(Word), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
  if (p->has_col) {
    if ((e = hipMalloc(&p->d_words_col, wc.size() * sizeof(Word))) != hipSuccess) return fail(e);
    if ((e = hipMalloc(&p->d_wbuf_col, wc.size() * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(p->d_words_col, wc.data(), wc.size() * sizeof(Word), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
  }
  *out = p;
  return VQN_OK;
}

int vqn_neus_pack_update(vqn_neus_pack* p, const float* const* sdf_w, const float* const* sdf_b, const float* const* col_w,
                         const float* const* col_b, void* stream) {
  VQN_CHECK_ARG(p != nullptr && sdf_w != nullptr && sdf_b != nullptr, "NULL argument");
  VQN_CHECK_ARG(!p->has_col || (col_w != nullptr && col_b != nullptr), "colour weights missing");
  hipStream_t st = (hipStream_t)stream;
  PtrTable t;
  memset(&t, 0, sizeof(t));
  for (int l = 0; l < p->sdf.n_lin; ++l) {
    VQN_CHECK_ARG(sdf_w[l] != nullptr && sdf_b[l] != nullptr, "NULL SDF layer pointer");
    t.p[2 * l] = sdf_w[l];
    t.p[2 * l + 1] = sdf_b[l];
  }
  pack_gather_kernel<<<(unsigned)((p->n_sdf + 255) / 256), 256, 0, st>>>(p->d_words_sdf, p->n_sdf, t, p->d_wbuf_sdf);
  VQN_LAUNCH_CHECK();
  if (p->has_col) {
    memset(&t, 0, sizeof(t));
    for (int l = 0; l < p->col.n_lin; ++l) {
      VQN_CHECK_ARG(col_w[l] != nullptr && col_b[l] != nullptr, "NULL colour layer pointer");
      t.p[2 * l] = col_w[l];
      t.p[2 * l + 1] = col_b[l];
    }
    pack_gather_kernel<<<(unsigned)((p->n_col + 255) / 256), 256, 0, st>>>(p->d_words_col, p->n_col, t, p->d_wbuf_col);
    VQN_LAUNCH_CHECK();
  }
  return VQN_OK;
}

const int32_t* vqn_neus_pack_sdf_desc(const vqn_neus_pack* p) { return p ? p->sdf_desc : nullptr; }
const int32_t* vqn_neus_pack_col_desc(const vqn_neus_pack* p) { return p ? p->col_desc : nullptr; }
const float* vqn_neus_pack_sdf_wbuf(const vqn_neus_pack* p) { return p ? p->d_wbuf_sdf : nullptr; }
const float* vqn_neus_pack_col_wbuf(const vqn_neus_pack* p) { return p ? p->d_wbuf_col : nullptr; }
int64_t vqn_neus_pack_sdf_floats(const vqn_neus_pack* p) { return p ? p->n_sdf : 0; }
int64_t vqn_neus_pack_col_floats(const vqn_neus_pack* p) { return p ? p->n_col : 0; }

void vqn_neus_pack_destroy(vqn_neus_pack* p) {
  if (!p) return;
  if (p->d_words_sdf) (void)hipFree(p->d_words_sdf);
  if (p->d_words_col) (void)hipFree(p->d_words_col);
  if (p->d_wbuf_sdf) (void)hipFree(p->d_wbuf_sdf);
  if (p->d_wbuf_col) (void)hipFree(p->d_wbuf_col);
  delete p;
}

}  // extern "C"
