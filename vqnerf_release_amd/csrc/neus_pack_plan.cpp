// vqn_neus_sdf_pack_plan / vqn_neus_col_pack_plan (include/vqnerf_hip.h): host-only, plain C++ (built into libvqnerf_hip.so and,
// by g++ -fsanitize=address,undefined, into the sanitizer check of tests/native/).
#include "neus_pack_plan.h"

using namespace vqn_pack;

extern "C" {

int64_t vqn_neus_sdf_pack_plan(const int32_t* dims, int n_lin, int skip, int multires, float scale, int max_tiles, int with_reverse,
                               int f16s, int32_t* desc_out, int32_t* words_out, int64_t words_cap) {
  SdfShape s;
  const int rc = sdf_shape(s, dims, n_lin, skip, multires, scale, max_tiles, f16s);
  if (rc != 0) return rc;
  std::vector<Word> words;
  int32_t desc[12 + 8 * VQN_MAX_SDF_LAYERS];
  sdf_plan(s, with_reverse != 0, f16s, words, desc);
  if (desc_out) memcpy(desc_out, desc, sizeof(desc));
  if (words_out && words_cap >= (int64_t)words.size()) memcpy(words_out, words.data(), words.size() * sizeof(Word));
  return (int64_t)words.size();
}

int64_t vqn_neus_col_pack_plan(int d_feature, int mode, int d_hidden, int n_layers, int d_out, int multires_view, int squeeze_out,
                               int feat_tiles, int f16s, int32_t* desc_out, int32_t* words_out, int64_t words_cap) {
  ColShape c;
  const int rc = col_shape(c, d_feature, mode, d_hidden, n_layers, d_out, multires_view, squeeze_out, feat_tiles, f16s);
  if (rc != 0) return rc;
  std::vector<Word> words;
  int32_t desc[16 + 8 * VQN_MAX_COL_LAYERS];
  col_plan(c, f16s, words, desc);
  if (desc_out) memcpy(desc_out, desc, sizeof(desc));
  if (words_out && words_cap >= (int64_t)words.size()) memcpy(words_out, words.data(), words.size() * sizeof(Word));
  return (int64_t)words.size();
}

}  // extern "C"
