// AddressSanitizer / UBSan check of the host-only C++ of the library (sanitizer runs on the CPU build only):
// the pack planner (csrc/neus_pack_plan.{h,cpp}) over a sweep of network shapes, f32 and split-precision, with every gather-table
// entry checked to lie inside its source matrix.  Built and run by tests/test_sanitizers.py:
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=all -Iinclude -Ivqnerf_release_amd/csrc
//       tests/native/pack_plan_asan.cpp vqnerf_release_amd/csrc/neus_pack_plan.cpp vqnerf_release_amd/csrc/error.cpp
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

struct vqn_chain_stack { int32_t kind, n_layers, widths[8], acts[8], skip_at, input, out_slot; };

extern "C" {
int64_t vqn_chain_pack_plan(int in_mode, int in_feats, int n_freqs, int n_stacks, const vqn_chain_stack* stacks, int32_t* desc_out,
                            int32_t* words_out, int64_t words_cap);
int64_t vqn_neus_sdf_pack_plan(const int32_t* dims, int n_lin, int skip, int multires, float scale, int max_tiles, int with_reverse,
                               int f16s, int32_t* desc_out, int32_t* words_out, int64_t words_cap);
int64_t vqn_neus_col_pack_plan(int d_feature, int mode, int d_hidden, int n_layers, int d_out, int multires_view, int squeeze_out,
                               int feat_tiles, int f16s, int32_t* desc_out, int32_t* words_out, int64_t words_cap);
const char* vqn_last_error(void);
}

static int fail(const char* what) {
  fprintf(stderr, "FAIL: %s (%s)\n", what, vqn_last_error());
  return 1;
}

int main() {
  int checked = 0;
  const int hidden[] = {33, 48, 64, 100, 160, 256};
  for (int f16s = 0; f16s < 3; ++f16s)
    for (int h : hidden)
      for (int n_hidden = 1; n_hidden <= 8; n_hidden += 3)
        for (int multires = 2; multires <= 10; multires += 4) {
          const int E = 3 + 6 * multires;
          if (E > 64) continue;
          for (int skip = -1; skip < n_hidden; skip += 2) {
            if (skip == 0) continue;
            if (skip > 0 && h <= E) continue;                       // the layer before the skip would have no outputs
            std::vector<int32_t> dims = {E};
            for (int l = 0; l < n_hidden; ++l) dims.push_back(h);
            dims.push_back(h + 1);
            const int n_lin = (int)dims.size() - 1;
            if (skip >= n_lin - 1) continue;
            int32_t desc[12 + 8 * 12];
            const int64_t n = vqn_neus_sdf_pack_plan(dims.data(), n_lin, skip, multires, 1.0f, 0, 1, f16s, desc, nullptr, 0);
            if (n <= 0 || n % 4) return fail("sdf plan size");
            std::vector<int32_t> words((size_t)n * 4);
            if (vqn_neus_sdf_pack_plan(dims.data(), n_lin, skip, multires, 1.0f, 0, 1, f16s, nullptr, words.data(), n) != n)
              return fail("sdf plan fill");
            for (int64_t i = 0; i < n; ++i) {
              const int32_t src = words[4 * i], i0 = words[4 * i + 1], i1 = words[4 * i + 2], kind = words[4 * i + 3];
              if (src < 0) continue;
              const int id = src & 0xff, l = id / 2;
              if (l >= n_lin) return fail("source layer out of range");
              const int out = (l + 1 == skip) ? dims[l + 1] - dims[0] : dims[l + 1];
              const int64_t lim = (id & 1) ? out : (int64_t)out * dims[l];
              if (i0 >= lim || (kind != 0 && i1 >= lim) || kind < 0 || kind > 5 || (kind != 0 && !f16s) || (f16s == 1 && kind > 2) || (f16s == 2 && kind != 0 && kind < 3))
                return fail("gather index outside its source matrix");
            }
            for (int l = 0; l < n_lin; ++l)
              for (int k = 3; k <= 6; ++k)
                if (desc[12 + 8 * l + k] >= n / 4) return fail("descriptor offset beyond the pack");
            ++checked;
          }
        }
  const int modes[] = {0, 1, 2};
  for (int f16s = 0; f16s < 3; ++f16s)
    for (int mode : modes)
      for (int d_feature : {32, 64, 100, 256})
        for (int h : {48, 64, 256})
          for (int n_layers = 1; n_layers <= 7; n_layers += 2) {
            const int mv = mode == 1 ? 0 : 4;
            int32_t desc[16 + 8 * 8];
            const int ft = (d_feature + 31) / 32;
            const int64_t n = vqn_neus_col_pack_plan(d_feature, mode, h, n_layers, 3, mv, 1, ft, f16s, desc, nullptr, 0);
            if (n <= 0 || n % 4) return fail("colour plan size");
            std::vector<int32_t> words((size_t)n * 4);
            if (vqn_neus_col_pack_plan(d_feature, mode, h, n_layers, 3, mv, 1, ft, f16s, nullptr, words.data(), n) != n)
              return fail("colour plan fill");
            ++checked;
          }
  // the layer-program builder of the Dense-stack kernel: encoder, head families, encoder + heads, odd widths
  for (int w : {64, 96, 128})
    for (int z : {128, 160, 256})
      for (int nf : {4, 10}) {
        vqn_chain_stack st[5] = {};
        st[0] = {0, 4, {w, w, w, w}, {1, 1, 1, 1}, 2, -1, -1};
        st[1] = {0, 3, {w, z, z}, {0, 1, 3}, -1, 0, 0};
        for (int h = 0; h < 3; ++h) st[2 + h] = {1, 3, {z, z / 2, h == 0 ? 3 : 1}, {1, 1, 3}, 1, 1, h + 1};
        for (int n_st : {2, 5}) {
          int32_t desc[16 + 16 * 16];
          const int64_t n = vqn_chain_pack_plan(1, 3 + 6 * nf, nf, n_st, st, desc, nullptr, 0);
          if (n <= 0 || n % 4) return fail("chain plan size");
          std::vector<int32_t> words((size_t)n * 4);
          if (vqn_chain_pack_plan(1, 3 + 6 * nf, nf, n_st, st, nullptr, words.data(), n) != n) return fail("chain plan fill");
          if (desc[6] < 1 || desc[6] > 156) return fail("chain plan rows");
          ++checked;
        }
        vqn_chain_stack hd[3];
        for (int h = 0; h < 3; ++h) hd[h] = {1, 3, {z, z / 2, 3}, {1, 1, 3}, 1, -1, h};
        if (vqn_chain_pack_plan(0, z, 0, 3, hd, nullptr, nullptr, 0) <= 0) return fail("head program");
        ++checked;
      }
  // rejected shapes must fail cleanly, not read out of bounds
  const int32_t bad[] = {39, 64, 64, 65};
  if (vqn_neus_sdf_pack_plan(bad, 3, 2, 6, 1.0f, 0, 1, 0, nullptr, nullptr, 0) != -2) return fail("skip into the last layer accepted");
  if (vqn_neus_sdf_pack_plan(nullptr, 3, -1, 6, 1.0f, 0, 1, 0, nullptr, nullptr, 0) != -1) return fail("NULL dims accepted");
  if (vqn_neus_sdf_pack_plan(bad, 13, -1, 6, 1.0f, 0, 1, 0, nullptr, nullptr, 0) != -2) return fail("13 layers accepted");
  if (vqn_neus_col_pack_plan(64, 3, 64, 2, 3, 4, 1, 2, 0, nullptr, nullptr, 0) != -2) return fail("bad mode accepted");
  printf("pack planner under ASan/UBSan: %d shapes ok\n", checked);
  return 0;
}
