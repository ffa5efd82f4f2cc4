// Host-only half of the C pack builder (no HIP dependency: also compiled by g++ under AddressSanitizer, tests/native/):
// the gather table + descriptor of the fused NeuS kernels' weight packs.  See csrc/neus_pack.hip for what a pack is; this is the
// C statement of vqnerf_release_amd/geo/packing.py (SdfPackPlan / ColPackPlan).
#pragma once
#include <stdint.h>
#include <string.h>

#include <utility>
#include <vector>

#include "vqn_neus_desc.h"

void vqn_set_error(const char* fmt, ...);

#define VQN_PLAN_CHECK(cond, code, msg)                            \
  do {                                                             \
    if (!(cond)) {                                                 \
      vqn_set_error("%s: %s: %s", __func__, (code) == -1 ? "bad argument" : "unsupported shape", msg); \
      return (code);                                               \
    }                                                              \
  } while (0)

namespace vqn_pack {


// one f32 word of the pack: kind 0 = copy src[i0] (src < 0: zero); kind 1 / 2 = the f16 hi / lo halves of the split-precision
// engine for the element pair (i0, i1): hi = f16(w), lo = f16((w - hi) * 2^11), packed low half first; kind 3 / 4 / 5 = the
// bf16 pieces p0 / p1 / p2 of the exact-split engine for the pair (truncation: w = p0 + p1 + p2 exactly), low half first
// Engine modes of the planners (the `f16s` argument of the C ABI, kept under that name): 0 = f32, 1 = f16 pair, 2 = bf16x3.
struct Word {
  int32_t src;      // -1: zero;  2 l: W_l;  2 l + 1: bias_l;  bit 30: the matrix is divided by sqrt(2) (the skip layer, fields.py:82)
  int32_t i0, i1;
  int32_t kind;
};
constexpr int32_t SKIP_SCALE = 1 << 30;

struct Seg { int rows; int n_valid; int base; };     // K segment: LDS rows, number of real features, first source column

inline int phi(int i) { return 2 * (i & 3) + 8 * (i >> 3) + ((i >> 2) & 1); }
inline int tiles_of(int n) { return (n + 31) / 32; }
inline int emb_rows_f32(int n) { return (((n + 1) / 2) + 3) / 4; }
inline int emb_rows_f16s(int n) { return 2 * ((n + 15) / 16); }
inline int emb_rows_x3(int n) { return 3 * ((n + 15) / 16); }
inline int emb_rows_mode(int n, int mode) { return mode == 0 ? emb_rows_f32(n) : (mode == 1 ? emb_rows_f16s(n) : emb_rows_x3(n)); }
inline int rows_per_tile(int mode) { return mode == 2 ? 6 : 4; }

// element (row, col) of the gathered matrix M -> flat index into the source array (or -1)
struct View {
  int32_t src;            // as in Word
  int rows, cols;         // shape of M
  int ld;                 // row stride of the source matrix W [out, in]
  int row0, col0;         // M[r][c] = W[row0 + r][col0 + c]   (transposed: W[col0 + c][row0 + r])
  bool transposed;
  int32_t at(int r, int c) const {
    if (r < 0 || r >= rows || c < 0 || c >= cols) return -1;
    return transposed ? (col0 + c) * ld + (row0 + r) : (row0 + r) * ld + (col0 + c);
  }
};

inline int seg_col(const Seg& s, int f) { return f < s.n_valid ? f + s.base : -1; }

// geo/packing.py: gemm_index -- [n_out_tiles][k rows of all segments][64 lanes][4]
inline void gemm_words(std::vector<Word>& out, const View& M, const std::vector<Seg>& segs) {
  const int nt = tiles_of(M.rows);
  for (int ot = 0; ot < nt; ++ot)
    for (const Seg& s : segs)
      for (int r = 0; r < s.rows; ++r)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const int f = 32 * (r >> 2) + 2 * (4 * (r & 3) + j) + (lane >> 5);
            const int32_t i = M.at(32 * ot + phi(lane & 31), seg_col(s, f));
            out.push_back({i < 0 ? -1 : M.src, i, 0, 0});
          }
}

// geo/packing.py: gemm_index_f16s + split_pack -- [n_out_tiles][steps, padded to whole 4-step blocks][hi | lo][64 lanes][8 halves]
inline void gemm_words_f16s(std::vector<Word>& out, const View& M, const std::vector<Seg>& segs) {
  const int nt = tiles_of(M.rows);
  std::vector<std::pair<const Seg*, int>> steps;            // (segment, local step) or (nullptr, 0) for padding
  for (const Seg& s : segs)
    for (int sl = 0; sl < s.rows / 2; ++sl) steps.push_back({&s, sl});
  while (steps.size() % 4) steps.push_back({nullptr, 0});
  for (int ot = 0; ot < nt; ++ot)
    for (auto& st : steps)
      for (int hl = 0; hl < 2; ++hl)
        for (int lane = 0; lane < 64; ++lane)
          for (int jw = 0; jw < 4; ++jw) {
            int32_t idx[2];
            for (int e = 0; e < 2; ++e) {
              const int jj = 2 * jw + e;
              idx[e] = -1;
              if (st.first) {
                const int f = 16 * st.second + 8 * (jj >> 2) + 4 * (lane >> 5) + (jj & 3);
                idx[e] = M.at(32 * ot + (lane & 31), seg_col(*st.first, f));
              }
            }
            out.push_back({(idx[0] < 0 && idx[1] < 0) ? -1 : M.src, idx[0], idx[1], 1 + hl});
          }
}

// geo/packing.py: gemm_index_x3 + split_pack_x3 -- [n_out_tiles][steps, padded to whole 2-step blocks][p0 | p1 | p2][64 lanes][8 bf16]
inline void gemm_words_x3(std::vector<Word>& out, const View& M, const std::vector<Seg>& segs) {
  const int nt = tiles_of(M.rows);
  std::vector<std::pair<const Seg*, int>> steps;
  for (const Seg& s : segs)
    for (int sl = 0; sl < s.rows / 3; ++sl) steps.push_back({&s, sl});
  while (steps.size() % 2) steps.push_back({nullptr, 0});
  for (int ot = 0; ot < nt; ++ot)
    for (auto& st : steps)
      for (int piece = 0; piece < 3; ++piece)
        for (int lane = 0; lane < 64; ++lane)
          for (int jw = 0; jw < 4; ++jw) {
            int32_t idx[2];
            for (int e = 0; e < 2; ++e) {
              const int jj = 2 * jw + e;
              idx[e] = -1;
              if (st.first) {
                const int f = 16 * st.second + 8 * (jj >> 2) + 4 * (lane >> 5) + (jj & 3);
                idx[e] = M.at(32 * ot + (lane & 31), seg_col(*st.first, f));
              }
            }
            out.push_back({(idx[0] < 0 && idx[1] < 0) ? -1 : M.src, idx[0], idx[1], 3 + piece});
          }
}

// geo/packing.py: bias_index / bias_index_f16s -- [n_tiles][2][16]
inline void bias_words(std::vector<Word>& out, int32_t src, int n_out, int first, bool f16s) {
  for (int ot = 0; ot < tiles_of(n_out); ++ot)
    for (int h = 0; h < 2; ++h)
      for (int k = 0; k < 16; ++k) {
        const int f = f16s ? 32 * ot + (k & 3) + 8 * (k >> 2) + 4 * h : 32 * ot + 2 * k + h;
        out.push_back({f < n_out ? src : -1, first + f, 0, 0});
      }
}

// geo/packing.py: rowdot_index / rowdot_index_f16s -- f32 images of whole rows in activation-image order
inline void rowdot_words(std::vector<Word>& out, const View& M, int n_rows, int mode) {
  for (int o = 0; o < M.rows; ++o) {
    if (mode == 0) {
      for (int r = 0; r < n_rows; ++r)
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < 4; ++j) {
            const int32_t i = M.at(o, 32 * (r >> 2) + 2 * (4 * (r & 3) + j) + h);
            out.push_back({i < 0 ? -1 : M.src, i, 0, 0});
          }
    } else {
      for (int sl = 0; sl < n_rows / (mode == 2 ? 3 : 2); ++sl)
        for (int h = 0; h < 2; ++h)
          for (int jj = 0; jj < 8; ++jj) {
            const int32_t i = M.at(o, 16 * sl + 8 * (jj >> 2) + 4 * h + (jj & 3));
            out.push_back({i < 0 ? -1 : M.src, i, 0, 0});
          }
    }
  }
}

struct SdfShape {
  int n_lin, skip, multires, E, emb_rows, max_tiles, feat_out;
  float scale;
  int in_dims[VQN_MAX_SDF_LAYERS], out_dims[VQN_MAX_SDF_LAYERS], tiles[VQN_MAX_SDF_LAYERS];
};

// geo/packing.py: SdfPackPlan.__init__ -- dims = [d0, hidden..., d_out] as in fields.py:24
inline int sdf_shape(SdfShape& s, const int32_t* dims, int n_lin, int skip, int multires, float scale, int max_tiles, int mode) {
  VQN_PLAN_CHECK(mode >= 0 && mode <= 2, -1, "engine mode: 0 f32, 1 f16 pair, 2 bf16x3");
  VQN_PLAN_CHECK(dims != nullptr, -1, "dims == NULL");
  VQN_PLAN_CHECK(n_lin >= 2 && n_lin <= VQN_MAX_SDF_LAYERS, -2, "2 <= n_lin <= 12");
  VQN_PLAN_CHECK(skip == -1 || (skip > 0 && skip < n_lin - 1), -2, "skip layer must be an interior layer (or -1)");
  VQN_PLAN_CHECK(dims[0] == 3 + 6 * multires && dims[0] <= 64 && multires > 0, -2, "dims[0] must be 3 + 6 multires <= 64");
  s.n_lin = n_lin; s.skip = skip; s.multires = multires; s.E = dims[0]; s.scale = scale;
  s.emb_rows = emb_rows_mode(s.E, mode);
  int mt = max_tiles > 0 ? max_tiles : 1;
  for (int l = 0; l < n_lin; ++l) {
    s.in_dims[l] = dims[l];
    s.out_dims[l] = (l + 1 == skip) ? dims[l + 1] - dims[0] : dims[l + 1];      // fields.py:38-41
    VQN_PLAN_CHECK(s.out_dims[l] > 0, -2, "layer width");
    s.tiles[l] = tiles_of(s.out_dims[l]);
  }
  s.feat_out = s.out_dims[n_lin - 1] - 1;
  s.tiles[n_lin - 1] = s.feat_out > 0 ? tiles_of(s.feat_out) : 0;
  for (int l = 0; l < n_lin; ++l) mt = s.tiles[l] > mt ? s.tiles[l] : mt;
  s.max_tiles = mt;
  return 0;
}

// geo/packing.py: SdfPackPlan._build + pack
inline void sdf_plan(const SdfShape& s, bool with_reverse, int mode, std::vector<Word>& words, int32_t* desc) {
  const bool f16s = mode != 0;               // (accumulator-order bias images: shared by the two 16-bit engines)
  const int rpt = rows_per_tile(mode);
  auto gemm = [&](const View& M, const std::vector<Seg>& segs) {
    mode == 0 ? gemm_words(words, M, segs) : (mode == 1 ? gemm_words_f16s(words, M, segs) : gemm_words_x3(words, M, segs));
  };
  memset(desc, 0, sizeof(int32_t) * (12 + 8 * VQN_MAX_SDF_LAYERS));
  int32_t lay[VQN_MAX_SDF_LAYERS][8];
  for (int l = 0; l < s.n_lin; ++l) {
    const int32_t init[8] = {s.tiles[l], 0, 0, -1, -1, -1, -1, 0};
    memcpy(lay[l], init, sizeof(init));
  }
  int32_t last_w_off = -1;
  for (int l = 0; l < s.n_lin; ++l) {
    const int in = s.in_dims[l], out = s.out_dims[l];
    const int rows_prev = l > 0 ? rpt * s.tiles[l - 1] : 0;
    const int prev = l > 0 ? s.out_dims[l - 1] : 0;
    const int32_t srcW = 2 * l | (l == s.skip ? SKIP_SCALE : 0);
    std::vector<Seg> segs;
    if (l == 0) segs = {{s.emb_rows, s.E, 0}};
    else if (l == s.skip) segs = {{rows_prev, prev, 0}, {s.emb_rows, s.E, prev}};
    else segs = {{rows_prev, in, 0}};
    if (l < s.n_lin - 1) {
      lay[l][3] = (int32_t)(words.size() / 4);
      gemm({srcW, out, in, in, 0, 0, false}, segs);
      lay[l][4] = (int32_t)(words.size() / 4);
      bias_words(words, 2 * l + 1, out, 0, f16s);
    } else {
      if (s.feat_out > 0) {                                    // feature rows = rows 1.. of the last layer
        lay[l][3] = (int32_t)(words.size() / 4);
        gemm({srcW, s.feat_out, in, in, 1, 0, false}, segs);
        lay[l][4] = (int32_t)(words.size() / 4);
        bias_words(words, 2 * l + 1, s.feat_out, 1, f16s);
      }
      last_w_off = (int32_t)(words.size() / 4);
      rowdot_words(words, {srcW, 1, in, in, 0, 0, false}, rows_prev, mode);
    }
    if (with_reverse && l < s.n_lin - 1) {
      const std::vector<Seg> ksegs = {{rpt * s.tiles[l], out, 0}};
      if (l >= 1) {                                            // rows = features of the previous activation
        lay[l][5] = (int32_t)(words.size() / 4);
        gemm({srcW, prev, out, in, 0, 0, true}, ksegs);
      }
      if (l == 0 || l == s.skip) {                             // rows = embedding features
        lay[l][6] = (int32_t)(words.size() / 4);
        gemm({srcW, s.E, out, in, l == s.skip ? prev : 0, 0, true}, ksegs);
      }
    }
  }
  // the sdf row's bias rides in the pack (component 0 of one float4)
  const int32_t last_b_off = (int32_t)(words.size() / 4);
  words.push_back({2 * (s.n_lin - 1) + 1, 0, 0, 0});
  for (int i = 0; i < 3; ++i) words.push_back({-1, 0, 0, 0});
  desc[0] = s.n_lin; desc[1] = s.skip; desc[2] = s.multires; desc[3] = s.E; desc[4] = s.emb_rows; desc[5] = s.max_tiles;
  memcpy(&desc[6], &s.scale, 4);
  desc[7] = last_w_off;
  desc[9] = last_b_off;
  for (int l = 0; l < s.n_lin; ++l) memcpy(&desc[12 + 8 * l], lay[l], sizeof(lay[l]));
}

struct ColShape { int n_lin, n_view, has_normal, extra, extra_rows, d_feature, squeeze, feat_tiles; int dims[VQN_MAX_COL_LAYERS + 1], tiles[VQN_MAX_COL_LAYERS]; };

// geo/packing.py: ColPackPlan -- input order [pts, view_embed, normals, feat] (fields.py:147-172); mode: 0 idr, 1 no_view_dir, 2 no_normal
inline int col_shape(ColShape& c, int d_feature, int mode, int d_hidden, int n_layers, int d_out, int multires_view, int squeeze_out, int feat_tiles,
              int engine) {
  VQN_PLAN_CHECK(engine >= 0 && engine <= 2, -1, "engine mode: 0 f32, 1 f16 pair, 2 bf16x3");
  VQN_PLAN_CHECK(mode >= 0 && mode <= 2, -2, "mode: 0 idr, 1 no_view_dir, 2 no_normal");
  c.n_view = (mode == 0 || mode == 2) ? 3 + 6 * multires_view : 0;
  c.has_normal = (mode == 0 || mode == 1) ? 1 : 0;
  c.extra = 3 + c.n_view + 3 * c.has_normal;
  c.extra_rows = emb_rows_mode(c.extra, engine);
  c.d_feature = d_feature; c.squeeze = squeeze_out ? 1 : 0; c.feat_tiles = feat_tiles;
  c.n_lin = n_layers + 1;
  VQN_PLAN_CHECK(c.n_lin >= 2 && c.n_lin <= VQN_MAX_COL_LAYERS && d_out == 3, -2, "2 <= colour layers <= 8, d_out == 3");
  VQN_PLAN_CHECK(feat_tiles == tiles_of(d_feature), -2, "feat_tiles must be ceil(d_feature / 32)");
  c.dims[0] = c.extra + d_feature;
  for (int l = 1; l <= n_layers; ++l) c.dims[l] = d_hidden;
  c.dims[c.n_lin] = d_out;
  for (int l = 0; l < c.n_lin; ++l) c.tiles[l] = tiles_of(c.dims[l + 1]);
  return 0;
}

inline void col_plan(const ColShape& c, int engine, std::vector<Word>& words, int32_t* desc) {
  const bool f16s = engine != 0;
  const int rpt = rows_per_tile(engine);
  memset(desc, 0, sizeof(int32_t) * (16 + 8 * VQN_MAX_COL_LAYERS));
  int32_t lay[VQN_MAX_COL_LAYERS][8];
  for (int l = 0; l < c.n_lin; ++l) {
    const int32_t init[8] = {c.tiles[l], 0, 0, -1, -1, -1, -1, 0};
    memcpy(lay[l], init, sizeof(init));
  }
  for (int l = 0; l < c.n_lin - 1; ++l) {
    std::vector<Seg> segs;
    if (l == 0) segs = {{rpt * c.feat_tiles, c.d_feature, c.extra}, {c.extra_rows, c.extra, 0}};
    else segs = {{rpt * c.tiles[l - 1], c.dims[l], 0}};
    const View M{2 * l, c.dims[l + 1], c.dims[l], c.dims[l], 0, 0, false};
    lay[l][3] = (int32_t)(words.size() / 4);
    engine == 0 ? gemm_words(words, M, segs) : (engine == 1 ? gemm_words_f16s(words, M, segs) : gemm_words_x3(words, M, segs));
    lay[l][4] = (int32_t)(words.size() / 4);
    bias_words(words, 2 * l + 1, c.dims[l + 1], 0, f16s);
  }
  const int L = c.n_lin - 1;
  const int32_t last_w_off = (int32_t)(words.size() / 4);
  rowdot_words(words, {2 * L, c.dims[L + 1], c.dims[L], c.dims[L], 0, 0, false}, rpt * c.tiles[L - 1], engine);
  const int32_t last_b_off = (int32_t)(words.size() / 4);
  for (int i = 0; i < 3; ++i) words.push_back({2 * L + 1, i, 0, 0});
  words.push_back({-1, 0, 0, 0});
  const int32_t head[8] = {c.n_lin, c.n_view, c.has_normal, c.extra, c.extra_rows, 3, c.squeeze, last_w_off};
  memcpy(desc, head, sizeof(head));
  desc[12] = last_b_off;
  for (int l = 0; l < c.n_lin; ++l) memcpy(&desc[16 + 8 * l], lay[l], sizeof(lay[l]));
}


}  // namespace vqn_pack
