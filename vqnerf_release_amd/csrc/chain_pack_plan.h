// Host-only C++ statement of the layer-program builder of the fused Dense-stack kernel (csrc/mlp_chain.hip): what
// vqnerf_release_amd/decomp/packing.py (ChainBuilder / ChainPlan, f32 kernels) does for the Python host, for callers of the C ABI
// (vqn_chain_pack_*).  No HIP dependency: also compiled by g++ under AddressSanitizer (tests/native/).
//
// A program is described by up to VQN_CHAIN_MAX_STACKS Dense stacks (include/vqnerf_hip.h: vqn_chain_stack):
//   kind 0  networks/mlp.py:24-50 -- Dense chain; after layer `skip_at` the output is concat(y, stack input)
//   kind 1  a reflectance head (nfr_unit.py:110-129): three Dense layers, the stack input concatenated into the last one
//           (skip_at = [1]), last width <= 4.  Built in the input-resident form: layer 1 is written in place over layer 0.
// What the stacks stand for in the reference: embedder.py:23-47 + mlp.py:24-50 + seq.py:24-38 as evaluated by
// vq_nfr.py:771-828 (_pred_enc_at, _pred_diff_at, _pred_spec_at, _pred_rough_at).
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <set>
#include <vector>

#include "neus_pack_plan.h"
#include "vqn_chain_desc.h"
#include "vqnerf_hip.h"

namespace vqn_chain {

using vqn_pack::Seg;
using vqn_pack::View;
using vqn_pack::Word;

struct Region {
  int row0 = -1, feats = 0, rows = 0, alloc_rows = 0;
};

struct Layer {
  int kind = 0;                 // 0 GEMM, 1 <= 4 outputs
  int weight = 0;               // index of its (kernel, bias) pair in the caller's arrays
  std::vector<int> segs;        // region ids (1 or 2), in Keras concat order
  int out = 0, act = 0, tiles = 0, dst = -1, over = -1, out_slot = -1;
  std::vector<int> live;        // region ids that must not be overwritten by dst
  int lds_w_off = 0, in_feats = 0;
};

struct Plan {
  std::vector<Region> regions;  // regions[0] = the program input
  std::vector<Layer> layers;
  int in_mode = 0, in_feats = 0, n_freqs = 0, in_stride = 0;
  int total_rows = 0, n_waves = 4, small_w4 = 0;
};

inline int new_region(Plan& p, int feats, int alloc_rows = -1) {
  Region r;
  r.feats = feats;
  r.rows = (feats + 7) / 8;
  r.alloc_rows = alloc_rows >= 0 ? alloc_rows : r.rows;
  p.regions.push_back(r);
  return (int)p.regions.size() - 1;
}

// packing.py: ChainBuilder.dense
inline int dense(Plan& p, int weight, std::vector<int> segs, int out, int act, std::vector<int> keep, int out_slot, int over = -1) {
  const int tiles = (out + 31) / 32;
  Layer L;
  L.kind = 0; L.weight = weight; L.segs = segs; L.out = out; L.tiles = tiles; L.out_slot = out_slot; L.over = over;
  L.act = act | (over >= 0 ? 0x100 : 0);
  L.dst = new_region(p, out, 4 * tiles);
  for (int s : segs)
    if (s != over) L.live.push_back(s);
  for (int k : keep) L.live.push_back(k);
  p.layers.push_back(L);
  return L.dst;
}

// packing.py: ChainBuilder.dense_small
inline void dense_small(Plan& p, int weight, std::vector<int> segs, int n_out, int act, int out_slot) {
  Layer L;
  L.kind = 1; L.weight = weight; L.segs = segs; L.out = n_out; L.act = act; L.tiles = n_out; L.out_slot = out_slot;
  p.layers.push_back(L);
}

// packing.py: ChainBuilder._assign_rows -- depth-first search over "row 0 or right after an already placed region"
inline bool assign_rows(Plan& p) {
  std::vector<int> gemm;
  for (size_t i = 0; i < p.layers.size(); ++i)
    if (p.layers[i].kind == 0) gemm.push_back((int)i);
  int best_rows = -1;
  std::vector<int> best_pos, pos;
  std::vector<int> placed = {0};
  p.regions[0].row0 = 0;
  struct Rec {
    Plan& p; std::vector<int>& gemm; int& best_rows; std::vector<int>& best_pos; std::vector<int>& pos; std::vector<int>& placed;
    void run(size_t i, int top) {
      if (best_rows >= 0 && top >= best_rows) return;
      if (i == gemm.size()) { best_rows = top; best_pos = pos; return; }
      Layer& L = p.layers[gemm[i]];
      const int need = p.regions[L.dst].alloc_rows;
      std::set<int> cset = {0};
      for (int r : placed) cset.insert(p.regions[r].row0 + p.regions[r].alloc_rows);
      std::vector<int> cands(cset.begin(), cset.end());
      if (L.over >= 0) cands = {p.regions[L.over].row0};
      for (int c : cands) {
        bool clash = false;
        for (int r : L.live) {
          const Region& R = p.regions[r];
          if (c < R.row0 + R.alloc_rows && R.row0 < c + need) { clash = true; break; }
        }
        if (clash) continue;
        p.regions[L.dst].row0 = c;
        placed.push_back(L.dst);
        pos.push_back(c);
        run(i + 1, std::max(top, c + need));
        pos.pop_back();
        placed.pop_back();
        p.regions[L.dst].row0 = -1;
      }
    }
  } rec{p, gemm, best_rows, best_pos, pos, placed};
  rec.run(0, p.regions[0].alloc_rows);
  if (best_rows < 0) return false;
  for (size_t i = 0; i < gemm.size(); ++i) p.regions[p.layers[gemm[i]].dst].row0 = best_pos[i];
  p.total_rows = best_rows;
  return true;
}

// Build the plan from the stack description.  Returns 0 / -1 / -2 (vqn_last_error set).
inline int build(Plan& p, int in_mode, int in_feats, int n_freqs, int n_stacks, const vqn_chain_stack* st) {
  VQN_PLAN_CHECK(st != nullptr && n_stacks >= 1 && n_stacks <= VQN_CHAIN_MAX_STACKS, -1, "1 <= n_stacks <= 8 stacks required");
  VQN_PLAN_CHECK(in_mode == 0 || in_mode == 1, -1, "in_mode: 0 raw features, 1 positional encoding of a 3-vector");
  VQN_PLAN_CHECK(in_feats >= 1 && (in_mode == 0 || in_feats == 3 + 6 * n_freqs), -2, "in_feats must be 3 + 6 n_freqs for the positional encoding");
  p.in_mode = in_mode; p.in_feats = in_feats; p.n_freqs = n_freqs; p.in_stride = in_mode == 1 ? 3 : in_feats;
  new_region(p, in_feats);
  std::vector<int> result(n_stacks, -1);          // region holding each stack's output (-1: it left through a small layer)
  int weight = 0, n_slots = 0;
  for (int si = 0; si < n_stacks; ++si) {
    const vqn_chain_stack& S = st[si];
    VQN_PLAN_CHECK(S.n_layers >= 1 && S.n_layers <= 8, -2, "1 <= layers per stack <= 8");
    VQN_PLAN_CHECK(S.input >= -1 && S.input < si && (S.input < 0 || result[S.input] >= 0), -1, "stack input must be -1 or an earlier stack with a resident output");
    VQN_PLAN_CHECK(S.out_slot >= -1 && S.out_slot < VQN_CHAIN_MAX_OUTS, -1, "out_slot must be -1 .. 3");
    for (int i = 0; i < S.n_layers; ++i)
      VQN_PLAN_CHECK(S.widths[i] >= 1 && S.acts[i] >= 0 && S.acts[i] <= 3, -2, "layer width / activation");
    const int x = S.input < 0 ? 0 : result[S.input];
    if (S.out_slot >= 0) n_slots = std::max(n_slots, S.out_slot + 1);
    // outputs of earlier stacks (and the program input) that a LATER stack still reads must survive this stack's layers
    std::vector<int> later;
    for (int k = -1; k < si; ++k) {
      bool needed = false;
      for (int s2 = si + 1; s2 < n_stacks; ++s2) needed = needed || st[s2].input == k;
      if (needed && (k < 0 || result[k] >= 0)) later.push_back(k < 0 ? 0 : result[k]);
    }
    const size_t first_layer = p.layers.size();
    if (S.kind == 1) {
      // nfr_unit.py: _head_program, input-resident form
      VQN_PLAN_CHECK(S.n_layers == 3 && S.widths[2] <= 4 && S.widths[1] <= 128 && S.out_slot >= 0, -2,
                     "a head is three layers, <= 128 wide in the middle, <= 4 outputs, with an output slot");
      const int y0 = dense(p, weight, {x}, S.widths[0], S.acts[0], {x}, -1);
      const int y1 = dense(p, weight + 1, {y0}, S.widths[1], S.acts[1], {x}, -1, y0);
      dense_small(p, weight + 2, {y1, x}, S.widths[2], S.acts[2], S.out_slot);
      weight += 3;
      for (size_t li = first_layer; li < p.layers.size(); ++li)
        for (int r : later) p.layers[li].live.push_back(r);
      continue;
    }
    VQN_PLAN_CHECK(S.kind == 0, -1, "stack kind must be 0 (Dense chain) or 1 (head)");
    VQN_PLAN_CHECK(S.skip_at >= -1 && S.skip_at < S.n_layers, -1, "skip_at");
    // packing.py: ChainBuilder.mlp
    std::vector<int> h = {x};
    int y = -1;
    bool left = false;
    for (int i = 0; i < S.n_layers; ++i) {
      const bool last = i == S.n_layers - 1;
      const bool later_skip = S.skip_at >= 0 && S.skip_at >= i && S.skip_at < S.n_layers - 1;
      std::vector<int> keep;
      if (later_skip) keep.push_back(x);
      if (last && S.widths[i] <= 4 && S.out_slot >= 0) {
        VQN_PLAN_CHECK(S.skip_at != i, -2, "a <= 4-output last layer cannot carry the skip concat");
        dense_small(p, weight + i, h, S.widths[i], S.acts[i], S.out_slot);
        left = true;
        break;
      }
      y = dense(p, weight + i, h, S.widths[i], S.acts[i], keep, last ? S.out_slot : -1);
      h = (S.skip_at == i) ? std::vector<int>{y, x} : std::vector<int>{y};
    }
    weight += S.n_layers;
    result[si] = (left || h.size() != 1) ? -1 : y;
    for (size_t li = first_layer; li < p.layers.size(); ++li)
      for (int r : later) p.layers[li].live.push_back(r);
  }
  VQN_PLAN_CHECK((int)p.layers.size() <= VQN_CHAIN_MAX_LAYERS && n_slots <= VQN_CHAIN_MAX_OUTS, -2, "more than 16 layers or 4 outputs");
  VQN_PLAN_CHECK(assign_rows(p), -2, "no LDS row assignment found");
  // packing.py: ChainPlan.__init__
  int off4 = 0;
  for (Layer& L : p.layers) {
    int rows = 0;
    L.in_feats = 0;
    for (int s : L.segs) { rows += p.regions[s].rows; L.in_feats += p.regions[s].feats; }
    if (L.kind == 1) { L.lds_w_off = off4; off4 += L.out * rows * 2; }
  }
  p.small_w4 = off4;
  const long lds = (long)p.total_rows * 1024 + 8 * 32 * 4 * 4 + 16L * p.small_w4;
  VQN_PLAN_CHECK(lds <= 160 * 1024, -2, "the program does not fit in 160 KB of LDS");
  p.n_waves = 2 * lds <= 160 * 1024 ? 4 : 8;
  return 0;
}

// packing.py: ChainPlan.pack -- the gather table (Keras layout: kernel [in, out], bias [out]) and the descriptor.  The
// <= 4-output layers' biases live IN the descriptor (bias4): `small_bias` lists (descriptor int index, weight id, count) for the
// caller to fill from the device biases.
struct SmallBias { int desc_pos, weight, n; };

inline void plan_words(const Plan& p, std::vector<Word>& words, int32_t* desc, std::vector<SmallBias>& small_bias) {
  memset(desc, 0, sizeof(int32_t) * (16 + 16 * VQN_CHAIN_MAX_LAYERS));
  const Region& in = p.regions[0];
  const int32_t head[10] = {(int32_t)p.layers.size(), p.in_mode, p.in_feats, in.rows, in.row0, p.n_freqs, p.total_rows, p.n_waves,
                            p.in_stride, p.small_w4};
  memcpy(desc, head, sizeof(head));
  for (size_t li = 0; li < p.layers.size(); ++li) {
    const Layer& L = p.layers[li];
    const int32_t srcW = 2 * L.weight, srcB = 2 * L.weight + 1;
    const View M{srcW, L.out, L.in_feats, L.out, 0, 0, true};          // M = kernel^T: M[o][i] = kernel[i][o]
    const int32_t w_off = (int32_t)(words.size() / 4);
    int32_t b_off = -1;
    if (L.kind == 0) {
      std::vector<Seg> segs;
      int base = 0;
      for (int s : L.segs) { segs.push_back({p.regions[s].rows, p.regions[s].feats, base}); base += p.regions[s].feats; }
      vqn_pack::gemm_words(words, M, segs);
      b_off = (int32_t)(words.size() / 4);
      vqn_pack::bias_words(words, srcB, L.out, 0, false);
    } else {
      // packing.py: _rowdot_index_segs -- [n_out][sum rows][2][4]
      for (int o = 0; o < L.out; ++o) {
        int base = 0;
        for (int s : L.segs) {
          const Region& R = p.regions[s];
          for (int r = 0; r < R.rows; ++r)
            for (int h = 0; h < 2; ++h)
              for (int j = 0; j < 4; ++j) {
                const int f = 32 * (r >> 2) + 2 * (4 * (r & 3) + j) + h;
                const int32_t i = f < R.feats ? M.at(o, f + base) : -1;
                words.push_back({i < 0 ? -1 : srcW, i, 0, 0});
              }
          base += R.feats;
        }
      }
      small_bias.push_back({(int)(16 + 16 * li + 12), L.weight, L.out});
    }
    const Region& A = p.regions[L.segs[0]];
    const int kB0 = L.segs.size() == 2 ? p.regions[L.segs[1]].row0 : 0, kB = L.segs.size() == 2 ? p.regions[L.segs[1]].rows : 0;
    const int dst0 = L.kind == 0 ? p.regions[L.dst].row0 : L.lds_w_off;
    const int32_t rec[12] = {L.kind, L.act, L.tiles, A.row0, A.rows, kB0, kB, dst0, w_off, b_off, L.out_slot, L.kind == 0 ? L.out : 0};
    memcpy(&desc[16 + 16 * li], rec, sizeof(rec));
  }
}

}  // namespace vqn_chain
