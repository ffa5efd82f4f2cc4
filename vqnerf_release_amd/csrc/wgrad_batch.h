// Problem table of the batched weight-gradient contraction launches (vqn_wgrad_partials_batched): blockIdx.y picks the problem,
// blockIdx.x / gridDim.x keep their meaning (the split over point tiles), so the kernel bodies are the single-problem ones.
#pragma once

constexpr int WG_MAX = 24;
struct WgProblem {
  const float* A; const float* B; float* ws; float* rs;
  int a_tiles, a_t0, a_nt, b_tiles, b_t0, b_nt;
};
struct WgTable { WgProblem p[WG_MAX]; };

// launches the f32-input kernels for `count` problems of one shape class (cls 0: a_nt <= 4 and b_nt <= 4; 1: a_nt <= 4; 2: the rest)
int vqn_wgrad_f32_batched_internal(const WgProblem* probs, int count, int cls, long n_point_tiles, long grid, void* stream);
