// vqn_chain_pack_plan (include/vqnerf_hip.h): host-only, plain C++ (built into libvqnerf_hip.so and, by g++
// -fsanitize=address,undefined, into the sanitizer check of tests/native/).
#include "chain_pack_plan.h"

extern "C" int64_t vqn_chain_pack_plan(int in_mode, int in_feats, int n_freqs, int n_stacks, const vqn_chain_stack* stacks,
                                       int32_t* desc_out, int32_t* words_out, int64_t words_cap) {
  vqn_chain::Plan p;
  const int rc = vqn_chain::build(p, in_mode, in_feats, n_freqs, n_stacks, stacks);
  if (rc != 0) return rc;
  std::vector<vqn_pack::Word> words;
  std::vector<vqn_chain::SmallBias> sb;
  int32_t desc[16 + 16 * VQN_CHAIN_MAX_LAYERS];
  vqn_chain::plan_words(p, words, desc, sb);
  if (desc_out) memcpy(desc_out, desc, sizeof(desc));
  if (words_out && words_cap >= (int64_t)words.size()) memcpy(words_out, words.data(), words.size() * sizeof(vqn_pack::Word));
  return (int64_t)words.size();
}
