// Layer programs + weight packs of the fused Dense-stack kernel built in C (C ABI: vqn_chain_pack_*): the handle half -- device
// memory of the pack and its gather table, the gather launch, the small layers' biases into the descriptor.  The program builder
// itself is host-only C++ (csrc/chain_pack_plan.h); tests/test_abi.py holds it to the Python builder's descriptors and packs.
#include "chain_pack_plan.h"
#include "pack_gather.h"

struct vqn_chain_pack {
  vqn_chain::Plan plan;
  std::vector<vqn_chain::SmallBias> small_bias;
  int32_t desc[16 + 16 * VQN_CHAIN_MAX_LAYERS];
  int64_t n = 0;
  int n_weights = 0;
  Word* d_words = nullptr;
  float* d_wbuf = nullptr;
};

extern "C" {

int vqn_chain_pack_create(int in_mode, int in_feats, int n_freqs, int n_stacks, const vqn_chain_stack* stacks, vqn_chain_pack** out) {
  VQN_CHECK_ARG(out != nullptr, "out == NULL");
  *out = nullptr;
  vqn_chain_pack* p = new vqn_chain_pack();
  const int rc = vqn_chain::build(p->plan, in_mode, in_feats, n_freqs, n_stacks, stacks);
  if (rc != 0) { delete p; return rc; }
  std::vector<Word> words;
  vqn_chain::plan_words(p->plan, words, p->desc, p->small_bias);
  p->n = (int64_t)words.size();
  for (const auto& L : p->plan.layers) p->n_weights = L.weight + 1 > p->n_weights ? L.weight + 1 : p->n_weights;
  hipError_t e;
  if ((e = hipMalloc(&p->d_words, words.size() * sizeof(Word))) != hipSuccess ||
      (e = hipMalloc(&p->d_wbuf, words.size() * sizeof(float))) != hipSuccess ||
      (e = hipMemcpy(p->d_words, words.data(), words.size() * sizeof(Word), hipMemcpyHostToDevice)) != hipSuccess) {
    vqn_set_error("vqn_chain_pack_create: HIP error %d (%s)", (int)e, hipGetErrorString(e));
    vqn_chain_pack_destroy(p);
    return VQN_EHIP;
  }
  *out = p;
  return VQN_OK;
}

int vqn_chain_pack_update(vqn_chain_pack* p, const float* const* kernels, const float* const* biases, void* stream) {
  VQN_CHECK_ARG(p != nullptr && kernels != nullptr && biases != nullptr, "NULL argument");
  VQN_CHECK_SHAPE(p->n_weights <= 16, "at most 16 layers");
  hipStream_t st = (hipStream_t)stream;
  PtrTable t;
  memset(&t, 0, sizeof(t));
  for (int i = 0; i < p->n_weights; ++i) {
    VQN_CHECK_ARG(kernels[i] != nullptr && biases[i] != nullptr, "NULL layer pointer");
    t.p[2 * i] = kernels[i];
    t.p[2 * i + 1] = biases[i];
  }
  pack_gather_kernel<<<(unsigned)((p->n + 255) / 256), 256, 0, st>>>(p->d_words, p->n, t, p->d_wbuf);
  VQN_LAUNCH_CHECK();
  for (const auto& sb : p->small_bias) {
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    VQN_HIP(hipMemcpyAsync(b4, biases[sb.weight], sizeof(float) * sb.n, hipMemcpyDeviceToHost, st));
    VQN_HIP(hipStreamSynchronize(st));
    memcpy(&p->desc[sb.desc_pos], b4, sizeof(float) * sb.n);
  }
  return VQN_OK;
}

int vqn_chain_pack_n_weights(const vqn_chain_pack* p) { return p ? p->n_weights : 0; }
const int32_t* vqn_chain_pack_desc(const vqn_chain_pack* p) { return p ? p->desc : nullptr; }
const float* vqn_chain_pack_wbuf(const vqn_chain_pack* p) { return p ? p->d_wbuf : nullptr; }
int64_t vqn_chain_pack_floats(const vqn_chain_pack* p) { return p ? p->n : 0; }

void vqn_chain_pack_destroy(vqn_chain_pack* p) {
  if (!p) return;
  if (p->d_words) (void)hipFree(p->d_words);
  if (p->d_wbuf) (void)hipFree(p->d_wbuf);
  delete p;
}

}  // extern "C"
