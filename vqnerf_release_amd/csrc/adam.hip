// One-launch Adam / AMSGrad over a table of parameter tensors with the step counters and the learning rate on the device, so that it
// lives inside a captured training step.  Two placements of epsilon, the reference uses both:
//   eps_mode 0 (torch.optim.Adam, nerf_runner.py:72):        p -= lr / (1 - b1^t) * m / (sqrt(vhat) / sqrt(1 - b2^t) + eps)
//   eps_mode 1 (Keras Adam(amsgrad=True), train_nfr.py:121-139 -> TF 2.4.1 ResourceApplyAdamWithAmsgrad):
//                                                            p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(vhat) + eps)
// i.e. Keras adds eps to the UN-debiased sqrt(vhat): an effective eps larger by 1 / sqrt(1 - b2^t) (31.6 x at step 1).  Stated in plain
// numpy in oracle/optim.py.  torch's own fused multi-tensor kernel hands each workgroup a 65,536-
// element chunk: the ~1 M parameters of the reflectance model make 16 workgroups and 85 us per launch (two launches: 0.17 of the
// captured step's 1.43 ms); here a workgroup takes 1,024 elements.
#include "common.h"
#include "vqnerf_hip.h"
#include <math.h>

namespace {

constexpr int ADAM_MAX = 56;
struct AdamEntry { float* p; const float* g; float* m; float* v; float* vmax; const float* step; long n; int blk0; int pad; };
struct AdamTable { AdamEntry e[ADAM_MAX]; int count; };

__global__ __launch_bounds__(256) void adam_kernel(const AdamTable tab, const float* __restrict__ lr_ptr, const double lr_host, const double beta1,
                                                   const double beta2, const double eps, const double weight_decay, const int maximize, const int eps_mode) {
  int ei = 0;
  for (int k = 1; k < tab.count; ++k)
    if ((int)blockIdx.x >= tab.e[k].blk0) ei = k;
  const AdamEntry& E = tab.e[ei];
  // bias corrections of this tensor's step count (already incremented), as torch's fused kernel takes them: in double, from a float count
  const double step = (double)E.step[0];
  const double bc1 = 1.0 - pow(beta1, step);
  const double bc2_sqrt = sqrt(1.0 - pow(beta2, step));
  const double lr = lr_ptr != nullptr ? (double)lr_ptr[0] : lr_host;
  const float step_size = (float)(eps_mode == 1 ? lr * bc2_sqrt / bc1 : lr / bc1);
  const long base = (long)(blockIdx.x - E.blk0) * 1024 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long i = base + 256 * k;
    if (i >= E.n) continue;
    float p = E.p[i], g = E.g[i], m = E.m[i], v = E.v[i];
    if (maximize) g = -g;
    if (weight_decay != 0.0) g = (float)((double)g + (double)p * weight_decay);
    m = (float)(beta1 * (double)m + (1.0 - beta1) * (double)g);
    v = (float)(beta2 * (double)v + (1.0 - beta2) * (double)g * (double)g);
    float denom;
    if (E.vmax != nullptr) {
      const float vm = fmaxf(E.vmax[i], v);
      E.vmax[i] = vm;
      denom = eps_mode == 1 ? (float)((double)sqrtf(vm) + eps) : (float)((double)sqrtf(vm) / bc2_sqrt + eps);
    } else denom = eps_mode == 1 ? (float)((double)sqrtf(v) + eps) : (float)((double)sqrtf(v) / bc2_sqrt + eps);
    p = __fsub_rn(p, __fdiv_rn(__fmul_rn(step_size, m), denom));             // (each step rounded: no contraction into an fma)
    E.p[i] = p; E.m[i] = m; E.v[i] = v;
  }
}

}  // namespace

extern "C" int vqn_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                             float* const* max_exp_avg_sq, const float* const* steps, const int64_t* numel, const float* lr_dev, double lr,
                             double beta1, double beta2, double eps, double weight_decay, int maximize, int eps_mode, void* stream) {
  VQN_CHECK_ARG(count >= 0 && params && grads && exp_avg && exp_avg_sq && steps && numel, "null pointer");
  VQN_CHECK_ARG(eps_mode == 0 || eps_mode == 1, "eps_mode: 0 = torch.optim.Adam, 1 = Keras Adam");
  for (int c0 = 0; c0 < count; c0 += ADAM_MAX) {
    AdamTable tab;
    memset(&tab, 0, sizeof(tab));
    tab.count = count - c0 < ADAM_MAX ? count - c0 : ADAM_MAX;
    long blocks = 0;
    for (int k = 0; k < tab.count; ++k) {
      const int i = c0 + k;
      VQN_CHECK_ARG(numel[i] >= 0 && (numel[i] == 0 || (params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && steps[i])), "entry: null tensor");
      AdamEntry& E = tab.e[k];
      E.p = params[i]; E.g = grads[i]; E.m = exp_avg[i]; E.v = exp_avg_sq[i];
      E.vmax = max_exp_avg_sq != nullptr ? max_exp_avg_sq[i] : nullptr;
      E.step = steps[i]; E.n = numel[i]; E.blk0 = (int)blocks;
      blocks += (numel[i] + 1023) / 1024;
    }
    if (blocks == 0) continue;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tab, lr_dev, lr, beta1, beta2, eps, weight_decay,
                       maximize, eps_mode);
    VQN_LAUNCH_CHECK();
  }
  return VQN_OK;
}
