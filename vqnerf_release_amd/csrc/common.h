// Shared helpers for the gfx950 kernels (HIP, wave64).  Not a public header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define VQN_OK 0
#define VQN_EARG (-1)     // bad argument (null pointer, negative size ...)
#define VQN_ESHAPE (-2)   // shape not supported by the kernels
#define VQN_EHIP (-3)     // HIP runtime error

void vqn_set_error(const char* fmt, ...);

#define VQN_CHECK_ARG(cond, msg)                         \
  do {                                                   \
    if (!(cond)) {                                       \
      vqn_set_error("%s: bad argument: %s", __func__, msg); \
      return VQN_EARG;                                   \
    }                                                    \
  } while (0)

#define VQN_CHECK_SHAPE(cond, msg)                       \
  do {                                                   \
    if (!(cond)) {                                       \
      vqn_set_error("%s: unsupported shape: %s", __func__, msg); \
      return VQN_ESHAPE;                                 \
    }                                                    \
  } while (0)

#define VQN_HIP(call)                                                        \
  do {                                                                       \
    hipError_t e_ = (call);                                                  \
    if (e_ != hipSuccess) {                                                  \
      vqn_set_error("%s: HIP error %d (%s) at %s:%d", __func__, (int)e_,     \
                    hipGetErrorString(e_), __FILE__, __LINE__);              \
      return VQN_EHIP;                                                       \
    }                                                                        \
  } while (0)

#define VQN_LAUNCH_CHECK() VQN_HIP(hipGetLastError())

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// streaming (non-temporal) accesses for data that is written once and read once (the per-workgroup activation stash of the
// fused NeuS kernels): keeps it from evicting the weight packs, which every workgroup re-reads, out of the 4 MB L2s
#ifdef __HIPCC__
__device__ __forceinline__ void st_stream(f32x4* p, const f32x4 v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ f32x4 ld_stream(const f32x4* p) { return __builtin_nontemporal_load(p); }
#endif

static inline int vqn_num_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}
