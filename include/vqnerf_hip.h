/* vqnerf_hip.h -- C ABI of libvqnerf_hip.so (gfx950 / MI355X kernels for the VQ-NeRF hot path).
 *
 * The reference (JiuTongBro/vqnerf_release) has no FFI layer: its hot path is sequences of
 * framework ops behind two Python classes.  Each entry point below replaces one such sequence
 * (cited as reference file:line) and is what a reference-side ctypes binding would call
 * (INTEGRATION.md shows that stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensor.data_ptr());
 *     the library allocates nothing persistent
 *   - `stream` is a hipStream_t (NULL = default stream); all work is enqueued asynchronously
 *   - return 0 on success; -1 bad argument, -2 unsupported shape, -3 HIP runtime error;
 *     vqn_last_error() returns the thread-local message of the last failure
 *   - re-entrant; no global mutable state besides that thread-local string
 *   - all floating point is fp32, indices are int64, tensors are dense row-major
 */
#ifndef VQNERF_HIP_H_
#define VQNERF_HIP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int vqn_version(void);
const char* vqn_last_error(void);

/* ---- VQ codebook (decomp/nerfvq_nfr3/nerfactor/networks/vq_layers.py) --------------------- */

/* Replaces vq_layers.py:277-301,346-349: distances = |x|^2 - 2 x C + |C|^2, optional code-dropout
 * mask (dropped codes get max(distances)), idx = argmax(-distances) (lowest index on ties),
 * quant = C^T[idx].
 *   x [N,D], codebook [D,K]; sel_mask [K] (1 keep / 0 drop) or NULL; ws: >= 4 bytes of device
 *   scratch, required iff sel_mask != NULL; idx [N] int64; quant [N,D] or NULL; dist [N,K] or NULL.
 * Summation order is fixed (see oracle/vq_strict.c); D % 4 == 0, K <= 128. */
int vqn_vq_assign(const float* x, int64_t N, int D, const float* codebook, int K, const float* sel_mask,
                  float* ws, int64_t* idx, float* quant, float* dist, void* stream);

/* Replaces vq_layers.py:304-309 (the two reductions that feed the EMAs):
 *   counts[k] = #{n : idx[n] == k}   (as float),   dw[d,k] = sum_n x[n,d] [idx[n] == k].
 * counts [K], dw [D,K] are overwritten. */
int vqn_vq_ema_stats(const float* x, const int64_t* idx, int64_t N, int D, int K, float* counts, float* dw,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VQNERF_HIP_H_ */
