// Layer program handed to the generic fused Dense-stack kernel (csrc/mlp_chain.hip).  Every field is 4 bytes so
// that the host (vqnerf_release_amd/decomp/packing.py) can build it as a flat int32 array; weight offsets are in
// float4 units into the pack buffer.  Public: the C ABI takes it as `const int32_t*`.
#pragma once
#include <stdint.h>

#define VQN_CHAIN_MAX_LAYERS 16
#define VQN_CHAIN_MAX_OUTS 4

struct ChainLayer {            // 16 ints
  int kind;                    // 0 = GEMM tiles (MFMA) -> LDS rows [dst_row0, +4*n_out_tiles);  1 = <= 4 outputs (VALU dots) -> HBM;
                               // 2 = load the input image again into rows [dst_row0, +in_rows)
  int act;                     // bits 0-7 eng::Act: 0 none, 1 relu, 2 softplus(beta=100), 3 sigmoid;  bit 8 (kind 0): the output is written
                               // in place over one of the layer's own K segments, after a barrier (<= one output tile per wave)
  int n_out_tiles;             // kind 0: ceil(out/32);  kind 1: number of outputs (1..4)
  int kA_row0, kA_rows;        // K segment A: LDS rows
  int kB_row0, kB_rows;        // K segment B (skip-concat input), kB_rows = 0 if none
  int dst_row0;                // kind 1: offset (float4 units) of this layer's weight image in the workgroup's LDS copy
  int w_off;                   // kind 0: A-fragment pack [n_out_tiles][kA+kB][64] float4;  kind 1: [n_out][kA+kB][2] float4
  int b_off;                   // kind 0: bias pack [n_out_tiles][2][4] float4
  int out_slot;                // >= 0: leaves for HBM output `out_slot` (kind 0: first out_feats features of dst)
  int out_feats;
  float bias4[4];              // kind 1 biases
};

struct ChainDesc {             // 16 + 16*16 = 272 ints
  int n_layers;
  int in_mode;                 // 0 = raw features [N, in_feats];  1 = posenc of a 3-vector, in_feats = 3 + 6*n_freqs
  int in_feats;
  int in_rows;                 // ceil(in_feats / 8)
  int in_row0;
  int n_freqs;
  int total_rows;              // LDS rows (1 KB each) the program uses
  int n_waves;                 // 4 (two workgroups per CU when LDS allows) or 8 (one 512-thread workgroup per CU)
  int in_stride;               // floats between consecutive input rows
  int small_w4;                // float4 count of all kind-1 weight images (copied to LDS once per workgroup, after the rows)
  int reserved[6];
  ChainLayer layers[VQN_CHAIN_MAX_LAYERS];
};
