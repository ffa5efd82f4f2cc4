// Network descriptors handed to the fused NeuS kernels.  Every field is 4 bytes so that the host
// side (Python) can build them as flat int32 arrays; offsets are in float4 units into the weight
// pack buffer.  Public: the C ABI takes them as `const int32_t*` (include/vqnerf_hip.h); vqn_neus_pack_* builds them in C.
#pragma once
#include <stdint.h>

#define VQN_MAX_SDF_LAYERS 12
#define VQN_MAX_COL_LAYERS 8

struct LayerDesc {
  int n_out_tiles;   // ceil(out_features / 32)
  int kA_rows;       // informational: K rows taken from the previous activation
  int kB_rows;       // informational: K rows taken from the embedding / extras
  int w_off;         // forward pack   [n_out_tiles][kA+kB][64] float4
  int b_off;         // bias pack      [n_out_tiles][2][4] float4
  int wT_off;        // reverse pack, rows = features of the previous activation  (or -1)
  int wTE_off;       // reverse pack, rows = embedding features (layer 0 and the skip layer) (or -1)
  int reserved;
};

struct SdfDesc {                       // 12 + 12*8 = 108 ints
  int n_lin;                           // number of linear layers (9 for the shipped net)
  int skip;                            // layer whose input is [h, embedding]/sqrt(2) (scale folded into the pack); -1 = none
  int multires;
  int emb_feats;                       // 3 + 6*multires
  int emb_rows;                        // ceil(ceil(emb_feats/2)/4)
  int max_tiles;                       // LDS buffer capacity in 32-feature tiles (>= every layer of both nets)
  float scale;
  int last_w_off;                      // sdf row of the last layer as a rowdot image [rows][2] float4
  float last_bias;                     // used when last_b_off == 0
  int last_b_off;                      // > 0: the sdf row's bias lives in the pack at this float4 offset (component 0):
                                       // re-packing after an optimiser step then needs no device -> host copy
  int reserved1, reserved2;
  LayerDesc layers[VQN_MAX_SDF_LAYERS];  // layers[n_lin-1] = the FEATURE rows of the last layer (n_out_tiles = 0 if none)
};

struct ColDesc {                       // 16 + 8*8 = 80 ints
  int n_lin;
  int n_view_feats;                    // 0 (mode no_view_dir) or 3 + 6*multires_view
  int has_normal;                      // 0 for mode no_normal
  int extra_feats;                     // 3 + n_view_feats + 3*has_normal
  int extra_rows;
  int d_out;                           // 3
  int squeeze_out;
  int last_w_off;                      // rowdot image [3][rows][2] float4
  float last_bias[4];                  // used when last_b_off == 0
  int last_b_off;                      // > 0: [b0, b1, b2, 0] in the pack at this float4 offset
  int reserved1, reserved2, reserved3;
  LayerDesc layers[VQN_MAX_COL_LAYERS];
};
