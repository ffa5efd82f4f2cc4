// Tile-program descriptor of csrc/tile_vm.hip: a list of ops over LDS "activation image" regions of one
// 32-point tile.  Every field is 4 bytes (host side: flat int32 array built by vqnerf_release_amd/geo/train_programs.py).
// Weight offsets are in float4 units into the pack buffer; tensor operands are indices into the pointer table
// handed to the launch.  Public (the C ABI takes it as `const int32_t*`).
//
// Global tensor formats
//   VEC  : [N, ld] row-major floats (points x few components)
//   TFMT : [n_tiles][F/32][32 features][32 points] floats -- feature-major inside a 32-point tile, so that
//          (a) an accumulator register (fixed feature, 32 lanes = 32 points) is one 128-byte store, and
//          (b) the weight-gradient kernel (csrc/wgrad.hip) reads MFMA operands with lanes = features directly.
#pragma once
#include <stdint.h>

#define VQN_VM_MAX_OPS 96
#define VQN_VM_MAX_TENSORS 96

enum VmKind : int {
  VM_LD_POSENC = 1,       // p0 x(VEC,3)  p1 dst_row0  p2 n_freqs  p3 feats  p4 store TFMT idx|-1  p5 scale(float)
  VM_LD_POSENC_JVP = 2,   // p0 x  p1 v(VEC,3)  p2 dst_row0  p3 n_freqs  p4 feats  p5 store TFMT|-1  p6 scale(float)
  VM_LD_T = 3,            // p0 TFMT idx  p1 dst_row0  p2 rows
  VM_LD_VEC = 4,          // p0 VEC idx  p1 dst_row0  p2 c (<= 8)  p3 scale(float)  p4 store TFMT|-1  p5 feature offset f0
  VM_LD_EXTRAS = 5,       // p0 x  p1 dirs  p2 normals|-1  p3 dst_row0  p4 n_view_freqs (0 = none)  p5 store TFMT|-1  p6 feats
  VM_GEMM = 6,            // p0 n_out_tiles p1 kA_row0 p2 kA_rows p3 kB_row0 p4 kB_rows p5 w_off p6 b_off|-1 p7 dst_row0|-1
                          // p8 epilogue  p9 act  p10 aux1 TFMT|-1  p11 aux2 TFMT|-1  p12 store TFMT|-1  p13 store2 TFMT|-1
                          // p14 accumulate-into-dst flag
  VM_ST_VEC = 7,          // p0 src_row0  p1 f0  p2 c (<= 4)  p3 VEC idx  p4 act  p5 scale(float)
  VM_POSENC_VJP = 8,      // p0 src_row0 (adjoint of the embedding)  p1 x  p2 out VEC(3)  p3 n_freqs  p4 scale(float)
};

enum VmEpi : int {
  VM_EPI_ACT = 0,         // y = act(acc)
  VM_EPI_MUL_DACT = 1,    // y = acc * act'(.)              act' from the activation OUTPUT stored in aux1
  VM_EPI_TANGENT = 2,     // y = act'(.) * acc ; store2 <- aux2 * acc * (act''/act')(.)
  VM_EPI_BWD2 = 3,        // y = acc * act'(.) + aux2
};

struct VmOp {
  int kind;
  int p[15];
};

struct VmDesc {            // 16 + 96*16 ints
  int n_ops;
  int total_rows;
  int n_waves;             // 4 or 8
  int n_tensors;
  int reserved[12];
  VmOp ops[VQN_VM_MAX_OPS];
};

struct VmTensor {
  float* ptr;
  int ld;                  // VEC: row stride in floats;  TFMT: number of 32-feature tiles
  int pad;
};
