// Kernel arguments shared by the two forms of the fused edge backward (csrc/fused_bwd.hip: three bf16 terms / one bf16 term;
// csrc/fused_bwd3.hip: two scaled fp16 terms).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/hgn_mp.h"

namespace hgn {

constexpr int FSLAB = 128 * 128 + 128;                // floats per (workgroup, layer): dW partial + bias partial (= wgrad.hip SLAB)

struct FusedArgs {
  hgn_mlp_bwd_t b;                                    // the data-gradient chain (n_dx == 1, residual, LayerNorm, ReLU sign words)
  const float* A[2];                                  // other operand of dW3, dW2: z2, z1 (row stride 128)
  float* slabs;                                       // [gridDim.x][2][FSLAB]
  long tiles;                                         // 64-row tiles
};

int launch_edge_bwd_fused3(const FusedArgs& fa, long grid, hipStream_t stream);      // csrc/fused_bwd3.hip

}  // namespace hgn
