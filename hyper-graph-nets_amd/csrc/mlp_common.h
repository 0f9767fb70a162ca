// Pieces shared by the fp32 (mlp.hip) and split-bf16 (mlp6.hip) backward kernels.
#pragma once
#include "hgn_device.h"
#include "../../include/hgn_mp.h"

namespace hgn {

__device__ __forceinline__ void relu_mask(Act& g, const float* __restrict__ zrow, int kq) {
  HGN_FOR_B(fb) {
    const f32x4 z = *reinterpret_cast<const f32x4*>(zrow + 16 * fb + 4 * kq);
#pragma unroll
    for (int u = 0; u < 4; ++u) g.v[fb][u] = z[u] > 0.f ? g.v[fb][u] : 0.f;
  }
}

// ReLU sign pattern of the lane's 32 values as one word (bit 4*fb + u <-> unit 16*fb + 4*kq + u; hgn_mlp_fwd_t.relu_bits)
__device__ __forceinline__ unsigned relu_bits_of(const Act& a) {
  unsigned m = 0;
  HGN_FOR_B(fb)
#pragma unroll
    for (int u = 0; u < 4; ++u) m |= (a.v[fb][u] > 0.f ? 1u : 0u) << (4 * fb + u);
  return m;
}

__device__ __forceinline__ void relu_mask_bits(Act& g, unsigned m) {
  HGN_FOR_B(fb)
#pragma unroll
    for (int u = 0; u < 4; ++u) g.v[fb][u] = (m >> (4 * fb + u)) & 1u ? g.v[fb][u] : 0.f;
}

// d_out_eff of the lane's row: d_out (optional) + the aggregation backward scattered back through the CSR row of the edge.
template <bool ACC>
__device__ __forceinline__ void load_dout(Act& g, const hgn_mlp_bwd_t& a, long rc, int kq) {
  // ACC: add d_out_eff to g (residual path; 128-wide, aligned);  otherwise g = d_out_eff
  if (ACC) {
    if (a.d_out) t_add(g, a.d_out + rc * a.ld_dout, kq);
  } else if (a.d_out) {
    const bool vec_out = (a.out_w == LAT) && ((a.ld_dout & 3) == 0);
    if (vec_out) t_load(g, a.d_out + rc * a.ld_dout, kq); else t_load_masked(g, a.d_out + rc * a.ld_dout, kq, a.out_w);
  } else {
    t_zero(g);
  }
  if (a.agg_dout) {
    const long r = a.agg_seg[rc];
    const int cnt = a.agg_rowptr[r + 1] - a.agg_rowptr[r];
    const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
    for (int slot = 0; slot < a.n_agg_ops; ++slot) {
      const float* ar = a.agg_dout + r * a.ld_agg + slot * LAT;
      const int op = a.agg_ops[slot];
      HGN_FOR_B(fb) {
        const int col = 16 * fb + 4 * kq;
        const f32x4 d = *reinterpret_cast<const f32x4*>(ar + col);
        if (op == HGN_OP_SUM) g.v[fb] += d;
        else if (op == HGN_OP_MEAN) g.v[fb] += d * inv;
        else {
          const int* ap = (op == HGN_OP_MAX ? a.agg_argmax : a.agg_argmin) + r * LAT + col;
#pragma unroll
          for (int u = 0; u < 4; ++u) g.v[fb][u] += ap[u] == (int)rc ? d[u] : 0.f;
        }
      }
    }
  }
}

}  // namespace hgn
