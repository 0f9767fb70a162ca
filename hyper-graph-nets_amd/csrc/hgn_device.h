// Device-side building blocks shared by the fused MLP kernels (gfx950 / CDNA4 only).
//
// Formulation.  Every dense product on the path is  Z[i][j] = sum_k X[i][k] * Wt[j][k]  (nn.Linear, weight
// [out][in]) or its transpose-weight twin  dX[i][k] = sum_j dZ[i][j] * Wt[j][k].  Both are evaluated in the
// TRANSPOSED form  Z^T = Wt * X^T  with v_mfma_f32_32x32x2_f32 (exact fp32 fma chain):
//   * MFMA "A" operand  = the weight block, read from LDS (row stride 129 floats -> conflict-free ds_read_b32);
//   * MFMA "B" operand  = activations of the 32 rows a wave owns, one row per lane, held in REGISTERS;
//   * MFMA "C/D"        = next activations, again one row per lane (col = lane&31), features in the 16 regs.
// Because the C/D map (row of the 32x32 block = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) is exactly the k-order in
// which we feed B, the output of one layer is the B operand of the next with no LDS round trip and no shuffle:
// activations never leave the register file between the three Linear layers, LayerNorm and the residual.
// LDS holds only weights (one 128x128 block, 66 KB, shared by the 4 waves of a workgroup).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hgn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LAT = 128;         // latent width (reference hard-codes 128: src/model/flag.py:57)
constexpr int LDW = 129;         // LDS row stride of a staged weight block (odd -> 32 lanes hit 32 banks)
constexpr int TILE_ROWS = 128;   // rows per workgroup (4 waves x 32)
constexpr int WG = 256;

// rho(s, h): position inside a 32-block that MFMA k-step s / C-register s maps to for lane half h
__device__ __forceinline__ constexpr int rho0(int s) { return (s & 3) + 8 * (s >> 2); }

// One contraction stage:  acc[ob] += Wblock(ob, cb) * b[cb]   for ob < nob, cb < ncb.
//  TR=false: LDS block holds Wt rows = OUTPUT features, cols = contraction index  (forward:  Z^T  = Wt  * X^T)
//  TR=true : LDS block holds Wt rows = CONTRACTION index, cols = output index     (backward: dX^T = Wt^T * dZ^T)
// wl = LDS base + lane offset:  TR=false: (lane&31)*LDW + 4*(lane>>5) ;  TR=true: 4*(lane>>5)*LDW + (lane&31)
template <bool TR>
__device__ __forceinline__ void mfma_stage(f32x16 (&acc)[4], const f32x16 (&b)[4], const float* __restrict__ wl,
                                           int nob, int ncb) {
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    if (cb < ncb) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int r = rho0(s);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
          if (ob < nob) {
            const int off = TR ? ((32 * cb + r) * LDW + 32 * ob) : (32 * ob * LDW + 32 * cb + r);
            acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[off], b[cb][s], acc[ob], 0, 0, 0);
          }
        }
      }
    }
  }
}

// Cooperative copy of a weight block W[r*ldw + c] (r < rows, c < cols; zero beyond) into LDS rows [0,rpad) x
// cols [0,cpad).  256 threads; consecutive lanes -> consecutive columns (coalesced global, conflict-free LDS).
__device__ __forceinline__ void stage_weight(float* __restrict__ wlds, const float* __restrict__ W, long ldw,
                                             int rows, int cols, int rpad, int cpad) {
  const int c = threadIdx.x & 127;
  const int r0 = threadIdx.x >> 7;
  if (c < cpad) {
#pragma unroll 8
    for (int r = r0; r < rpad; r += 2) {
      float v = 0.f;
      if (r < rows && c < cols) v = W[(long)r * ldw + c];
      wlds[r * LDW + c] = v;
    }
  }
}

// B operand of the first stage: the lane's own row x[0..kw) (kw <= 128), zero beyond kw.
__device__ __forceinline__ void load_bfrag(f32x16 (&b)[4], const float* __restrict__ xrow, int kw, int h,
                                           bool vec) {
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = 32 * cb + 8 * q + 4 * h;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (vec) {
        if (col < kw) v = *reinterpret_cast<const f32x4*>(xrow + col);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (col + u < kw) v[u] = xrow[col + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) b[cb][4 * q + u] = v[u];
    }
  }
}

// C-layout helpers: the lane owns one row; register (ob, 4g+u) <-> column 32*ob + 8*g + 4*h + u.
#define HGN_FOR_C(ob, g) \
  _Pragma("unroll") for (int ob = 0; ob < 4; ++ob) _Pragma("unroll") for (int g = 0; g < 4; ++g)

__device__ __forceinline__ void c_load(f32x16 (&a)[4], const float* __restrict__ row, int h, int w = 128) {
  HGN_FOR_C(ob, g) {
    const int col = 32 * ob + 8 * g + 4 * h;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col < w) v = *reinterpret_cast<const f32x4*>(row + col);
#pragma unroll
    for (int u = 0; u < 4; ++u) a[ob][4 * g + u] = v[u];
  }
}
__device__ __forceinline__ void c_add(f32x16 (&a)[4], const float* __restrict__ row, int h) {
  HGN_FOR_C(ob, g) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 32 * ob + 8 * g + 4 * h);
#pragma unroll
    for (int u = 0; u < 4; ++u) a[ob][4 * g + u] += v[u];
  }
}
__device__ __forceinline__ void c_store(const f32x16 (&a)[4], float* __restrict__ row, int h) {
  HGN_FOR_C(ob, g) {
    f32x4 v;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = a[ob][4 * g + u];
    *reinterpret_cast<f32x4*>(row + 32 * ob + 8 * g + 4 * h) = v;
  }
}
// scalar, width-masked variants for unaligned / narrow rows (encoder inputs, decoder outputs)
__device__ __forceinline__ void c_load_masked(f32x16 (&a)[4], const float* __restrict__ row, int h, int w) {
  HGN_FOR_C(ob, g) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int col = 32 * ob + 8 * g + 4 * h + u;
      a[ob][4 * g + u] = (col < w) ? row[col] : 0.f;
    }
  }
}
__device__ __forceinline__ void c_store_masked(const f32x16 (&a)[4], float* __restrict__ row, int h, int w) {
  HGN_FOR_C(ob, g) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int col = 32 * ob + 8 * g + 4 * h + u;
      if (col < w) row[col] = a[ob][4 * g + u];
    }
  }
}
__device__ __forceinline__ void c_zero(f32x16 (&a)[4]) {
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int s = 0; s < 16; ++s) a[ob][s] = 0.f;
}
// Sum over the 128 features of the lane's row: 64 in-lane values + the partner lane (lane ^ 32).
__device__ __forceinline__ float row_sum(const f32x16 (&a)[4]) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) { s0 += a[0][s]; s1 += a[1][s]; s2 += a[2][s]; s3 += a[3][s]; }
  float t = (s0 + s1) + (s2 + s3);
  return t + __shfl_xor(t, 32);
}

}  // namespace hgn
