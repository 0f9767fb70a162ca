// Device-side building blocks shared by the fused MLP kernels (gfx950 / CDNA4 only).
//
// Formulation.  Every dense product on the path is  Z[i][j] = sum_k X[i][k] * Wt[j][k]  (nn.Linear, weight
// [out][in]) or its transpose-weight twin  dX[i][k] = sum_j dZ[i][j] * Wt[j][k].  Both are evaluated in the
// TRANSPOSED form  Z^T = Wt * X^T  with v_mfma_f32_32x32x2_f32 (exact fp32 fma chain):
//   * MFMA "A" operand  = the weight block, read from LDS.  The block is staged by LDS-DMA (global_load_lds_dwordx4,
//                         no VGPRs, 16 instructions per wave) into unpadded 512-byte rows whose 16-byte groups are
//                         XOR-swizzled with the row index: forward reads are conflict-free ds_read_b128 (one read feeds
//                         four MFMAs), transposed (backward) reads are conflict-free ds_read_b32;
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
constexpr int LDW = 128;         // LDS row stride of a staged weight block (unpadded; 16-B groups XOR-swizzled by row)
constexpr int TILE_ROWS = 128;   // rows per workgroup (4 waves x 32)
constexpr int WG = 256;

// rho(s, h): position inside a 32-block that MFMA k-step s / C-register s maps to for lane half h
__device__ __forceinline__ constexpr int rho0(int s) { return (s & 3) + 8 * (s >> 2); }

// LDS image of a weight block: element (r, c) lives at float index  r*128 + (((c>>2) ^ (r&31)) << 2) + (c&3).
__device__ __forceinline__ int wswz(int r, int c) { return r * LDW + ((((c >> 2) ^ (r & 31)) << 2) | (c & 3)); }

// One contraction stage:  acc[ob] += Wblock(ob, cb) * b[cb]   for ob < nob, cb < ncb.
//  TR=false: LDS block holds Wt rows = OUTPUT features, cols = contraction index  (forward:  Z^T  = Wt  * X^T)
//  TR=true : LDS block holds Wt rows = CONTRACTION index, cols = output index     (backward: dX^T = Wt^T * dZ^T)
// MFMA k-step s of contraction block cb uses contraction index 32*cb + rho0(s) + 4*h  (h = lane>>5), which is also
// the feature a C/D register s of block cb holds -- so b[] can be the previous stage's accumulators.
template <bool TR>
__device__ __forceinline__ void mfma_stage(f32x16 (&acc)[4], const f32x16 (&b)[4], const float* __restrict__ wlds,
                                           int nob, int ncb) {
  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  if (!TR) {
    // lane (m,h) needs Wt[32*ob + m][32*cb + 8*q + 4*h + u], u = 0..3  ->  ONE ds_read_b128 per (ob, cb, q)
    const float* base = wlds + m * LDW;
    const int p = m ^ h;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      if (cb < ncb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int xo = (p ^ (8 * cb + 2 * q)) << 2;            // ((8cb + 2q + h) ^ m) << 2
          f32x4 a[4];
#pragma unroll
          for (int ob = 0; ob < 4; ++ob)
            if (ob < nob) a[ob] = *reinterpret_cast<const f32x4*>(base + 32 * LDW * ob + xo);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
              if (ob < nob) acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ob][u], b[cb][4 * q + u], acc[ob], 0, 0, 0);
          }
        }
      }
    }
  } else {
    // lane (m,h) needs Wt[32*cb + rho0(s) + 4*h][32*ob + m]; rho0(s) never has bit 2 set, so the swizzled address splits
    // into a compile-time part and four per-lane bases (one per value of rho0(s) & 3).
    const int pl = (m >> 2) ^ (4 * h);
    const float* bv[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bv[v] = wlds + 4 * h * LDW + (m & 3) + ((pl ^ v) << 2);
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      if (cb < ncb) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          const int r0 = rho0(s);
#pragma unroll
          for (int ob = 0; ob < 4; ++ob) {
            if (ob < nob) {
              const int imm = (32 * cb + r0) * LDW + (((8 * ob) ^ (r0 & 24)) << 2);
              acc[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[r0 & 3][imm], b[cb][s], acc[ob], 0, 0, 0);
            }
          }
        }
      }
    }
  }
}

// Cooperative copy of a weight block W[r*ldw + c] (r < rows, c < cols; zero beyond) into the swizzled LDS image, rows
// [0,rpad) x cols [0,cpad).  Full, 16-byte-aligned 128x128 blocks go by LDS-DMA: each wave-instruction moves two rows
// (1 KiB, lane-linear in LDS); the swizzle is applied to the per-lane SOURCE address.  Completion is covered by the
// vmcnt(0) hipcc emits ahead of the following __syncthreads().
__device__ __forceinline__ void stage_weight(float* __restrict__ wlds, const float* __restrict__ W, long ldw,
                                             int rows, int cols, int rpad, int cpad) {
  const bool dma = rows == 128 && cols == 128 && (ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
  if (dma) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = wave; i < 64; i += WG / 64) {
      const int r = 2 * i + (lane >> 5);
      const int g = (lane & 31) ^ (r & 31);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(W + (long)r * ldw + 4 * g),
          (__attribute__((address_space(3))) void*)(wlds + i * 256), 16, 0, 0);
    }
    return;
  }
  const int c = threadIdx.x & 127;
  const int r0 = threadIdx.x >> 7;
  if (c < cpad) {
#pragma unroll 4
    for (int r = r0; r < rpad; r += 2) {
      float v = 0.f;
      if (r < rows && c < cols) v = W[(long)r * ldw + c];
      wlds[wswz(r, c)] = v;
    }
  }
}

// Workgroup barrier WITHOUT the vmcnt(0) drain that __syncthreads() carries: used before a weight block is restaged,
// where the only requirement is that every wave has finished READING the LDS image (its ds_reads have returned once the
// MFMAs that consumed them were issued).  Outstanding global stores of saved activations stay in flight across it.
__device__ __forceinline__ void wg_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
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
