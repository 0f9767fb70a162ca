// Device-side building blocks shared by the fused MLP kernels (gfx950 / CDNA4 only).
//
// Formulation.  Every dense product on the path is  Z[i][j] = sum_k X[i][k] * Wt[j][k]  (nn.Linear, weight
// [out][in]) or its transpose-weight twin  dX[i][k] = sum_j dZ[i][j] * Wt[j][k].  Both are evaluated in the
// TRANSPOSED form  Z^T = Wt * X^T  with v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, 32-cycle issue):
//   * MFMA "A" operand  = the weight block, read from LDS.  The block is staged by LDS-DMA (global_load_lds_dwordx4,
//                         no VGPRs) into unpadded 512-byte rows whose 16-byte groups are XOR-swizzled with the row
//                         index: forward reads are conflict-free ds_read_b128 (one read feeds four MFMAs), transposed
//                         (backward) reads are conflict-free ds_read_b32;
//   * MFMA "B" operand  = activations of the 16 rows a wave owns, held in REGISTERS: lane (n = lane&15, kq = lane>>4)
//                         holds, for every 16-feature block fb, features 16*fb + 4*kq + r (r = 0..3) of row n;
//   * MFMA "C/D"        = next activations in exactly the same map (col = lane&15 is the row, reg r of block fb is
//                         feature 16*fb + 4*(lane>>4) + r).
// Because the C/D map equals the k-order in which B is fed, the output of one layer is the B operand of the next with
// no LDS round trip and no shuffle: activations never leave the register file between the three Linear layers,
// LayerNorm and the residual.  LDS holds only weights (one 128x128 block, 64 KB, shared by the 8 waves of a
// workgroup).  A wave needs 32 + 32 activation registers, so four waves fit per SIMD (<= 128 VGPRs): while one wave
// streams its rows in or out, three others keep the matrix pipe fed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LAT = 128;         // latent width (reference hard-codes 128: src/model/flag.py:57)
constexpr int LDW = 128;         // LDS row stride of a staged weight block (unpadded; 16-B groups XOR-swizzled by row)
constexpr int NB = 8;            // 16-feature blocks per latent row
constexpr int WAVE_ROWS = 16;    // rows per wave
constexpr int WG = 512;          // 8 waves
constexpr int TILE_ROWS = 128;   // rows per workgroup

// LDS image of a weight block: element (r, c) lives at float index  r*128 + (((c>>2) ^ (r&31)) << 2) + (c&3).
__device__ __forceinline__ int wswz(int r, int c) { return r * LDW + ((((c >> 2) ^ (r & 31)) << 2) | (c & 3)); }

struct Act { f32x4 v[NB]; };     // one wave-row tile of activations: v[fb][r] = feature 16*fb + 4*kq + r of row n

// One contraction stage:  acc[ob] += Wblock(ob, cb) * b[cb]   for ob < nob, cb < ncb   (blocks of 16).
//  TR=false: LDS block holds Wt rows = OUTPUT features, cols = contraction index  (forward:  Z^T  = Wt  * X^T)
//  TR=true : LDS block holds Wt rows = CONTRACTION index, cols = output index     (backward: dX^T = Wt^T * dZ^T)
template <bool TR>
__device__ __forceinline__ void mfma_stage(Act& acc, const Act& b, const float* __restrict__ wlds, int nob, int ncb) {
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, kq = lane >> 4;
  if (!TR) {
    // lane (m,kq) needs Wt[16*ob + m][16*cb + 4*kq + r], r = 0..3  ->  ONE ds_read_b128 per (ob, cb).
    // row & 31 = m + 16*(ob & 1);  16-byte group = 4*cb + kq.
    const float* base = wlds + m * LDW;
    const int pe = (m ^ kq) << 2;                 // ob even:  ((4cb + kq) ^ m)        << 2  =  pe ^ (16cb)
    const int po = ((m + 16) ^ kq) << 2;          // ob odd :  ((4cb + kq) ^ (m + 16)) << 2  =  po ^ (16cb)
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      if (cb < ncb) {
        const int xe = pe ^ (16 * cb), xo = po ^ (16 * cb);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          f32x4 a[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ob = 4 * half + q;
            if (ob < nob) a[q] = *reinterpret_cast<const f32x4*>(base + 16 * LDW * ob + ((ob & 1) ? xo : xe));
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int ob = 4 * half + q;
              if (ob < nob) acc.v[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][r], b.v[cb][r], acc.v[ob], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);     // keep the weight fragments of later blocks out of the register file
        }
      }
    }
  } else {
    // lane (m,kq) needs Wt[16*cb + 4*kq + r][16*ob + m]  (ds_read_b32).  With oc = ob ^ 4*(cb&1) the swizzled float
    // index is  (16cb + r)*128 + ((oc & 4) << 4)   [compile time]
    //         +  4kq*128 + (((oc & 3) ^ kq) << 4) + (((m >> 2) ^ r) << 2) + (m & 3)   [per lane, 16 variants].
    int off[4][4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) off[v][r] = 4 * kq * LDW + ((v ^ kq) << 4) + ((((m >> 2) ^ r)) << 2) + (m & 3);
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      if (cb < ncb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int ob = 0; ob < NB; ++ob) {
            if (ob < nob) {
              const int oc = ob ^ (4 * (cb & 1));
              const int imm = (16 * cb + r) * LDW + ((oc & 4) << 4);
              acc.v[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wlds[imm + off[oc & 3][r]], b.v[cb][r], acc.v[ob], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
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
    // 32-bit offsets from the (wave-uniform) block pointer: one VGPR per in-flight DMA address
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned ld = (unsigned)ldw;
#pragma unroll
    for (unsigned i = wave; i < 64; i += WG / 64) {
      const unsigned r = 2 * i + (lane >> 5);
      const unsigned g = (lane & 31) ^ (r & 31);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + (r * ld + 4 * g)),
                                       (__attribute__((address_space(3))) void*)(wlds + i * 256), 16, 0, 0);
    }
    return;
  }
  const int c = threadIdx.x & 127;
  const int r0 = threadIdx.x >> 7;
  if (c < cpad) {
#pragma unroll 4
    for (int r = r0; r < rpad; r += WG / 128) {
      float v = 0.f;
      if (r < rows && c < cols) v = W[(long)r * ldw + c];
      wlds[wswz(r, c)] = v;
    }
  }
}

// Workgroup barrier WITHOUT the vmcnt(0) drain that __syncthreads() carries: used before a weight block is restaged,
// where the only requirement is that every wave has finished READING the LDS image.  Outstanding global stores of
// saved activations stay in flight across it.
__device__ __forceinline__ void wg_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// Row tile of this workgroup.  Workgroups b and b+8 run on the same XCD (round-robin dispatch; used for speed only,
// never for correctness): give every XCD one CONTIGUOUS range of row tiles, so that the node rows gathered by
// neighbouring edge tiles (same graph, receiver-sorted) are served by that XCD's own 4 MiB L2.
__device__ __forceinline__ long xcd_tile() {
  const long nt = gridDim.x, b = blockIdx.x;
  const long q = nt >> 3, r = nt & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Row-tile <-> global memory.  The lane owns 16 bytes per 16-feature block: row n, columns 16*fb + 4*kq .. +3.
#define HGN_FOR_B(fb) _Pragma("unroll") for (int fb = 0; fb < NB; ++fb)

__device__ __forceinline__ void t_load(Act& a, const float* __restrict__ row, int kq) {
  HGN_FOR_B(fb) a.v[fb] = *reinterpret_cast<const f32x4*>(row + 16 * fb + 4 * kq);
}
// width-limited (w multiple of 4, 16-byte aligned row): zero beyond w
__device__ __forceinline__ void t_load_w(Act& a, const float* __restrict__ row, int kq, int w) {
  HGN_FOR_B(fb) {
    const int col = 16 * fb + 4 * kq;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col < w) v = *reinterpret_cast<const f32x4*>(row + col);
    a.v[fb] = v;
  }
}
// scalar, any width / alignment (encoder inputs, decoder outputs)
__device__ __forceinline__ void t_load_masked(Act& a, const float* __restrict__ row, int kq, int w) {
  HGN_FOR_B(fb) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int col = 16 * fb + 4 * kq + u;
      a.v[fb][u] = (col < w) ? row[col] : 0.f;
    }
  }
}
__device__ __forceinline__ void t_add(Act& a, const float* __restrict__ row, int kq) {
  HGN_FOR_B(fb) a.v[fb] += *reinterpret_cast<const f32x4*>(row + 16 * fb + 4 * kq);
}
__device__ __forceinline__ void t_store(const Act& a, float* __restrict__ row, int kq) {
  HGN_FOR_B(fb) *reinterpret_cast<f32x4*>(row + 16 * fb + 4 * kq) = a.v[fb];
}
__device__ __forceinline__ void t_store_masked(const Act& a, float* __restrict__ row, int kq, int w) {
  HGN_FOR_B(fb) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int col = 16 * fb + 4 * kq + u;
      if (col < w) row[col] = a.v[fb][u];
    }
  }
}
__device__ __forceinline__ void t_zero(Act& a) {
  HGN_FOR_B(fb) a.v[fb] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// Sum over the 128 features of a row: 32 in-lane values, then the three partner lanes (same n, other kq).
__device__ __forceinline__ float row_sum(const Act& a) {
  float s0 = 0.f, s1 = 0.f;
  HGN_FOR_B(fb) {
    s0 += a.v[fb][0] + a.v[fb][1];
    s1 += a.v[fb][2] + a.v[fb][3];
  }
  float t = s0 + s1;
  t += __shfl_xor(t, 16);
  t += __shfl_xor(t, 32);
  return t;
}

}  // namespace hgn
