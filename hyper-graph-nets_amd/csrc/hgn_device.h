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
// LayerNorm and the residual.  LDS holds only weights, HALF a 128x128 block at a time (32 KB: 64 contraction columns in the
// forward form, 64 contraction rows in the transposed form), shared by the 4 waves of a 64-row workgroup.  A wave needs
// 32 + 32 activation registers (<= 128 VGPRs), so a CU holds FOUR independent workgroups (16 waves, 4 per SIMD, 128 KB
// LDS): while some stream their rows in or out or wait for a weight DMA, the others keep the matrix pipe fed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// gfx950 only.  Beyond the instructions (v_mfma_f32_16x16x32_f16 / _bf16, global_load_lds_dwordx4, v_permlane16/32_swap,
// ds_read_b64_tr_b16) the kernels lean on gfx9-family memory behaviour that the HIP memory model does not promise: the "last
// workgroup reduces" tickets (csrc/mlp.hip: ln_reduce_kernel, csrc/wgrad.hip) publish with sc1 write-through stores counted in
// vmcnt and read with sc1 loads instead of device-scope fences (a fence writes back / invalidates a whole L2: measured 16 us
// against 10), and every counted `s_waitcnt vmcnt(N)` assumes in-order return of vector memory.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "csrc/: written for gfx950 (MI355X) only -- build with --offload-arch=gfx950"
#endif

namespace hgn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LAT = 128;         // latent width (reference hard-codes 128: src/model/flag.py:57)
constexpr int LDW = 128;         // LDS row stride of a staged weight block (unpadded; 16-B groups XOR-swizzled by row)
constexpr int NB = 8;            // 16-feature blocks per latent row
constexpr int WAVE_ROWS = 16;    // rows per wave
constexpr int WG = 256;          // 4 waves
constexpr int TILE_ROWS = 64;    // rows per workgroup
constexpr int HALF = 64;         // contraction indices per staged half block
constexpr int WLDS_FLOATS = 64 * 128;   // 32 KB

struct Act { f32x4 v[NB]; };     // one wave-row tile of activations: v[fb][r] = feature 16*fb + 4*kq + r of row n

// ---------------------------------------------------------------------------------------------------------------------
// Forward form (TR=false).  LDS image of a half block: [128 output rows][64 contraction cols], 256-byte rows, the sixteen
// 16-byte groups of a row XOR-swizzled with (row & 15):  (r, c) -> r*64 + (((c>>2) ^ (r&15)) << 2) + (c&3).
// Lane (m = lane&15, kq = lane>>4) reads Wt[16*ob + m][64*H + 16*cb + 4*kq + 0..3] with ONE conflict-free ds_read_b128
// per (ob, cb); the four values feed four MFMAs (k-steps r = 0..3 of contraction block 4*H + cb).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wswz_n(int r, int c) { return r * HALF + ((((c >> 2) ^ (r & 15)) << 2) | (c & 3)); }

template <int H>
__device__ __forceinline__ void mfma_half_n(Act& acc, const Act& b, const float* __restrict__ wlds, int nob, int ncb) {
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, kq = lane >> 4;
  const float* base = wlds + m * HALF;
  const int p = (m ^ kq) << 2;                       // ((4cb + kq) ^ m) << 2  =  p ^ (16cb)
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    if (cb < ncb) {
      const int xo = p ^ (16 * cb);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ob = 4 * half + q;
          if (ob < nob) a[q] = *reinterpret_cast<const f32x4*>(base + 16 * HALF * ob + xo);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int ob = 4 * half + q;
            if (ob < nob)
              acc.v[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][r], b.v[4 * H + cb][r], acc.v[ob], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the weight fragments of later blocks out of the register file
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Transposed form (TR=true).  LDS image of a half block: [64 contraction rows][128 output cols], 512-byte rows, 16-byte
// groups XOR-swizzled with (row & 31):  (r, c) -> r*128 + (((c>>2) ^ (r&31)) << 2) + (c&3).
// Lane (m, kq) reads Wt[64*H + 16*cb + 4*kq + r][16*ob + m] with conflict-free ds_read_b32.  With oc = ob ^ 4*(cb&1) the
// swizzled index is (16cb + r)*128 + ((oc & 4) << 4) [compile time] + one of 16 per-lane offsets.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wswz_t(int r, int c) { return r * LDW + ((((c >> 2) ^ (r & 31)) << 2) | (c & 3)); }

template <int H>
__device__ __forceinline__ void mfma_half_t(Act& acc, const Act& b, const float* __restrict__ wlds, int nob, int ncb) {
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, kq = lane >> 4;
  int off[4][4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int r = 0; r < 4; ++r) off[v][r] = 4 * kq * LDW + ((v ^ kq) << 4) + ((((m >> 2) ^ r)) << 2) + (m & 3);
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    if (cb < ncb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
          if (ob < nob) {
            const int oc = ob ^ (4 * (cb & 1));
            const int imm = (16 * cb + r) * LDW + ((oc & 4) << 4);
            acc.v[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(wlds[imm + off[oc & 3][r]], b.v[4 * H + cb][r], acc.v[ob], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Staging of a half block W[r*ldw + c], r < rows, c < cols (zero beyond, up to rpad x cpad).  Full, 16-byte-aligned halves
// go by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, lane-linear in LDS, swizzle applied to the per-lane
// SOURCE address, no VGPR traffic); completion is covered by the vmcnt(0) ahead of the following __syncthreads().
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_half_n(float* __restrict__ wlds, const float* __restrict__ W, long ldw, int rows,
                                             int cols, int rpad, int cpad) {
  // forward form: up to 128 rows x 64 cols
  const bool dma = rows == 128 && cols == HALF && (ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
  if (dma) {
    unsigned lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));      // opaque: the 8 DMA addresses are recomputed per call instead of being kept live
    const unsigned wave = threadIdx.x >> 6;
    const unsigned ld = (unsigned)ldw;
#pragma unroll
    for (unsigned i = wave; i < 32; i += WG / 64) {              // instruction i fills rows 4i .. 4i+3
      const unsigned r = 4 * i + (lane >> 4);
      const unsigned g = (lane & 15) ^ (r & 15);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + (r * ld + 4 * g)),
                                       (__attribute__((address_space(3))) void*)(wlds + i * 256), 16, 0, 0);
    }
    return;
  }
  const int c = threadIdx.x & 63;
  const int r0 = threadIdx.x >> 6;
  if (c < cpad) {
#pragma unroll 4
    for (int r = r0; r < rpad; r += WG / 64) {
      float v = 0.f;
      if (r < rows && c < cols) v = W[(long)r * ldw + c];
      wlds[wswz_n(r, c)] = v;
    }
  }
}

__device__ __forceinline__ void stage_half_t(float* __restrict__ wlds, const float* __restrict__ W, long ldw, int rows,
                                             int cols, int rpad, int cpad) {
  // transposed form: up to 64 rows x 128 cols
  const bool dma = rows == HALF && cols == 128 && (ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
  if (dma) {
    unsigned lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const unsigned wave = threadIdx.x >> 6;
    const unsigned ld = (unsigned)ldw;
#pragma unroll
    for (unsigned i = wave; i < 32; i += WG / 64) {              // instruction i fills rows 2i, 2i+1
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
      wlds[wswz_t(r, c)] = v;
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

// Row tile of this workgroup: the launch order.  All workgroups then sweep the row arrays front to back TOGETHER, one coherent stream
// through HBM.  (Rounds 1-3 gave every XCD one contiguous range of tiles -- workgroups b and b + 8 share an XCD under round-robin dispatch
// -- so that the node rows gathered by neighbouring edge tiles would be served by that XCD's own L2: eight concurrent sweeps.  Measured
// against each other on the final kernels, A/B on one box: edge forward 1.016 -> 0.987 ms, pre-projection 0.108 -> 0.104 ms, step
// 59.19 / 59.03 -> 58.87 / 58.82 ms in favour of the single sweep; the gathered rows come from the Infinity Cache either way.  The paired
// segment sums, whose second visit of a row depends on L2, keep the XCD ranges: csrc/segment.hip.)
#ifndef HGN_XCD_RUN
#define HGN_XCD_RUN 0
#endif
__device__ __forceinline__ long xcd_tile() {
#if HGN_XCD_RUN > 0
  // Runs of HGN_XCD_RUN consecutive tiles per XCD inside a window of 8 runs (workgroup b is dispatched to XCD b & 7): still ONE sweep
  // through the row arrays -- the chip's concurrent workgroups cover the same span of tiles as in launch order -- but the node rows
  // gathered by neighbouring edge tiles are found in the XCD's own L2.  Whole windows only; the tail keeps the launch order.
  constexpr unsigned RUN = HGN_XCD_RUN, WIN = 8 * RUN;
  const unsigned b = blockIdx.x;
  if (b >= gridDim.x / WIN * WIN) return b;
  const unsigned xcd = b & 7u, slot = b >> 3;
  return (long)(((slot / RUN) * 8u + xcd) * RUN + slot % RUN);
#else
  return blockIdx.x;
#endif
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
// Row-contiguous store of a wave's 16 x 128 fp32 tile.  In the MFMA C layout a store instruction writes 64 B of each of 16
// rows -- 64 quarter-line requests; measured on the edge forward (-DHGN_ABL=512: the same bytes as whole rows) that pattern
// costs 7 % of the kernel.  Here the tile goes through 4 KB of LDS private to the wave, half a row (256 B) at a time: written
// in the C layout (chunk c = 4 (fb & 3) + kq of row n at slot c ^ n: 16 lanes of one kq hit 16 distinct 16-byte slots), read
// back with lane l holding chunk l & 15 of row l >> 4 (+ 4, 8, 12), stored as 4 rows x 256 B = 8 whole lines per instruction.
// Same values, same addresses: only the lane that carries each 16 bytes changes.  `st`: this wave's 1024 floats.
__device__ __forceinline__ void t_store_rows(const Act& a, float* __restrict__ base, long row0, long ld, long M, float* st) {
  const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  const int r = lane >> 4, c = lane & 15;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(st + n * 64 + 4 * ((4 * q + kq) ^ n)) = a.v[4 * h + q];
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(st + (r + 4 * q) * 64 + 4 * (c ^ (r + 4 * q)));
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (row0 + r + 4 * q < M) *reinterpret_cast<f32x4*>(base + (row0 + r + 4 * q) * ld + 64 * h + 4 * c) = v[q];
  }
}
// Row access with a UNIFORM base pointer and a 32-bit per-lane byte offset (the caller bounds the array to 4 GiB): the address is one
// VGPR next to a scalar register pair (global_load / global_store with saddr) -- no 64-bit vector arithmetic per access and no 64-bit
// pointer per array for the allocator to keep alive.
__device__ __forceinline__ void t_load32(Act& a, const float* __restrict__ base, unsigned byte_off) {
  const char* p = reinterpret_cast<const char*>(base);
  HGN_FOR_B(fb) a.v[fb] = *reinterpret_cast<const f32x4*>(p + (byte_off + 64u * fb));
}
__device__ __forceinline__ void t_store32(const Act& a, float* __restrict__ base, unsigned byte_off) {
  char* p = reinterpret_cast<char*>(base);
  HGN_FOR_B(fb) *reinterpret_cast<f32x4*>(p + (byte_off + 64u * fb)) = a.v[fb];
}
// t_store_rows for a 128-float row stride with 32-bit offsets.  `row0`: first of the wave's 16 rows; FULL: all 16 rows exist (every
// tile of a launch but the last): no per-row tests, and a STATIC number of store instructions (8) for counted waits.
template <bool FULL>
__device__ __forceinline__ void t_store_rows32(const Act& a, float* __restrict__ base, unsigned row0, unsigned M, float* st) {
  const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  const unsigned r = lane >> 4, c = lane & 15;
  char* p = reinterpret_cast<char*>(base);
  const unsigned off = (row0 + r) * 512u + 16u * c;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(st + n * 64 + 4 * ((4 * q + kq) ^ n)) = a.v[4 * h + q];
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const f32x4*>(st + (r + 4 * q) * 64 + 4 * (c ^ (r + 4 * q)));
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (FULL || row0 + r + 4 * q < M) *reinterpret_cast<f32x4*>(p + (off + 2048u * q + 256u * h)) = v[q];
  }
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
// Sum of x over the 16 lanes of a DPP row (= the 16 rows n of one k-quarter): four rotate-and-add steps on the VALU
// (v_add_f32 with row_ror DPP control), no LDS crossbar traffic; every lane of the row ends with the full sum.
__device__ __forceinline__ float row16_sum(float x) {
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x128, 0xf, 0xf, true));   // row_ror:8
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x124, 0xf, 0xf, true));   // row_ror:4
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x122, 0xf, 0xf, true));   // row_ror:2
  x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x121, 0xf, 0xf, true));   // row_ror:1
  return x;
}

// Sums of the 32 values of an Act over the 16 lanes of a DPP row (= the 16 rows n of one k-quarter) by a TRANSPOSING butterfly: in each
// of four steps a lane keeps half of its values, hands the other half to a partner lane and adds what the partner hands back, so that
// afterwards lane n holds 2 of the 32 sums -- those of values j = 2 n, 2 n + 1 of the numbering j = 4 fb + w, i.e. of
//   fb = n >> 1,  w = 2 (n & 1) + {0, 1}      (feature 16 fb + 4 kq + w).
// Partners: lane ^ 8 (row_ror:8), 7 - lane within the half (row_half_mirror), lane ^ 2, lane ^ 1 (quad_perm) -- each flips the bit the
// step selects by and keeps the higher ones.  Steps 1 and 2 select by lane bits 3 and 2, which the DPP bank mask can express (a bank =
// four lanes of a row): out = lo + partner's lo, written in the lanes whose bit is 0, then out = hi + partner's hi in the others -- two
// write-masked v_add_f32_dpp per output and no select (bfly8, inline assembly: the compiler has no form for a DPP add that keeps other
// lanes of its destination).  Steps 3 and 4 select inside a quad: two v_cndmask and one add with a DPP operand per output.
// 50 + 18 instructions for the 32 sums, against 8 per VALUE (row16_sum on every value) for sums that every lane then holds sixteen-fold.
// The assembly's operands must be VALU results (the hazard recogniser does not look into inline assembly: the block opens with the two
// wait states a DPP read of a fresh VALU result needs; a matrix-pipe result would need more).
#define HGN_BFLY_ROW(o, a, CTRL, MASK) "v_add_f32_dpp %" #o ", %" #a ", %" #a " " CTRL " row_mask:0xf bank_mask:" MASK "\n\t"
#define HGN_BFLY8(CTRL, MLO, MHI)                                                                                              \
  asm("s_nop 1\n\t"                                                                                                            \
      HGN_BFLY_ROW(0, 8, CTRL, MLO) HGN_BFLY_ROW(1, 9, CTRL, MLO) HGN_BFLY_ROW(2, 10, CTRL, MLO) HGN_BFLY_ROW(3, 11, CTRL, MLO)     \
      HGN_BFLY_ROW(4, 12, CTRL, MLO) HGN_BFLY_ROW(5, 13, CTRL, MLO) HGN_BFLY_ROW(6, 14, CTRL, MLO) HGN_BFLY_ROW(7, 15, CTRL, MLO)   \
      HGN_BFLY_ROW(0, 16, CTRL, MHI) HGN_BFLY_ROW(1, 17, CTRL, MHI) HGN_BFLY_ROW(2, 18, CTRL, MHI) HGN_BFLY_ROW(3, 19, CTRL, MHI)   \
      HGN_BFLY_ROW(4, 20, CTRL, MHI) HGN_BFLY_ROW(5, 21, CTRL, MHI) HGN_BFLY_ROW(6, 22, CTRL, MHI) HGN_BFLY_ROW(7, 23, CTRL, MHI)   \
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])                 \
      : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(lo[4]), "v"(lo[5]), "v"(lo[6]), "v"(lo[7]),                        \
        "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(hi[4]), "v"(hi[5]), "v"(hi[6]), "v"(hi[7]))
// o[k] = lo[k] + partner's lo[k] in the lanes whose selecting bit is 0, hi[k] + partner's hi[k] in the others
template <int BIT>
__device__ __forceinline__ void bfly8(const float (&lo)[8], const float (&hi)[8], float (&o)[8]) {
  static_assert(BIT == 3 || BIT == 2, "bank masks select by lane bits 3 and 2 only");
  if constexpr (BIT == 3) HGN_BFLY8("row_ror:8", "0x3", "0xc");
  else HGN_BFLY8("row_half_mirror", "0x5", "0xa");
}

__device__ __forceinline__ void row16_sums_transposed(const Act& a, float (&out)[2]) {
  const int lane = threadIdx.x & 63;
  const bool b1 = lane & 2, b0 = lane & 1;
  // (every lane has a partner under these controls: no `old` value to keep, so none is materialised)
  auto dpp = [](float x, auto ctrl) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true)); };
  float s16[2][8], s8[8], s4[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {                       // values of feature blocks 2 h, 2 h + 1 with those of 2 h + 4, 2 h + 5
    float lo[8], hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { lo[k] = a.v[2 * h + (k >> 2)][k & 3]; hi[k] = a.v[2 * h + 4 + (k >> 2)][k & 3]; }
    bfly8<3>(lo, hi, s16[h]);
  }
  bfly8<2>(s16[0], s16[1], s8);                       // s8[4 fb + w], fb < 2
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float lo = s8[w], hi = s8[4 + w];
    s4[w] = (b1 ? hi : lo) + dpp(b1 ? lo : hi, std::integral_constant<int, 0x4E>{});               // quad_perm [2,3,0,1]
  }
#pragma unroll
  for (int w = 0; w < 2; ++w) {
    const float lo = s4[w], hi = s4[w + 2];
    out[w] = (b0 ? hi : lo) + dpp(b0 ? lo : hi, std::integral_constant<int, 0xB1>{});              // quad_perm [1,0,3,2]
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Butterflies across the four 16-lane rows of a wavefront (lanes l, l ^ 16, l ^ 32, l ^ 48 hold the four quarters of one tile row)
// on gfx950's v_permlane16_swap / v_permlane32_swap: both operands of a step arrive by ONE vector instruction (swap rows 1 <-> 0',
// 3 <-> 2' of two copies of the value, then halves) instead of a ds_bpermute round trip through the LDS crossbar (~120 cycles of
// latency each, two per reduction, in the dependent chain of every LayerNorm and every row scale).  Same operands per lane as
// x op shfl_xor(x, 16), then op shfl_xor(., 32): commutative ops give the same bits.
// ---------------------------------------------------------------------------------------------------------------------
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <class Op>
__device__ __forceinline__ float rows4_reduce(float t, Op op) {
  u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  t = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
  return op(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float rows4_sum(float t) { return rows4_reduce(t, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float rows4_max(float t) { return rows4_reduce(t, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ int rows4_min_i(int t) {
  u32x2_t r = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false);
  t = min((int)r[0], (int)r[1]);
  r = __builtin_amdgcn_permlane32_swap((unsigned)t, (unsigned)t, false, false);
  return min((int)r[0], (int)r[1]);
}

// Sum over the 128 features of a row: 32 in-lane values, then the three partner lanes (same n, other kq).
__device__ __forceinline__ float row_sum(const Act& a) {
  float s0 = 0.f, s1 = 0.f;
  HGN_FOR_B(fb) {
    float p = a.v[fb][0] + a.v[fb][1], q = a.v[fb][2] + a.v[fb][3];
    asm("" : "+v"(p), "+v"(q));      // four scalar adds per block: packed adds would pair (0, 2) with (1, 3) and pay four moves to get there
    s0 += p;
    s1 += q;
  }
  float t = s0 + s1;
  return rows4_sum(t);
}

// Sum over the row of a * b, the products rounded on their own (no fused multiply-add), in the summation order of row_sum.
__device__ __forceinline__ float row_dot(const Act& a, const Act& b) {
  float s0 = 0.f, s1 = 0.f;
  HGN_FOR_B(fb) {
    float p = __fmul_rn(a.v[fb][0], b.v[fb][0]) + __fmul_rn(a.v[fb][1], b.v[fb][1]);
    float q = __fmul_rn(a.v[fb][2], b.v[fb][2]) + __fmul_rn(a.v[fb][3], b.v[fb][3]);
    asm("" : "+v"(p), "+v"(q));
    s0 += p;
    s1 += q;
  }
  float t = s0 + s1;
  return rows4_sum(t);
}

// row_sum of the squares, with the squares rounded on their own (no fused multiply-add): the same bits as row_sum of a tile
// that holds a * a, without that tile
__device__ __forceinline__ float row_sum_sq(const Act& a) {
  float s0 = 0.f, s1 = 0.f;
  HGN_FOR_B(fb) {
    s0 += __fmul_rn(a.v[fb][0], a.v[fb][0]) + __fmul_rn(a.v[fb][1], a.v[fb][1]);
    s1 += __fmul_rn(a.v[fb][2], a.v[fb][2]) + __fmul_rn(a.v[fb][3], a.v[fb][3]);
  }
  float t = s0 + s1;
  return rows4_sum(t);
}

// ---------------------------------------------------------------------------------------------------------------------
// One 128-wide contraction block = two staged halves.  `between()` is called once, after the first half's DMA has been
// issued and before the wait, so the caller's own global loads fly together with it.
//   gemm_n: acc[ob] += W[out rows < rows][k < kw] * b          (W = &Wt[0][k0], forward form)
//   gemm_t: acc[ob] += W[j < jw][cols < cols]^T * b            (W = &Wt[0][c0], transposed form; contraction over rows j)
// ---------------------------------------------------------------------------------------------------------------------
template <class F>
__device__ __forceinline__ void gemm_n(Act& acc, const Act& b, float* __restrict__ wlds, const float* __restrict__ W,
                                       long ldw, int rows, int kw, F&& between) {
  const int nob = (rows + 15) >> 4, ncb = (kw + 15) >> 4;
  wg_barrier_lds();
  stage_half_n(wlds, W, ldw, rows, min(kw, HALF), 16 * nob, 16 * min(ncb, 4));
  between();
  __syncthreads();
  mfma_half_n<0>(acc, b, wlds, nob, min(ncb, 4));
  if (ncb > 4) {
    wg_barrier_lds();
    stage_half_n(wlds, W + HALF, ldw, rows, kw - HALF, 16 * nob, 16 * (ncb - 4));
    __syncthreads();
    mfma_half_n<1>(acc, b, wlds, nob, ncb - 4);
  }
}

template <class F>
__device__ __forceinline__ void gemm_t(Act& acc, const Act& b, float* __restrict__ wlds, const float* __restrict__ W,
                                       long ldw, int jw, int cols, F&& between) {
  const int nob = (cols + 15) >> 4, ncb = (jw + 15) >> 4;
  wg_barrier_lds();
  stage_half_t(wlds, W, ldw, min(jw, HALF), cols, 16 * min(ncb, 4), 16 * nob);
  between();
  __syncthreads();
  mfma_half_t<0>(acc, b, wlds, nob, min(ncb, 4));
  if (ncb > 4) {
    wg_barrier_lds();
    stage_half_t(wlds, W + (long)HALF * ldw, ldw, jw - HALF, cols, 16 * (ncb - 4), 16 * nob);
    __syncthreads();
    mfma_half_t<1>(acc, b, wlds, nob, ncb - 4);
  }
}

}  // namespace hgn
