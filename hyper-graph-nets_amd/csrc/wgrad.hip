// Weight / bias / LayerNorm-affine gradients of the fused MLPs, plus the flat Adam step.
//
// dW[j][k] = sum_i G[i][j] * A[i][k] contracts over ROWS (edges / nodes).  Each workgroup owns a contiguous chunk of rows and
// keeps its 32x128 slice-per-wave of dW in accumulators across the whole chunk; chunk partials go to slabs that a second
// kernel adds in fixed order (deterministic, no float atomics -- MI355X global float atomics run at ~1.3 TB/s and are order
// dependent).  Kernels, in the order they are tried for a task:
//   wgrad6s_kernel   (default) split-bf16 products; operand rows loaded into registers, split ONCE and handed to all waves as
//                    ready-made bf16x8 operand vectors through LDS
//   wgrad6_kernel    the previous form (LDS-DMA ring of fp32 tiles, every wave splits what it reads; HGN_WGRAD_RESPLIT=1)
//   wgrad_dma_kernel plain fp32 MFMAs from an LDS-DMA ring (HGN_FP32_MFMA=1)
//   wgrad_kernel     register-staged fp32 fallback for narrow / gathered operands (encoder inputs) and LayerNorm-affine tasks
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hgn_host.h"
#include "split_bf16.h"

namespace hgn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int WGW = 256;                          // 4 waves: wave w owns dW rows [32w, 32w+32)
// row of a 32x32 MFMA C/D tile held by register r of lane half h:  rho0(r) + 4*h
__device__ __forceinline__ constexpr int rho0(int r) { return (r & 3) + 8 * (r >> 2); }

constexpr int WT_ROWS = 32;                       // rows per LDS tile
constexpr int SLAB = 128 * 128 + 128;             // floats per (task, chunk): dW partial + colsum partial

struct WTaskDev {
  int type; const float* A; long lda; int K; const int* idxA; const float* G; long ldg; float* slab;
};
struct WArgs { WTaskDev t[HGN_MAX_WTASK]; long M; long rows_per_chunk; int n_chunks; int task0; };

__device__ __forceinline__ void wt_load_tile(float4 (&ra)[4], float4 (&rg)[4], const WTaskDev& t, long row0, long row_end,
                                             bool vecA) {
  // thread -> (row = tid>>3, 16-byte column group (tid&7) + 8*q)
  const int r = threadIdx.x >> 3;
  const long row = row0 + r;
  const bool ok = row < row_end;
  const long arow = ok ? (t.idxA ? (long)t.idxA[row] : row) : 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = ((threadIdx.x & 7) + 8 * q) * 4;
    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
      vg = *reinterpret_cast<const float4*>(t.G + row * t.ldg + c);
      if (vecA) {
        if (c < t.K) va = *reinterpret_cast<const float4*>(t.A + arow * t.lda + c);
      } else {
        const float* p = t.A + arow * t.lda;
        va.x = c + 0 < t.K ? p[c + 0] : 0.f; va.y = c + 1 < t.K ? p[c + 1] : 0.f;
        va.z = c + 2 < t.K ? p[c + 2] : 0.f; va.w = c + 3 < t.K ? p[c + 3] : 0.f;
      }
    }
    ra[q] = va; rg[q] = vg;
  }
}

__device__ __forceinline__ void wt_store_tile(float* __restrict__ As, float* __restrict__ Gs, const float4 (&ra)[4],
                                              const float4 (&rg)[4]) {
  const int r = threadIdx.x >> 3;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = ((threadIdx.x & 7) + 8 * q) * 4;
    *reinterpret_cast<float4*>(As + r * 128 + c) = ra[q];
    *reinterpret_cast<float4*>(Gs + r * 128 + c) = rg[q];
  }
}

__global__ __launch_bounds__(WGW, 2) void wgrad_kernel(const WArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][WT_ROWS * 128];   // [buf][A|G][32][128]  = 64 KB
  const WTaskDev t = a.t[a.task0 + blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 31, h = lane >> 5;
  const long row_beg = (long)blockIdx.x * a.rows_per_chunk;
  const long row_end = min(a.M, row_beg + a.rows_per_chunk);
  float* slab = t.slab + (long)blockIdx.x * SLAB;
  const bool vecA = ((t.lda & 3) == 0) && ((t.K & 3) == 0) && ((reinterpret_cast<uintptr_t>(t.A) & 15) == 0);
  const int ncb = (t.K + 31) >> 5;
  const int cs_col = threadIdx.x & 127, cs_half = threadIdx.x >> 7;

  f32x16 acc[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
  float cs0 = 0.f, cs1 = 0.f;       // column sums: type 0: sum G ; type 1: sum G*A (cs0) and sum G (cs1)
  float4 ra[4], rg[4];
  if (row_beg < row_end) {
    wt_load_tile(ra, rg, t, row_beg, row_end, vecA);
    wt_store_tile(lds[0][0], lds[0][1], ra, rg);
  }
  __syncthreads();
  int buf = 0;
  for (long r0 = row_beg; r0 < row_end; r0 += WT_ROWS) {
    const bool more = r0 + WT_ROWS < row_end;
    if (more) wt_load_tile(ra, rg, t, r0 + WT_ROWS, row_end, vecA);      // in flight during the MFMAs below
    const float* As = lds[buf][0];
    const float* Gs = lds[buf][1];
    if (t.type == 0) {
#pragma unroll
      for (int s = 0; s < WT_ROWS / 2; ++s) {
        const float ga = Gs[(2 * s + h) * 128 + 32 * wave + m];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
          if (kb < ncb) acc[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, As[(2 * s + h) * 128 + 32 * kb + m], acc[kb], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) cs0 += Gs[(16 * cs_half + r) * 128 + cs_col];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float g = Gs[(16 * cs_half + r) * 128 + cs_col];
        cs0 += g * As[(16 * cs_half + r) * 128 + cs_col];
        cs1 += g;
      }
    }
    if (more) wt_store_tile(lds[buf ^ 1][0], lds[buf ^ 1][1], ra, rg);
    __syncthreads();
    buf ^= 1;
  }
  // ---- write the chunk partial ---------------------------------------------------------------------------
  if (t.type == 0) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(32 * wave + rho0(r) + 4 * h) * 128 + 32 * kb + m] = acc[kb][r];
  }
  float* red = &lds[0][0][0];
  __syncthreads();
  red[threadIdx.x] = cs0;
  red[256 + threadIdx.x] = cs1;
  __syncthreads();
  if (threadIdx.x < 128) {
    if (t.type == 0) slab[128 * 128 + threadIdx.x] = red[threadIdx.x] + red[128 + threadIdx.x];
    else {
      slab[threadIdx.x] = red[threadIdx.x] + red[128 + threadIdx.x];                       // dgamma partial
      slab[128 * 128 + threadIdx.x] = red[256 + threadIdx.x] + red[256 + 128 + threadIdx.x];   // dbeta partial
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// LDS-DMA pipelined variant for the common case (both operands 128 wide, 16-byte aligned rows, no gather): row tiles of
// A and G go global -> LDS by global_load_lds_dwordx4 (no VGPRs) into a 4-slot ring, three tiles in flight per workgroup
// behind COUNTED vmcnt waits and raw s_barrier (a __syncthreads() would drain the ring: it carries vmcnt(0)).  All LDS is
// one array (a second __shared__ object makes hipcc wait vmcnt(0) before every ds_read).
// ----------------------------------------------------------------------------------------------------------------
constexpr int DT = 16;       // rows per ring slot
constexpr int NS = 4;        // ring slots

__device__ __forceinline__ void wg_dma_issue(float* __restrict__ slotA, float* __restrict__ slotG, const WTaskDev& t,
                                             long tile_row0, long M) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (unsigned q = 0; q < 2; ++q) {
    const unsigned i = wave * 2 + q;                       // 8 wave-instructions of 1 KiB cover the 16x128 tile
    long gr = tile_row0 + 2 * i + (lane >> 5);
    gr = gr < M ? gr : M - 1;                              // clamp (rows past the chunk end are masked in the MFMA loop)
    const unsigned c = 4 * (lane & 31);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t.A + gr * t.lda + c),
                                     (__attribute__((address_space(3))) void*)(slotA + i * 256), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t.G + gr * t.ldg + c),
                                     (__attribute__((address_space(3))) void*)(slotG + i * 256), 16, 0, 0);
  }
}

__global__ __launch_bounds__(WGW, 2) void wgrad_dma_kernel(const WArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[NS * 2 * DT * 128];      // [slot][A|G][16][128] = 64 KB
  const WTaskDev t = a.t[a.task0 + blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 31, h = lane >> 5;
  const long row_beg = (long)blockIdx.x * a.rows_per_chunk;
  const long row_end = min(a.M, row_beg + a.rows_per_chunk);
  float* slab = t.slab + (long)blockIdx.x * SLAB;
  const int ntiles = row_beg < row_end ? (int)((row_end - row_beg + DT - 1) / DT) : 0;
  const int cs_col = threadIdx.x & 127, cs_half = threadIdx.x >> 7;

  f32x16 acc[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
  float cs0 = 0.f;
  for (int j = 0; j < NS - 1 && j < ntiles; ++j)
    wg_dma_issue(lds + (j * 2) * DT * 128, lds + (j * 2 + 1) * DT * 128, t, row_beg + (long)j * DT, a.M);
  for (int it = 0; it < ntiles; ++it) {
    const int younger = min(NS - 2, ntiles - 1 - it);        // tiles issued after tile `it` by this wave (4 DMAs each)
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // tile `it` landed for every wave; slot (it-1)%NS is free
    if (it + NS - 1 < ntiles) {
      const int sl = (it + NS - 1) % NS;
      wg_dma_issue(lds + (sl * 2) * DT * 128, lds + (sl * 2 + 1) * DT * 128, t, row_beg + (long)(it + NS - 1) * DT, a.M);
    }
    const float* As = lds + ((it % NS) * 2) * DT * 128;
    const float* Gs = As + DT * 128;
    const long r0 = row_beg + (long)it * DT;
    const bool partial = r0 + DT > row_end;
#pragma unroll
    for (int s = 0; s < DT / 2; ++s) {
      float ga = Gs[(2 * s + h) * 128 + 32 * wave + m];
      if (partial && r0 + 2 * s + h >= row_end) ga = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        acc[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga, As[(2 * s + h) * 128 + 32 * kb + m], acc[kb], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < DT / 2; ++r) {
      const int rr = (DT / 2) * cs_half + r;
      const float g = Gs[rr * 128 + cs_col];
      cs0 += (partial && r0 + rr >= row_end) ? 0.f : g;
    }
  }
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[(32 * wave + rho0(r) + 4 * h) * 128 + 32 * kb + m] = acc[kb][r];
  __builtin_amdgcn_s_barrier();
  lds[threadIdx.x] = cs0;
  __syncthreads();
  if (threadIdx.x < 128) slab[128 * 128 + threadIdx.x] = lds[threadIdx.x] + lds[128 + threadIdx.x];
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-bf16 variant of the DMA kernel: dW = G^T A with the contraction over ROWS on v_mfma_f32_16x16x32_bf16 -- both operands
// are read from the row-major fp32 LDS tiles "transposed" (a lane takes 8 consecutive rows of one feature: eight conflict-free
// ds_read_b32), split into three bf16 terms in registers and multiplied with six MFMAs (see csrc/mlp6.hip for the arithmetic).
// 32 rows per contraction block = two 16-row ring slots (a "pair"); the 4-slot ring holds the pair being multiplied and the
// pair in flight.  16-byte groups of a row are XOR-swizzled with (row >> 3) so that the four 8-row groups of a read hit
// different bank halves.  Per pair and wave: 80 values split (VALU) against 96 MFMAs of 16 cycles -- half the cycles of the
// fp32 MFMA formulation (256 MFMAs of 64... cycles per 32 rows), which leaves the kernel HBM-bound.
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wg_dma_issue6(float* __restrict__ slotA, float* __restrict__ slotG, const WTaskDev& t,
                                              long tile_row0, long M, int parity) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (unsigned q = 0; q < 2; ++q) {
    const unsigned i = wave * 2 + q;                       // 8 wave-instructions of 1 KiB cover the 16x128 tile
    const unsigned r = 2 * i + (lane >> 5);                // row inside the slot
    long gr = tile_row0 + r;
    gr = gr < M ? gr : M - 1;
    const unsigned kg = 2 * parity + (r >> 3);             // 8-row group inside the 32-row pair
    const unsigned c = 4 * ((lane & 31) ^ (kg << 2));      // LDS group position (lane & 31) holds source group ^ (kg << 2)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t.A + gr * t.lda + c),
                                     (__attribute__((address_space(3))) void*)(slotA + i * 256), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(t.G + gr * t.ldg + c),
                                     (__attribute__((address_space(3))) void*)(slotG + i * 256), 16, 0, 0);
  }
}

__device__ __forceinline__ void split3v(const float (&v)[8], bf16x8 (&s)[3]) {
  hgn_split::eight(v, s);
}

#if HGN_LAB   // the previous split-bf16 weight-gradient kernel: laboratory build only (HGN_WGRAD_RESPLIT=1)
__global__ __launch_bounds__(WGW, 2) void wgrad6_kernel(const WArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[NS * 2 * DT * 128];      // [slot][A|G][16][128] = 64 KB, NS = 4
  const WTaskDev t = a.t[a.task0 + blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kg = lane >> 4;
  const long row_beg = (long)blockIdx.x * a.rows_per_chunk;
  const long row_end = min(a.M, row_beg + a.rows_per_chunk);
  float* slab = t.slab + (long)blockIdx.x * SLAB;
  const int ntiles = row_beg < row_end ? (int)((row_end - row_beg + DT - 1) / DT) : 0;
  const int npairs = (ntiles + 1) >> 1;
  const int cs_col = threadIdx.x & 127, cs_half = threadIdx.x >> 7;

  f32x4 acc[2][8];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float cs0 = 0.f;
  auto issue_pair = [&](int p) {
    const int s0 = 2 * (p & 1);
    wg_dma_issue6(lds + (s0 * 2) * DT * 128, lds + (s0 * 2 + 1) * DT * 128, t, row_beg + (long)(2 * p) * DT, a.M, 0);
    wg_dma_issue6(lds + ((s0 + 1) * 2) * DT * 128, lds + ((s0 + 1) * 2 + 1) * DT * 128, t, row_beg + (long)(2 * p + 1) * DT, a.M, 1);
  };
  if (npairs > 0) issue_pair(0);
  // this lane's 8 rows of the pair: rows 8kg .. 8kg+7 -> slot kg >> 1, slot rows 8(kg & 1) + j; swizzle value kg << 2
  const int slot_of = kg >> 1, r_base = 8 * (kg & 1), swz = kg << 2;
  for (int p = 0; p < npairs; ++p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // pair p landed for every wave; the slots of pair p-1 are free
    if (p + 1 < npairs) issue_pair(p + 1);
    const int s0 = 2 * (p & 1);
    const float* As = lds + ((s0 + slot_of) * 2) * DT * 128 + r_base * 128;
    const float* Gs = As + DT * 128;
    const long r0 = row_beg + (long)p * 2 * DT + 8 * kg;       // global row of j = 0
    // G operand (this wave's 32 dW rows = two 16-feature blocks), zeroed beyond the chunk end
    bf16x8 gs[2][3];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int f = 32 * wave + 16 * mb + m;
      const int col = (((f >> 2) ^ swz) << 2) | (f & 3);
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (r0 + j < row_end) ? Gs[j * 128 + col] : 0.f;
      split3v(v, gs[mb]);
    }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const int f = 16 * nb + m;
      const int col = (((f >> 2) ^ swz) << 2) | (f & 3);
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = As[j * 128 + col];
      bf16x8 as[3];
      split3v(v, as);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        f32x4 c = acc[mb][nb];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][2], as[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], as[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], as[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], as[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], as[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], as[0], c, 0, 0, 0);
        acc[mb][nb] = c;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // column sums of G (bias gradient): thread (cs_half, cs_col) sums rows [16 cs_half, 16 cs_half + 16) of the pair
    {
      const float* Gp = lds + ((s0 + cs_half) * 2 + 1) * DT * 128;
      const long rr0 = row_beg + (long)p * 2 * DT + 16 * cs_half;
#pragma unroll
      for (int r = 0; r < DT; ++r) {
        const int sw = (2 * cs_half + (r >> 3)) << 2;
        const float g = Gp[r * 128 + ((((cs_col >> 2) ^ sw) << 2) | (cs_col & 3))];
        cs0 += (rr0 + r < row_end) ? g : 0.f;
      }
    }
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(32 * wave + 16 * mb + 4 * kg + r) * 128 + 16 * nb + m] = acc[mb][nb][r];
  __builtin_amdgcn_s_barrier();
  lds[threadIdx.x] = cs0;
  __syncthreads();
  if (threadIdx.x < 128) slab[128 * 128 + threadIdx.x] = lds[threadIdx.x] + lds[128 + threadIdx.x];
}

#endif

// ---------------------------------------------------------------------------------------------------------------------
// Split-bf16 weight gradients, every value split ONCE: the two operand tiles of a 32-row contraction block (A and G, 32 x 128
// fp32 each) are loaded straight into registers -- a lane takes four features of eight consecutive rows (eight 16-byte loads,
// a half wave covers one whole 512-byte row) --, split into three bf16 terms there and written to LDS as ready-made MFMA
// operand vectors (eight rows of one feature = one bf16x8), layout [array][split][row group][feature] so that operand reads
// are linear ds_read_b128.  The rows of block p+1 are in flight while block p is multiplied.  Against wgrad6_kernel (each of
// the four waves re-reads and re-splits the whole A tile): 32 instead of 80 values split per lane and block, 30 b128 reads
// instead of 80 b32 reads; 48 KB of LDS, two workgroups per CU (the launch has 512).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int OPS_VEC = 2 * 3 * 4 * 128;          // bf16x8 vectors per block: [A|G][split][row group][feature]

// NP = 3 (two scaled fp16 terms, three products; csrc/split_bf16.h): the contraction runs over ROWS, so a scale must be uniform over the
// 32 rows of a block: every producer wave leaves the largest magnitude of its 16 rows in LDS before the barrier that frees the
// operand image, all waves derive the block's two exponents (eA, eG) from the four values, the block is multiplied from ZERO
// accumulators on the matrix pipe and added to the fp32 accumulators on the vector pipe with the factor 2^-(eA + eG) (exact): no
// running scale, no overflow however the gradient magnitude moves along the rows.
template <int NP>          // 6 / 3: fp32-accurate split products; 1: single bf16 product (hgn_set_matmul_products)
__global__ __launch_bounds__(WGW, 2) void wgrad6s_kernel(const WArgs a) {
  __shared__ __attribute__((aligned(16))) bf16x8 ops[OPS_VEC];                 // 48 KB
  __shared__ float blkmax[2][2];                                                // [array][producer wave of the array]
  constexpr int NSP = NP == 6 ? 3 : (NP == 3 ? 2 : 1);
  const WTaskDev t = a.t[a.task0 + blockIdx.y];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kg = lane >> 4;
  // Workgroup x takes the 32-row blocks x, x + n_chunks, x + 2 n_chunks, ...: all workgroups sweep the two arrays TOGETHER, front to back
  // (one coherent stream through HBM instead of 512 private ones: -3 % per launch, -0.5 ms per step against contiguous chunks).
  const long nblk_all = (a.M + 31) / 32;
  const long row_end = a.M;
  float* slab = t.slab + (long)blockIdx.x * SLAB;
  const int nblocks = nblk_all > (long)blockIdx.x ? (int)((nblk_all - (long)blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
  // producer role of this lane: array (waves 0,1: A; waves 2,3: G), row group of the block, feature quad
  const int arr = wave >> 1, kgp = 2 * (wave & 1) + (lane >> 5), q = lane & 31;
  const float* src = arr ? t.G : t.A;
  const long ld = arr ? t.ldg : t.lda;

  f32x4 acc[2][8];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};             // column sums of G (bias gradient) of this lane's quad and row group
  // TWO blocks of operand rows in flight per workgroup (two register buffers; a launch has 512 workgroups = two per CU, so the registers
  // are there): with one, a workgroup had 32 KB under way for the length of a product phase and waited out the rest of the memory
  // latency at every block (50 requests in flight per CU, profiles/r05_memory_counters.md): dW1e 0.249 -> 0.235 ms, a node-level task
  // 0.099 -> 0.095.  Every fetch inside the loop is unconditional (rows past the end are clamped to the last row and never used).
  f32x4 xa[8], xb[8];
  auto fetch = [&](int p, f32x4 (&x)[8]) {
    const long r0 = ((long)p * gridDim.x + blockIdx.x) * 32 + 8 * kgp;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long r = min(r0 + j, a.M - 1);
      x[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + r * ld + 4 * q));      // streamed once (see csrc/segment.hip: stream_load4)
    }
  };
  if (nblocks > 0) {
    fetch(0, xa);
    fetch(1, xb);
  }
  auto body = [&](int p, f32x4 (&x)[8]) {
    const long r0 = ((long)p * gridDim.x + blockIdx.x) * 32 + 8 * kgp;
    if (arr) {                                      // rows past the chunk end contribute nothing
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (r0 + j >= row_end) x[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) cs += x[j];
    }
    if constexpr (NP == 3) {
      float mx = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(x[j][0]), fabsf(x[j][1]))), fmaxf(fabsf(x[j][2]), fabsf(x[j][3])));
#pragma unroll
      for (int sh = 32; sh >= 1; sh >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
      if (lane == 0) blkmax[arr][wave & 1] = mx;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                   // every wave has finished reading the operands of block p-1
    float unscale = 1.f, unscale2 = 1.f, sc = 1.f;
    if constexpr (NP == 3) {
      const int eA = hgn_split::scale_exp_of(fmaxf(blkmax[0][0], blkmax[0][1]));
      const int eG = hgn_split::scale_exp_of(fmaxf(blkmax[1][0], blkmax[1][1]));
      sc = hgn_split::pow2f(arr ? eG : eA);
      unscale = hgn_split::pow2f(-eA);                 // (two factors: their product 2^-(eA + eG) need not be a normal number)
      unscale2 = hgn_split::pow2f(-eG);
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = x[j][f];
      bf16x8 sp[3];
      if constexpr (NP == 3) hgn_split::eight16(v, sc, sp); else split3v(v, sp);
#pragma unroll
      for (int sidx = 0; sidx < NSP; ++sidx) ops[((arr * 3 + sidx) * 4 + kgp) * 128 + 4 * q + f] = sp[sidx];
    }
    fetch(p + 2, x);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // operands of block p complete
    bf16x8 gs[2][3];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int sidx = 0; sidx < NSP; ++sidx) gs[mb][sidx] = ops[((3 + sidx) * 4 + kg) * 128 + 32 * wave + 16 * mb + m];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      bf16x8 as[3];
#pragma unroll
      for (int sidx = 0; sidx < NSP; ++sidx) as[sidx] = ops[(sidx * 4 + kg) * 128 + 16 * nb + m];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        if constexpr (NP == 3) {                  // the block's product from zero (smallest terms first), then into the fp32 accumulators
          typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
          f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, gs[mb][1]), __builtin_bit_cast(f16x8, as[0]), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, gs[mb][0]), __builtin_bit_cast(f16x8, as[1]), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, gs[mb][0]), __builtin_bit_cast(f16x8, as[0]), c, 0, 0, 0);
          acc[mb][nb] += (c * unscale) * unscale2;
          continue;
        }
        f32x4 c = acc[mb][nb];
        if (NP != 1) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][2], as[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], as[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], as[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], as[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], as[1], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], as[0], c, 0, 0, 0);
        acc[mb][nb] = c;
      }
    }
  };
  for (int p = 0; p < nblocks; p += 2) {
    body(p, xa);
    if (p + 1 < nblocks) body(p + 1, xb);
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(32 * wave + 16 * mb + 4 * kg + r) * 128 + 16 * nb + m] = acc[mb][nb][r];
  // bias gradient: the four row groups of a feature quad live in four G-producer lanes -> fixed-order sum through LDS
  __syncthreads();
  float* csl = reinterpret_cast<float*>(ops);
  if (arr) *reinterpret_cast<f32x4*>(csl + kgp * 128 + 4 * q) = cs;
  __syncthreads();
  if (threadIdx.x < 128)
    slab[128 * 128 + threadIdx.x] = (csl[threadIdx.x] + csl[128 + threadIdx.x]) + (csl[256 + threadIdx.x] + csl[384 + threadIdx.x]);
}

struct RTaskDev { int type; int K; int n_out; int acc; int n_chunks; float* dW; long ldw; float* db; const float* slab; long stride; };
struct RArgs { RTaskDev t[HGN_MAX_WTASK]; };
struct RArgsBatch { RTaskDev t[HGN_MAX_WRED]; };      // hgn_slab_reduce_batch: the pending sums of several calls in one launch

// Fixed-order sum of the chunk slabs.  512 threads = 64 consecutive slab elements x 8 chunk groups: group g adds the chunks
// c = g, g + 8, ... (four loads in flight), the eight partial sums are combined through LDS in a fixed tree -- eight times
// the loads in flight of a one-thread-per-element loop, which at ~170 chunks per task was pure latency (60 us per launch).
constexpr int RED_ELEMS = 64, RED_GROUPS = 8;
template <class ARGS>
__global__ __launch_bounds__(RED_ELEMS * RED_GROUPS) void wgrad_reduce_kernel(const ARGS a) {
  __shared__ float part[RED_GROUPS][RED_ELEMS];
  const RTaskDev t = a.t[blockIdx.y];
  const int el = threadIdx.x & (RED_ELEMS - 1), grp = threadIdx.x / RED_ELEMS;
  const int e = blockIdx.x * RED_ELEMS + el;
  const bool is_bias = e >= 128 * 128;
  const int j = is_bias ? e - 128 * 128 : e >> 7, k = e & 127;
  bool live = e < SLAB;
  if (live) {
    if (t.type == 0) live = j < t.n_out && (is_bias ? t.db != nullptr : k < t.K);
    else live = is_bias ? t.db != nullptr : e < 128;
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (live) {
    const float* p = t.slab + e;
    int c = grp;
    const long st = t.stride;
    for (; c + 3 * RED_GROUPS < t.n_chunks; c += 4 * RED_GROUPS) {
      s0 += p[(long)c * st]; s1 += p[(long)(c + RED_GROUPS) * st];
      s2 += p[(long)(c + 2 * RED_GROUPS) * st]; s3 += p[(long)(c + 3 * RED_GROUPS) * st];
    }
    for (; c < t.n_chunks; c += RED_GROUPS) s0 += p[(long)c * st];
  }
  part[grp][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp != 0 || !live) return;
  const float s = ((part[0][el] + part[1][el]) + (part[2][el] + part[3][el])) + ((part[4][el] + part[5][el]) + (part[6][el] + part[7][el]));
  float* dst;
  if (t.type == 0) dst = is_bias ? t.db + j : t.dW + (long)j * t.ldw + k;
  else dst = is_bias ? t.db + j : t.dW + e;
  *dst = t.acc ? *dst + s : s;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, float gscale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;          // torch.optim.Adam: (sqrt(v)/sqrt(bc2)) + eps
  p[i] -= (lr / bc1) * (mi / denom);
}

// Chunk counts: the MFMA tasks (type 0) get one full round of workgroups (2 per CU x 256 CUs) split evenly over the
// tasks; the LayerNorm-affine tasks (type 1: no MFMA, pure streaming) are launched separately with many small chunks.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                const int* __restrict__ step, float gscale) {
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const double t = (double)(*step);
    bc[0] = (float)(1.0 - pow((double)b1, t));
    bc[1] = (float)sqrt(1.0 - pow((double)b2, t));
  }
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  p[i] -= (lr / bc[0]) * (mi / (sqrtf(vi) / bc[1] + eps));
}
__global__ void step_incr_kernel(int* step) { *step += 1; }

constexpr int WG_CHUNKS = 512;                    // workgroups of one weight-gradient launch: two per CU (768 = three per CU measured no faster; wgrad6s_kernel is built for two)
static int chunks_mfma(long M, int n_tasks) {
  long c = WG_CHUNKS / (n_tasks > 0 ? n_tasks : 1);
  const long by_rows = (M + 63) / 64;
  if (c > by_rows) c = by_rows;
  if (c < 1) c = 1;
  return (int)c;
}
static int chunks_ln(long M) {
  long c = (M + 255) / 256;
  if (c > 1024) c = 1024;
  if (c < 1) c = 1;
  return (int)c;
}

int launch_slab_reduce(const SlabReduceTask* tasks, int n_tasks, hipStream_t stream) {
  if (!tasks || n_tasks < 1 || n_tasks > HGN_MAX_WTASK) return hgn_fail(HGN_E_INVALID, "launch_slab_reduce: bad task list");
  RArgs ra;
  for (int i = 0; i < n_tasks; ++i) {
    const SlabReduceTask& t = tasks[i];
    ra.t[i] = {t.type, t.K, t.n_out, t.acc, t.n_chunks, t.dW, t.ldw, t.db, t.slab, t.chunk_stride};
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel<RArgs>, dim3((SLAB + RED_ELEMS - 1) / RED_ELEMS, (unsigned)n_tasks), dim3(RED_ELEMS * RED_GROUPS), 0,
                     stream, ra);
  return hgn_check_launch("slab reduce");
}

}  // namespace hgn

using namespace hgn;

extern "C" int hgn_wgrad_workspace_bytes(int64_t M, int n_tasks, size_t* bytes) {
  if (!bytes || M < 0 || n_tasks < 0 || n_tasks > HGN_MAX_WTASK) return hgn_fail(HGN_E_INVALID, "hgn_wgrad_workspace_bytes: bad argument");
  // The DMA-eligible MFMA tasks share WG_CHUNKS chunks in total, the other MFMA tasks (narrow or gathered operands: their launch
  // follows the first one) WG_CHUNKS of their own, LN tasks use up to 1024 small ones each
  size_t mf = (size_t)2 * WG_CHUNKS * SLAB * sizeof(float);
  size_t ln = (size_t)chunks_ln(M) * SLAB * sizeof(float) * (size_t)n_tasks;
  *bytes = mf + ln + 256;
  return HGN_OK;
}

static int wgrad_impl(const hgn_wtask_t* tasks, int n_tasks, int64_t M, void* workspace, size_t ws_bytes, hgn_wred_task_t* red_out,
                      void* stream) {
  if (n_tasks == 0) return HGN_OK;
  size_t need = 0;
  if (!tasks || hgn_wgrad_workspace_bytes(M, n_tasks, &need) != HGN_OK || !workspace || ws_bytes < need)
    return hgn_fail(HGN_E_INVALID, "hgn_mlp_wgrad: bad tasks / workspace too small");
  // order: DMA-eligible GEMM tasks, other GEMM tasks, then LayerNorm-affine tasks
  auto dma_ok = [](const hgn_wtask_t& t) {
    return t.type == 0 && t.K == 128 && !t.idxA && (t.lda & 3) == 0 && ((uintptr_t)t.A & 15) == 0;
  };
  int order[HGN_MAX_WTASK], nd = 0, ng = 0, n1 = 0;
  for (int i = 0; i < n_tasks; ++i) if (dma_ok(tasks[i])) order[nd++] = i;
  for (int i = 0; i < n_tasks; ++i) if (tasks[i].type == 0 && !dma_ok(tasks[i])) order[nd + ng++] = i;
  for (int i = 0; i < n_tasks; ++i) if (tasks[i].type == 1) order[nd + ng + n1++] = i;
  const int n0 = nd + ng;
  if (n0 + n1 != n_tasks) return hgn_fail(HGN_E_INVALID, "hgn_mlp_wgrad: bad task type");
  for (int i = 0; i < n_tasks; ++i)
    if (!valid_products(tasks[i].products) || tasks[i].products != tasks[0].products || tasks[i].flags != tasks[0].flags)
      return hgn_fail(HGN_E_INVALID, "hgn_mlp_wgrad: products (0, 6, 3, 1 or 2) and flags must agree over the tasks of one launch");
  // The generic kernel (narrow / gathered operands: the encoders' first layers) keeps ONE 16 KB row tile in flight per workgroup: its
  // launch gets all WG_CHUNKS workgroups for its own tasks instead of the share the whole call would leave it (an edge encoder's
  // dW1 ran on 170 workgroups beside dW3 / dW2: 0.44-1.9 ms for 0.6-1.3 GB).
  const int nch0 = chunks_mfma(M, nd > 0 ? nd : 1), nchg = chunks_mfma(M, ng > 0 ? ng : 1), nch1 = chunks_ln(M);
  WArgs wa; RArgs ra;
  wa.M = M;
  float* slab = (float*)workspace;
  size_t off = 0;
  for (int q = 0; q < n_tasks; ++q) {
    const hgn_wtask_t& t = tasks[order[q]];
    if (!t.A || !t.G || !t.dW || t.K < 1 || t.K > 128 || t.n_out < 1 || t.n_out > 128 || (t.ldg & 3) ||
        ((uintptr_t)t.G & 15))
      return hgn_fail(HGN_E_INVALID, "hgn_mlp_wgrad: bad task");
    const int nch = q < nd ? nch0 : (q < n0 ? nchg : nch1);
    wa.t[q] = {t.type, t.A, (long)t.lda, t.K, t.idxA, t.G, (long)t.ldg, slab + off};
    ra.t[q] = {t.type, t.K, t.n_out, t.accumulate ? 1 : 0, nch, t.dW, (long)t.ldw, t.db, slab + off, (long)SLAB};
    off += (size_t)nch * SLAB;
  }
  auto rows_per = [&](int nch, int mult) {
    long rpc = (M + nch - 1) / nch;
    rpc = (rpc + mult - 1) / mult * mult;
    return rpc < mult ? (long)mult : rpc;
  };
  ProfScope ps(g_prof_tag == 1 ? 11 : 4, (double)M * n0, (hipStream_t)stream);
  if (nd) {
    const bool fp32_only = (tasks[0].flags & HGN_F_FP32_MFMA) != 0;
    wa.n_chunks = nch0; wa.task0 = 0;
    if (fp32_only) {
      wa.rows_per_chunk = rows_per(nch0, DT);
      hipLaunchKernelGGL(wgrad_dma_kernel, dim3((unsigned)nch0, (unsigned)nd), dim3(WGW), 0, (hipStream_t)stream, wa);
    } else {                                    // split-bf16 products: 32-row contraction blocks
      wa.rows_per_chunk = rows_per(nch0, 2 * DT);
#if HGN_LAB
      static const bool resplit = getenv("HGN_WGRAD_RESPLIT") != nullptr;     // the previous kernel, laboratory build only
      if (resplit) hipLaunchKernelGGL(wgrad6_kernel, dim3((unsigned)nch0, (unsigned)nd), dim3(WGW), 0, (hipStream_t)stream, wa);
      else
#endif
      if (bwd_products(tasks[0].products) == 1) hipLaunchKernelGGL(wgrad6s_kernel<1>, dim3((unsigned)nch0, (unsigned)nd), dim3(WGW), 0, (hipStream_t)stream, wa);
      else if (bwd_products(tasks[0].products) == 3) hipLaunchKernelGGL(wgrad6s_kernel<3>, dim3((unsigned)nch0, (unsigned)nd), dim3(WGW), 0, (hipStream_t)stream, wa);
      else hipLaunchKernelGGL(wgrad6s_kernel<6>, dim3((unsigned)nch0, (unsigned)nd), dim3(WGW), 0, (hipStream_t)stream, wa);
    }
  }
  if (ng) {
    wa.n_chunks = nchg; wa.rows_per_chunk = rows_per(nchg, WT_ROWS); wa.task0 = nd;
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)nchg, (unsigned)ng), dim3(WGW), 0, (hipStream_t)stream, wa);
  }
  if (n1) {
    wa.n_chunks = nch1; wa.rows_per_chunk = rows_per(nch1, WT_ROWS); wa.task0 = n0;
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)nch1, (unsigned)n1), dim3(WGW), 0, (hipStream_t)stream, wa);
  }
  if (red_out) {                                      // the caller takes the sums later (hgn_slab_reduce_batch); entry i <-> tasks[i]
    for (int q = 0; q < n_tasks; ++q) {
      const RTaskDev& r = ra.t[q];
      red_out[order[q]] = {r.type, r.K, r.n_out, r.acc, r.n_chunks, 0, r.dW, (int64_t)r.ldw, r.db, r.slab, (int64_t)r.stride};
    }
    return hgn_check_launch("hgn_mlp_wgrad_partial");
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel<RArgs>, dim3((SLAB + RED_ELEMS - 1) / RED_ELEMS, (unsigned)n_tasks), dim3(RED_ELEMS * RED_GROUPS), 0,
                     (hipStream_t)stream, ra);
  return hgn_check_launch("hgn_mlp_wgrad");
}
extern "C" int hgn_mlp_wgrad(const hgn_wtask_t* tasks, int n_tasks, int64_t M, void* workspace, size_t ws_bytes, void* stream) {
  return wgrad_impl(tasks, n_tasks, M, workspace, ws_bytes, nullptr, stream);
}
extern "C" int hgn_mlp_wgrad_partial(const hgn_wtask_t* tasks, int n_tasks, int64_t M, void* workspace, size_t ws_bytes,
                                     hgn_wred_task_t* red, void* stream) {
  if (!red) return hgn_fail(HGN_E_INVALID, "hgn_mlp_wgrad_partial: null descriptor array");
  return wgrad_impl(tasks, n_tasks, M, workspace, ws_bytes, red, stream);
}
extern "C" int hgn_slab_reduce_batch(const hgn_wred_task_t* red, int n, void* stream) {
  if (n == 0) return HGN_OK;
  if (!red || n < 0 || n > HGN_MAX_WRED) return hgn_fail(HGN_E_INVALID, "hgn_slab_reduce_batch: bad task list");
  RArgsBatch ra;
  for (int i = 0; i < n; ++i) {
    const hgn_wred_task_t& t = red[i];
    if (!t.dW || !t.slab || t.n_chunks < 1 || t.K < 1 || t.K > 128 || t.n_out < 1 || t.n_out > 128 || (t.type != 0 && t.type != 1))
      return hgn_fail(HGN_E_INVALID, "hgn_slab_reduce_batch: bad task");
    for (int j = 0; j < i; ++j)
      if (red[j].dW == t.dW || (t.db && red[j].db == t.db))
        return hgn_fail(HGN_E_INVALID, "hgn_slab_reduce_batch: two sums of one batch share a target");
    ra.t[i] = {t.type, t.K, t.n_out, t.accumulate ? 1 : 0, t.n_chunks, t.dW, (long)t.ldw, t.db, t.slab, (long)t.chunk_stride};
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel<RArgsBatch>, dim3((SLAB + RED_ELEMS - 1) / RED_ELEMS, (unsigned)n), dim3(RED_ELEMS * RED_GROUPS), 0,
                     (hipStream_t)stream, ra);
  return hgn_check_launch("hgn_slab_reduce_batch");
}

extern "C" int hgn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, int32_t step, float grad_scale, void* stream) {
  if (n == 0) return HGN_OK;
  if (!p || !g || !m || !v || n < 0 || step < 1) return hgn_fail(HGN_E_INVALID, "hgn_adam_step: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  ProfScope ps(9, (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr,
                     beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale);
  return hgn_check_launch("hgn_adam_step");
}

extern "C" int hgn_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                 float eps, int32_t* step_dev, float grad_scale, void* stream) {
  if (!step_dev) return hgn_fail(HGN_E_INVALID, "hgn_adam_step_dev: null step counter");
  hipLaunchKernelGGL(step_incr_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  if (n == 0) return hgn_check_launch("hgn_adam_step_dev");
  if (!p || !g || !m || !v || n < 0) return hgn_fail(HGN_E_INVALID, "hgn_adam_step_dev: bad argument");
  ProfScope ps(9, (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr,
                     beta1, beta2, eps, step_dev, grad_scale);
  return hgn_check_launch("hgn_adam_step_dev");
}
