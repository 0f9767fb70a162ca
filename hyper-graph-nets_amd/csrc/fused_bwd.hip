// Edge-block backward with the weight gradients IN THE SAME PASS (include/hgn_mp.h: hgn_edge_bwd_fused).
// Semantics: autograd of GraphNet._update_edge_features (src/migration/graphnet.py:22-32) -- LayerNorm backward -> W3^T -> relu'
// -> W2^T -> relu' -> W1e^T + residual -- plus dW3 = dz3^T z2, dW2 = dz2^T z1, their bias sums and the LayerNorm-affine gradients.
//
// Why one pass.  The separate kernels hand dz3 / dz2 from the data-gradient chain (csrc/mlp6.hip: mlp6_bwd_kernel) to the weight-
// gradient kernel (csrc/wgrad.hip: wgrad6s_kernel) through HBM: 1 KB per edge row written and read back.  Here they never leave
// the chip: 3.1 KB per row (read d(e'), x-hat, z2, z1, sign words; write de, dz1) instead of 5.6 KB for the two launches.
//
// Structure (second version; the first one -- tools/lab/fused_bwd_v1.hip -- staged half a weight block at a time and waited for
// every one of its six DMAs per tile with the chain idle: 1.57 ms per launch at 1.19 M rows, the parts of the tile time ADDED UP).
// One PERSISTENT 8-wave workgroup per CU walks a contiguous range of 64-row tiles:
//   waves 0-3  "chain":  the data-gradient chain for 16 rows each.  The 3-way bf16 split of dz3 / dz2 that each product needs
//                        anyway is also written to LDS as the "G" operand of the weight gradients (eight consecutive rows of
//                        one feature = one bf16x8 vector).  They issue no weight DMA and wait for none: a barrier tells them
//                        that the piece they are about to read has landed.
//   waves 4-7  "wgrad":  keep dW3 and dW2 (2 x 128 x 128 fp32 = 128 accumulator registers per lane) for the whole row range;
//                        load the rows of the other operand (z2 / z1) one layer ahead, split them once, publish them as "A"
//                        operand vectors (32 rows at a time) and run dW += G^T A on v_mfma_f32_16x16x32_bf16 (six products,
//                        contraction over rows).  They also run the WEIGHT RING: the packed transposed weights stream through
//                        three 24 KB LDS slots (one contraction block of 32 features each: [split][output block][lane][8]) in
//                        the fixed order W3^T, W2^T, W1e^T, W3^T, ... -- 12 pieces per tile, piece p + 2 issued (LDS-DMA, 6 per
//                        wave) at the barrier that opens piece p, i.e. two product phases ahead of its use, and retired by a
//                        COUNTED s_waitcnt vmcnt(N) in front of the barrier that opens piece p itself.  The DMA is issued from
//                        inline assembly: the compiler neither tracks it nor drains it at its own waits.
// Twelve phases per tile, one barrier each (as many as the first version had), none of them behind an exposed DMA.  Chain: phase
// 4 (3 - l) + c multiplies contraction block c of layer l = 3, 2, 1 (ring piece = phase number); phase 0 opens with the LayerNorm
// backward and the split of dz3 into G, phases 4 and 8 with the ReLU mask and the split of dz2 into G / the store of dz1; phases
// 8-10 also issue the next tile's row loads, a third each.  Weight-gradient waves: see the schedule above wg_fetches() -- blocks
// in phases 0, 3, 5, 7, operand publishes in 1, 4, 6, 11, the DMA of ring piece P + 2 in EVERY phase P.
// So the matrix pipe runs the chain's products and the weight gradients side by side and the HBM rows of both roles are in flight
// half a layer (wgrad) or one tile (chain) ahead of their use.
// LDS: 72 KB ring + 48 KB G image (64 rows) + 24 KB A vectors (32 rows) + 4.5 KB LayerNorm partials / weights = 148.5 KB.
// Per-workgroup partial results go to slabs that the existing fixed-order reductions add (deterministic, no float atomics).
#include <cstdlib>
#include <type_traits>
#include "hgn_device.h"
#include "hgn_host.h"
#include "mlp_common.h"
#include "mlp6_device.h"
#include "fused_args.h"

// Diagnostic build only (-DHGN_FUSED_STAMPS, tools/fusedstamps.py): shader-clock stamps of one mid-launch workgroup's waves 0 (chain)
// and 4 (weight gradients) at every phase boundary of its 11th tile.  In the shipped library FSTAMP() is empty.  Read them with
// care: a stamp is an s_memtime plus a full lgkmcnt drain (~300-1000 cycles under load) -- the stamped tile takes 56 k cycles,
// an unstamped one 46 k; per-phase ORDER of magnitude only.  What each part costs is measured by the HGN_FEXP ablations.
#ifdef HGN_FUSED_STAMPS
namespace hgn { __device__ unsigned long long g_fstamps[3 * 64]; }
#define FSTAMP(role, idx)                                                                                      \
  do {                                                                                                         \
    if (blockIdx.x == 37 && (threadIdx.x & 63) == 0 && tile == t_beg + 10) g_fstamps[(role) * 64 + (idx)] = clock64(); \
  } while (0)
#else
#define FSTAMP(role, idx) do {} while (0)
#endif

#ifndef HGN_FEXP
#define HGN_FEXP 0      // diagnostic builds only (compile-time ablations for timing; the results are then WRONG): 1 no weight DMA, 2 no operand
                        // fetch, 4 no dz1 / de stores, 8 no chain row loads, 32 no weight-gradient blocks, 64 no chain products, 128 no publish,
                        // 16 / 256 chain row loads / stores in a whole-row pattern (same bytes, an eighth of the cache-line requests)
#endif

namespace hgn {

constexpr int FT = 512;                               // threads: 8 waves
constexpr int PIECE_BYTES = 3 * 8 * 1024;             // one contraction block of a packed block: [split][output block][lane][8 bf16]
constexpr int RING_BYTES = 3 * PIECE_BYTES;           // 72 KB
constexpr int G_BYTES = 3 * 64 * 256;                 // [split][row 0..63][feature] bf16, row-major and swizzled: 48 KB
constexpr int A_BYTES = 3 * 4 * 128 * 16;             // [split][row group 0..3][feature] bf16x8: 32 rows, 24 KB
constexpr int G_OFF = RING_BYTES, A_OFF = G_OFF + G_BYTES, LN_OFF = A_OFF + A_BYTES, LNG_OFF = LN_OFF + 4 * 256 * 4;
constexpr int FUSED_LDS = LNG_OFF + 128 * 4;
static_assert(FUSED_LDS <= 160 * 1024, "one workgroup per CU");
// (FusedArgs, FSLAB: csrc/fused_args.h)

__device__ __forceinline__ void bar_lds() {           // every wave's LDS traffic issued so far is complete; global traffic stays in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
// ... and all but this wave's KEEP most recently issued vector-memory operations have completed (they retire in order)
template <int KEEP>
__device__ __forceinline__ void bar_keep() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP) : "memory");
  __builtin_amdgcn_s_barrier();
}

template <int KEEP>
__device__ __forceinline__ void wait_keep() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP) : "memory");
}

__device__ __forceinline__ unsigned opaque(unsigned v) { return opaque_u(v); }

// (glds_piece<NP>: csrc/mlp6_device.h -- the LDS-DMA of one ring piece by one of the four waves that share it, one inline-assembly
// statement outside the compiler's s_waitcnt bookkeeping; completion: a counted vmcnt of the issuing wave, then a barrier)

__device__ __forceinline__ void split3v8(const float (&v)[8], bf16x8 (&s)[3]) {
  hgn_split::eight(v, s);
}

// ---- the G image (dz3 / dz2 as weight-gradient operand): ROW-MAJOR bf16, [split][row 0..63][feature 0..127], 256-byte rows whose
// sixteen 16-byte chunks are XOR-swizzled with the row:  off(row, ch) = 256 row + 16 (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))).
// The chain lane (row, feature quarter kq) holds, per 16-feature block, four consecutive features = 8 bytes of one row: 24
// ds_write_b64 per layer (2-way bank conflicts).  The weight-gradient waves need the TRANSPOSE (eight consecutive rows of one
// feature per lane): gfx950's ds_read_b64_tr_b16 delivers a 4-row x 16-column block column-major, conflict-free on this image --
// two of them make one bf16x8 operand.  (The first layout stored operand vectors and was written two bytes at a time: 288
// ds_write_b16 per layer and lane, 2.3 k cycles of the chain's issue time, and an LDS queue full of conflicting writes behind
// which the first weight-gradient block of every tile ran at a seventh of its speed.)
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int G_SPLIT_BYTES = 64 * 256;               // one split of the image: 16 KB

// gw[c] / gw[4 + c]: the lane's byte addresses (inside smem) of the chunks that hold features 32 c + 4 kq .. + 3 and 32 c + 16 + 4 kq .. + 3
__device__ __forceinline__ void g_write_addrs(unsigned (&gw)[8], int wave, int n, int kq) {
  const unsigned a = n & 3, b = (n >> 2) & 3, u = (unsigned)(kq >> 1) ^ b;
  const unsigned rowb = (unsigned)(G_OFF + (16 * wave + n) * 256 + 8 * (kq & 1));
#pragma unroll
  for (unsigned c = 0; c < 4; ++c) {
    gw[c] = rowb + 64u * (c ^ a) + 16u * u;            // chunk 4 c + (kq >> 1), swizzled
    gw[4 + c] = rowb + 64u * (c ^ a) + 16u * (u ^ 2u); // chunk 4 c + 2 + (kq >> 1)
  }
}
template <int NP>
__device__ __forceinline__ void write_gops(unsigned char* __restrict__ smem, const unsigned (&gw)[8], const bf16x8 (&xs)[3][4]) {
#pragma unroll
  for (int s = 0; s < (NP == 1 ? 1 : 3); ++s)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      *reinterpret_cast<bf16x4*>(smem + gw[c] + s * G_SPLIT_BYTES) = __builtin_shufflevector(xs[s][c], xs[s][c], 0, 1, 2, 3);
      *reinterpret_cast<bf16x4*>(smem + gw[4 + c] + s * G_SPLIT_BYTES) = __builtin_shufflevector(xs[s][c], xs[s][c], 4, 5, 6, 7);
    }
}
// Weight-gradient side.  Lane l = 16 kg + 4 q + p of wave ww reads, for its 16-feature block mb and row half hh, the 8 bytes
// of row 8 kg + 4 hh + q (of the 32-row block), columns 32 ww + 16 mb + 4 p .. + 3; the transposed read hands lane 16 kg + i
// feature 32 ww + 16 mb + i of those four rows.  gr[2 mb + hh]: the lane's byte address for row block 0, split 0.
__device__ __forceinline__ void g_read_addrs(unsigned (&gr)[4], unsigned ww, int lane) {
  const unsigned kg = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
  for (unsigned mb = 0; mb < 2; ++mb)
#pragma unroll
    for (unsigned hh = 0; hh < 2; ++hh) {
      const unsigned r = 8 * kg + 4 * hh + q;
      const unsigned ch = 4 * ww + 2 * mb + (p >> 1);
      gr[2 * mb + hh] = (unsigned)G_OFF + 256u * r + 16u * (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))) + 8u * (p & 1);
    }
}
__device__ __forceinline__ bf16x8 g_read(const unsigned char* __restrict__ smem, unsigned lo, unsigned hi, int off) {
  const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + lo + off));
  const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + hi + off));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 z = __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, z);
}

// acc[ob] += (piece of the packed block: contraction block C) * x, six products per output block, in the order of mfma_half6_sb
// of the first version and of csrc/mlp6_device.h (smallest terms first; per accumulator c = 0..3 in sequence): the same bits.
// The fragments of output block ob + 1 are read while block ob multiplies (scheduling barriers keep the reads where they are
// written: at this kernel's register budget the unconstrained scheduler hoists dozens of fragment reads and then spills).
// START 0: acc += ...; 1: acc = 0 + ... (an inline constant as the first product's C operand: no zero fill); 2: acc = init + ... (the
// first product reads its C operand from `init`'s registers and writes acc's: no copy)
template <int C, int NP, int START = 0>
__device__ __forceinline__ void sweep_piece(Act& acc, const bf16x8 (&xs)[3][4], const unsigned char* __restrict__ lp /*slot + 16 lane*/,
                                            const Act* init = nullptr) {
  // two output blocks at a time (two independent accumulation chains), the SIX fragments of the next pair read before the
  // current pair's twelve products are issued: 192 matrix-pipe cycles between a ds_read_b128 and its use (one block ahead, 96
  // cycles, left an exposed LDS round trip of 100-300 cycles per block: the sweeps ran at half the matrix rate)
  if (HGN_FEXP & 64) return;
  constexpr int NSP = NP == 1 ? 1 : 3;
  bf16x8 fr[2][2][3];
  auto load_pair = [&](int g, bf16x8 (&f)[2][3]) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int s = 0; s < NSP; ++s) f[k][s] = *reinterpret_cast<const bf16x8*>(lp + (s * 8 + 2 * g + k) * 1024);
  };
  load_pair(0, fr[0]);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g + 1 < 4) load_pair(g + 1, fr[(g + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 (&a)[2][3] = fr[g & 1];               // a[k][0] hi, [1] mid, [2] lo of output block 2 g + k
    f32x4 t0, t1;
    if constexpr (START == 1) { t0 = f32x4{0.f, 0.f, 0.f, 0.f}; t1 = f32x4{0.f, 0.f, 0.f, 0.f}; }
    else if constexpr (START == 2) { t0 = init->v[2 * g]; t1 = init->v[2 * g + 1]; }
    else { t0 = acc.v[2 * g]; t1 = acc.v[2 * g + 1]; }
    if (NP == 1) {
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], xs[0][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], xs[0][C], t1, 0, 0, 0);
    } else {
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][2], xs[0][C], t0, 0, 0, 0);      // smallest terms first
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][2], xs[0][C], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], xs[2][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], xs[2][C], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][1], xs[1][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][1], xs[1][C], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][1], xs[0][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][1], xs[0][C], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], xs[1][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], xs[1][C], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], xs[0][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], xs[0][C], t1, 0, 0, 0);
    }
    acc.v[2 * g] = t0; acc.v[2 * g + 1] = t1;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dW_layer += G^T A over the 32 rows of block `blk` of the tile; wgrad wave ww owns dW rows [32 ww, 32 ww + 32).
// G: rows 32 blk .. 32 blk + 31 of the row-major image (transposed reads), A: the 32-row vector image.  The A vectors of feature block nb + 1 are read while
// block nb multiplies.
template <int NP>
__device__ __forceinline__ void wgrad_block(f32x4 (&acc)[2][8], f32x4 (&cs)[2], const unsigned char* __restrict__ smem, const unsigned (&gr)[4],
                                            const bf16x8* __restrict__ ap /*lane base: A image*/, int blk) {
  constexpr int NS = NP == 1 ? 1 : 3;
  bf16x8 gs[2][3];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int s = 0; s < NS; ++s) gs[mb][s] = g_read(smem, gr[2 * mb], gr[2 * mb + 1], s * G_SPLIT_BYTES + blk * 32 * 256);
  // bias gradient = column sums of G over the 32 rows: the same G^T fragments against an operand of ONES on the matrix pipe (the
  // three split terms add up to the fp32 value exactly; every column of the 16 x 16 result holds the sums) -- three products per
  // 16 features instead of 24 conversions + 24 additions per lane on the vector pipe, which this kernel is short of
  {
    bf16x8 ones;
#pragma unroll
    for (int p = 0; p < 8; ++p) ones[p] = (__bf16)1.0f;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x4 c = cs[mb];
      if (NP != 1) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][2], ones, c, 0, 0, 0);      // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], ones, c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], ones, c, 0, 0, 0);
      cs[mb] = c;
    }
  }
  bf16x8 as[2][3];
#pragma unroll
  for (int s = 0; s < NS; ++s) as[0][s] = ap[(s * 4) * 128];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    if (nb + 1 < 8) {
#pragma unroll
      for (int s = 0; s < NS; ++s) as[(nb + 1) & 1][s] = ap[(s * 4) * 128 + 16 * (nb + 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 (&a)[3] = as[nb & 1];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x4 c = acc[mb][nb];
      if (NP != 1) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][2], a[0], c, 0, 0, 0);      // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], a[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], a[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], a[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], a[1], c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], a[0], c, 0, 0, 0);
      acc[mb][nb] = c;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// (t_load32 / t_store32: hgn_device.h -- a uniform base pointer + a 32-bit per-lane byte offset; eligibility bounds these arrays to 4 GiB)

#if HGN_FEXP & (16 | 256)
// timing experiments only (WRONG rows): the same bytes with every instruction covering two whole rows (8 cache lines) instead of
// 64 B of each of 16 rows (64 quarter-line touches)
__device__ __forceinline__ void t_load32c(Act& a, const float* __restrict__ base, unsigned byte_off) {
  const char* p = reinterpret_cast<const char*>(base);
  HGN_FOR_B(fb) a.v[fb] = *reinterpret_cast<const f32x4*>(p + (byte_off + 1024u * fb));
}
__device__ __forceinline__ void t_store32c(const Act& a, float* __restrict__ base, unsigned byte_off) {
  char* p = reinterpret_cast<char*>(base);
  HGN_FOR_B(fb) *reinterpret_cast<f32x4*>(p + (byte_off + 1024u * fb)) = a.v[fb];
}
#endif

// ---- the weight ring (wgrad waves) ----------------------------------------------------------------------------------------------
// Piece q of a tile (q = 0..11): layer q / 4 (W3^T, W2^T, W1e^T), contraction block q % 4, ring slot q % 3 (12 = 0 mod 3: the slot
// of a piece does not depend on the tile).  DPW = DMA instructions per wgrad wave and piece.
template <int NP> struct Ring { static constexpr int DPW = NP == 1 ? 2 : 6; };

template <int Q, int NP>
__device__ __forceinline__ void dma_piece(const __bf16* __restrict__ pk3, const __bf16* __restrict__ pk2, const __bf16* __restrict__ pk1,
                                          unsigned lds_base, unsigned ww, unsigned voff /*dma_lane_off*/) {
  constexpr int layer = Q / 4, c = Q % 4, half = c >> 1, cl = c & 1, slot = Q % 3;
  const __bf16* blk = layer == 0 ? pk3 : (layer == 1 ? pk2 : pk1);
  // tile (split 0, output block ww) of contraction block c: uniform -> a scalar register pair
  const __bf16* src = blk + (half * HALF_BF16 + (cl * 8) * TILE_BF16) + ww * TILE_BF16;
  if (!(HGN_FEXP & 1)) glds_piece<NP>(src, voff, lds_base + slot * PIECE_BYTES + ww * 1024);
}

// Schedule of the weight-gradient waves over the 12 phases of a tile T (all four waves run the SAME instruction stream: wave w
// fetches and publishes row group w of whichever 32-row block is due; one code path instead of one per row block is what
// compiles without scratch, and 40 KB of instructions instead of 63-74 KB):
//   phase 0  dW3 += G3[rows 0-31]^T  A          (A = z2 rows 0-31 of T, published in phase 11 of T - 1)
//         1  publish z2 rows 32-63 (buffer xb);  fetch xb <- z1 rows 32-63 of T
//         3  dW3 += G3[rows 32-63]^T A
//         4  publish z1 rows 0-31  (buffer xa);  fetch xa <- z2 rows 0-31 of T + 1
//         5  dW2 += G2[rows 0-31]^T  A
//         6  publish z1 rows 32-63 (xb);         fetch xb <- z2 rows 32-63 of T + 1
//         7  dW2 += G2[rows 32-63]^T A
//        11  publish z2 rows 0-31 of T + 1 (xa); fetch xa <- z1 rows 0-31 of T + 1
// and in EVERY phase P the DMA of ring piece P + 2, between the publish and the fetch.  The blocks sit in the phases in which the
// chain has the most work of its own (first piece of a layer + G rows, last piece + ReLU mask + split), the publishes in the rest.
constexpr bool wg_fetches(int p) { const int q = ((p % 12) + 12) % 12; return q == 1 || q == 4 || q == 6 || q == 11; }
// vector-memory operations a wgrad wave issues AFTER the DMA of piece P (issued in phase P - 2): the fetch (8 loads) of phase P - 2,
// the DMA of piece P + 1 (phase P - 1) and the fetch of phase P - 1
template <int P, int NP>
struct Keep { static constexpr int value = Ring<NP>::DPW + 8 * (wg_fetches(P - 2) ? 1 : 0) + 8 * (wg_fetches(P - 1) ? 1 : 0); };

// ---- the weight-gradient waves (4-7) of edge_bwd_fused_kernel -----------------------------------------------------------------
template <int NP>
__device__ __forceinline__ void wgrad_role(const FusedArgs& fa, unsigned char* __restrict__ smem, long t_beg, long t_end) {
  const hgn_mlp_bwd_t& a = fa.b;
  const long M = a.M;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned ww = (unsigned)__builtin_amdgcn_readfirstlane((tid >> 6) - 4);
  const int m = lane & 15, kg = lane >> 4;
  unsigned gr[4];
  g_read_addrs(gr, ww, lane);
  const bf16x8* ap = reinterpret_cast<const bf16x8*>(smem + opaque((unsigned)(A_OFF + (kg * 128 + m) * 16)));
  // producer role: wave ww loads row group ww (8 rows) of a 32-row block, lane l features 2 l, 2 l + 1 (a wave instruction = one whole
  // 512-byte row), and publishes 2 x 3 operand vectors
  bf16x8* apub = reinterpret_cast<bf16x8*>(smem + opaque((unsigned)(A_OFF + (ww * 128 + 2 * lane) * 16)));
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;      // LDS byte address of the ring
  const __bf16* pk3 = reinterpret_cast<const __bf16*>(a.W3pk_t);
  const __bf16* pk2 = reinterpret_cast<const __bf16*>(a.W2pk_t);
  const __bf16* pk1 = reinterpret_cast<const __bf16*>(a.dx[0].Wpk_t);
  f32x4 acc[2][2][8];
  f32x4 cs[2][2];
#pragma unroll
  for (int l = 0; l < 2; ++l)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      cs[l][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) acc[l][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // A operand (z2 for layer 3, z1 for layer 2): rows are loaded more than a layer ahead into one of two buffers (xa: row block 0,
  // xb: row block 1), split once and published.  Rows past the end are clamped (their G rows are zero).
  f32x2 xa[8], xb[8];
  auto fetch = [&](f32x2 (&x)[8], int l, long tile, int blk) {
    const long r0 = tile * TILE_ROWS + blk * 32 + (long)ww * 8;
    const char* A = reinterpret_cast<const char*>(fa.A[l]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned r = (unsigned)min(r0 + j, M - 1);
      x[j] = *reinterpret_cast<const f32x2*>(A + (r * (LAT * 4u) + 8u * (unsigned)lane));
    }
  };
  auto publish = [&](f32x2 (&x)[8]) {
    if (HGN_FEXP & 128) return;
    // (the rows were fetched phases ago; without this fence the compiler converts them right behind the loads -- where the wait
    // for them stalls a weight-gradient block and, through the barrier, the whole workgroup)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = x[j][f];
      bf16x8 sp[3];
      split3v8(v, sp);
#pragma unroll
      for (int s2 = 0; s2 < (NP == 1 ? 1 : 3); ++s2) apub[s2 * 4 * 128 + f] = sp[s2];
    }
  };
  // One phase: [wait: piece P has landed] barrier; publish; DMA of piece P + 2; fetch; weight-gradient block.  The order inside a
  // phase is what Keep<> counts.
  auto phase = [&](auto P_, long tile) {
    constexpr int P = decltype(P_)::value;
    FSTAMP(1, 4 * P);
    bar_keep<Keep<P, NP>::value>();
    FSTAMP(1, 4 * P + 1);
    if constexpr (P == 1 || P == 6) publish(xb);
    if constexpr (P == 4 || P == 11) publish(xa);
    FSTAMP(1, 4 * P + 2);
    dma_piece<(P + 2) % 12, NP>(pk3, pk2, pk1, lds_base, ww, dma_lane_off((unsigned)lane));
    if constexpr (P == 1 && !(HGN_FEXP & 2)) fetch(xb, 1, tile, 1);
    if constexpr (P == 4 && !(HGN_FEXP & 2)) fetch(xa, 0, tile + 1, 0);     // (past the last tile: clamped rows -- keeps the operation counts static)
    if constexpr (P == 6 && !(HGN_FEXP & 2)) fetch(xb, 0, tile + 1, 1);
    if constexpr (P == 11 && !(HGN_FEXP & 2)) fetch(xa, 1, tile + 1, 0);
    FSTAMP(1, 4 * P + 3);
    if constexpr (P == 0 && !(HGN_FEXP & 32)) wgrad_block<NP>(acc[0], cs[0], smem, gr, ap, 0);
    if constexpr (P == 3 && !(HGN_FEXP & 32)) wgrad_block<NP>(acc[0], cs[0], smem, gr, ap, 1);
    if constexpr (P == 5 && !(HGN_FEXP & 32)) wgrad_block<NP>(acc[1], cs[1], smem, gr, ap, 0);
    if constexpr (P == 7 && !(HGN_FEXP & 32)) wgrad_block<NP>(acc[1], cs[1], smem, gr, ap, 1);
  };
  // prologue = phases 10 and 11 of the tile before the first one: both buffers' z2 rows, piece 0, publish z2 rows 0-31, piece 1,
  // fetch z1 rows 0-31 (so that the first barrier's count is the steady-state one)
  fetch(xa, 0, t_beg, 0);
  fetch(xb, 0, t_beg, 1);
  dma_piece<0, NP>(pk3, pk2, pk1, lds_base, ww, dma_lane_off((unsigned)lane));
  publish(xa);
  dma_piece<1, NP>(pk3, pk2, pk1, lds_base, ww, dma_lane_off((unsigned)lane));
  fetch(xa, 1, t_beg, 0);
  bar_lds();                                          // (S)
  for (long tile = t_beg; tile < t_end; ++tile) {
    phase(std::integral_constant<int, 0>{}, tile);
    phase(std::integral_constant<int, 1>{}, tile);
    phase(std::integral_constant<int, 2>{}, tile);
    phase(std::integral_constant<int, 3>{}, tile);
    phase(std::integral_constant<int, 4>{}, tile);
    phase(std::integral_constant<int, 5>{}, tile);
    phase(std::integral_constant<int, 6>{}, tile);
    phase(std::integral_constant<int, 7>{}, tile);
    phase(std::integral_constant<int, 8>{}, tile);
    phase(std::integral_constant<int, 9>{}, tile);
    phase(std::integral_constant<int, 10>{}, tile);
    phase(std::integral_constant<int, 11>{}, tile);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the two pieces issued past the end have landed before the workgroup's LDS is released
  bar_lds();                                          // (E)
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    float* slab = fa.slabs + ((long)blockIdx.x * 2 + l) * FSLAB;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(32 * ww + 16 * mb + 4 * kg + r) * 128 + 16 * nb + m] = acc[l][mb][nb][r];
      if (m == 0) {                                   // every column of the ones-product holds the sums: column 0 stores them
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[128 * 128 + 32 * ww + 16 * mb + 4 * kg + r] = cs[l][mb][r];
      }
    }
  }
}

template <int NP>
__global__ __launch_bounds__(FT, 2) void edge_bwd_fused_kernel(const FusedArgs fa) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[FUSED_LDS];
  float* lnl = reinterpret_cast<float*>(smem + LN_OFF);
  float* lng = reinterpret_cast<float*>(smem + LNG_OFF);
  const hgn_mlp_bwd_t& a = fa.b;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the role split is a scalar branch
  const long M = a.M;
  // this workgroup's tiles: workgroups b, b + 8, ... share an XCD (round-robin dispatch; speed only): XCD-major order
  const long G = gridDim.x, bx = blockIdx.x;
  const long q8 = G >> 3, r8 = G & 7, xc = bx & 7, ix = bx >> 3;
  const long pos = (xc < r8 ? xc * (q8 + 1) : r8 * (q8 + 1) + (xc - r8) * q8) + ix;
  const long t_beg = pos * fa.tiles / G, t_end = (pos + 1) * fa.tiles / G;
  if (tid < 128) lng[tid] = a.ln_g[tid];              // LayerNorm weights: read from LDS per tile (visible after the first barrier)

  if (wave < 4) {
    // ================================= data-gradient chain =================================
    const hgn_dx_t d = a.dx[0];
    const bool has_dout = a.d_out != nullptr, has_agg = a.agg_dout != nullptr;
    // Four row tiles of registers and when each is free (a load can only land where nothing lives):
    //   xh   : x-hat of the tile.  Consumed by the LayerNorm backward at the tile's start, so the NEXT tile's rows are fetched right
    //          behind it -- eleven phases (~ 14 us) ahead of their use.
    //   dout : d(e') rows of the tile, added into `geff` at the tile's start: the next tile's rows are fetched in phase 1.
    //   geff : the receiver's d(agg) row on arrival, then d_out_eff = d(e') + d(agg) -- alive until phase 8, where it starts the last
    //          layer's accumulators (the skip connection); the next tile's d(agg) rows (cache resident) are gathered in phase 9.
    //   g    : the gradient on its way down the chain (LayerNorm backward -> dz3 -> dz2 -> dz1 -> de), products accumulating in place.
    // The first version of this schedule fetched x-hat in phase 8, d(e') in 9 and the rest in 10: the youngest of them was two phases
    // (~ 3 us) old when the next tile needed it, and the ablation without these loads ran 0.27 ms faster.
    Act g, xh, dout, geff;
    unsigned pf_m1 = 0, pf_m2 = 0;
    float pf_rstd = 0.f;
    int seg_next = 0;
    bf16x8 xs[3][4];
    int n = lane & 15, kq = lane >> 4;
    float lnacc_g[2] = {0.f, 0.f}, lnacc_b[2] = {0.f, 0.f};      // LayerNorm-affine gradient partials of the lane's 2 + 2 features, over all tiles
    const unsigned ld_dout4 = (unsigned)a.ld_dout * 4u;
    auto row_of = [&](long tile, int n_) -> unsigned {
      const long row = tile * TILE_ROWS + wave * WAVE_ROWS + n_;
      return (unsigned)(row < M ? row : M - 1);
    };
    auto fetch_xhat = [&](long tile, int n_, int kq_) {
      if (!(HGN_FEXP & 8)) t_load32(xh, a.xhat, row_of(tile, n_) * (LAT * 4u) + 16u * kq_);
    };
    auto fetch_dout = [&](long tile, int n_, int kq_) {
      if (has_dout && !(HGN_FEXP & 8)) t_load32(dout, a.d_out, row_of(tile, n_) * ld_dout4 + 16u * kq_);
    };
    auto fetch_small = [&](long tile, int n_, int kq_) {         // sign words, 1 / sigma
      const long row = tile * TILE_ROWS + wave * WAVE_ROWS + n_;
      const unsigned rc = row_of(tile, n_);
      const unsigned* bits = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(a.relu_bits) + (rc * 32u + 4u * kq_));
      pf_m1 = bits[0];
      pf_m2 = bits[4];
      pf_rstd = row < M ? a.rstd[rc] : 0.f;           // rows past the end contribute nothing to any gradient
    };
    auto fetch_agg = [&](int seg, int kq_) {          // `sum` aggregation backward: the receiver's row (cache-resident gather; 64-bit offset)
      if (has_agg && !(HGN_FEXP & 8)) {
        const char* ar = reinterpret_cast<const char*>(a.agg_dout) + ((long)seg * a.ld_agg * 4 + 16 * kq_);
        HGN_FOR_B(fb) geff.v[fb] = *reinterpret_cast<const f32x4*>(ar + 64 * fb);
      }
    };
    auto seg_of = [&](long tile) -> int {
      const long row = tile * TILE_ROWS + wave * WAVE_ROWS + n;
      return has_agg ? a.agg_seg[row < M ? row : M - 1] : 0;
    };
    if (!has_dout) t_zero(dout);
    if (!has_agg) t_zero(geff);
    fetch_xhat(t_beg, n, kq);
    fetch_dout(t_beg, n, kq);
    fetch_agg(seg_of(t_beg), kq);
    fetch_small(t_beg, n, kq);
    bar_lds();                                        // (S) LayerNorm weights in LDS; pairs with the wgrad waves' first barrier
    for (long tile = t_beg; tile < t_end; ++tile) {
      // everything per-lane is re-derived from an opaque lane id inside the loop: otherwise the compiler hoists two dozen loop-
      // invariant addresses out of the loop and spills them
      const int lane_i = (int)opaque((unsigned)lane);
      n = lane_i & 15; kq = lane_i >> 4;
      const long row = tile * TILE_ROWS + wave * WAVE_ROWS + n;
      const bool valid = row < M;
      FSTAMP(0, 0);
      seg_next = seg_of(tile + 1);                    // consumed by the gather in phase 9
      const unsigned mb1 = pf_m1, mb2 = pf_m2;
      // ---- LayerNorm backward -> dz3 (g) ------------------------------------------------------------------------------------
      {
        HGN_FOR_B(fb) geff.v[fb] += dout.v[fb];       // d_out_eff (either part may be zeros); `dout` is free from here
        // LayerNorm-affine gradient partials of this wave's 16 rows: column sums of d_out_eff * x-hat and of d_out_eff by the
        // transposing butterfly (hgn_device.h: 90 instructions per array; the row16_sum form was ~640 for the two, a third of this
        // role's vector instructions), accumulated in registers over the tiles -- no LDS, no exec-masked stores.
        if (tile + 1 == t_end && (M & (TILE_ROWS - 1)) != 0)      // rows past the end (the launch's last tile only; uniform)
          HGN_FOR_B(fb) geff.v[fb] = valid ? geff.v[fb] : f32x4{0.f, 0.f, 0.f, 0.f};
        {
          HGN_FOR_B(fb) g.v[fb] = geff.v[fb] * xh.v[fb];
          float sg[2], sb[2];
          row16_sums_transposed(g, sg);
          row16_sums_transposed(geff, sb);
          lnacc_g[0] += sg[0]; lnacc_g[1] += sg[1];
          lnacc_b[0] += sb[0]; lnacc_b[1] += sb[1];
        }
        HGN_FOR_B(fb) g.v[fb] = geff.v[fb] * *reinterpret_cast<const f32x4*>(lng + 16 * fb + 4 * kq);
        const float m1 = row_sum(g) * (1.f / LAT);
        const float m2 = row_dot(g, xh) * (1.f / LAT);
        const float r = pf_rstd;
        HGN_FOR_B(fb) g.v[fb] = r * (g.v[fb] - m1 - xh.v[fb] * m2);
      }
      fetch_xhat(tile + 1, n, kq);                    // x-hat of the NEXT tile: its registers are free for the rest of this one
      FSTAMP(0, 1);
      // ---- layers 3 and 2: ONE copy of the code (a run-time loop: fully unrolled the kernel is 63 KB of instructions; the
      // counters show no instruction-cache misses either way, the loop is kept for build time and register pressure).
      // Entering layer `li`: g = the gradient to multiply (dz3, dz2).  Once it is split into operand vectors its registers are free:
      // the products accumulate INTO g (the first piece from an inline 0) -- no second tile, no zero fill, no copy per layer.
      // li = 0: g <- W3^T dz3; 1: g <- W2^T dz2.  Ring slot of piece (li, c): (4 li + c) mod 3.
#pragma unroll 1
      for (int li = 0; li < 2; ++li) {
        split3(g, xs);
        unsigned gw[8];
        g_write_addrs(gw, wave, (int)opaque((unsigned)n), kq);   // (8 address registers, alive for the two write sites of this layer only)
        if (wave < 2) write_gops<NP>(smem, gw, xs);             // rows 0-31 of G: free since the previous layer's first weight-gradient block
        const unsigned char* ring = smem + opaque((unsigned)(lane_i * 16 + (li ? 1 : 0) * PIECE_BYTES));
        FSTAMP(0, 2 + 8 * li);
        bar_lds();                                              // ---- phase 4 li
        FSTAMP(0, 3 + 8 * li);
        if (wave >= 2) write_gops<NP>(smem, gw, xs);            // rows 32-63: free now (the previous layer's second block is done)
        sweep_piece<0, NP, 1>(g, xs, ring);
        // (one array per phase: one burst of all of them holds up every other memory instruction of the CU -- the ring's DMA among
        // them -- for thousands of cycles)
        if (li == 0) fetch_dout(tile + 1, (int)opaque((unsigned)n), kq);       // phase 1: d(e') of the next tile
        ring = smem + opaque((unsigned)(lane_i * 16 + (li ? 2 : 1) * PIECE_BYTES));
        FSTAMP(0, 4 + 8 * li); bar_lds(); FSTAMP(0, 5 + 8 * li);
        sweep_piece<1, NP>(g, xs, ring);                        // ---- phase 4 li + 1
        ring = smem + opaque((unsigned)(lane_i * 16 + (li ? 0 : 2) * PIECE_BYTES));
        FSTAMP(0, 6 + 8 * li); bar_lds(); FSTAMP(0, 7 + 8 * li);
        sweep_piece<2, NP>(g, xs, ring);                        // ---- phase 4 li + 2
        ring = smem + opaque((unsigned)(lane_i * 16 + (li ? 1 : 0) * PIECE_BYTES));
        FSTAMP(0, 8 + 8 * li); bar_lds(); FSTAMP(0, 9 + 8 * li);
        sweep_piece<3, NP>(g, xs, ring);                        // ---- phase 4 li + 3
        relu_mask_bits(g, li == 0 ? mb2 : mb1);                 // dz2 / dz1
        if (li == 0) fetch_small(tile + 1, (int)opaque((unsigned)n), kq);      // (mb1 / mb2 hold this tile's words; pf_* the next tile's)
      }
      // ---- layer 1: de = d_out_eff + dz1 W1e, accumulated in g from `geff` (the first product's C operand) -----------------
      // (dz1: whole 64-row tiles are stored; rows past M land in the padding the caller provides)
#if HGN_FEXP & 256
      t_store32c(g, a.dz1, ((unsigned)(row - n) + (lane_i >> 5)) * (LAT * 4u) + 16u * (lane_i & 31));
#else
      if (!(HGN_FEXP & 4)) t_store32(g, a.dz1, (unsigned)row * (LAT * 4u) + 16u * kq);
#endif
      split3(g, xs);
      const unsigned char* ring = smem + opaque((unsigned)(lane_i * 16));
      FSTAMP(0, 18); bar_lds(); FSTAMP(0, 19); sweep_piece<0, NP, 2>(g, xs, ring + 2 * PIECE_BYTES, &geff);                // ---- phase 8
      FSTAMP(0, 20); bar_lds(); FSTAMP(0, 21);
      if (has_agg) fetch_agg(seg_next, kq); else t_zero(geff);      // phase 9: `geff` has started the accumulators; the next tile's d(agg) rows
      sweep_piece<1, NP>(g, xs, ring + 0 * PIECE_BYTES);                // ---- phase 9
      FSTAMP(0, 22); bar_lds(); FSTAMP(0, 23); sweep_piece<2, NP>(g, xs, ring + 1 * PIECE_BYTES);                // ---- phase 10
      FSTAMP(0, 24); bar_lds(); FSTAMP(0, 25); sweep_piece<3, NP>(g, xs, ring + 2 * PIECE_BYTES);                // ---- phase 11
      FSTAMP(0, 26);
#if HGN_FEXP & 256
      t_store32c(g, d.dx, ((unsigned)(row - n) + (lane_i >> 5)) * ((unsigned)d.ld * 4u) + 16u * (lane_i & 31));
#else
      if (valid && !(HGN_FEXP & 4)) t_store32(g, d.dx, (unsigned)row * ((unsigned)d.ld * 4u) + 16u * kq);
#endif
    }
    {                                                 // lane (n, kq) holds features 16 (n >> 1) + 4 kq + 2 (n & 1) + {0, 1} (row16_sums_transposed)
      const int f0 = 16 * (n >> 1) + 4 * kq + 2 * (n & 1);
      lnl[wave * 256 + f0] = lnacc_g[0]; lnl[wave * 256 + f0 + 1] = lnacc_g[1];
      lnl[wave * 256 + 128 + f0] = lnacc_b[0]; lnl[wave * 256 + 128 + f0 + 1] = lnacc_b[1];
    }
    bar_lds();                                        // (E) every chain wave's LayerNorm partials are in LDS
    const float sum = (lnl[tid] + lnl[256 + tid]) + (lnl[512 + tid] + lnl[768 + tid]);
    a.ln_ws[(long)blockIdx.x * 256 + tid] = sum;
    if (blockIdx.x == 0 && tid == 0) { reinterpret_cast<unsigned*>(a.ln_ws)[-256] = 0u; reinterpret_cast<unsigned*>(a.ln_ws)[-255] = gridDim.x; }               // ticket of ln_reduce_kernel (csrc/mlp.hip)
  } else {
    // ================================= weight gradients of layers 3 and 2, and the weight ring =================================
    wgrad_role<NP>(fa, smem, t_beg, t_end);
  }
}

}  // namespace hgn

using namespace hgn;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static long fused_grid(int64_t M) {
  static const long cap = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
    }
    return (long)cus;                                 // one 8-wave workgroup per CU (148.5 KB of LDS each)
  }();
  const long tiles = (M + TILE_ROWS - 1) / TILE_ROWS;
  return tiles < cap ? tiles : cap;
}

extern "C" int hgn_edge_bwd_fused_workspace_bytes(int64_t M, size_t* bytes) {
  if (!bytes || M < 0) return hgn_fail(HGN_E_INVALID, "hgn_edge_bwd_fused_workspace_bytes: bad argument");
  *bytes = (size_t)fused_grid(M) * 2 * FSLAB * sizeof(float) + 256;
  return HGN_OK;
}

extern "C" int hgn_edge_bwd_fused_eligible(const hgn_mlp_bwd_t* a) {
  if (!a || !hgn_mlp_bwd6_eligible(a)) return 0;          // (includes the HGN_F_FP32_MFMA flag of the call)
  if (a->n_dx != 1 || !a->dx[0].residual || a->dx[0].K != 128 || a->seg_dz1 || !a->dz1) return 0;
  if (a->agg_dout && (a->n_agg_ops != 1 || a->agg_ops[0] != HGN_OP_SUM)) return 0;      // several aggregates (pna): the two-launch path
  const int64_t ldmax = a->ld_dout > a->dx[0].ld ? a->ld_dout : a->dx[0].ld;
  // 32-bit row offsets inside the kernel for the edge-row arrays (d(e'), x-hat, z1 / z2, dz1, de); the d(agg) gather is 64-bit
  if ((a->M + TILE_ROWS) * (ldmax > 128 ? ldmax : 128) * 4 >= ((int64_t)1 << 32)) return 0;
  return 1;
}

static int edge_bwd_fused_impl(const hgn_mlp_bwd_t* a, const hgn_wfuse_t* w, void* workspace, size_t ws_bytes, hgn_wred_task_t* red_out,
                               void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!a || !w) return hgn_fail(HGN_E_INVALID, "hgn_edge_bwd_fused: null args");
  if (a->M == 0) return HGN_OK;
  size_t need = 0;
  if (a->M < 0 || hgn_edge_bwd_fused_workspace_bytes(a->M, &need) != HGN_OK || !workspace || ws_bytes < need ||
      !aligned16(workspace))
    return hgn_fail(HGN_E_INVALID, "hgn_edge_bwd_fused: workspace missing or too small");
  if (!hgn_edge_bwd_fused_eligible(a)) return hgn_fail(HGN_E_INVALID, "hgn_edge_bwd_fused: arguments not eligible (see hgn_edge_bwd_fused_eligible)");
  if ((!a->d_out && !a->agg_dout) || !a->d_gamma || !a->d_beta)
    return hgn_fail(HGN_E_INVALID, "hgn_edge_bwd_fused: gradient inputs / LayerNorm outputs missing");
  if (a->agg_dout) {
    if (a->n_agg_ops < 1 || a->n_agg_ops > 4 || !a->agg_seg || !a->agg_rowptr || (a->ld_agg & 3) || !aligned16(a->agg_dout))
      return hgn_fail(HGN_E_INVALID, "hgn_edge_bwd_fused: bad aggregation-backward descriptor");
    for (int i = 0; i < a->n_agg_ops; ++i) {
      if (a->agg_ops[i] < 0 || a->agg_ops[i] > 3) return hgn_fail(HGN_E_INVALID, "Invalid operation type!");
      if ((a->agg_ops[i] == HGN_OP_MAX && !a->agg_argmax) || (a->agg_ops[i] == HGN_OP_MIN && !a->agg_argmin))
        return hgn_fail(HGN_E_INVALID, "hgn_edge_bwd_fused: max/min need the saved arg index");
    }
  }
  if (!w->z2 || !w->z1 || !aligned16(w->z2) || !aligned16(w->z1) || !w->dW3 || !w->dW2 || !a->dz1)
    return hgn_fail(HGN_E_INVALID, "hgn_edge_bwd_fused: bad weight-gradient operands (dz1 must be written: dW1 is the caller's launch)");
  const long G = fused_grid(a->M);
  FusedArgs fa;
  fa.b = *a;
  fa.b.ln_ws += 256;                                  // the kernel's slab 0 lies behind the header slab (ticket of ln_reduce_kernel)
  fa.A[0] = w->z2;
  fa.A[1] = w->z1;
  fa.slabs = (float*)workspace;
  fa.tiles = (a->M + TILE_ROWS - 1) / TILE_ROWS;
  ProfScope ps(14, (double)a->M, stream);
  if (bwd_products(a->products) == 3) { if (launch_edge_bwd_fused3(fa, G, stream) != HGN_OK) return HGN_E_LAUNCH; }
  else if (bwd_products(a->products) == 1) hipLaunchKernelGGL((edge_bwd_fused_kernel<1>), dim3((unsigned)G), dim3(FT), 0, stream, fa);
  else hipLaunchKernelGGL((edge_bwd_fused_kernel<6>), dim3((unsigned)G), dim3(FT), 0, stream, fa);
  if (hgn_check_launch("hgn_edge_bwd_fused") != HGN_OK) return HGN_E_LAUNCH;
  // fixed-order sums of the per-workgroup partials: two weight gradients + biases, and the LayerNorm-affine gradients
  SlabReduceTask rt[2];
  float* dW[2] = {w->dW3, w->dW2};
  float* db[2] = {w->db3, w->db2};
  for (int l = 0; l < 2; ++l) {
    rt[l].type = 0; rt[l].K = 128; rt[l].n_out = 128; rt[l].acc = w->accumulate ? 1 : 0; rt[l].n_chunks = (int)G;
    rt[l].dW = dW[l]; rt[l].ldw = 128; rt[l].db = db[l]; rt[l].slab = fa.slabs + (long)l * FSLAB;
    rt[l].chunk_stride = 2L * FSLAB;
  }
  if (red_out) {
    for (int l = 0; l < 2; ++l)
      red_out[l] = {rt[l].type, rt[l].K, rt[l].n_out, rt[l].acc, rt[l].n_chunks, 0, rt[l].dW, (int64_t)rt[l].ldw, rt[l].db, rt[l].slab,
                    (int64_t)rt[l].chunk_stride};
  } else if (launch_slab_reduce(rt, 2, stream) != HGN_OK) return HGN_E_LAUNCH;
  // LayerNorm partial slabs: ln_ws holds hgn_mlp_bwd_ln_workspace_bytes(M) bytes = (tiles + parts) slabs; G <= tiles
  if (!(a->flags & HGN_F_DEFER_LN) &&
      launch_ln_reduce(fa.b.ln_ws, G, fa.b.ln_ws + G * 256, a->d_gamma, a->d_beta, a->ln_accumulate, stream) != HGN_OK) return HGN_E_LAUNCH;
  return hgn_check_launch("hgn_edge_bwd_fused (reductions)");
}
extern "C" int hgn_edge_bwd_fused(const hgn_mlp_bwd_t* a, const hgn_wfuse_t* w, void* workspace, size_t ws_bytes, void* stream) {
  return edge_bwd_fused_impl(a, w, workspace, ws_bytes, nullptr, stream);
}
extern "C" int hgn_edge_bwd_fused_partial(const hgn_mlp_bwd_t* a, const hgn_wfuse_t* w, void* workspace, size_t ws_bytes,
                                          hgn_wred_task_t* red, void* stream) {
  if (!red) return hgn_fail(HGN_E_INVALID, "hgn_edge_bwd_fused_partial: null descriptor array");
  return edge_bwd_fused_impl(a, w, workspace, ws_bytes, red, stream);
}

#ifdef HGN_FUSED_STAMPS
extern "C" int hgn_debug_fused_stamps(unsigned long long* host192) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host192, HIP_SYMBOL(hgn::g_fstamps), 3 * 64 * 8) == hipSuccess ? HGN_OK : HGN_E_LAUNCH;
}
#endif
