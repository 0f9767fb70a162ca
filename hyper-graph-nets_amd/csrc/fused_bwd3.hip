// Edge-block backward with the weight gradients in the same pass -- product mode 3 (two scaled fp16 terms, three MFMAs per
// product; csrc/mlp6_device.h: Prod<3>).  Same contract, phases, roles and weight ring as csrc/fused_bwd.hip (read its header first:
// one persistent 8-wave workgroup per CU, waves 0-3 the data-gradient chain, waves 4-7 the weight gradients and the ring; twelve
// phases per 64-row tile, one barrier each, counted waits).  What differs is how the two roles share dz3 / dz2:
//
//   * The chain scales every gradient row by its own power of two before it splits it (fp16 has 5 exponent bits; the rows of a
//     gradient tensor differ by orders of magnitude) -- exact for the chain's products, whose columns ARE the rows, but useless as
//     the weight-gradient operand, which contracts over rows and needs ONE scale per 32-row block.  So the chain hands over the
//     fp32 rows themselves (8 ds_write_b128 per layer instead of 24 ds_write_b64: the chain is the role short of issue slots) and
//     the weight-gradient waves split what they read: lane (m, kg) of wave ww takes G[rows 8 kg .. 8 kg + 7][feature 32 ww + 16 mb + m]
//     -- exactly its MFMA A fragment -- as 16 ds_read_b32 from an XOR-swizzled row-major image (conflict-free both ways).
//   * The other operand (z2 / z1 rows) is published per 8-row group with the group's own scale 2^eA[grp] (the wave that fetched
//     the group knows its largest magnitude; no agreement between waves is needed at publish time).
//   * Per block each weight-gradient wave picks T = min over the four row groups of (sG[grp] + eA[grp]), sG from the largest
//     magnitude of ITS 32 x 32 piece of G, scales group grp of G by 2^(T - eA[grp]) (never above 2^15 by construction), multiplies
//     the block from ZERO accumulators on the matrix pipe and adds it to the fp32 accumulators with the factor 2^-T on the vector
//     pipe.  No running scale, no overflow, nothing to agree on between waves or tiles.
//   * Bias gradients: column sums of the fp32 G values the wave holds anyway (16 adds per block).
//   * The rows of z2 / z1 travel HBM -> LDS by LDS-DMA (two 16 KB landing buffers) instead of through 32 registers per lane that
//     were live for the whole tile: the weight-gradient role now holds 128 fp32 accumulators AND the operands of a block.
//   * The published operand image is double-buffered (2 x 16 KB): the last weight-gradient block of a tile (dW2, rows 32-63) then
//     needs nothing that is overwritten before the NEXT tile's first barrier and runs in phase 11 -- beside the chain's LayerNorm
//     backward of the next tile, the longest stretch in which the weight-gradient waves used to wait (phase stamps: 4.6 k of a
//     tile's 34 k cycles) -- instead of in phase 7, where the chain waited 3 k cycles for it.
// LDS: ring 3 x 16 KB + G 32 KB (fp32, 64 rows) + A 2 x 16 KB + 4.5 KB LayerNorm partials + 2 x 16 KB landing buffers = 148.6 KB.
#include <cstdlib>
#include <type_traits>
#include "hgn_device.h"
#include "hgn_host.h"
#include "mlp_common.h"
#include "mlp6_device.h"
#include "fused_args.h"

#ifndef HGN_FEXP
#define HGN_FEXP 0      // diagnostic builds only (see csrc/fused_bwd.hip)
#endif
// Diagnostic build only (-DHGN_FUSED_STAMPS, tools/fusedstamps.py): shader-clock stamps of one mid-launch workgroup's waves 0 (chain) and 4
// (weight gradients) at the phase boundaries of its 11th tile.  A stamp is an s_memtime plus an lgkmcnt drain: order of magnitude only.
#ifdef HGN_FUSED_STAMPS
namespace hgn { __device__ unsigned long long g_fstamps[3 * 64]; __device__ unsigned long long g_wstamps[8 * 32]; }
#define FSTAMP(role, idx)                                                                                      \
  do {                                                                                                         \
    if (blockIdx.x == 37 && (threadIdx.x & 63) == 0 && tile == t_beg + 10) g_fstamps[(role) * 64 + (idx)] = clock64(); \
  } while (0)
// WSTAMP(p): every wave's arrival at the barrier of phase p (12 x 8 entries behind the FSTAMP block); WSTAMP2: wgrad waves, after their
// counted vmcnt wait and before the barrier itself
#define WSTAMP(p)                                                                                              \
  do {                                                                                                         \
    if (blockIdx.x == 37 && (threadIdx.x & 63) == 0 && tile == t_beg + 10) g_wstamps[(threadIdx.x >> 6) * 32 + (p)] = clock64(); \
  } while (0)
#else
#define FSTAMP(role, idx) do {} while (0)
#define WSTAMP(p) do {} while (0)
#endif

namespace hgn {
namespace f3 {

constexpr int NP = 3;
constexpr int FT = 512;                               // threads: 8 waves
constexpr int PIECE_BYTES = 2 * 8 * 1024;             // one contraction block of a packed block: [split][output block][lane][8 fp16]
constexpr int RING_BYTES = 3 * PIECE_BYTES;           // 48 KB
constexpr int G_BYTES = 64 * 512;                     // fp32 rows of dz3 / dz2, 16-byte chunks XOR-swizzled with the row: 32 KB
constexpr int A_BYTES = 2 * 4 * 128 * 16;             // [split][row group 0..3][feature] fp16x8: 32 rows, 16 KB -- TWO of them (A_OFF + buf * A_BYTES)
constexpr int X_BYTES = 4 * 8 * 512;                  // raw fp32 rows of the other operand on their way in: 4 row groups x 8 rows, 16 KB per buffer
constexpr int G_OFF = RING_BYTES, A_OFF = G_OFF + G_BYTES, EA_OFF = A_OFF + 2 * A_BYTES, LN_OFF = EA_OFF + 32, LNG_OFF = LN_OFF + 4 * 256 * 4;
constexpr int XA_OFF = LNG_OFF + 128 * 4, XB_OFF = XA_OFF + X_BYTES;
constexpr int FUSED_LDS = XB_OFF + X_BYTES;
static_assert(FUSED_LDS <= 160 * 1024, "one workgroup per CU");

__device__ __forceinline__ void bar_lds() {           // every wave's LDS traffic issued so far is complete; global traffic stays in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
template <int KEEP>
__device__ __forceinline__ void bar_keep() {          // ... and all but this wave's KEEP youngest vector-memory operations have completed
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP) : "memory");
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ unsigned opaque(unsigned v) { return opaque_u(v); }

// ---- the G image: fp32, row-major, 512-byte rows; chunk c (16 bytes) of row r lives at chunk c ^ sw(r), sw(r) = (r & 7) ^ (((r >> 3) & 1) << 2).
// Chain lane (row n of its wave, feature quarter kq) writes chunk 4 fb + kq of its row per 16-feature block fb: the eight lanes
// n = 0..7 of a write group hit eight different chunk slots.  Weight-gradient lane (m, kg) reads word m & 3 of chunk
// 8 ww + 4 mb + (m >> 2) of row 8 kg + j: for a fixed j the row groups kg = 0..3 alternate between the two halves of the 32 banks.
__device__ __forceinline__ unsigned g_swz(unsigned r) { return (r & 7u) ^ (((r >> 3) & 1u) << 2); }

__device__ __forceinline__ void g_write(unsigned char* __restrict__ smem, int wave, int n, int kq, const Act& g) {
  const unsigned r = 16u * wave + n;
  const unsigned base = (unsigned)G_OFF + 512u * r;
  const unsigned sw = g_swz(r);
  HGN_FOR_B(fb) *reinterpret_cast<f32x4*>(smem + base + 16u * ((4u * fb + kq) ^ sw)) = g.v[fb];
}

// acc[ob] += (piece of the packed block: contraction block C) * x, three products per output block, smallest terms first, two output
// blocks at a time with the fragments of the next pair read before the current pair's products are issued (see fused_bwd.hip).
// START 0: acc += ...; 1: acc = 0 + ...; 2: acc = init + ...
template <int C, int START = 0>
__device__ __forceinline__ void sweep_piece(Act& acc, const bf16x8 (&xs)[3][4], const unsigned char* __restrict__ lp /*slot + 16 lane*/,
                                            const Act* init = nullptr) {
  if (HGN_FEXP & 64) return;
  bf16x8 fr[2][2][2];
  auto load_pair = [&](int g, bf16x8 (&f)[2][2]) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int s = 0; s < 2; ++s) f[k][s] = *reinterpret_cast<const bf16x8*>(lp + (s * 8 + 2 * g + k) * 1024);
  };
  load_pair(0, fr[0]);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g + 1 < 4) load_pair(g + 1, fr[(g + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 (&a)[2][2] = fr[g & 1];               // a[k][0] hi, [1] lo of output block 2 g + k
    f32x4 t0, t1;
    if constexpr (START == 1) { t0 = f32x4{0.f, 0.f, 0.f, 0.f}; t1 = f32x4{0.f, 0.f, 0.f, 0.f}; }
    else if constexpr (START == 2) { t0 = init->v[2 * g]; t1 = init->v[2 * g + 1]; }
    else { t0 = acc.v[2 * g]; t1 = acc.v[2 * g + 1]; }
    t0 = mfma_f16(a[0][1], xs[0][C], t0);
    t1 = mfma_f16(a[1][1], xs[0][C], t1);
    t0 = mfma_f16(a[0][0], xs[1][C], t0);
    t1 = mfma_f16(a[1][0], xs[1][C], t1);
    t0 = mfma_f16(a[0][0], xs[0][C], t0);
    t1 = mfma_f16(a[1][0], xs[0][C], t1);
    acc.v[2 * g] = t0; acc.v[2 * g + 1] = t1;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dW_layer += G^T A over the 32 rows of block `blk` of the tile (see the header).  acc: TRUE scale, fp32.
__device__ __forceinline__ void wgrad_block(f32x4 (&acc)[2][8], float (&bs)[2], const unsigned char* __restrict__ smem, unsigned gaddr /*opaque (ww, kg, m)*/,
                                            const bf16x8* __restrict__ ap0 /*lane base: A image 0*/, const int* __restrict__ ea_lds0, int kg, int blk, int abuf) {
  const bf16x8* ap = ap0 + abuf * (A_BYTES / 16);       // (abuf, blk: compile-time constants at every call site)
  const int* ea_lds = ea_lds0 + 4 * abuf;
  // ---- this lane's 2 x 8 values of G: rows 32 blk + 8 kg + j, features 32 ww + 16 mb + m.  Swizzled chunk of (j, mb):
  // ((m >> 2) ^ (j & 3)) + 4 (mb ^ (j >> 2) ^ (kg & 1)): eight lane bases (j & 3, parity), everything else an immediate offset.
  // One 16-feature half (mb) at a time -- read, column sums, scale, split -- so that only its eight values are live beside the accumulators.
  // The block's scale, per half: T = min over the four row groups of (sG + eA), sG of this lane's row group over the half's 16
  // features, eA of the group from its publisher; group kg of G enters at 2^(T - eA[kg]) <= 2^sG.
  bf16x8 gs[2][3];
  float unscale[2], unscale2[2];
  const int ea = ea_lds[kg];
  unsigned ga = gaddr;
  asm volatile("" : "+v"(ga));                         // (derived here, per block: hoisted out of the tile loop the eight lane bases are eight spills)
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const unsigned m = (ga >> 2) & 15u;                // (packed by the caller: bits 2..5 = m, 6..7 = kg, 8.. = ww)
    const unsigned kgu = (ga >> 6) & 3u, wwu = ga >> 8;
    const unsigned rowb = (unsigned)G_OFF + 4096u * kgu + 128u * wwu + 4u * (m & 3u);
    const unsigned q = m >> 2, par = kgu & 1u;
    float gv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned lane_b = rowb + 16u * ((q ^ (unsigned)(j & 3)) + 4u * ((unsigned)(mb ^ (j >> 2)) ^ par));
      gv[j] = *reinterpret_cast<const float*>(smem + lane_b + 512u * (32u * blk + j));
    }
    float sum = 0.f, mx = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sum += gv[j]; mx = fmaxf(mx, fabsf(gv[j])); }
    bs[mb] += sum;                                     // bias gradient: column sums (fp32, true scale)
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x128, 0xf, 0xf, true)));   // row_ror:8 (over m: the 16 lanes of a row group)
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x124, 0xf, 0xf, true)));   // row_ror:4
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x122, 0xf, 0xf, true)));   // row_ror:2
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x121, 0xf, 0xf, true)));   // row_ror:1
    int T = hgn_split::scale_exp_shared(mx) + ea;
    T = rows4_min_i(T);
    hgn_split::eight16(gv, hgn_split::pow2f(T - ea), gs[mb]);   // (T - ea <= sG: no overflow; a group far below the block's largest goes to zero)
    // 2^-T in two factors (|T| <= 2 SCALE_CLAMP: 2^-T itself need not be a normal number).  T is the same in every lane: scalar registers.
    const int Tu = __builtin_amdgcn_readfirstlane(-T);
    const int T1 = Tu < -hgn_split::SCALE_CLAMP ? -hgn_split::SCALE_CLAMP : (Tu > hgn_split::SCALE_CLAMP ? hgn_split::SCALE_CLAMP : Tu);
    unscale[mb] = hgn_split::pow2f(T1);
    unscale2[mb] = hgn_split::pow2f(Tu - T1);
#if !(HGN_FEXP & 512)
    asm volatile("" : "+v"(ga) : "v"(gs[mb][0]), "v"(gs[mb][1]));      // the second half's reads start when the first half's values are dead (8 registers)
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  // ---- products from zero, then into the accumulators
  bf16x8 as[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) as[0][s] = ap[(s * 4) * 128];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    if (nb + 1 < 8) {
#pragma unroll
      for (int s = 0; s < 2; ++s) as[(nb + 1) & 1][s] = ap[(s * 4) * 128 + 16 * (nb + 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 (&a)[2] = as[nb & 1];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
      c = mfma_f16(gs[mb][1], a[0], c);                // smallest terms first
      c = mfma_f16(gs[mb][0], a[1], c);
      c = mfma_f16(gs[mb][0], a[0], c);
      acc[mb][nb] += (c * unscale[mb]) * unscale2[mb];
      // (the sum is "used" here: left alone, the compiler parks the sixteen products of a block in scratch and adds all four blocks
      // of a tile at the loop's end)
      asm volatile("" : "+v"(acc[mb][nb]));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the weight ring (wgrad waves): piece q of a tile (q = 0..11): layer q / 4 (W3^T, W2^T, W1e^T), contraction block q % 4, slot q % 3

template <int Q>
__device__ __forceinline__ void dma_piece(const __bf16* __restrict__ pk3, const __bf16* __restrict__ pk2, const __bf16* __restrict__ pk1,
                                          unsigned lds_base, unsigned ww, unsigned voff /*dma_lane_off2*/) {
  constexpr int layer = Q / 4, c = Q % 4, half = c >> 1, cl = c & 1, slot = Q % 3;
  const __bf16* blk = layer == 0 ? pk3 : (layer == 1 ? pk2 : pk1);
  const __bf16* src = blk + (half * HALF_BF16 + (cl * 8) * TILE_BF16) + ww * TILE_BF16;
  if (!(HGN_FEXP & 1)) glds_piece2(src, voff, lds_base + slot * PIECE_BYTES + ww * 1024);      // ww in {0, 1}: tiles ww, ww + 2, ww + 4, ww + 6 of both splits
}

// Schedule of the weight-gradient waves: blocks in phases 0, 3, 7, 11 -- the phases in which the chain has the most work of its own
// (sweep + G rows; sweep + ReLU mask + split; the same + dz1 store; last sweep + the next tile's LayerNorm backward) --, publishes in
// 1, 4, 6, 11, and in EVERY phase P the DMA of ring piece P + 2.
// Vector memory retires in order: a wave that issues both the ring's DMA (L2 hits, needed two phases later) and the operand rows'
// DMA (HBM, needed five phases later) waits for the rows whenever it waits for a piece (phase stamps: 1.7 k cycles at two barriers
// of a 34 k tile).  So the four waves split the two duties: waves 4, 5 ("ring") issue the whole piece (8 instructions each) and
// retire it with vmcnt(8) -- only the next piece may fly --; waves 6, 7 ("rows") issue the operand fetches (two row groups = 8
// instructions each) and wait, with vmcnt(8) as well, only in front of the barrier of the phase that publishes a buffer: the fetch
// that filled it is five to seven phases old, the one younger fetch may fly.
constexpr bool wg_fetches(int p) { const int q = ((p % 12) + 12) % 12; return q == 1 || q == 4 || q == 6 || q == 11; }
constexpr int WAVE_DMA = 8;                           // LDS-DMA instructions of one ring piece per ring wave = of one fetch per rows wave

// RING: this wave's memory duty (the arithmetic is the same for all four): the weight ring (waves 4, 5) or the operand rows (6, 7).  Two
// instantiations = two straight-line tile loops, each with its own static operation counts (check_fused_counts.py).
template <bool RING>
__device__ __forceinline__ void wgrad_role(const FusedArgs& fa, unsigned char* __restrict__ smem, long t_beg, long t_end) {
  const hgn_mlp_bwd_t& a = fa.b;
  const long M = a.M;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned ww = (unsigned)__builtin_amdgcn_readfirstlane((tid >> 6) - 4);
  const int m = lane & 15, kg = lane >> 4;
  // (ww, kg, m) in one opaque register: wgrad_block derives its G addresses from it inside each block (nothing address-like lives across phases)
  const unsigned gaddr = opaque(((unsigned)ww << 8) | ((unsigned)kg << 6) | ((unsigned)m << 2));
  const bf16x8* ap = reinterpret_cast<const bf16x8*>(smem + opaque((unsigned)(A_OFF + (kg * 128 + m) * 16)));
  const int* ea_lds = reinterpret_cast<const int*>(smem + EA_OFF);
  // producer role: wave ww loads row group ww (8 rows) of a 32-row block, lane l features 2 l, 2 l + 1, and publishes 2 x 2 operand vectors
  bf16x8* apub = reinterpret_cast<bf16x8*>(smem + opaque((unsigned)(A_OFF + (ww * 128 + 2 * lane) * 16)));
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const __bf16* pk3 = reinterpret_cast<const __bf16*>(a.W3pk_t);
  const __bf16* pk2 = reinterpret_cast<const __bf16*>(a.W2pk_t);
  const __bf16* pk1 = reinterpret_cast<const __bf16*>(a.dx[0].Wpk_t);
  f32x4 acc[2][2][8];
  float bs[2][2];
#pragma unroll
  for (int l = 0; l < 2; ++l)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      bs[l][mb] = 0.f;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) acc[l][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // The other operand's rows: a row group (8 rows x 512 B) of a 32-row block arrives by four LDS-DMA instructions (two whole rows
  // each; lane l: row 2 i + (l >> 5), bytes 16 (l & 31)) in its 4 KB of a landing buffer, five to seven phases before wave `grp`
  // publishes it.
  auto fetch = [&](int buf_off, int l, long tile, int blk) {      // rows waves only (ww in {2, 3}): row groups 2 (ww - 2), 2 (ww - 2) + 1 = 16 rows
    const long r0 = tile * TILE_ROWS + blk * 32 + (long)(ww - 2u) * 16 + (lane >> 5);
    const char* A = reinterpret_cast<const char*>(fa.A[l]);
    const unsigned dst = lds_base + (unsigned)buf_off + (ww - 2u) * 8192u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                     // (one address register at a time: eight live ones are four spills in the block phases)
      unsigned vo = (unsigned)min(r0 + 2 * i, M - 1) * (LAT * 4u) + 16u * (unsigned)(lane & 31);
      // (nt: these rows are read once, by this CU alone -- a streaming fill leaves the L2 to the weight ring and the gathered rows every CU
      // re-reads: 1.028 -> 0.993 ms on one box; the same hint on the chain's own row loads or on the dz1 / de stores gains nothing or loses)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(vo), "s"(A), "s"(dst + 1024u * i) : "memory", "m0");
    }
  };
  auto ring_piece = [&](auto Q_) {                    // ring waves only: operand tiles ww + 2 k of the piece
    constexpr int Q = decltype(Q_)::value;
    dma_piece<Q>(pk3, pk2, pk1, lds_base, ww, dma_lane_off2((unsigned)lane));
  };
  auto publish = [&](int buf_off, int abuf) {
    if (HGN_FEXP & 128) return;
    const f32x2* xp = reinterpret_cast<const f32x2*>(smem + opaque((unsigned)buf_off + ww * 4096u + 8u * (unsigned)lane));
    f32x2 x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = xp[64 * j];    // row j of the group, features 2 lane, 2 lane + 1
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fmaxf(fabsf(x[j][0]), fabsf(x[j][1])));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x128, 0xf, 0xf, true)));   // row_ror:8, 4, 2, 1: over the 16 lanes of a row
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x124, 0xf, 0xf, true)));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x122, 0xf, 0xf, true)));
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(mx), 0x121, 0xf, 0xf, true)));
    mx = rows4_max(mx);                                   // ... then over the four rows
    const int ea = hgn_split::scale_exp_shared(mx);
    const float sc = hgn_split::pow2f(ea);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = x[j][f];
      bf16x8 sp[3];
      hgn_split::eight16(v, sc, sp);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) apub[abuf * (A_BYTES / 16) + s2 * 4 * 128 + f] = sp[s2];
    }
    reinterpret_cast<int*>(smem + EA_OFF)[4 * abuf + ww] = ea;      // (every lane, the same word: a predicated store would put a branch into the counted phases)
  };
  auto phase = [&](auto P_, long tile) {
    constexpr int P = decltype(P_)::value;
    FSTAMP(1, 4 * P);
    WSTAMP(P);
    if constexpr (RING || wg_fetches(P)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAVE_DMA) : "memory");   // RING: piece P has landed (piece P + 1
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                              // may fly); rows: the buffer published below
    WSTAMP(12 + P);
    __builtin_amdgcn_s_barrier();
    FSTAMP(1, 4 * P + 1);
    if constexpr (P == 1 || P == 6) publish(XB_OFF, 1);      // rows 32-63 of z2 / z1 -> image 1
    if constexpr (P == 4 || P == 11) publish(XA_OFF, 0);     // rows 0-31 of z1 / the next tile's z2 -> image 0
    FSTAMP(1, 4 * P + 2);
    if constexpr (RING) ring_piece(std::integral_constant<int, (P + 2) % 12>{});
    else if constexpr (!(HGN_FEXP & 2)) {
      if constexpr (P == 1) fetch(XB_OFF, 1, tile, 1);
      if constexpr (P == 4) fetch(XA_OFF, 0, tile + 1, 0);   // (past the last tile: clamped rows -- keeps the operation counts static)
      if constexpr (P == 6) fetch(XB_OFF, 0, tile + 1, 1);
      if constexpr (P == 11) fetch(XA_OFF, 1, tile + 1, 0);
    }
    FSTAMP(1, 4 * P + 3);
    if constexpr (P == 0 && !(HGN_FEXP & 32)) wgrad_block(acc[0], bs[0], smem, gaddr, ap, ea_lds, kg, 0, 0);
    if constexpr (P == 3 && !(HGN_FEXP & 32)) wgrad_block(acc[0], bs[0], smem, gaddr, ap, ea_lds, kg, 1, 1);
    if constexpr (P == 7 && !(HGN_FEXP & 32)) wgrad_block(acc[1], bs[1], smem, gaddr, ap, ea_lds, kg, 0, 0);      // (G rows 0-31 hold dz2 from barrier 4 on, image 0 z1 from phase 4 to 11)
    // dW2 rows 32-63: G rows 32-63 (dz2) stay until the next tile's waves 2, 3 write behind barrier 0, image 1 until phase 1 of the next tile
    if constexpr (P == 11 && !(HGN_FEXP & 32)) wgrad_block(acc[1], bs[1], smem, gaddr, ap, ea_lds, kg, 1, 1);
  };
  // prologue = phases 10 and 11 of the tile before the first one: both landing buffers' z2 rows and piece 0, a barrier of the whole
  // workgroup (the groups a wave publishes were fetched by another), publish z2 rows 0-31, piece 1, fetch z1 rows 0-31
  if constexpr (RING) ring_piece(std::integral_constant<int, 0>{});
  else { fetch(XA_OFF, 0, t_beg, 0); fetch(XB_OFF, 0, t_beg, 1); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bar_lds();                                          // (P) pairs with the chain's extra barrier
  publish(XA_OFF, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... read before the next fetch overwrites the buffer (every wave reads its own group: a barrier)
  __builtin_amdgcn_s_barrier();                       // (Q)
  if constexpr (RING) ring_piece(std::integral_constant<int, 1>{});
  else fetch(XA_OFF, 1, t_beg, 0);
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
  const int lane_e = (int)opaque(threadIdx.x & 63u), m_e = lane_e & 15, kg_e = lane_e >> 4;      // (re-derived: nothing lane-like lives across the loop)
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    float* slab = fa.slabs + ((long)blockIdx.x * 2 + l) * FSLAB;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(32 * ww + 16 * mb + 4 * kg_e + r) * 128 + 16 * nb + m_e] = acc[l][mb][nb][r];
      float s = bs[l][mb];                            // the four row groups of a feature: fixed-order sum over kg
      s = rows4_sum(s);
      if (kg_e == 0) slab[128 * 128 + 32 * ww + 16 * mb + m_e] = s;
    }
  }
}

__global__ __launch_bounds__(FT, 2) void edge_bwd_fused3_kernel(const FusedArgs fa) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[FUSED_LDS];
  float* lnl = reinterpret_cast<float*>(smem + LN_OFF);
  float* lng = reinterpret_cast<float*>(smem + LNG_OFF);
  const hgn_mlp_bwd_t& a = fa.b;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long M = a.M;
  const long G = gridDim.x, bx = blockIdx.x;
  const long q8 = G >> 3, r8 = G & 7, xc = bx & 7, ix = bx >> 3;
  const long pos = (xc < r8 ? xc * (q8 + 1) : r8 * (q8 + 1) + (xc - r8) * q8) + ix;
  const long t_beg = pos * fa.tiles / G, t_end = (pos + 1) * fa.tiles / G;
  if (tid < 128) lng[tid] = a.ln_g[tid];

  if (wave < 4) {
    // ================================= data-gradient chain (see fused_bwd.hip for the register plan) =================================
    const hgn_dx_t d = a.dx[0];
    const bool has_dout = a.d_out != nullptr, has_agg = a.agg_dout != nullptr;
    // block scales of the three transposed packs (uniform)
    const int sw3 = pack_scale_exp(reinterpret_cast<const __bf16*>(a.W3pk_t));
    const int sw2 = pack_scale_exp(reinterpret_cast<const __bf16*>(a.W2pk_t));
    const int sw1 = pack_scale_exp(reinterpret_cast<const __bf16*>(d.Wpk_t));
    Act g, xh, dout, geff;
    unsigned pf_m1 = 0, pf_m2 = 0;
    float pf_rstd = 0.f;
    int seg_next = 0;
    bf16x8 xs[3][4];
    int n = lane & 15, kq = lane >> 4;
    float lnacc_g[2] = {0.f, 0.f}, lnacc_b[2] = {0.f, 0.f};
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
    auto fetch_small = [&](long tile, int n_, int kq_) {
      const long row = tile * TILE_ROWS + wave * WAVE_ROWS + n_;
      const unsigned rc = row_of(tile, n_);
      const unsigned* bits = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(a.relu_bits) + (rc * 32u + 4u * kq_));
      pf_m1 = bits[0];
      pf_m2 = bits[4];
      pf_rstd = row < M ? a.rstd[rc] : 0.f;
    };
    auto fetch_agg = [&](int seg, int kq_) {
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
    bar_lds();                                        // (P), (Q): the weight-gradient waves' prologue barriers
    bar_lds();
    bar_lds();                                        // (S)
    for (long tile = t_beg; tile < t_end; ++tile) {
      const int lane_i = (int)opaque((unsigned)lane);
      n = lane_i & 15; kq = lane_i >> 4;
      const long row = tile * TILE_ROWS + wave * WAVE_ROWS + n;
      const bool valid = row < M;
      FSTAMP(0, 0);
      seg_next = seg_of(tile + 1);
      const unsigned mb1 = pf_m1, mb2 = pf_m2;
      // ---- LayerNorm backward -> dz3 (g) ------------------------------------------------------------------------------------
      {
        HGN_FOR_B(fb) geff.v[fb] += dout.v[fb];
        if (tile + 1 == t_end && (M & (TILE_ROWS - 1)) != 0)
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
      fetch_xhat(tile + 1, n, kq);
      FSTAMP(0, 1);
      // ---- layers 3 and 2 (li = 0: g <- W3^T dz3; 1: g <- W2^T dz2).  Ring slot of piece (li, c): (4 li + c) mod 3.
      // G rows: waves 0, 1 (rows 0-31) write before the phase's barrier -- free since the previous layer's first weight-gradient
      // block --, waves 2, 3 (rows 32-63) after it (the previous layer's second block is done).
#pragma unroll 1
      for (int li = 0; li < 2; ++li) {
        if (wave < 2) g_write(smem, wave, (int)opaque((unsigned)n), kq, g);
        const int T = split_np<NP>(g, xs) + (li ? sw2 : sw3);      // the row's scale + the block's
        const unsigned char* ring = smem + opaque((unsigned)(lane_i * 16 + (li ? 1 : 0) * PIECE_BYTES));
        FSTAMP(0, 2 + 8 * li); WSTAMP(4 * li);
        bar_lds();                                              // ---- phase 4 li
        FSTAMP(0, 3 + 8 * li);
        if (wave >= 2) g_write(smem, wave, (int)opaque((unsigned)n), kq, g);
        sweep_piece<0, 1>(g, xs, ring);
        if (li == 0) fetch_dout(tile + 1, (int)opaque((unsigned)n), kq);       // phase 1: d(e') of the next tile
        ring = smem + opaque((unsigned)(lane_i * 16 + (li ? 2 : 1) * PIECE_BYTES));
        FSTAMP(0, 4 + 8 * li); WSTAMP(4 * li + 1); bar_lds(); FSTAMP(0, 5 + 8 * li);
        sweep_piece<1>(g, xs, ring);                            // ---- phase 4 li + 1
        ring = smem + opaque((unsigned)(lane_i * 16 + (li ? 0 : 2) * PIECE_BYTES));
        FSTAMP(0, 6 + 8 * li); WSTAMP(4 * li + 2); bar_lds(); FSTAMP(0, 7 + 8 * li);
        sweep_piece<2>(g, xs, ring);                            // ---- phase 4 li + 2
        ring = smem + opaque((unsigned)(lane_i * 16 + (li ? 1 : 0) * PIECE_BYTES));
        FSTAMP(0, 8 + 8 * li); WSTAMP(4 * li + 3); bar_lds(); FSTAMP(0, 9 + 8 * li);
        sweep_piece<3>(g, xs, ring);                            // ---- phase 4 li + 3
        scale_act(g, -T);                                       // back to the true scale
        relu_mask_bits(g, li == 0 ? mb2 : mb1);                 // dz2 / dz1
        if (li == 0) fetch_small(tile + 1, (int)opaque((unsigned)n), kq);
      }
      // ---- layer 1: de = d_out_eff + dz1 W1e, accumulated in g from `geff` ---------------------------------------------------
      if (!(HGN_FEXP & 4)) t_store32(g, a.dz1, (unsigned)row * (LAT * 4u) + 16u * kq);
      const int T1 = split_np<NP, false>(g, xs, &geff, sw1) + sw1;    // (capped by what the skip connection holds)
      scale_act(geff, T1);                                      // the skip connection enters at the products' scale
      const unsigned char* ring = smem + opaque((unsigned)(lane_i * 16));
      FSTAMP(0, 18); WSTAMP(8); bar_lds(); FSTAMP(0, 19); sweep_piece<0, 2>(g, xs, ring + 2 * PIECE_BYTES, &geff);                // ---- phase 8
      FSTAMP(0, 20); WSTAMP(9); bar_lds(); FSTAMP(0, 21);
      if (has_agg) fetch_agg(seg_next, kq); else t_zero(geff);      // phase 9
      sweep_piece<1>(g, xs, ring + 0 * PIECE_BYTES);                // ---- phase 9
      FSTAMP(0, 22); WSTAMP(10); bar_lds(); FSTAMP(0, 23); sweep_piece<2>(g, xs, ring + 1 * PIECE_BYTES);     // ---- phase 10
      FSTAMP(0, 24); WSTAMP(11); bar_lds(); FSTAMP(0, 25); sweep_piece<3>(g, xs, ring + 2 * PIECE_BYTES);     // ---- phase 11
      FSTAMP(0, 26);
      scale_act(g, -T1);
      if (valid && !(HGN_FEXP & 4)) t_store32(g, d.dx, (unsigned)row * ((unsigned)d.ld * 4u) + 16u * kq);
    }
    {
      const int f0 = 16 * (n >> 1) + 4 * kq + 2 * (n & 1);
      lnl[wave * 256 + f0] = lnacc_g[0]; lnl[wave * 256 + f0 + 1] = lnacc_g[1];
      lnl[wave * 256 + 128 + f0] = lnacc_b[0]; lnl[wave * 256 + 128 + f0 + 1] = lnacc_b[1];
    }
    bar_lds();                                        // (E)
    const float sum = (lnl[tid] + lnl[256 + tid]) + (lnl[512 + tid] + lnl[768 + tid]);
    a.ln_ws[(long)blockIdx.x * 256 + tid] = sum;
    if (blockIdx.x == 0 && tid == 0) { reinterpret_cast<unsigned*>(a.ln_ws)[-256] = 0u; reinterpret_cast<unsigned*>(a.ln_ws)[-255] = gridDim.x; }               // ticket of ln_reduce_kernel (csrc/mlp.hip)
  } else {
    if (wave < 6) wgrad_role<true>(fa, smem, t_beg, t_end);      // (uniform)
    else wgrad_role<false>(fa, smem, t_beg, t_end);
  }
}

}  // namespace f3

#ifdef HGN_FUSED_STAMPS
}  // namespace hgn
extern "C" int hgn_debug_fused_stamps(unsigned long long* host192) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host192, HIP_SYMBOL(hgn::g_fstamps), 3 * 64 * 8) == hipSuccess ? HGN_OK : HGN_E_LAUNCH;
}
extern "C" int hgn_debug_fused_wave_stamps(unsigned long long* host256) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(host256, HIP_SYMBOL(hgn::g_wstamps), 8 * 32 * 8) == hipSuccess ? HGN_OK : HGN_E_LAUNCH;
}
namespace hgn {
#endif

int launch_edge_bwd_fused3(const FusedArgs& fa, long grid, hipStream_t stream) {
  hipLaunchKernelGGL(f3::edge_bwd_fused3_kernel, dim3((unsigned)grid), dim3(f3::FT), 0, stream, fa);
  return hgn_check_launch("hgn_edge_bwd_fused (two-term fp16 products)");
}

}  // namespace hgn
