// Device building blocks of the split-bf16 kernels (csrc/mlp6.hip, csrc/fused_bwd.hip): operand-tile geometry of the packed
// weights (hgn_pack_bf16x3), the 3-way bf16 split, LDS-DMA staging of half a packed block and the six-product MFMA sweep.
#pragma once
#include "hgn_device.h"
#include "mlp_common.h"
#include "split_bf16.h"

// Diagnostic builds only (tools/build_ablations.sh): compile-time ablation mask of the split-bf16 edge kernels.
//   1 no weight DMA   2 no MFMA   4 no stores of saved activations / intermediate gradients   8 no row loads
//  32 (experiment, not an ablation) raised wave priority around the product sweeps
//  64 (experiment) operand fragments read one output block ahead of the products (mfma_half6_pipe)
// 256 (comparison) the compiler-scheduled product sweep (mfma_half6) instead of the pair-ahead one (mfma_half6_pipe2, the default)
//  16 time stamps (s_memrealtime, 10 ns) of one mid-launch workgroup's wave 0 through the forward kernel (tools/fwdstamps.py)
#ifndef HGN_ABL
#define HGN_ABL 0
#endif
#if HGN_ABL & 16
#ifndef HGN_STAMP_BLOCK
#define HGN_STAMP_BLOCK 5000
#endif
namespace hgn {
static __device__ unsigned long long g_hgn_stamps[256];      // (one copy per translation unit: only csrc/mlp6.hip reads its own)
static __device__ int g_hgn_stamp_n;
}
#define HGN_STAMP()                                                                      \
  do {                                                                                   \
    if (blockIdx.x == HGN_STAMP_BLOCK && threadIdx.x == 0 && g_hgn_stamp_n < 120) {      \
      g_hgn_stamps[128 + g_hgn_stamp_n] = clock64();      /* shader cycles: with the 100 MHz stamps, the in-kernel clock */ \
      g_hgn_stamps[g_hgn_stamp_n++] = wall_clock64();                                    \
    }                                                                                    \
  } while (0)
#else
#define HGN_STAMP() do {} while (0)
#endif

namespace hgn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TILE_BF16 = 512;                     // one 16 x 32 operand tile: 64 lanes x 8 bf16 = 1 KiB
constexpr int HALF_TILES = 3 * 2 * 8;              // splits x contraction blocks of the half x output blocks
constexpr int HALF_BF16 = HALF_TILES * TILE_BF16;  // 48 KB
constexpr int BLOCK_BF16 = 2 * HALF_BF16;          // one packed 128 x 128 block: 96 KB (= HGN_PACK_BLOCK_BYTES)

// ----------------------------------------------------------------------------------------------------------
// device building blocks
// ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3(const Act& x, bf16x8 (&s)[3][4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {                     // csrc/split_bf16.h: two values per instruction
    const f32x4 &q0 = x.v[2 * c], &q1 = x.v[2 * c + 1];
    const float v[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
    bf16x8 t[3];
    hgn_split::eight(v, t);
    s[0][c] = t[0]; s[1][c] = t[1]; s[2][c] = t[2];
  }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// Product modes (template parameter NP of every kernel here = hgn_mlp_fwd_t.products):
//   6  three bf16 terms per operand, six v_mfma_f32_16x16x32_bf16 per product (fp32 accurate; no range limits)
//   3  TWO fp16 terms per operand, three v_mfma_f32_16x16x32_f16 per product.  fp16 carries 11 significant bits, so hi + lo hold
//      22 bits + the sign of lo = the 24 bits of an fp32 value to within 2^-24 relative, and hi*hi + hi*lo + lo*hi leaves out only
//      the lo*lo term (2^-22 of the product at most, 2^-24 typically): the accuracy of the six-product bf16 form at half the matrix work, two
//      thirds of the weight traffic and operand registers.  What fp16 lacks is RANGE (5 exponent bits): every operand is
//      therefore scaled by a power of two first -- the weights per packed block at pack time (hgn_pack_t.transposed & 4: header
//      in the image), the rows of the other operand per row at split time (the largest |x| of the row goes to [2^14, 2^15)) --
//      and the accumulators are scaled back afterwards.  Scaling by a power of two is exact and commutes with every rounding
//      of the fp32 accumulation, so  unscale(scale(acc) + products of scaled operands)  has the bits of  acc + products.
//   1 / 2  one bf16 / one fp16 product (reduced precision, opt-in)
template <int NP> struct Prod {
  static constexpr int NSPLIT = NP == 6 ? 3 : (NP == 3 ? 2 : 1);
  static constexpr bool SCALED = NP == 3;
  static constexpr bool F16 = NP == 2 || NP == 3;
};
constexpr int PK_SCALE_BYTE = 2 * 16 * 1024;       // header of a scaled pack {int exponent sw: tiles hold W * 2^sw}: first word of the (unused) third split
using hgn_split::pow2f;
using hgn_split::scale_exp_of;
__device__ __forceinline__ int pack_scale_exp(const __bf16* __restrict__ pk) {      // uniform: one scalar load
  return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(pk) + PK_SCALE_BYTE);
}
// largest magnitude of the row a lane holds a quarter of (lanes n, n + 16, n + 32, n + 48 hold row n)
__device__ __forceinline__ float row_max_abs(const Act& x) {
  float m = 0.f;
  HGN_FOR_B(fb) {
    m = fmaxf(m, fmaxf(fabsf(x.v[fb][0]), fabsf(x.v[fb][1])));
    m = fmaxf(m, fmaxf(fabsf(x.v[fb][2]), fabsf(x.v[fb][3])));
  }
  return rows4_max(m);
}
// operand split of a product mode: 6 -> three bf16 terms; 3 -> two fp16 terms of the row scaled by 2^s (returns s); 1 -> the leading
// bf16 term; 2 -> fp16 (bit patterns in the bf16 slots)
// `acc` + `sw` (scaled mode): the accumulator the products of this row will be added to at scale 2^(s + sw) and the weight block's
// exponent -- s is capped so that what `acc` already holds survives the scaling (split_bf16.h: SCALE_EASY); without them the
// accumulator starts from zero or holds products of the same row.
template <int NP, bool BRANCH = true>
__device__ __forceinline__ int split_np(const Act& x, bf16x8 (&s)[3][4], const Act* acc = nullptr, int sw = 0) {
  if constexpr (NP == 2) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) s[0][c][j] = __builtin_bit_cast(__bf16, (_Float16)x.v[2 * c + (j >> 2)][j & 3]);
    return 0;
  } else if constexpr (NP == 3) {
    int e = scale_exp_of(row_max_abs(x));
    if constexpr (BRANCH) {
      if (acc != nullptr && __builtin_amdgcn_ballot_w64(e + sw > hgn_split::SCALE_EASY) != 0)
        e = min(e, hgn_split::acc_room(row_max_abs(*acc)) - sw);
    } else {                                     // (the fused backward's chain: a branch there costs registers, the twenty operations do not)
      const int room = hgn_split::acc_room(row_max_abs(*acc)) - sw;
      e = e + sw > hgn_split::SCALE_EASY ? min(e, room) : e;
    }
    const float sc = pow2f(e);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 &q0 = x.v[2 * c], &q1 = x.v[2 * c + 1];
      const float v[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
      bf16x8 t[3];
      hgn_split::eight16(v, sc, t);
      s[0][c] = t[0]; s[1][c] = t[1];
    }
    return e;
  } else {
    split3(x, s);
    return 0;
  }
}
// scaled modes: acc <- acc * 2^e (exact)
__device__ __forceinline__ void scale_act(Act& a, int e) {
  // e = a row's exponent + a block's: |e| <= 240.  One multiplication per value while every lane's factor is a normal number (always,
  // short of rows below 2^-100 of an ordinary weight block): a wave vote, so the common path has no second multiplication.
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(e < -hgn_split::SCALE_CLAMP || e > hgn_split::SCALE_CLAMP) == 0, 1)) {
    const float f = pow2f(e);
    HGN_FOR_B(fb) a.v[fb] *= f;
  } else {
    const int e1 = e < -hgn_split::SCALE_CLAMP ? -hgn_split::SCALE_CLAMP : (e > hgn_split::SCALE_CLAMP ? hgn_split::SCALE_CLAMP : e);
    const float f1 = pow2f(e1), f2 = pow2f(e - e1);
    HGN_FOR_B(fb) a.v[fb] = (a.v[fb] * f1) * f2;
  }
}
__device__ __forceinline__ void scale4(f32x4& v, int e) {      // the same for one register quadruple (column-split kernels)
  const int e1 = e < -hgn_split::SCALE_CLAMP ? -hgn_split::SCALE_CLAMP : (e > hgn_split::SCALE_CLAMP ? hgn_split::SCALE_CLAMP : e);
  v = (v * pow2f(e1)) * pow2f(e - e1);
}
// one product of a mode against the fragments a[0] (hi) a[1] (mid / lo) a[2] (lo) of one operand tile, smallest terms first
template <int NP>
__device__ __forceinline__ f32x4 mfma_np(const bf16x8 (&a)[3], const bf16x8& x0, const bf16x8& x1, const bf16x8& x2, f32x4 t) {
  if constexpr (NP == 6) {
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], x0, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], x2, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], x1, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], x0, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], x1, t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], x0, t, 0, 0, 0);
  } else if constexpr (NP == 3) {
    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[1]), __builtin_bit_cast(f16x8, x0), t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, x1), t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, x0), t, 0, 0, 0);
  } else if constexpr (NP == 2) {
    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[0]), __builtin_bit_cast(f16x8, x0), t, 0, 0, 0);
  } else {
    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], x0, t, 0, 0, 0);
  }
  return t;
}
// one reduced-precision product: bf16 (NP = 1) or fp16 (NP = 2) operands, fp32 accumulation
template <int NP>
__device__ __forceinline__ f32x4 mfma_one(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (NP == 2)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_f16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// NP = 1 (single bf16 product) / NP = 2 (single fp16 product: the leading third of the pack holds fp16 bit patterns, hgn_pack_bf16x3
// with transposed | 2): only the leading split of the weights is staged (the first third of the half's tiles)
template <int NP>
__device__ __forceinline__ void stage_half6(__bf16* __restrict__ lds, const __bf16* __restrict__ gsrc) {
  unsigned lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: scalar tile offsets, LDS bases straight into M0
#pragma unroll
  for (unsigned i = wave; i < (unsigned)(HALF_TILES / 3 * Prod<NP>::NSPLIT); i += WG / 64)          // one operand tile (1 KiB) per wave instruction
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + i * TILE_BF16 + lane * 8),
                                     (__attribute__((address_space(3))) void*)(lds + i * TILE_BF16), 16, 0, 0);
}

// Whole packed block (both halves: 96 KB, contiguous in memory and in LDS) by a workgroup of NWV waves
template <int NP, int NWV>
__device__ __forceinline__ void stage_block6(__bf16* __restrict__ lds, const __bf16* __restrict__ gsrc) {
  unsigned lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
  const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr unsigned PER_HALF = HALF_TILES / 3 * Prod<NP>::NSPLIT;
#pragma unroll
  for (unsigned h = 0; h < 2; ++h)
#pragma unroll
    for (unsigned i = wave; i < PER_HALF; i += NWV)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (h * HALF_TILES + i) * TILE_BF16 + lane * 8),
                                       (__attribute__((address_space(3))) void*)(lds + (h * HALF_TILES + i) * TILE_BF16), 16, 0, 0);
}

// A wave owns NS sub-tiles of 16 rows; every operand fragment read from LDS is multiplied with all of them.
template <int HALF, int NS, int NP>
__device__ __forceinline__ void mfma_half6(Act (&acc)[NS], const bf16x8 (&xs)[NS][3][4], const __bf16* __restrict__ lds) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int cl = 0; cl < 2; ++cl) {
    const int c = 2 * HALF + cl;
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
      const bf16x8 a_hi = *reinterpret_cast<const bf16x8*>(lds + ((0 * 2 + cl) * 8 + ob) * TILE_BF16 + lane * 8);
      if constexpr (NP == 1 || NP == 2) {
#pragma unroll
        for (int u = 0; u < NS; ++u) acc[u].v[ob] = mfma_one<NP>(a_hi, xs[u][0][c], acc[u].v[ob]);
        continue;
      }
      if constexpr (NP == 3) {
        const bf16x8 a3[3] = {a_hi, *reinterpret_cast<const bf16x8*>(lds + ((1 * 2 + cl) * 8 + ob) * TILE_BF16 + lane * 8), a_hi};
#pragma unroll
        for (int u = 0; u < NS; ++u) acc[u].v[ob] = mfma_np<3>(a3, xs[u][0][c], xs[u][1][c], xs[u][1][c], acc[u].v[ob]);
        continue;
      }
      const bf16x8 a_mi = *reinterpret_cast<const bf16x8*>(lds + ((1 * 2 + cl) * 8 + ob) * TILE_BF16 + lane * 8);
      const bf16x8 a_lo = *reinterpret_cast<const bf16x8*>(lds + ((2 * 2 + cl) * 8 + ob) * TILE_BF16 + lane * 8);
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        f32x4 t = acc[u].v[ob];
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, xs[u][0][c], t, 0, 0, 0);      // smallest terms first
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, xs[u][2][c], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_mi, xs[u][1][c], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_mi, xs[u][0][c], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, xs[u][1][c], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, xs[u][0][c], t, 0, 0, 0);
        acc[u].v[ob] = t;
      }
    }
  }
}

// The same sweep for one sub-tile with the fragments of output block i + 1 read while block i multiplies (scheduling barriers keep
// the reads where they are written).  Experiment (HGN_ABL & 64), see DESIGN.md section 5.8.
template <int HALF, int NP>
__device__ __forceinline__ void mfma_half6_pipe(Act& acc, const bf16x8 (&xs)[3][4], const __bf16* __restrict__ lds) {
  const int lane = threadIdx.x & 63;
  const __bf16* lp = lds + lane * 8;
  constexpr int NSP = Prod<NP>::NSPLIT;
  bf16x8 fr[2][3];
#pragma unroll
  for (int s = 0; s < NSP; ++s) fr[0][s] = *reinterpret_cast<const bf16x8*>(lp + ((s * 2 + 0) * 8 + 0) * TILE_BF16);
#pragma unroll
  for (int i = 0; i < 16; ++i) {                   // i = 8 * cl + ob
    const int cl = i >> 3, ob = i & 7, c = 2 * HALF + cl;
    if (i + 1 < 16) {
      const int cl1 = (i + 1) >> 3, ob1 = (i + 1) & 7;
#pragma unroll
      for (int s = 0; s < NSP; ++s) fr[(i + 1) & 1][s] = *reinterpret_cast<const bf16x8*>(lp + ((s * 2 + cl1) * 8 + ob1) * TILE_BF16);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 (&a)[3] = fr[i & 1];              // a[0] hi, a[1] mid, a[2] lo
    f32x4 t = acc.v[ob];
    if constexpr (NP != 6) {
      t = mfma_np<NP>(a, xs[0][c], xs[1][c], xs[2][c], t);
    } else {
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], xs[0][c], t, 0, 0, 0);      // smallest terms first
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[2][c], t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], xs[1][c], t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], xs[0][c], t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[1][c], t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[0][c], t, 0, 0, 0);
    }
    acc.v[ob] = t;
    __builtin_amdgcn_sched_barrier(0);
  }
}
// Two output blocks at a time (two independent accumulation chains, as the compiler pairs them anyway) with the SIX operand
// fragments of the next pair read from LDS before the current pair's twelve products are issued: a whole pair (192 matrix-pipe
// cycles) of distance between a `ds_read_b128` and its use, where the unconstrained schedule reads just in time and starts
// every pair with an exposed LDS round trip.  24 more registers during the sweep, which is not where the kernels peak.
template <int HALF, int NP>
__device__ __forceinline__ void mfma_half6_pipe2(Act& acc, const bf16x8 (&xs)[3][4], const __bf16* __restrict__ lds) {
  const int lane = threadIdx.x & 63;
  const __bf16* lp = lds + lane * 8;
  constexpr int NSP = Prod<NP>::NSPLIT;
  bf16x8 fr[2][2][3];
  auto load_pair = [&](int g, bf16x8 (&f)[2][3]) {
    const int cl = g >> 2, ob0 = 2 * (g & 3);
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int s = 0; s < NSP; ++s) {
        if (HGN_ABL & 1024) f[k][s] = xs[s][k];        // (timing experiment: no fragment reads)
        else f[k][s] = *reinterpret_cast<const bf16x8*>(lp + ((s * 2 + cl) * 8 + ob0 + k) * TILE_BF16);
      }
  };
  load_pair(0, fr[0]);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int cl = g >> 2, ob0 = 2 * (g & 3), c = 2 * HALF + cl;
    if (g + 1 < 8) load_pair(g + 1, fr[(g + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 (&a)[2][3] = fr[g & 1];           // a[k][0] hi, [1] mid, [2] lo of output block ob0 + k
    f32x4 t0 = acc.v[ob0], t1 = acc.v[ob0 + 1];
    if constexpr (NP == 1 || NP == 2) {
      t0 = mfma_one<NP>(a[0][0], xs[0][c], t0);
      t1 = mfma_one<NP>(a[1][0], xs[0][c], t1);
    } else if constexpr (NP == 3) {                   // smallest terms first, the two accumulation chains interleaved
      t0 = mfma_f16(a[0][1], xs[0][c], t0);
      t1 = mfma_f16(a[1][1], xs[0][c], t1);
      t0 = mfma_f16(a[0][0], xs[1][c], t0);
      t1 = mfma_f16(a[1][0], xs[1][c], t1);
      t0 = mfma_f16(a[0][0], xs[0][c], t0);
      t1 = mfma_f16(a[1][0], xs[0][c], t1);
    } else {
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][2], xs[0][c], t0, 0, 0, 0);      // smallest terms first
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][2], xs[0][c], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], xs[2][c], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], xs[2][c], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][1], xs[1][c], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][1], xs[1][c], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][1], xs[0][c], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][1], xs[0][c], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], xs[1][c], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], xs[1][c], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][0], xs[0][c], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][0], xs[0][c], t1, 0, 0, 0);
    }
    acc.v[ob0] = t0; acc.v[ob0 + 1] = t1;
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int HALF, int NS, int NP>
__device__ __forceinline__ void mfma_half6_sel(Act (&acc)[NS], const bf16x8 (&xs)[NS][3][4], const __bf16* __restrict__ lds) {
  // one sub-tile per wave: the pair-ahead sweep (forward -0.7 %, backward -1.6 % at 1.19 M rows against the compiler's own
  // schedule of mfma_half6; same products in the same order per accumulator: identical bits)
  if constexpr (NS == 1 && (HGN_ABL & 64) != 0) mfma_half6_pipe<HALF, NP>(acc[0], xs[0], lds);
  else if constexpr (NS == 1 && (HGN_ABL & 256) == 0) mfma_half6_pipe2<HALF, NP>(acc[0], xs[0], lds);
  else mfma_half6<HALF, NS, NP>(acc, xs, lds);
}

// acc[u][ob] += Wblock * b[u] for one packed 128 x 128 block.  `between()` runs after the first half's DMA has been issued and
// before the wait (the caller's own global loads fly with it); `b` is split after the wait, so `between` may load it.
// `post_split(b)` runs once `b` has been split into the bf16 operands: its registers are free from there to the end of the block,
// e.g. as the landing zone of a load the caller needs right after the block (two product halves and a DMA wait ahead of its use).
template <int NS, int NP, class F, class G>
__device__ __forceinline__ void gemm6(Act (&acc)[NS], Act (&b)[NS], __bf16* __restrict__ lds, const __bf16* __restrict__ pk,
                                      F&& between, G&& post_split) {
  bf16x8 xs[NS][3][4];
  int T[NS];                                      // scaled modes: acc[u] is carried at 2^T[u] through the block (row scale + block scale)
  int sw = 0;
  if constexpr (Prod<NP>::SCALED) sw = pack_scale_exp(pk);
  HGN_STAMP();                                    // 0: block entered
  wg_barrier_lds();
  HGN_STAMP();                                    // 1: stage free
  if (!(HGN_ABL & 1)) stage_half6<NP>(lds, pk);
  between();
  HGN_STAMP();                                    // 2: DMA + caller's loads issued
  __syncthreads();
  HGN_STAMP();                                    // 3: landed
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    T[u] = split_np<NP>(b[u], xs[u], &acc[u], sw) + sw;
    if constexpr (Prod<NP>::SCALED) scale_act(acc[u], T[u]);
  }
  post_split(b);
  HGN_STAMP();                                    // 4: split
  if (HGN_ABL & 32) __builtin_amdgcn_s_setprio(2);
  if (!(HGN_ABL & 2)) mfma_half6_sel<0, NS, NP>(acc, xs, lds);
  if (HGN_ABL & 32) __builtin_amdgcn_s_setprio(0);
  HGN_STAMP();                                    // 5: products of half 0 issued
  wg_barrier_lds();
  HGN_STAMP();                                    // 6: every wave done with half 0
  if (!(HGN_ABL & 1)) stage_half6<NP>(lds, pk + HALF_BF16);
  __syncthreads();
  HGN_STAMP();                                    // 7: half 1 landed
  if (HGN_ABL & 32) __builtin_amdgcn_s_setprio(2);
  if (!(HGN_ABL & 2)) mfma_half6_sel<1, NS, NP>(acc, xs, lds);
  if (HGN_ABL & 32) __builtin_amdgcn_s_setprio(0);
  HGN_STAMP();                                    // 8: products of half 1 issued
  if constexpr (Prod<NP>::SCALED) {
#pragma unroll
    for (int u = 0; u < NS; ++u) scale_act(acc[u], -T[u]);
  }
}
template <int NS, int NP, class F>
__device__ __forceinline__ void gemm6(Act (&acc)[NS], const Act (&b)[NS], __bf16* __restrict__ lds, const __bf16* __restrict__ pk,
                                      F&& between) {
  gemm6<NS, NP>(acc, const_cast<Act (&)[NS]>(b), lds, pk, between, [](Act (&)[NS]) {});
}

// ---- quarter-pipelined weight ring (throughput kernels: csrc/mlp6.hip mlp6_fwd_edge_kernel) ------------------------------------------
// gemm6 stages half a packed block, WAITS for it, multiplies, and does so twice per block: a workgroup spends ~1 us per wait with
// nothing to do, six times per tile (the ablation without weight DMA ran 22 % faster).  Here the 48 KB stage is a ring of two 24 KB
// slots, one CONTRACTION BLOCK ("piece": [split][output block][lane][8 bf16], as in csrc/fused_bwd.hip) each: while piece q multiplies
// out of slot q & 1 the LDS-DMA of piece q + 1 -- of the NEXT packed block when q = 3 -- is in flight into the other slot, which every
// wave finished reading before the barrier that opened piece q.  One barrier per piece (four per block, as before) and no exposed DMA:
//   wait (own DMA of piece q landed) ; barrier ; issue DMA of piece q + 1 ; [q = 0: split the operand] ; 48 NS products of piece q.
// The DMA is issued from inline assembly (glds_piece): issued through the builtin the compiler orders every later LDS read of the
// array behind it with a vmcnt(0), which is the wait this structure exists to remove.  Waits are explicit s_waitcnt vmcnt(KEEP):
// KEEP = vector-memory operations the wave issued AFTER that DMA that may stay in flight (they retire in order).
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int PIECE_BYTES6 = 3 * 8 * 1024;

__device__ __forceinline__ unsigned opaque_u(unsigned v) {      // one per-lane base register + immediates, never a register per address
  asm volatile("" : "+v"(v));
  return v;
}
template <int KEEP>
__device__ __forceinline__ void wait_vm_keep() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP) : "memory");
}

// The LDS-DMA of one piece by one of the four waves that share it (global_load_lds_dwordx4: lane l copies 16 bytes from
// sbase + voff(l) + imm to M0 + imm + 16 l; 1 KiB per instruction -- the immediate offset applies to BOTH addresses).  Wave ww copies
// operand tiles ww + 4 k, k = 0..5, i.e. split k / 2, output block ww + 4 (k & 1): in the packed block those lie
// (16 (k / 2) + 4 (k & 1)) KiB behind tile (split 0, output block ww), in the slot 4 k KiB.  The two tiles of a split are 4 KiB apart
// on both sides, so one M0 and one scalar base serve both with immediates -2 KiB / +2 KiB around a lane offset that carries +2 KiB
// (dma_lane_off): no vector instruction per DMA.  M0 is written in the statement that reads it and declared clobbered, like SCC
// (s_add_u32); one wait state between an M0 write and the DMA that reads it.  NP != 6: only the leading split is staged.
__device__ __forceinline__ unsigned dma_lane_off(unsigned lane) { return opaque_u(lane * 16u + 0x800u); }

template <int NP>
__device__ __forceinline__ void glds_piece(const void* sbase /*wave-uniform: source of tile (0, ww)*/, unsigned voff /*dma_lane_off*/,
                                           unsigned lds_dst /*wave-uniform: LDS byte address of tile (0, ww) in the slot*/) {
  const unsigned m0v = lds_dst + 0x800u;
  if constexpr (Prod<NP>::NSPLIT == 1) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048"
                 : : "v"(voff), "s"(sbase), "s"(m0v) : "memory", "m0");
  } else if constexpr (Prod<NP>::NSPLIT == 2) {
    asm volatile("" : "+s"(sbase));
    const void* s1 = static_cast<const unsigned char*>(sbase) + 0x4000;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:-2048\n\tglobal_load_lds_dwordx4 %0, %2 offset:2048"
                 : : "v"(voff), "s"(sbase), "s"(s1), "s"(m0v) : "memory", "scc", "m0");
  } else {
    asm volatile("" : "+s"(sbase));                  // the two bases below are made here (four scalar adds), not kept in registers per piece
    const void* s1 = static_cast<const unsigned char*>(sbase) + 0x4000;
    const void* s2 = static_cast<const unsigned char*>(sbase) + 0x8000;
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:-2048\n\tglobal_load_lds_dwordx4 %0, %2 offset:2048\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3 offset:-2048\n\tglobal_load_lds_dwordx4 %0, %3 offset:2048"
                 : : "v"(voff), "s"(sbase), "s"(s1), "s"(s2), "s"(m0v) : "memory", "scc", "m0");
  }
}
// The same piece by one of TWO waves (csrc/fused_bwd3.hip: the ring waves): wave ww in {0, 1} copies tiles ww + 2 k, k = 0..3, of both
// splits -- 2 KiB apart on both sides, so with a lane offset that carries +4 KiB (dma_lane_off2) the immediates -4096, -2048, 0, +2048
// reach all four from one M0 and one scalar base per split.  Two-split modes only.
__device__ __forceinline__ unsigned dma_lane_off2(unsigned lane) { return opaque_u(lane * 16u + 0x1000u); }
__device__ __forceinline__ void glds_piece2(const void* sbase /*wave-uniform: source of tile (0, ww)*/, unsigned voff /*dma_lane_off2*/,
                                            unsigned lds_dst /*wave-uniform: LDS byte address of tile (0, ww) in the slot*/) {
  const unsigned m0v = lds_dst + 0x1000u;
  asm volatile("" : "+s"(sbase));
  const void* s1 = static_cast<const unsigned char*>(sbase) + 0x4000;
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %1 offset:-4096\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048\n\t"
               "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\t"
               "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %0, %2 offset:-4096\n\tglobal_load_lds_dwordx4 %0, %2 offset:-2048\n\t"
               "global_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %0, %2 offset:2048"
               : : "v"(voff), "s"(sbase), "s"(s1), "s"(m0v) : "memory", "scc", "m0");
}
// piece c (contraction block) of the packed block `blk` -> ring slot `slot` of the ring at LDS byte address lds_base
template <int NP>
__device__ __forceinline__ void dma_piece6(const __bf16* __restrict__ blk, int c, unsigned lds_base, int slot, unsigned ww, unsigned voff) {
  const __bf16* src = blk + ((c >> 1) * HALF_BF16 + ((c & 1) * 8) * TILE_BF16) + ww * TILE_BF16;      // uniform: a scalar register pair
  if (!(HGN_ABL & 1)) glds_piece<NP>(src, voff, lds_base + slot * PIECE_BYTES6 + ww * 1024);
}

// acc[u][ob] += (piece: contraction block C) * x[u], six products per output block and sub-tile in the order of mfma_half6 (the same
// bits); the three fragments of output block ob + 1 are read while block ob multiplies (12 NS products: 192 NS matrix-pipe cycles).
template <int C, int NS, int NP>
__device__ __forceinline__ void sweep_piece6(Act (&acc)[NS], const bf16x8 (&xs)[NS][3][4], const unsigned char* __restrict__ lp /*slot + 16 lane*/) {
  if (HGN_ABL & 2) return;
  constexpr int NSP = Prod<NP>::NSPLIT;
  bf16x8 fr[2][3];
#pragma unroll
  for (int s = 0; s < NSP; ++s) fr[0][s] = *reinterpret_cast<const bf16x8*>(lp + (s * 8) * 1024);
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) {
    if (ob + 1 < NB) {
#pragma unroll
      for (int s = 0; s < NSP; ++s) fr[(ob + 1) & 1][s] = *reinterpret_cast<const bf16x8*>(lp + (s * 8 + ob + 1) * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bf16x8 (&a)[3] = fr[ob & 1];               // a[0] hi, a[1] mid, a[2] lo
    if constexpr (NS == 2 && NP == 6) {             // two independent accumulation chains, interleaved
      f32x4 t0 = acc[0].v[ob], t1 = acc[1].v[ob];
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], xs[0][0][C], t0, 0, 0, 0);      // smallest terms first
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], xs[1][0][C], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[0][2][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[1][2][C], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], xs[0][1][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], xs[1][1][C], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], xs[0][0][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], xs[1][0][C], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[0][1][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[1][1][C], t1, 0, 0, 0);
      t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[0][0][C], t0, 0, 0, 0);
      t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[1][0][C], t1, 0, 0, 0);
      acc[0].v[ob] = t0; acc[1].v[ob] = t1;
    } else if constexpr (NS == 2 && NP == 3) {      // two independent accumulation chains, interleaved; smallest terms first
      f32x4 t0 = acc[0].v[ob], t1 = acc[1].v[ob];
      t0 = mfma_f16(a[1], xs[0][0][C], t0);
      t1 = mfma_f16(a[1], xs[1][0][C], t1);
      t0 = mfma_f16(a[0], xs[0][1][C], t0);
      t1 = mfma_f16(a[0], xs[1][1][C], t1);
      t0 = mfma_f16(a[0], xs[0][0][C], t0);
      t1 = mfma_f16(a[0], xs[1][0][C], t1);
      acc[0].v[ob] = t0; acc[1].v[ob] = t1;
    } else {
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        f32x4 t = acc[u].v[ob];
        if constexpr (NP != 6) {
          t = mfma_np<NP>(a, xs[u][0][C], xs[u][1][C], xs[u][2][C], t);
        } else {
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], xs[u][0][C], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[u][2][C], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], xs[u][1][C], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], xs[u][0][C], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[u][1][C], t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[u][0][C], t, 0, 0, 0);
        }
        acc[u].v[ob] = t;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// One packed block through the two-slot ring.  On entry the DMA of the block's piece 0 is in flight into slot 0 (issued during the
// block before: `pk_next` there) unless `first`, in which case it is issued here behind a barrier.  `between()`: the caller's loads
// that must have LANDED before the products (operand rows, accumulator start values).  `at_piece(q, b)`: called at the start of piece
// q = 0..3, right after the DMA of the following piece has been issued (q = 0: after `b` has been split -- its registers are free from
// there on); KEEP1..3 = the vector-memory operations at_piece(0..2) issue: they stay in flight across the wait that follows them (a
// later wait drains them: vector memory retires in order, and the next DMA is younger than they are).  `pk_next` (nullable): the block
// whose piece 0 is fetched while this block's piece 3 multiplies.
// The wait for piece 0 + the "use" of everything loaded so far (see below): a function of its own so that a caller whose path to
// this point has a static operation count can run it INSIDE that path (KEEP > 0), before the path joins one whose count differs.
template <int NS, int KEEP>
__device__ __forceinline__ void gemm6q_landed(Act (&acc)[NS], Act (&b)[NS]) {
  wait_vm_keep<KEEP>();
  // The compiler does not see that wait (nor the DMAs): it would put its OWN counted waits in front of the first use of whatever
  // was loaded -- somewhere inside the sweeps, where its count (its loads only) also drains the younger DMAs it knows nothing
  // about.  Every such register is "used" here, before the next DMA is issued: the compiler's waits land here, already satisfied.
#pragma unroll
  for (int u = 0; u < NS; ++u)
    HGN_FOR_B(fb) asm volatile("" : "+v"(acc[u].v[fb]), "+v"(b[u].v[fb]));
}

// PREWAITED: the caller has run gemm6q_landed itself (and between() is empty).
template <int NS, int NP, bool PREWAITED, int KEEP1, int KEEP2, int KEEP3, class F, class G>
__device__ __forceinline__ void gemm6q(Act (&acc)[NS], Act (&b)[NS], __bf16* __restrict__ lds, const __bf16* __restrict__ pk,
                                       const __bf16* __restrict__ pk_next, bool first, F&& between, G&& at_piece) {
  bf16x8 xs[NS][3][4];
  int T[NS];
  int sw = 0;
  if constexpr (Prod<NP>::SCALED) sw = pack_scale_exp(pk);
  const unsigned lane = threadIdx.x & 63;
  const unsigned ww = (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)reinterpret_cast<unsigned char*>(lds);
  const unsigned char* ring = reinterpret_cast<const unsigned char*>(lds);
  if (first) {
    wg_barrier_lds();
    dma_piece6<NP>(pk, 0, lds_base, 0, ww, dma_lane_off(lane));
  }
  between();
  if constexpr (!PREWAITED) gemm6q_landed<NS, 0>(acc, b);
  __builtin_amdgcn_s_barrier();                     // ---- piece 0 has landed for every wave; slot 1 is free
  dma_piece6<NP>(pk, 1, lds_base, 1, ww, dma_lane_off(lane));
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    T[u] = split_np<NP>(b[u], xs[u], &acc[u], sw) + sw;
    if constexpr (Prod<NP>::SCALED) scale_act(acc[u], T[u]);
  }
  at_piece(0, b);
  sweep_piece6<0, NS, NP>(acc, xs, ring + opaque_u(lane * 16u));
  wait_vm_keep<KEEP1>();
  __builtin_amdgcn_s_barrier();                     // ---- piece 1 landed; every wave is done with slot 0
  dma_piece6<NP>(pk, 2, lds_base, 0, ww, dma_lane_off(lane));
  at_piece(1, b);
  sweep_piece6<1, NS, NP>(acc, xs, ring + opaque_u(lane * 16u + PIECE_BYTES6));
  wait_vm_keep<KEEP2>();
  __builtin_amdgcn_s_barrier();                     // ---- piece 2 landed; slot 1 free
  dma_piece6<NP>(pk, 3, lds_base, 1, ww, dma_lane_off(lane));
  at_piece(2, b);
  sweep_piece6<2, NS, NP>(acc, xs, ring + opaque_u(lane * 16u));
  wait_vm_keep<KEEP3>();
  __builtin_amdgcn_s_barrier();                     // ---- piece 3 landed; slot 0 free
  if (pk_next) dma_piece6<NP>(pk_next, 0, lds_base, 0, ww, dma_lane_off(lane));
  at_piece(3, b);
  sweep_piece6<3, NS, NP>(acc, xs, ring + opaque_u(lane * 16u + PIECE_BYTES6));
  if constexpr (Prod<NP>::SCALED) {
#pragma unroll
    for (int u = 0; u < NS; ++u) scale_act(acc[u], -T[u]);
  }
}

// ---- latency form (small launches: at most a tile or two per CU, csrc/mlp6.hip: mlp6_fwd_kernel<1, NP, 5>) -----------------------
// A workgroup that has its CU to itself (rollout: one 1 600-node graph = 146 edge tiles, 25 node tiles) spends most of a block
// of gemm6 waiting for its own weight DMA: issue -> 2 us of L2 latency -> barrier -> products, twice per block.  Here NL extra
// waves do nothing but stream the packed halves through a ring of three 48 KB LDS slots, two halves ahead of the products, and
// retires each with a counted wait of its own (its vector-memory queue holds the DMAs and nothing else); the four compute waves
// meet it at ONE barrier per half and never wait for a transfer.  Slot of half h = h mod 3; at barrier h every compute wave has
// finished the products of half h - 1, whose slot the loader then refills with half h + 2.
template <int NP> struct LatRing { static constexpr int PER = HALF_TILES / 3 * Prod<NP>::NSPLIT; };   // DMA instructions per half

// loader wave l of NL issues the operand tiles l, l + NL, ... of a half
template <int NP, int NL>
__device__ __forceinline__ void lat_issue_half(__bf16* __restrict__ slot, const __bf16* __restrict__ gsrc, unsigned l) {
  unsigned lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
#pragma unroll
  for (unsigned i = 0; i < (unsigned)(LatRing<NP>::PER / NL); ++i)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (i * NL + l) * TILE_BF16 + lane * 8),
                                     (__attribute__((address_space(3))) void*)(slot + (i * NL + l) * TILE_BF16), 16, 0, 0);
}
template <int N> __device__ __forceinline__ void lat_wait_keep() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// A loader wave (l of NL): `next()` yields the packed blocks in the order the compute waves multiply them (nullptr = no more).
// `extra_after` (optional): the compute waves run two barriers of their own right after the barrier that opens half
// `extra_after` (mlp6_fwd_cs_kernel: between the LayerNorm epilogue and the post-projection blocks); the loader joins them.
// `xbar` (column-split kernels in the scaled product mode 3: the four compute waves exchange the row maxima of a block's operand
// through LDS and run one barrier of their own for it, right BEFORE the barrier that opens the block's first half): 0 never, 1 in
// front of every block up to and including the one of half `extra_after` (the post blocks behind it share one operand), 2 in front
// of the first block only; `extra_n`: barriers of the compute waves behind half `extra_after`.
template <int NP, int NL, class Next>
__device__ __forceinline__ void lat_loader(__bf16* __restrict__ lds, unsigned l, Next&& next, int extra_after = -1, int xbar = 0, int extra_n = 2) {
  static_assert(LatRing<NP>::PER % NL == 0, "whole tiles per loader wave");
  constexpr int MINE = LatRing<NP>::PER / NL;
  int j = 0, slot = 0;
  auto open = [&](int h, bool last) {               // half h has landed (a younger one may still fly); then its barrier
    if ((h & 1) == 0 && (xbar == 1 || (xbar == 2 && h == 0)) && !(extra_after >= 0 && h >= extra_after)) __builtin_amdgcn_s_barrier();
    if (last) lat_wait_keep<0>(); else lat_wait_keep<MINE>();
    __builtin_amdgcn_s_barrier();
    if (h == extra_after) for (int k = 0; k < extra_n; ++k) __builtin_amdgcn_s_barrier();
  };
  for (const __bf16* pk = next(); pk != nullptr; pk = next()) {
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      if (j >= 2) open(j - 2, false);
      lat_issue_half<NP, NL>(lds + slot * HALF_BF16, pk + half * HALF_BF16, l);
      slot = slot == 2 ? 0 : slot + 1;
      ++j;
    }
  }
  open(j - 2, false);                               // barrier H - 2
  open(j - 1, true);                                // barrier H - 1
}

// A compute wave's block: the same products in the same order as gemm6 (identical bits); `slot` is the ring position of the
// block's first half and is advanced by two.
template <int NP, class F, class G>
__device__ __forceinline__ void gemm6_lat(Act (&acc)[1], Act (&b)[1], __bf16* __restrict__ lds, int& slot, const __bf16* __restrict__ pk,
                                          F&& between, G&& post_split) {
  bf16x8 xs[1][3][4];
  int sw = 0;
  if constexpr (Prod<NP>::SCALED) sw = pack_scale_exp(pk);
  HGN_STAMP();                                      // 0: block entered
  between();
  HGN_STAMP();                                      // 1: caller's loads issued
  wg_barrier_lds();                                 // half 2 b has landed
  HGN_STAMP();                                      // 2: past the barrier of half 0
  const int T = split_np<NP>(b[0], xs[0], &acc[0], sw) + sw;
  if constexpr (Prod<NP>::SCALED) scale_act(acc[0], T);
  post_split(b);
  HGN_STAMP();                                      // 3: split (the caller's loads have arrived)
  if (!(HGN_ABL & 2)) mfma_half6_sel<0, 1, NP>(acc, xs, lds + slot * HALF_BF16);
  slot = slot == 2 ? 0 : slot + 1;
  HGN_STAMP();                                      // 4: products of half 0 issued
  wg_barrier_lds();                                 // half 2 b + 1 has landed
  HGN_STAMP();                                      // 5: past the barrier of half 1
  if (!(HGN_ABL & 2)) mfma_half6_sel<1, 1, NP>(acc, xs, lds + slot * HALF_BF16);
  slot = slot == 2 ? 0 : slot + 1;
  HGN_STAMP();                                      // 6: products of half 1 issued
  if constexpr (Prod<NP>::SCALED) scale_act(acc[0], -T);
}

// The same block for a BIG workgroup (NWV waves, one per CU) that can afford a 96 KB stage: both halves are fetched at once, so
// a block costs one DMA wait and two barriers instead of two and four, and the weights are fetched once per 16 NWV rows.
template <int NS, int NP, int NWV, class F, class G>
__device__ __forceinline__ void gemm6_big(Act (&acc)[NS], Act (&b)[NS], __bf16* __restrict__ lds, const __bf16* __restrict__ pk,
                                          F&& between, G&& post_split) {
  bf16x8 xs[NS][3][4];
  static_assert(!Prod<NP>::SCALED, "big workgroups (laboratory build): unscaled product modes only");
  wg_barrier_lds();
  if (!(HGN_ABL & 1)) stage_block6<NP, NWV>(lds, pk);
  between();
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NS; ++u) split_np<NP>(b[u], xs[u]);
  post_split(b);
  if (!(HGN_ABL & 2)) mfma_half6<0, NS, NP>(acc, xs, lds);
  __builtin_amdgcn_sched_barrier(0);              // (no fragment of the second half in registers before the first is done)
  {
    // (an address register of its own for the upper half: DS immediates reach 64 KB, the stage is 96 KB)
    unsigned up = HALF_BF16 * 2;
    asm volatile("" : "+v"(up));
    const __bf16* lds1 = reinterpret_cast<const __bf16*>(reinterpret_cast<const char*>(lds) + up);
    if (!(HGN_ABL & 2)) mfma_half6<1, NS, NP>(acc, xs, lds1);
  }
}

__device__ __forceinline__ void relu6(Act& a) {
  HGN_FOR_B(fb)
#pragma unroll
  for (int u = 0; u < 4; ++u) a.v[fb][u] = fmaxf(a.v[fb][u], 0.f);
}

// Rows of a workgroup: NS sub-tiles of 64 consecutive rows, sub-tile u of wave w = rows 64u + 16w .. +15 of the tile (each
// sub-tile is a contiguous 64-row block, which the in-kernel segment sums rely on).
template <int NS, int NWV = WG / 64>
struct Rows {
  long row[NS], rc[NS]; bool valid[NS]; long tile_row0;
  __device__ __forceinline__ Rows(long M) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int SUB = WAVE_ROWS * NWV;               // rows of one sub-tile (64 for the 4-wave workgroups)
    tile_row0 = xcd_tile() * (SUB * NS);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      row[u] = tile_row0 + u * SUB + wave * WAVE_ROWS + (lane & 15);
      valid[u] = row[u] < M;
      rc[u] = valid[u] ? row[u] : M - 1;
    }
  }
};

}  // namespace hgn
