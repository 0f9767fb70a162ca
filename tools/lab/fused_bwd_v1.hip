// Edge-block backward with the weight gradients IN THE SAME PASS (include/hgn_mp.h: hgn_edge_bwd_fused).
//
// Why.  The separate kernels hand dz3 / dz2 / dz1 from the data-gradient chain (csrc/mlp6.hip: mlp6_bwd_kernel) to the weight-
// gradient kernel (csrc/wgrad.hip: wgrad6s_kernel) through HBM: 1.5 KB per edge row written, 3 KB read back, of the ~10.6 KB
// per edge and layer the whole step moved -- and both kernels were bound by exactly that row traffic.  Here dz3 / dz2 never
// leave the chip: 3.6 KB per row (read d(e'), x-hat, z2, z1, e; write de, dz1) instead of 6.1 KB for the two launches.
//
// How.  One PERSISTENT 8-wave workgroup per CU walks a contiguous range of 64-row tiles:
//   waves 0-3  "chain":   the data-gradient chain of mlp6_bwd_kernel for 16 rows each (LayerNorm backward -> W3^T -> relu' ->
//                         W2^T -> relu' -> W1e^T + residual), packed weights staged half a block at a time by LDS-DMA.  The
//                         3-way bf16 split of dz3 / dz2 / dz1 that each product needs anyway is ALSO written to LDS as the
//                         "G" operand of the weight gradients: eight consecutive rows of one feature = one bf16x8 vector.
//   waves 4-7  "wgrad":   keep dW3, dW2, dW1e (3 x 128 x 128 fp32 = 192 accumulator registers per lane) for the whole row
//                         range; per layer they load the 64 rows of the other operand (z2 / z1 / e) from HBM, split them
//                         once, publish them as "A" operand vectors and run dW += G^T A on v_mfma_f32_16x16x32_bf16
//                         (contraction over rows, six products, as in wgrad6s_kernel).
// Every SIMD hosts one wave of each kind; both follow the SAME barrier sequence (4 per layer: stage free / half landed / stage
// free / half landed), so the matrix pipe runs the chain's product for layer l and the weight gradient of layer l side by
// side, and the chain's weight-DMA waits coincide with the wgrad waves' row loads and splits.
// LDS: 48 KB weight stage + 48 KB G vectors + 48 KB A vectors + 4 KB LayerNorm partials = 148 KB of the CU's 160 KB.
// Per-workgroup partial results go to slabs that the existing fixed-order reductions add (deterministic, no float atomics).
#include <cstdlib>
#include "hgn_device.h"
#include "hgn_host.h"
#include "mlp_common.h"
#include "mlp6_device.h"

namespace hgn {

constexpr int FT = 512;                         // threads: 8 waves
constexpr int OPS64 = 3 * 8 * 128;              // bf16x8 vectors of one operand array of a 64-row tile: [split][row group][feature]
constexpr int FSLAB = 128 * 128 + 128;          // floats per (workgroup, layer): dW partial + bias partial (= wgrad.hip SLAB)
constexpr int FUSED_LDS = HALF_BF16 * 2 + 2 * OPS64 * 16 + 4 * 256 * 4;

struct FusedArgs {
  hgn_mlp_bwd_t b;                              // the data-gradient chain (n_dx == 1, residual, LayerNorm, ReLU sign words)
  const float* A[2]; long ldA[2];               // other operand of dW3, dW2: z2, z1
  float* slabs;                                 // [gridDim.x][2][FSLAB]
  long tiles;                                   // 64-row tiles
  int dbg;                                      // diagnostic ablations (HGN_FUSED_DBG): 1 no G writes, 2 no A publish, 4 no chain MFMA,
                                                // 8 no wgrad MFMA, 16 no weight DMA, 32 no row loads in the LayerNorm prologue
};

__device__ __forceinline__ void bar_lds() {     // every wave's LDS traffic issued so far is complete; global traffic stays in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ void bar_all() {     // ... and this wave's global loads / LDS-DMA have landed
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ void split3v8(const float (&v)[8], bf16x8 (&s)[3]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    const float r1 = v[j] - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    s[0][j] = h; s[1][j] = m; s[2][j] = (__bf16)r2;
  }
}

// The chain lane (row n of wave `wave`, feature quarter kq) publishes its split values as G operand: vector (split, row group
// 2*wave + n/8, feature), element n%8.
// (Every LDS address below is ONE per-lane base register, made opaque to the optimiser, plus a compile-time offset below the
// 64 KB reach of the DS instructions' immediate field -- left alone, the compiler materialises a register per (split, block,
// feature block) combination of the 148 KB image and spills them.)
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

template <int NP>
__device__ __forceinline__ void write_gops(unsigned char* __restrict__ gbase /*lane base inside the G image*/, const bf16x8 (&xs)[3][4]) {
  __bf16* gb = reinterpret_cast<__bf16*>(gbase);
#pragma unroll
  for (int s = 0; s < (NP == 1 ? 1 : 3); ++s)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int fofs = 32 * c + (j < 4 ? j : 16 + (j - 4));          // + 4 kq: in the lane base
        gb[(s * 8 * 128 + fofs) * 8] = xs[s][c][j];
      }
}

// dW_layer += G^T A over the 32 rows of block `blk` of the tile; wave ww owns dW rows [32 ww, 32 ww + 32).
// The A vectors of feature block nb + 1 are read while block nb multiplies (double-buffered; the sched_barriers keep the compiler
// from sinking the reads to their use, where every block would start with an exposed LDS round trip).
template <int NP>
__device__ __forceinline__ void wgrad_block(f32x4 (&acc)[2][8], float (&cs)[2], const bf16x8* __restrict__ gp /*lane base: G image*/,
                                            const bf16x8* __restrict__ ap /*lane base: A image*/, int blk) {
  constexpr int NS = NP == 1 ? 1 : 3;
  bf16x8 gs[2][3];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int s = 0; s < NS; ++s) gs[mb][s] = gp[(s * 8 + blk * 4) * 128 + 16 * mb];
  bf16x8 as[2][3];
#pragma unroll
  for (int s = 0; s < NS; ++s) as[0][s] = ap[(s * 8 + blk * 4) * 128];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {                 // bias gradient: the three split terms add up to the fp32 value exactly
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      float v = (float)gs[mb][0][p];
      if (NP != 1) v += (float)gs[mb][1][p] + (float)gs[mb][2][p];
      t += v;
    }
    cs[mb] += t;
  }
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    if (nb + 1 < 8) {
#pragma unroll
      for (int s = 0; s < NS; ++s) as[(nb + 1) & 1][s] = ap[(s * 8 + blk * 4) * 128 + 16 * (nb + 1)];
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

// mfma_half6 of mlp6_device.h with the operand fragments of one output block at a time (scheduling barrier per block): at the
// 256-register budget of this kernel the unconstrained scheduler hoists dozens of fragment reads and then spills.
template <int HALFI, int NP>
__device__ __forceinline__ void mfma_half6_sb(Act& acc, const bf16x8 (&xs)[3][4], const __bf16* __restrict__ lds) {
  const int lane = threadIdx.x & 63;
  const __bf16* lp = lds + lane * 8;
  bf16x8 fr[2][3];
#pragma unroll
  for (int s = 0; s < (NP == 1 ? 1 : 3); ++s) fr[0][s] = *reinterpret_cast<const bf16x8*>(lp + ((s * 2 + 0) * 8 + 0) * TILE_BF16);
#pragma unroll
  for (int i = 0; i < 16; ++i) {                   // i = 8 * cl + ob
    const int cl = i >> 3, ob = i & 7, c = 2 * HALFI + cl;
    if (i + 1 < 16) {
      const int cl1 = (i + 1) >> 3, ob1 = (i + 1) & 7;
#pragma unroll
      for (int s = 0; s < (NP == 1 ? 1 : 3); ++s)
        fr[(i + 1) & 1][s] = *reinterpret_cast<const bf16x8*>(lp + ((s * 2 + cl1) * 8 + ob1) * TILE_BF16);
    }
    __builtin_amdgcn_sched_barrier(0);             // issued here, ahead of this block's products
    const bf16x8 (&a)[3] = fr[i & 1];              // a[0] hi, a[1] mid, a[2] lo
    f32x4 t = acc.v[ob];
    if (NP == 1) {
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], xs[0][c], t, 0, 0, 0);
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

// (t_load32 / t_store32: csrc/hgn_device.h)

// Counted wait: everything but the `keep` most recently issued vector-memory operations has completed (they retire in order).
template <int KEEP>
__device__ __forceinline__ void bar_keep() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP) : "memory");
  __builtin_amdgcn_s_barrier();
}

template <int NP, int DBG>          // DBG: compile-time ablation mask of the diagnostic instantiation (HGN_FUSED_DBG), 0 in the product
__global__ __launch_bounds__(FT, 2) void edge_bwd_fused_kernel(const FusedArgs fa) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[FUSED_LDS];
  __bf16* wst = reinterpret_cast<__bf16*>(smem);
  float* lnl = reinterpret_cast<float*>(smem + HALF_BF16 * 2 + 2 * OPS64 * 16);
  const hgn_mlp_bwd_t& a = fa.b;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the role split is a scalar branch
  const long M = a.M;
  // this workgroup's tiles: workgroups b, b + 8, ... share an XCD (round-robin dispatch; speed only): XCD-major order
  const long G = gridDim.x, bx = blockIdx.x;
  const long q8 = G >> 3, r8 = G & 7, xc = bx & 7, ix = bx >> 3;
  const long pos = (xc < r8 ? xc * (q8 + 1) : r8 * (q8 + 1) + (xc - r8) * q8) + ix;
  const long t_beg = pos * fa.tiles / G, t_end = (pos + 1) * fa.tiles / G;

  if (wave < 4) {
    // ================================= data-gradient chain =================================
    const hgn_dx_t d = a.dx[0];
    const __bf16* pk3 = reinterpret_cast<const __bf16*>(a.W3pk_t);
    const __bf16* pk2 = reinterpret_cast<const __bf16*>(a.W2pk_t);
    const __bf16* pk1 = reinterpret_cast<const __bf16*>(d.Wpk_t);
    const bool has_dout = a.d_out != nullptr;
    Act g[1], t[1], gout;
    // next tile's d(e') and x-hat rows are loaded into gout / g while this tile's last product runs (both are dead by then)
    unsigned pf_m1 = 0, pf_m2 = 0, pf_seg = 0;
    bf16x8 xs[1][3][4];
    int n = lane & 15, kq = lane >> 4;
    // LayerNorm-affine gradient partials: the 16-row sums of a tile go through 1 KB of LDS per wave (written BEFORE the layer's
    // weight DMA is issued: the compiler waits for a pending LDS-DMA before any LDS access it cannot tell apart from the stage)
    // and are accumulated over the tiles in four registers per lane
    float lnacc[4] = {0.f, 0.f, 0.f, 0.f};
    float* lnw = nullptr;
    const unsigned ld_dout4 = (unsigned)a.ld_dout * 4u;
    unsigned char* gbase = nullptr;
    auto prefetch = [&](long tile) {                  // 8 (+ 8 with d_out) row loads, 2 sign-word loads (+ 1 receiver id)
      const long row = tile * TILE_ROWS + wave * WAVE_ROWS + n;
      const unsigned rc = (unsigned)(row < M ? row : M - 1);
      t_load32(g[0], a.xhat, rc * (LAT * 4u) + 16u * kq);
      if (has_dout) t_load32(gout, a.d_out, rc * ld_dout4 + 16u * kq);
      const unsigned* bits = reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(a.relu_bits) + (rc * 32u + 4u * kq));
      pf_m1 = bits[0];
      pf_m2 = bits[4];
      pf_seg = a.agg_dout ? (unsigned)a.agg_seg[rc] : 0u;      // receiver of the row: its d(agg) row is gathered at the tile's start
    };
    if (t_beg < t_end && !(DBG & 32)) prefetch(t_beg);
    for (long tile = t_beg; tile < t_end; ++tile) {
      // everything per-lane is re-derived from an opaque lane id inside the loop: otherwise the compiler hoists two dozen loop-
      // invariant 64-bit addresses and the 32 LayerNorm weights of the lane out of the loop and spills them
      const int lane_i = (int)opaque((unsigned)lane);
      n = lane_i & 15; kq = lane_i >> 4;
      // row n of this wave = row group 2 * wave + n / 8, element n % 8 of the G vectors; features 4 kq + ... of every 16-block
      gbase = smem + (unsigned)(HALF_BF16 * 2 + ((2 * wave + (n >> 3)) * 128 + 4 * kq) * 16 + (n & 7) * 2);
      lnw = reinterpret_cast<float*>(smem + (unsigned)(HALF_BF16 * 2 + 2 * OPS64 * 16 + (wave * 256 + 4 * kq) * 4));
      const long row = tile * TILE_ROWS + wave * WAVE_ROWS + n;
      const bool valid = row < M;
      const long rc = valid ? row : M - 1;
      // ---- layer 3: LayerNorm backward -> dz3 (g); t = W3^T dz3 ------------------------------------------------------
      bar_lds();
      const unsigned mb1 = pf_m1, mb2 = pf_m2;
      {
        Act& xh = t[0];
        if (DBG & 32) { t_zero(gout); t_zero(xh); }
        else {
          xh = g[0];
          if (!has_dout) t_zero(gout);
          if (a.agg_dout) {                           // `sum` aggregation backward: the receiver's d(agg) row (cache-resident gather)
            const unsigned r = pf_seg;
            const char* ar = reinterpret_cast<const char*>(a.agg_dout) + (r * ((unsigned)a.ld_agg * 4u) + 16u * kq);
            HGN_FOR_B(fb) gout.v[fb] += *reinterpret_cast<const f32x4*>(ar + 64 * fb);
          }
        }
        HGN_FOR_B(fb) {                               // LayerNorm-affine gradient partials of this wave's rows
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            float pb = valid ? gout.v[fb][w] : 0.f;
            const float pg = row16_sum(pb * xh.v[fb][w]);
            pb = row16_sum(pb);
            if (n == 0) { lnw[16 * fb + w] = pg; lnw[128 + 16 * fb + w] = pb; }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        HGN_FOR_B(fb) g[0].v[fb] = gout.v[fb] * *reinterpret_cast<const f32x4*>(a.ln_g + 16 * fb + 4 * kq);
        const float m1 = row_sum(g[0]) * (1.f / LAT);
        float q0 = 0.f, q1 = 0.f;
        HGN_FOR_B(fb) {
          q0 += g[0].v[fb][0] * xh.v[fb][0] + g[0].v[fb][1] * xh.v[fb][1];
          q1 += g[0].v[fb][2] * xh.v[fb][2] + g[0].v[fb][3] * xh.v[fb][3];
        }
        float qs = q0 + q1;
        qs += __shfl_xor(qs, 16);
        qs += __shfl_xor(qs, 32);
        const float m2 = qs * (1.f / LAT);
        const float r = valid ? a.rstd[rc] : 0.f;     // rows past the end contribute nothing to any weight gradient
        HGN_FOR_B(fb) g[0].v[fb] = r * (g[0].v[fb] - m1 - xh.v[fb] * m2);
      }
      if (!(DBG & 16)) stage_half6<NP>(wst, pk3);       // (after the LayerNorm partials went to LDS, see above)
      split3(g[0], xs[0]);
      t_zero(t[0]);
      bar_all();
      if (!(DBG & 1)) write_gops<NP>(gbase, xs[0]);
#pragma unroll
      for (int k = 0; k < 4; ++k) lnacc[k] += lnl[wave * 256 + lane + 64 * k];      // this wave's own partials of this tile
      if (!(DBG & 4)) mfma_half6_sb<0, NP>(t[0], xs[0], wst);
      bar_lds();
      if (!(DBG & 16)) stage_half6<NP>(wst, pk3 + HALF_BF16);
      bar_all();
      if (!(DBG & 4)) mfma_half6_sb<1, NP>(t[0], xs[0], wst);
      relu_mask_bits(t[0], mb2);                      // dz2
      // ---- layer 2: g = W2^T dz2 -------------------------------------------------------------------------------------
      bar_lds();
      if (!(DBG & 16)) stage_half6<NP>(wst, pk2);
      split3(t[0], xs[0]);
      t_zero(g[0]);
      bar_all();
      if (!(DBG & 1)) write_gops<NP>(gbase, xs[0]);
      if (!(DBG & 4)) mfma_half6_sb<0, NP>(g[0], xs[0], wst);
      bar_lds();
      if (!(DBG & 16)) stage_half6<NP>(wst, pk2 + HALF_BF16);
      bar_all();
      if (!(DBG & 4)) mfma_half6_sb<1, NP>(g[0], xs[0], wst);
      relu_mask_bits(g[0], mb1);                      // dz1
      // (whole 64-row tiles are stored: rows past M land in the padding the caller provides)
      t_store32(g[0], a.dz1, (unsigned)row * (LAT * 4u) + 16u * kq);
      // ---- layer 1: de = d_out_eff + dz1 W1e -------------------------------------------------------------------------
      bar_lds();
      if (!(DBG & 16)) stage_half6<NP>(wst, pk1);
      split3(g[0], xs[0]);
      t[0] = gout;                                    // the skip connection is the accumulator's start value
      // g and gout are dead: the next tile's rows start their way now and stay in flight across the next barrier (counted
      // wait: only the weight DMA issued before them has to have landed) and the first product half; the wait for the second
      // weight half then completes them (vector memory retires in order).  Past the last tile the clamped rows are unused.
      if (DBG & 32) bar_all();
      else {
        prefetch(tile + 1);
        if (a.agg_dout) { if (has_dout) bar_keep<19>(); else bar_keep<11>(); }
        else { if (has_dout) bar_keep<18>(); else bar_keep<10>(); }
      }
      if (!(DBG & 4)) mfma_half6_sb<0, NP>(t[0], xs[0], wst);
      bar_lds();
      if (!(DBG & 16)) stage_half6<NP>(wst, pk1 + HALF_BF16);
      bar_all();
      if (!(DBG & 4)) mfma_half6_sb<1, NP>(t[0], xs[0], wst);
      if (valid) t_store32(t[0], d.dx, (unsigned)row * ((unsigned)d.ld * 4u) + 16u * kq);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) lnl[wave * 256 + lane + 64 * k] = lnacc[k];
    bar_lds();                                        // (E) every chain wave's LayerNorm partials are in LDS
    const float sum = (lnl[tid] + lnl[256 + tid]) + (lnl[512 + tid] + lnl[768 + tid]);
    a.ln_ws[(long)blockIdx.x * 256 + tid] = sum;
    if (blockIdx.x == 0 && tid == 0) reinterpret_cast<unsigned*>(a.ln_ws)[-256] = 0u;     // ticket of ln_reduce_kernel (csrc/mlp.hip)
  } else {
    // ================================= weight gradients of layers 3 and 2 =================================
    const int ww = wave - 4, tw = tid - 256;
    const int blkp = tw >> 7, kgp = (tw >> 5) & 3, qd = tw & 31;      // producer role: 8 rows x 4 features of the A operand
    const int m = lane & 15, kg = lane >> 4;
    const bf16x8* gp = reinterpret_cast<const bf16x8*>(smem + opaque((unsigned)(HALF_BF16 * 2 + (kg * 128 + 32 * ww + m) * 16)));
    const bf16x8* ap = reinterpret_cast<const bf16x8*>(smem + opaque((unsigned)(HALF_BF16 * 2 + OPS64 * 16 + (kg * 128 + m) * 16)));
    bf16x8* apub = reinterpret_cast<bf16x8*>(smem + opaque((unsigned)(HALF_BF16 * 2 + OPS64 * 16 + ((blkp * 4 + kgp) * 128 + 4 * qd) * 16)));
    f32x4 acc[2][2][8];
    float cs[2][2];
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        cs[l][mb] = 0.f;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[l][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    // A operand (z2 for layer 3, z1 for layer 2) of a whole 64-row tile: this lane loads 4 features of 8 consecutive rows one
    // phase ahead, splits them once and publishes 4 x 3 operand vectors.
    f32x4 x[8];
    auto fetch = [&](int l, long tile) {
      const long r0 = tile * TILE_ROWS + blkp * 32 + kgp * 8;
      const char* A = reinterpret_cast<const char*>(fa.A[l]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned r = (unsigned)min(r0 + j, M - 1);
        x[j] = *reinterpret_cast<const f32x4*>(A + (r * (LAT * 4u) + 16u * qd));
      }
    };
    auto publish = [&]() {
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = x[j][f];
        bf16x8 sp[3];
        split3v8(v, sp);
#pragma unroll
        for (int s2 = 0; s2 < (NP == 1 ? 1 : 3); ++s2) apub[s2 * 8 * 128 + f] = sp[s2];
      }
    };
    // Interleaving with the chain (same four barriers per layer; the matrix pipe alternates between the two kinds of wave):
    //   chain:  VALU (next dz, split) | write G(l), product half 0 | wait for weight half 1 | product half 1
    //   wgrad:  dW(l+1) rows 32-63    | publish A(l), fetch next A | dW(l) rows 0-31        | -
    const bool pub = !(DBG & 2), mm = !(DBG & 8);
    if (t_beg < t_end && pub) fetch(0, t_beg);
    for (long tile = t_beg; tile < t_end; ++tile) {
      // chain layer 3 / dW3 (A = z2)
      bar_lds();
      bar_lds();
      if (pub) { publish(); fetch(1, tile); }
      bar_lds();
      if (mm) wgrad_block<NP>(acc[0], cs[0], gp, ap, 0);
      bar_lds();
      // chain layer 2 / dW2 (A = z1)
      bar_lds();
      if (mm) wgrad_block<NP>(acc[0], cs[0], gp, ap, 1);
      bar_lds();
      if (pub) { publish(); if (tile + 1 < t_end) fetch(0, tile + 1); }
      bar_lds();
      if (mm) wgrad_block<NP>(acc[1], cs[1], gp, ap, 0);
      bar_lds();
      // chain layer 1: no weight gradient here (dW1e = dz1^T e goes through the streaming kernel: dz1 is in memory anyway)
      bar_lds();
      if (mm) wgrad_block<NP>(acc[1], cs[1], gp, ap, 1);
      bar_lds();
      bar_lds();
      bar_lds();
    }
    bar_lds();                                        // (E)
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      float* slab = fa.slabs + ((long)blockIdx.x * 2 + l) * FSLAB;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[(32 * ww + 16 * mb + 4 * kg + r) * 128 + 16 * nb + m] = acc[l][mb][nb][r];
        float v = cs[l][mb];                          // the four row groups of a feature live in four lanes
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (kg == 0) slab[128 * 128 + 32 * ww + 16 * mb + m] = v;
      }
    }
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
    return (long)cus;                                 // one 8-wave workgroup per CU (148 KB of LDS each)
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
  static const bool off = getenv("HGN_NO_FUSED_BWD") != nullptr || getenv("HGN_FP32_MFMA") != nullptr;
  if (off || !a || !hgn_mlp_bwd6_eligible(a)) return 0;
  if (a->n_dx != 1 || !a->dx[0].residual || a->dx[0].K != 128 || a->seg_dz1 || !a->dz1) return 0;
  if (a->agg_dout && (a->n_agg_ops != 1 || a->agg_ops[0] != HGN_OP_SUM)) return 0;      // several aggregates (pna): the two-launch path
  const int64_t ldmax = a->ld_dout > a->dx[0].ld ? a->ld_dout : a->dx[0].ld;
  if (a->M * (ldmax > 128 ? ldmax : 128) * 4 >= ((int64_t)1 << 32)) return 0;      // 32-bit row offsets inside the kernel
  return 1;
}

extern "C" int hgn_edge_bwd_fused(const hgn_mlp_bwd_t* a, const hgn_wfuse_t* w, void* workspace, size_t ws_bytes, void* stream_) {
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
  fa.b.ln_ws += 256;                                  // slab 0 lies behind the header slab (ticket of ln_reduce_kernel)
  fa.A[0] = w->z2; fa.ldA[0] = 128;
  fa.A[1] = w->z1; fa.ldA[1] = 128;
  fa.slabs = (float*)workspace;
  fa.tiles = (a->M + TILE_ROWS - 1) / TILE_ROWS;
  fa.dbg = 0;
  ProfScope ps(14, (double)a->M, stream);
  if (bwd_products(a->products) == 1) hipLaunchKernelGGL((edge_bwd_fused_kernel<1, 0>), dim3((unsigned)G), dim3(FT), 0, stream, fa);
  else {
#if HGN_LAB   // laboratory build only: compile-time ablation instantiations (HGN_FUSED_DBG), one ablation each
    static const int dbg = getenv("HGN_FUSED_DBG") ? atoi(getenv("HGN_FUSED_DBG")) : 0;
    fa.dbg = dbg;
    switch (dbg) {
      case 2: hipLaunchKernelGGL((edge_bwd_fused_kernel<6, 2>), dim3((unsigned)G), dim3(FT), 0, stream, fa); break;
      case 4: hipLaunchKernelGGL((edge_bwd_fused_kernel<6, 4>), dim3((unsigned)G), dim3(FT), 0, stream, fa); break;
      case 8: hipLaunchKernelGGL((edge_bwd_fused_kernel<6, 8>), dim3((unsigned)G), dim3(FT), 0, stream, fa); break;
      case 12: hipLaunchKernelGGL((edge_bwd_fused_kernel<6, 12>), dim3((unsigned)G), dim3(FT), 0, stream, fa); break;
      case 16: hipLaunchKernelGGL((edge_bwd_fused_kernel<6, 16>), dim3((unsigned)G), dim3(FT), 0, stream, fa); break;
      case 32: hipLaunchKernelGGL((edge_bwd_fused_kernel<6, 32>), dim3((unsigned)G), dim3(FT), 0, stream, fa); break;
      case 63: hipLaunchKernelGGL((edge_bwd_fused_kernel<6, 63>), dim3((unsigned)G), dim3(FT), 0, stream, fa); break;
      default: hipLaunchKernelGGL((edge_bwd_fused_kernel<6, 0>), dim3((unsigned)G), dim3(FT), 0, stream, fa);
    }
#else
    hipLaunchKernelGGL((edge_bwd_fused_kernel<6, 0>), dim3((unsigned)G), dim3(FT), 0, stream, fa);
#endif
  }
  if (hgn_check_launch("hgn_edge_bwd_fused") != HGN_OK) return HGN_E_LAUNCH;
  // fixed-order sums of the per-workgroup partials: three weight gradients + biases, and the LayerNorm-affine gradients
  SlabReduceTask rt[2];
  float* dW[2] = {w->dW3, w->dW2};
  float* db[2] = {w->db3, w->db2};
  for (int l = 0; l < 2; ++l) {
    rt[l].type = 0; rt[l].K = 128; rt[l].n_out = 128; rt[l].acc = w->accumulate ? 1 : 0; rt[l].n_chunks = (int)G;
    rt[l].dW = dW[l]; rt[l].ldw = 128; rt[l].db = db[l]; rt[l].slab = fa.slabs + (long)l * FSLAB;
    rt[l].chunk_stride = 2L * FSLAB;
  }
  if (launch_slab_reduce(rt, 2, stream) != HGN_OK) return HGN_E_LAUNCH;
  // LayerNorm partial slabs: ln_ws holds hgn_mlp_bwd_ln_workspace_bytes(M) bytes = (tiles + parts) slabs; G <= tiles
  if (launch_ln_reduce(fa.b.ln_ws, G, fa.b.ln_ws + G * 256, a->d_gamma, a->d_beta, a->ln_accumulate, stream) != HGN_OK) return HGN_E_LAUNCH;
  return hgn_check_launch("hgn_edge_bwd_fused (reductions)");
}
