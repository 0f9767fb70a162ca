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
  const float* A[3]; long ldA[3];               // other operand of dW3, dW2, dW1e: z2, z1, e
  float* slabs;                                 // [gridDim.x][3][FSLAB]
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

// dW_layer += G^T A over the 32 rows of block `blk` of the tile; wave ww owns dW rows [32 ww, 32 ww + 32)
template <int NP>
__device__ __forceinline__ void wgrad_block(f32x4 (&acc)[2][8], float (&cs)[2], const bf16x8* __restrict__ gp /*lane base: G image*/,
                                            const bf16x8* __restrict__ ap /*lane base: A image*/, int blk) {
  bf16x8 gs[2][3];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int s = 0; s < (NP == 1 ? 1 : 3); ++s) gs[mb][s] = gp[(s * 8 + blk * 4) * 128 + 16 * mb];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {               // bias gradient: the three split terms add up to the fp32 value exactly
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
    bf16x8 as[3];
#pragma unroll
    for (int s = 0; s < (NP == 1 ? 1 : 3); ++s) as[s] = ap[(s * 8 + blk * 4) * 128 + 16 * nb];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x4 c = acc[mb][nb];
      if (NP != 1) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][2], as[0], c, 0, 0, 0);      // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], as[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], as[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], as[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], as[1], c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], as[0], c, 0, 0, 0);
      acc[mb][nb] = c;
    }
    __builtin_amdgcn_sched_barrier(0);           // keep the operand vectors of later blocks out of the register file
  }
}

template <int NP>
__global__ __launch_bounds__(FT, 2) void edge_bwd_fused_kernel(const FusedArgs fa) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[FUSED_LDS];
  __bf16* wst = reinterpret_cast<__bf16*>(smem);
  bf16x8* gops = reinterpret_cast<bf16x8*>(smem + HALF_BF16 * 2);
  bf16x8* aops = gops + OPS64;
  float* lnl = reinterpret_cast<float*>(aops + OPS64);
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
    const int n = lane & 15, kq = lane >> 4;
    const hgn_dx_t d = a.dx[0];
    const __bf16* pk3 = reinterpret_cast<const __bf16*>(a.W3pk_t);
    const __bf16* pk2 = reinterpret_cast<const __bf16*>(a.W2pk_t);
    const __bf16* pk1 = reinterpret_cast<const __bf16*>(d.Wpk_t);
    Act g[1], t[1], gout;
    bf16x8 xs[1][3][4];
    // row n of this wave = row group 2 * wave + n / 8, element n % 8 of the G vectors; features 4 kq + ... of every 16-block
    unsigned char* gbase = smem + opaque((unsigned)(HALF_BF16 * 2 + ((2 * wave + (n >> 3)) * 128 + 4 * kq) * 16 + (n & 7) * 2));
    float* lnw = reinterpret_cast<float*>(smem + opaque((unsigned)(HALF_BF16 * 2 + 2 * OPS64 * 16 + (wave * 256 + 4 * kq) * 4)));
    float lnacc[4] = {0.f, 0.f, 0.f, 0.f};
    for (long tile = t_beg; tile < t_end; ++tile) {
      const long row = tile * TILE_ROWS + wave * WAVE_ROWS + n;
      const bool valid = row < M;
      const long rc = valid ? row : M - 1;
      const unsigned mb1 = a.relu_bits[rc * 8 + kq], mb2 = a.relu_bits[rc * 8 + 4 + kq];
      // ---- layer 3: LayerNorm backward -> dz3 (g); t = W3^T dz3 ------------------------------------------------------
      bar_lds();
      if (!(fa.dbg & 16)) stage_half6<NP>(wst, pk3);
      {
        Act& xh = t[0];
        if (!(fa.dbg & 32)) { load_dout<false>(gout, a, rc, kq); t_load(xh, a.xhat + rc * LAT, kq); }
        else { t_zero(gout); t_zero(xh); }
        HGN_FOR_B(fb) {                               // LayerNorm-affine gradient partials of this wave's rows
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            float pb = valid ? gout.v[fb][w] : 0.f;
            float pg = row16_sum(pb * xh.v[fb][w]);
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
      split3(g[0], xs[0]);
      t_zero(t[0]);
      bar_all();
      if (!(fa.dbg & 1)) write_gops<NP>(gbase, xs[0]);
#pragma unroll
      for (int k = 0; k < 4; ++k) lnacc[k] += lnl[wave * 256 + lane + 64 * k];      // this wave's own partials of this tile
      if (!(fa.dbg & 4)) mfma_half6<0, 1, NP>(t, xs, wst);
      bar_lds();
      if (!(fa.dbg & 16)) stage_half6<NP>(wst, pk3 + HALF_BF16);
      bar_all();
      if (!(fa.dbg & 4)) mfma_half6<1, 1, NP>(t, xs, wst);
      relu_mask_bits(t[0], mb2);                      // dz2
      // ---- layer 2: g = W2^T dz2 -------------------------------------------------------------------------------------
      bar_lds();
      if (!(fa.dbg & 16)) stage_half6<NP>(wst, pk2);
      split3(t[0], xs[0]);
      t_zero(g[0]);
      bar_all();
      if (!(fa.dbg & 1)) write_gops<NP>(gbase, xs[0]);
      if (!(fa.dbg & 4)) mfma_half6<0, 1, NP>(g, xs, wst);
      bar_lds();
      if (!(fa.dbg & 16)) stage_half6<NP>(wst, pk2 + HALF_BF16);
      bar_all();
      if (!(fa.dbg & 4)) mfma_half6<1, 1, NP>(g, xs, wst);
      relu_mask_bits(g[0], mb1);                      // dz1
      if (a.dz1 && valid) t_store(g[0], a.dz1 + row * LAT, kq);
      // ---- layer 1: de = d_out_eff + dz1 W1e -------------------------------------------------------------------------
      bar_lds();
      if (!(fa.dbg & 16)) stage_half6<NP>(wst, pk1);
      split3(g[0], xs[0]);
      t[0] = gout;                                    // the skip connection is the accumulator's start value
      bar_all();
      if (!(fa.dbg & 1)) write_gops<NP>(gbase, xs[0]);
      if (!(fa.dbg & 4)) mfma_half6<0, 1, NP>(t, xs, wst);
      bar_lds();
      if (!(fa.dbg & 16)) stage_half6<NP>(wst, pk1 + HALF_BF16);
      bar_all();
      if (!(fa.dbg & 4)) mfma_half6<1, 1, NP>(t, xs, wst);
      if (valid) t_store(t[0], d.dx + row * d.ld, kq);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) lnl[wave * 256 + lane + 64 * k] = lnacc[k];
    bar_lds();                                        // (E) every chain wave's LayerNorm partials are in LDS
    const float sum = (lnl[tid] + lnl[256 + tid]) + (lnl[512 + tid] + lnl[768 + tid]);
    a.ln_ws[(long)blockIdx.x * 256 + tid] = sum;
  } else {
    // ================================= weight gradients =================================
    const int ww = wave - 4, tw = tid - 256;
    const int blkp = tw >> 7, kgp = (tw >> 5) & 3, qd = tw & 31;      // producer role: 8 rows x 4 features of the A operand
    const int m = lane & 15, kg = lane >> 4;
    const bf16x8* gp = reinterpret_cast<const bf16x8*>(smem + opaque((unsigned)(HALF_BF16 * 2 + (kg * 128 + 32 * ww + m) * 16)));
    const bf16x8* ap = reinterpret_cast<const bf16x8*>(smem + opaque((unsigned)(HALF_BF16 * 2 + OPS64 * 16 + (kg * 128 + m) * 16)));
    bf16x8* apub = reinterpret_cast<bf16x8*>(smem + opaque((unsigned)(HALF_BF16 * 2 + OPS64 * 16 + ((blkp * 4 + kgp) * 128 + 4 * qd) * 16)));
    f32x4 acc[3][2][8];
    float cs[3][2];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        cs[l][mb] = 0.f;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[l][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    // A operand of layer l for the whole 64-row tile: this lane loads 4 features of 8 consecutive rows, splits them once and
    // publishes 4 x 3 operand vectors.  Loaded and consumed inside the window in which the chain waits for its weight DMA, so
    // the 32 row registers are dead again while the accumulators work.
    auto publish = [&](int l, long tile) {
      const long r0 = tile * TILE_ROWS + blkp * 32 + kgp * 8;
      const float* A = fa.A[l] + 4 * qd;
      const long ld = fa.ldA[l];
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4* pub = reinterpret_cast<bf16x4*>(apub);
#pragma unroll
      for (int h = 0; h < 2; ++h) {                   // rows 4h .. 4h+3 of the group = one half (8 bytes) of each operand vector
        f32x4 x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long r = min(r0 + 4 * h + j, M - 1);
          x[j] = *reinterpret_cast<const f32x4*>(A + r * ld);
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          bf16x4 sp[3];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = x[j][f];
            const __bf16 hi = (__bf16)v;
            const float r1 = v - (float)hi;
            const __bf16 mi = (__bf16)r1;
            sp[0][j] = hi; sp[1][j] = mi; sp[2][j] = (__bf16)(r1 - (float)mi);
          }
#pragma unroll
          for (int s2 = 0; s2 < (NP == 1 ? 1 : 3); ++s2) pub[(s2 * 8 * 128 + f) * 2 + h] = sp[s2];
        }
      }
    };
    // Interleaving with the chain (same four barriers per layer; the matrix pipe alternates between the two kinds of wave):
    //   chain:  VALU (next dz, split) | write G(l), product half 0 | wait for weight half 1 | product half 1
    //   wgrad:  dW(l+1) block 1       | publish A(l)               | dW(l) block 0          | -
    // G(l) / A(l) are written between the 2nd and 3rd barrier and last read before the 2nd barrier of the NEXT layer.
    bool pending = false;                             // block 1 of the previous layer still to be multiplied
    for (long tile = t_beg; tile < t_end; ++tile) {
#pragma unroll
      for (int l = 0; l < 3; ++l) {                   // layer 3 (A = z2), layer 2 (A = z1), layer 1 (A = e)
        bar_lds();
        if (pending && !(fa.dbg & 8)) wgrad_block<NP>(acc[(l + 2) % 3], cs[(l + 2) % 3], gp, ap, 1);
        bar_lds();
        if (!(fa.dbg & 2)) publish(l, tile);
        bar_lds();
        if (!(fa.dbg & 8)) wgrad_block<NP>(acc[l], cs[l], gp, ap, 0);
        bar_lds();
        pending = true;
      }
    }
    if (pending && !(fa.dbg & 8)) wgrad_block<NP>(acc[2], cs[2], gp, ap, 1);
    bar_lds();                                        // (E)
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      float* slab = fa.slabs + ((long)blockIdx.x * 3 + l) * FSLAB;
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
  *bytes = (size_t)fused_grid(M) * 3 * FSLAB * sizeof(float) + 256;
  return HGN_OK;
}

extern "C" int hgn_edge_bwd_fused_eligible(const hgn_mlp_bwd_t* a) {
  static const bool off = getenv("HGN_NO_FUSED_BWD") != nullptr || getenv("HGN_FP32_MFMA") != nullptr;
  if (off || !a || !hgn_mlp_bwd6_eligible(a)) return 0;
  if (a->n_dx != 1 || !a->dx[0].residual || a->dx[0].K != 128 || a->seg_dz1) return 0;
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
  if (!w->z2 || !w->z1 || !w->x || (w->ldx & 3) || !aligned16(w->z2) || !aligned16(w->z1) || !aligned16(w->x) || !w->dW3 ||
      !w->dW2 || !w->dW1 || w->ldw1 < 128)
    return hgn_fail(HGN_E_INVALID, "hgn_edge_bwd_fused: bad weight-gradient operands");
  const long G = fused_grid(a->M);
  FusedArgs fa;
  fa.b = *a;
  fa.A[0] = w->z2; fa.ldA[0] = 128;
  fa.A[1] = w->z1; fa.ldA[1] = 128;
  fa.A[2] = w->x; fa.ldA[2] = w->ldx;
  fa.slabs = (float*)workspace;
  fa.tiles = (a->M + TILE_ROWS - 1) / TILE_ROWS;
  static const int dbg = getenv("HGN_FUSED_DBG") ? atoi(getenv("HGN_FUSED_DBG")) : 0;
  fa.dbg = dbg;
  ProfScope ps(14, (double)a->M, stream);
  if (matmul_products() == 1) hipLaunchKernelGGL(edge_bwd_fused_kernel<1>, dim3((unsigned)G), dim3(FT), 0, stream, fa);
  else hipLaunchKernelGGL(edge_bwd_fused_kernel<6>, dim3((unsigned)G), dim3(FT), 0, stream, fa);
  if (hgn_check_launch("hgn_edge_bwd_fused") != HGN_OK) return HGN_E_LAUNCH;
  // fixed-order sums of the per-workgroup partials: three weight gradients + biases, and the LayerNorm-affine gradients
  SlabReduceTask rt[3];
  float* dW[3] = {w->dW3, w->dW2, w->dW1};
  long ldw[3] = {128, 128, (long)w->ldw1};
  float* db[3] = {w->db3, w->db2, w->db1};
  for (int l = 0; l < 3; ++l) {
    rt[l].type = 0; rt[l].K = 128; rt[l].n_out = 128; rt[l].acc = w->accumulate ? 1 : 0; rt[l].n_chunks = (int)G;
    rt[l].dW = dW[l]; rt[l].ldw = ldw[l]; rt[l].db = db[l]; rt[l].slab = fa.slabs + (long)l * FSLAB;
    rt[l].chunk_stride = 3L * FSLAB;
  }
  if (launch_slab_reduce(rt, 3, stream) != HGN_OK) return HGN_E_LAUNCH;
  // LayerNorm partial slabs: ln_ws holds hgn_mlp_bwd_ln_workspace_bytes(M) bytes = (tiles + parts) slabs; G <= tiles
  if (launch_ln_reduce(a->ln_ws, G, a->ln_ws + G * 256, a->d_gamma, a->d_beta, a->ln_accumulate, stream) != HGN_OK) return HGN_E_LAUNCH;
  return hgn_check_launch("hgn_edge_bwd_fused (reductions)");
}
