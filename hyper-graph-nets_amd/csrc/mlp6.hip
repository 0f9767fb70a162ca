// Fused MLP kernels with SPLIT-bf16 matrix products: every fp32 128x128 product of csrc/mlp.hip evaluated as six bf16 MFMAs
// (v_mfma_f32_16x16x32_bf16, fp32 accumulation) on a 3-way bf16 split of both operands.
//   x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) captures all 24 significand bits of an
//   fp32 value exactly; w likewise; the six products with i + j <= 4 leave out terms below 2^-24 relative.  Measured on the
//   MLP shapes: max relative error 2.6e-7 against fp64, vs 4.5e-7 for a plain fp32 product -- fp32 accuracy at 16/6 = 2.7x
//   the fp32 MFMA rate, which moves these kernels from "MFMA pipe and row traffic equally loaded" to row-traffic bound.
// Formulation is the one of hgn_device.h (transposed product, weights = A operand from LDS, the wave's 16 rows = B operand in
// registers, C/D map == next layer's B map).  For the bf16 MFMA the B operand of contraction block c (32 features) takes
// its 8 values per lane from the lane's own fp32 registers -- features 32c + 4q + j (j < 4) and 32c + 16 + 4q + (j - 4) --
// i.e. a fixed permutation of the contraction index, which is folded into the PACKED weights (hgn_pack_bf16x3): operand
// tiles of 16 x 32 bf16 stored in lane order, so staging is a straight LDS-DMA copy and every ds_read_b128 is linear.
// LDS: half a block (two contraction blocks x three splits x eight output blocks = 48 KB) at a time, 3 workgroups per CU.
#include <cstdlib>
#include <type_traits>
#include "hgn_device.h"
#include "hgn_host.h"
#include "mlp_common.h"
#include "mlp6_device.h"

namespace hgn {

// ----------------------------------------------------------------------------------------------------------
// packing: [half][split][cl][ob][lane][8];  forward form: tile(ob, c)[m][p] = W[16ob + m][feat(c, p)]  (contraction over the
// input features);  transposed form: tile(ob, c)[m][p] = W[feat(c, p)][16ob + m]  (contraction over the outputs)
// ----------------------------------------------------------------------------------------------------------
struct PackArgs { hgn_pack_t d[HGN_MAX_PACK]; };

// Scaled two-term fp16 images (hgn_pack_t.transposed & 4, product mode 3): the block's power-of-two scale first -- the exponent sw
// with max|W| * 2^sw in [2^14, 2^15) (an all-zero block: 15), written to the image's header (mlp6_device.h: PK_SCALE_BYTE) by one
// workgroup per block; the pack kernel behind it on the stream reads it.  Blocks of the other forms: nothing to do.
__device__ __forceinline__ void pack_scale_block(const hgn_pack_t& d) {
  if (!(d.transposed & 4)) return;
  __shared__ float part[4];
  const int transposed = d.transposed & 1;
  float m = 0.f;
  for (int i = threadIdx.x; i < d.n_out * d.n_in; i += blockDim.x) {
    const int o = i / d.n_in, k = i - o * d.n_in;
    m = fmaxf(m, fabsf(d.W[(long)o * d.ldw + k]));
  }
  (void)transposed;                                 // (the extents name the same sub-matrix in both forms)
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) m = fmaxf(m, __shfl_xor(m, sh));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    *reinterpret_cast<int*>(reinterpret_cast<char*>(d.out) + PK_SCALE_BYTE) = scale_exp_of(m);
  }
}
__global__ void pack_scale_kernel(const PackArgs a) { pack_scale_block(a.d[blockIdx.x]); }
__global__ void pack_scale_table_kernel(const hgn_pack_t* __restrict__ table) { pack_scale_block(table[blockIdx.x]); }

__device__ __forceinline__ void pack_block(const hgn_pack_t& d);
__global__ void pack_bf16x3_kernel(const PackArgs a) { pack_block(a.d[blockIdx.y]); }
__global__ void pack_bf16x3_table_kernel(const hgn_pack_t* __restrict__ table) { pack_block(table[blockIdx.y]); }
__device__ __forceinline__ void pack_block(const hgn_pack_t& d) {
  const float* __restrict__ W = d.W;
  const long ldw = d.ldw;
  const int n_out = d.n_out, n_in = d.n_in, transposed = d.transposed & 1, f16 = d.transposed & 2, f16x2 = d.transposed & 4;
  __bf16* __restrict__ out = reinterpret_cast<__bf16*>(d.out);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;        // one (half, cl, ob, lane, j)
  if (i >= 2 * 2 * 8 * 64 * 8) return;
  const int j = i & 7, l = (i >> 3) & 63, ob = (i >> 9) & 7, cl = (i >> 12) & 1, half = i >> 13;
  const int m = l & 15, q = l >> 4, c = 2 * half + cl;
  const int feat = 32 * c + (j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4));
  const int o = transposed ? feat : 16 * ob + m, k = transposed ? 16 * ob + m : feat;     // W[o][k]
  float w = 0.f;
  if (o < n_out && k < n_in) w = W[(long)o * ldw + k];
  const __bf16 h = (__bf16)w;
  const float r1 = w - (float)h;
  const __bf16 mi = (__bf16)r1;
  const float r2 = r1 - (float)mi;
  const __bf16 lo = (__bf16)r2;
  __bf16* base = out + (long)half * HALF_BF16 + ((cl * 8 + ob) * 64 + l) * 8 + j;
  if (f16x2) {                     // two fp16 terms of W * 2^sw (the third split's region holds the header: not written here)
    const float ws = __builtin_amdgcn_ldexpf(w, pack_scale_exp(out));
    const _Float16 hf = (_Float16)ws;
    const _Float16 lf = (_Float16)(ws - (float)hf);
    base[0 * 2 * 8 * TILE_BF16] = __builtin_bit_cast(__bf16, hf);
    base[1 * 2 * 8 * TILE_BF16] = __builtin_bit_cast(__bf16, lf);
    return;
  }
  if (f16) {                       // reduced-precision forward (one fp16 product): the leading third holds fp16 bit patterns
    const _Float16 hf = (_Float16)w;
    base[0 * 2 * 8 * TILE_BF16] = __builtin_bit_cast(__bf16, hf);
    base[1 * 2 * 8 * TILE_BF16] = (__bf16)0.f;
    base[2 * 2 * 8 * TILE_BF16] = (__bf16)0.f;
    return;
  }
  base[0 * 2 * 8 * TILE_BF16] = h;
  base[1 * 2 * 8 * TILE_BF16] = mi;
  base[2 * 2 * 8 * TILE_BF16] = lo;
}

// ----------------------------------------------------------------------------------------------------------
// forward (output 128 wide): same contract as mlp_fwd_kernel.  NS = 2: 128-row workgroup tiles, 2 waves / SIMD -- half the
// weight DMA, LDS operand reads and barriers per row (0.635 -> 0.554 ms for 594 048 edge rows in tools/micro/bf16x6_mlp.hip).
// ----------------------------------------------------------------------------------------------------------
// NWV = 12 ("big": one 12-wave workgroup per CU on a 192-row tile instead of three 4-wave workgroups on 64 rows each): the three
// groups share ONE weight stage, which can then hold a whole 96 KB block -- a third of the L2 -> LDS traffic, one DMA wait and
// two barriers per block instead of two and four (gemm6_big).  Same per-row arithmetic, bit for bit.
#if HGN_ABL & 512
// timing experiment only (WRONG rows): every store instruction covers two whole rows instead of 64 B of each of 16 rows
__device__ __forceinline__ void t_store_rows_x(const Act& a, float* __restrict__ p0, long ld) {
  const int lane = threadIdx.x & 63;
  float* p = p0 + (lane >> 5) * ld + (lane & 31) * 4;
  HGN_FOR_B(fb) *reinterpret_cast<f32x4*>(p + 2 * fb * ld) = a.v[fb];
}
#define HGN_TSTORE(act, base, ld, u) do { if (R.row[u] - (threadIdx.x & 15) + 16 <= a.M) t_store_rows_x(act, (base) + (R.row[u] - (threadIdx.x & 15)) * (long)(ld), ld); } while (0)
#else
// (128-row workgroups -- every big launch: rows go out as whole cache lines through the wave's LDS stage, hgn_device.h: t_store_rows)
#define HGN_TSTORE(act, base, ld, u) do {                                                                        \
    if constexpr (STAGED) t_store_rows(act, base, R.row[u] - (threadIdx.x & 15), ld, a.M, stage_lds[threadIdx.x >> 6]); \
    else if (R.valid[u]) t_store(act, (base) + R.row[u] * (ld), kq);                                             \
  } while (0)
#endif
// Forward launches of at most LAT_MAX_TILES 64-row tiles (one per CU): the latency form (mlp6_fwd_kernel<1, NP, 4 + LAT_LOADERS>, linear6_fwd_kernel<NP, true>).
constexpr long LAT_MAX_TILES = 256;
constexpr int LAT_LOADERS = 2;                     // loader waves of the latency form (1 / 2 / 4 measured: DESIGN section 9 f4)
// NWV = 5 / 6 / 8 ("latency form", small launches): four compute waves on a 64-row tile + 1 / 2 / 4 loader waves that stream the packed weights
// through a ring of three 48 KB slots (mlp6_device.h: lat_loader / gemm6_lat).  Same per-row arithmetic, bit for bit.
template <int NS, int NP, int NWV = WG / 64>
__global__ __launch_bounds__(64 * NWV, NWV != WG / 64 ? 1 : (NS == 1 ? 3 : 2)) void mlp6_fwd_kernel(const hgn_mlp_fwd_t a) {
  constexpr bool LATF = NWV == 5 || NWV == 6 || NWV == 8;      // 1, 2 or 4 loader waves (LAT is the latent width)
  constexpr int NL = LATF ? NWV - 4 : 1;
  constexpr bool BIG = NWV != WG / 64 && !LATF;
  constexpr int GROUPS = BIG ? NWV / 4 : NS;          // 64-row sub-tiles of the workgroup's tile
  static_assert(!BIG || (NS == 1 && NWV % 4 == 0), "big workgroups: one 16-row sub-tile per wave, whole 64-row groups");
  static_assert(!LATF || NS == 1, "latency form: one 16-row sub-tile per compute wave");
  constexpr int LDS_BF16 = LATF ? 3 * HALF_BF16 : BIG ? (BLOCK_BF16 > GROUPS * SEG_LDS_FLOATS * 2 ? BLOCK_BF16 : GROUPS * SEG_LDS_FLOATS * 2) : HALF_BF16;
  __shared__ __attribute__((aligned(16))) __bf16 lds[LDS_BF16];
  __shared__ int seg_ids_lds[GROUPS][SEG_PRE_INTS];   // see SegPre
  static_assert(HALF_BF16 * 2 >= SEG_LDS_FLOATS * 4, "the weight stage doubles as the segment-sum tile");
  constexpr bool STAGED = NS == 2 && !BIG;            // 2 workgroups / CU: 48 KB weight stage + 16 KB store stage each
  __shared__ __attribute__((aligned(16))) float stage_lds[STAGED ? WG / 64 : 1][STAGED ? 1024 : 4];
  const int kq = (threadIdx.x & 63) >> 4;
  if constexpr (LATF) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4) {       // a loader wave: the blocks in the order of the code below
      int si = 0, k0 = 0, tail = 0;
      lat_loader<NP, NL>(lds, (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) - 4u, [&]() -> const __bf16* {
        if (si < a.n_src) {
          const __bf16* p = reinterpret_cast<const __bf16*>(a.src[si].Wpk) + (long)(k0 >> 7) * BLOCK_BF16;
          k0 += 128;
          if (k0 >= a.src[si].K) { ++si; k0 = 0; }
          return p;
        }
        ++tail;
        if (tail == 1) return reinterpret_cast<const __bf16*>(a.W2pk);
        if (tail == 2) return reinterpret_cast<const __bf16*>(a.W3pk);
        return tail - 3 < a.n_post ? reinterpret_cast<const __bf16*>(a.post_pk[tail - 3]) : nullptr;
      });
      return;
    }
  }
  const Rows<NS, LATF ? WG / 64 : NWV> R(a.M);
  const int grp = BIG ? (int)(threadIdx.x >> 8) : 0, ltid = BIG ? (int)(threadIdx.x & 255) : (int)threadIdx.x;
  int ring_slot = 0;
  auto block = [&](Act (&acc_)[NS], Act (&b_)[NS], const __bf16* pk_, auto&& between_, auto&& post_) {
    if constexpr (LATF) gemm6_lat<NP>(acc_, b_, lds, ring_slot, pk_, between_, post_);
    else if constexpr (BIG) gemm6_big<NS, NP, NWV>(acc_, b_, lds, pk_, between_, post_);
    else gemm6<NS, NP>(acc_, b_, lds, pk_, between_, post_);
  };
  auto nothing = [](Act (&)[NS]) {};
  HGN_STAMP();                                      // kernel entered
  // Every INDEX the kernel gathers through is loaded here, ahead of everything: the dependent row loads below then cost one
  // memory round trip instead of two (measured on 1.19 M edge rows: the first block's loads took 12.6 of the workgroup's 75 us
  // when each `add` source was index load -> wait -> row loads -> wait -> add, one after the other).
  int add_row[NS][HGN_MAX_ADD];
#pragma unroll
  for (int u = 0; u < NS; ++u)
#pragma unroll
    for (int i = 0; i < HGN_MAX_ADD; ++i) add_row[u][i] = i < a.n_add ? a.add[i].idx[R.rc[u]] : 0;
  SegPre seg_pre[NS];
  if (a.seg_out) {
    if constexpr (BIG) seg_pre[0].load(a.seg_ids, R.tile_row0 + grp * TILE_ROWS, a.M, ltid);
    else {
#pragma unroll
      for (int u = 0; u < NS; ++u) seg_pre[u].load(a.seg_ids, R.tile_row0 + u * TILE_ROWS, a.M);
    }
  }

  Act acc[NS], b[NS];
  bool first = true;
  for (int si = 0; si < a.n_src; ++si) {
    const hgn_src_t s = a.src[si];
    const __bf16* pk = reinterpret_cast<const __bf16*>(s.Wpk);
    const bool vec = ((s.ld & 3) == 0) && ((s.K & 3) == 0) && ((reinterpret_cast<uintptr_t>(s.x) & 15) == 0);
    for (int k0 = 0; k0 < s.K; k0 += 128) {
      const int kw = min(128, s.K - k0);
      block(acc, b, pk + (long)(k0 >> 7) * BLOCK_BF16, [&] {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          const long srow = s.idx ? (long)s.idx[R.rc[u]] : R.rc[u];
          const float* xr = s.x + srow * s.ld + k0;
          // narrow / unaligned sources (encoder inputs): zero-extended to the 128-wide block whose pack is zero padded
          if (HGN_ABL & 8) t_zero(b[u]);
          else if (vec) { if (kw == 128) t_load(b[u], xr, kq); else t_load_w(b[u], xr, kq, kw); } else t_load_masked(b[u], xr, kq, kw);
          if (first) {                              // acc = b1 + P0[row] + P1[row]: all loads issued before the first add
            if (a.n_add == 0) t_load(acc[u], a.b1, kq);
            else {
              static_assert(HGN_MAX_ADD == 2, "two gathered pre-projections at most");
              Act p1;
              t_load(acc[u], a.add[0].P + (long)add_row[u][0] * a.add[0].ld, kq);
              if (a.n_add > 1) t_load(p1, a.add[1].P + (long)add_row[u][1] * a.add[1].ld, kq);
              __builtin_amdgcn_sched_barrier(0);    // (the row loads above are in flight before the bias loads and the adds)
              HGN_FOR_B(fb) acc[u].v[fb] = *reinterpret_cast<const f32x4*>(a.b1 + 16 * fb + 4 * kq) + acc[u].v[fb];
              if (a.n_add > 1) HGN_FOR_B(fb) acc[u].v[fb] += p1.v[fb];
            }
            if (a.seg_out) { if constexpr (BIG) seg_pre[0].stash(seg_ids_lds[grp], ltid); else seg_pre[u].stash(seg_ids_lds[u]); }
          }
        }
        first = false;
      }, nothing);
    }
  }
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    unsigned m1 = 0;
    if (a.relu_bits) m1 = relu_with_bits(acc[u]); else relu_int(acc[u]);
    if (!(HGN_ABL & 4) && a.z1) HGN_TSTORE(acc[u], a.z1, LAT, u);
    if (!(HGN_ABL & 4) && a.relu_bits && R.valid[u]) a.relu_bits[R.row[u] * 8 + kq] = m1;
  }
  block(b, acc, reinterpret_cast<const __bf16*>(a.W2pk), [&] {
#pragma unroll
    for (int u = 0; u < NS; ++u) t_load(b[u], a.b2, kq);
  }, nothing);
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    unsigned m2 = 0;
    if (a.relu_bits) m2 = relu_with_bits(b[u]); else relu_int(b[u]);
    if (!(HGN_ABL & 4) && a.z2) HGN_TSTORE(b[u], a.z2, LAT, u);
    if (!(HGN_ABL & 4) && a.relu_bits && R.valid[u]) a.relu_bits[R.row[u] * 8 + 4 + kq] = m2;
  }
  block(acc, b, reinterpret_cast<const __bf16*>(a.W3pk), [&] {
#pragma unroll
    for (int u = 0; u < NS; ++u) t_load(acc[u], a.b3, kq);
  }, [&](Act (&free_b)[NS]) {                       // the residual rows arrive while the last block multiplies
    if (a.res) {
#pragma unroll
      for (int u = 0; u < NS; ++u) t_load(free_b[u], a.res + R.rc[u] * a.ld_res, kq);
    }
  });
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    if (a.ln_g) {
      const float mean = row_sum(acc[u]) * (1.f / LAT);
      HGN_FOR_B(fb) acc[u].v[fb] -= mean;
      const float var = row_sum_sq(acc[u]) * (1.f / LAT);
      const float rstd = 1.f / sqrtf(var + 1e-5f);
      HGN_FOR_B(fb) acc[u].v[fb] *= rstd;
      if (!(HGN_ABL & 4) && a.xhat) HGN_TSTORE(acc[u], a.xhat, LAT, u);
      if (a.rstd && R.valid[u] && kq == 0) a.rstd[R.row[u]] = rstd;
      HGN_FOR_B(fb) {
        const f32x4 gm = *reinterpret_cast<const f32x4*>(a.ln_g + 16 * fb + 4 * kq);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(a.ln_b + 16 * fb + 4 * kq);
        acc[u].v[fb] = acc[u].v[fb] * gm + bt;
      }
    }
    if (a.res) HGN_FOR_B(fb) acc[u].v[fb] += b[u].v[fb];
    HGN_TSTORE(acc[u], a.out, a.ld_out, u);
  }
  HGN_STAMP();                                      // epilogue stores issued
  if (a.seg_out) {                                  // (waits for none of the stores above: see SegPre)
    if constexpr (BIG) {                              // every 4-wave group walks its own 64-row sub-tile, side by side
      tile_segment_sum(acc[0], reinterpret_cast<float*>(lds) + grp * SEG_LDS_FLOATS, a.seg_ids, a.seg_out, a.ld_seg_out,
                       R.tile_row0 + grp * TILE_ROWS, a.M, seg_ids_lds[grp], ltid);
    } else {
#pragma unroll
      for (int u = 0; u < NS; ++u)
        if (R.tile_row0 + u * TILE_ROWS < a.M)          // uniform over the workgroup
          tile_segment_sum(acc[u], reinterpret_cast<float*>(lds), a.seg_ids, a.seg_out, a.ld_seg_out, R.tile_row0 + u * TILE_ROWS, a.M,
                           seg_ids_lds[u]);
    }
  }
  HGN_STAMP();                                      // segment sums done
  // ---- post-projection blocks (hgn_mlp_fwd_t.post_*): the output rows are the NEXT edge block's node operand -- its pre-projection
  // P = [h W1s^T | h W1r^T] (and the zero fill of its aggregate buffer) while the rows are still in registers: the same products in the
  // same order as hgn_linear_fwd6 on the stored rows (bit-identical), without reading them back and without the launch.
  if constexpr (!BIG) {
    if (a.n_post > 0) {
      for (int pb = 0; pb < a.n_post; ++pb) {
        block(b, acc, reinterpret_cast<const __bf16*>(a.post_pk[pb]), [&] {
#pragma unroll
          for (int u = 0; u < NS; ++u) t_zero(b[u]);
        }, nothing);
#pragma unroll
        for (int u = 0; u < NS; ++u)
          if (R.valid[u]) t_store(b[u], a.post_out + R.row[u] * a.ld_post + 128 * pb, kq);
      }
      if (a.post_zero) {
#pragma unroll
        for (int u = 0; u < NS; ++u)
          if (R.valid[u]) { t_zero(b[u]); t_store(b[u], a.post_zero + R.row[u] * a.ld_post_zero, kq); }
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------------
// The TRAINING EDGE BLOCK as a kernel of its own (graphnet.py:22-32 on >= FWD128_MIN_ROWS receiver-sorted edge rows: 15 of these
// launches are a quarter of a training step).  Same tiles (128 rows, two 16-row sub-tiles per wave), same products in the same order
// and the same row sums as mlp6_fwd_kernel<2, NP> -- bit-identical results -- with everything the general kernel decides at run time
// decided at launch (launch_mlp6_fwd checks): ONE 128-wide ungathered source, two gathered pre-projections, LayerNorm, residual, all
// saves present, every row array 128 floats wide and below 4 GiB.  What that buys is the instruction stream: row addresses are one
// 32-bit VGPR next to a scalar base (no 64-bit vector arithmetic, no per-source loop, no width / alignment cases), the stores of all
// tiles but the launch's last carry no per-row test, ReLU + sign word cost three instructions per value instead of four and a half.
// ----------------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(WG, 2) void mlp6_fwd_edge_kernel(const hgn_mlp_fwd_t a) {
  constexpr int NS = 2;
  __shared__ __attribute__((aligned(16))) __bf16 lds[HALF_BF16];
  __shared__ int seg_ids_lds[NS][SEG_PRE_INTS];
  __shared__ __attribute__((aligned(16))) float stage_lds[WG / 64][1024];
  const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned M = (unsigned)a.M;
  const unsigned tile_row0 = (unsigned)xcd_tile() * (NS * TILE_ROWS);
  const bool full = tile_row0 + NS * TILE_ROWS <= M;                 // uniform: every tile but the launch's last
  unsigned row0[NS];                                                  // the wave's first row of sub-tile u (uniform)
#pragma unroll
  for (int u = 0; u < NS; ++u) row0[u] = tile_row0 + u * TILE_ROWS + wave * WAVE_ROWS;
  // Per-lane row numbers are RE-DERIVED from an opaque lane id wherever they are needed (three instructions) instead of living in
  // registers from kernel entry to the epilogue: at this kernel's budget every long-lived per-lane value is a spill, and a spill's
  // reload is a vector-memory operation in the middle of the counted waits.
  auto lane_now = [&] { return opaque_u(threadIdx.x & 63); };
  auto row_clamped = [&](int u, unsigned ln) { const unsigned r = row0[u] + (ln & 15u); return r < M ? r : M - 1; };
  auto row_valid = [&](int u, unsigned ln) { return row0[u] + (ln & 15u) < M; };
  unsigned rc[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) rc[u] = row_clamped(u, lane);
  // gather indices first: the dependent row loads then cost one memory round trip, not two
  int add_row[NS][2];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    add_row[u][0] = a.add[0].idx[rc[u]];
    add_row[u][1] = a.add[1].idx[rc[u]];
  }
  SegPre seg_pre[NS];
  if (a.seg_out) {
#pragma unroll
    for (int u = 0; u < NS; ++u) seg_pre[u].load(a.seg_ids, (long)tile_row0 + u * TILE_ROWS, a.M);
  }
  float* st = stage_lds[wave];
  auto nothing = [](int, Act (&)[NS]) {};
  Act acc[NS], b[NS];
  const __bf16* pk1 = reinterpret_cast<const __bf16*>(a.src[0].Wpk);
  const __bf16* pk2 = reinterpret_cast<const __bf16*>(a.W2pk);
  const __bf16* pk3 = reinterpret_cast<const __bf16*>(a.W3pk);
  gemm6q<NS, NP, false, 0, 0, 0>(acc, b, lds, pk1, pk2, true, [&] {
    // acc = (b1 + P0[snd]) + P1[rcv] per sub-tile.  Both e tiles (HBM) go out first, then the gathered pre-projection rows (cache
    // resident) one sub-tile at a time: with all six tiles in flight plus the bias vectors the allocator spills, and scratch traffic
    // in this kernel would sit in the same in-order queue as every store.
    Act p1;
    t_load32(b[0], a.src[0].x, rc[0] * 512u + 16u * kq);
    t_load32(b[1], a.src[0].x, rc[1] * 512u + 16u * kq);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      t_load(acc[u], a.add[0].P + (long)add_row[u][0] * a.add[0].ld, kq);
      t_load(p1, a.add[1].P + (long)add_row[u][1] * a.add[1].ld, kq);
      __builtin_amdgcn_sched_barrier(0);
      HGN_FOR_B(fb) acc[u].v[fb] = *reinterpret_cast<const f32x4*>(a.b1 + 16 * fb + 4 * kq) + acc[u].v[fb];
      HGN_FOR_B(fb) acc[u].v[fb] += p1.v[fb];
      if (u + 1 < NS) {
        // the next sub-tile's gather addresses "depend" on these sums: its loads cannot be issued (into yet another 64 registers)
        // before this sub-tile's temporaries are dead -- scheduling barriers alone do not hold the adds in place
        int i0 = add_row[u + 1][0], i1 = add_row[u + 1][1];
        asm volatile("" : "+v"(i0), "+v"(i1)
                     : "v"(acc[u].v[0][0]), "v"(acc[u].v[0][2]), "v"(acc[u].v[1][0]), "v"(acc[u].v[1][2]), "v"(acc[u].v[2][0]), "v"(acc[u].v[2][2]),
                       "v"(acc[u].v[3][0]), "v"(acc[u].v[3][2]), "v"(acc[u].v[4][0]), "v"(acc[u].v[4][2]), "v"(acc[u].v[5][0]), "v"(acc[u].v[5][2]),
                       "v"(acc[u].v[6][0]), "v"(acc[u].v[6][2]), "v"(acc[u].v[7][0]), "v"(acc[u].v[7][2]));
        add_row[u + 1][0] = i0; add_row[u + 1][1] = i1;
      }
    }
    if (a.seg_out) {
      seg_pre[0].stash(seg_ids_lds[0]);
      seg_pre[1].stash(seg_ids_lds[1]);
    }
  }, [&](int q, Act (&free_b)[NS]) {                // the next block's accumulators start from its bias: loaded during the last piece
    if (q == 3) {                                   // (three quarters of the operand vectors are dead by then), into the registers of
#pragma unroll                                      // the operand rows this block has split: nothing is loaded between two blocks
      for (int u = 0; u < NS; ++u) t_load(free_b[u], a.b2, kq);
    }
  });
  // Between two blocks: ReLU, sign words, the saved activation rows -- and the wait for the next block's first piece, whose DMA was
  // issued BEFORE these stores.  In a full tile their number is static (per sub-tile 8 row stores + 1 sign-word store): the wait
  // leaves them in flight.  Path by path (no join in between: the compiler's own bookkeeping takes the smaller count at a join).
  constexpr int SAVE_OPS = NS * (NB + 1);
  auto save = [&](auto full_, Act (&z)[NS], Act (&next_acc)[NS], float* __restrict__ dst, unsigned word) {
    constexpr bool FULL = decltype(full_)::value;
    const unsigned ln = lane_now();
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const unsigned m = relu_with_bits(z[u]);
      t_store_rows32<FULL>(z[u], dst, row0[u], M, st);
      if (FULL || row_valid(u, ln)) a.relu_bits[row_clamped(u, ln) * 8u + word + (ln >> 4)] = m;
    }
    gemm6q_landed<NS, FULL ? SAVE_OPS : 0>(next_acc, z);
  };
  if (full) save(std::true_type{}, acc, b, a.z1, 0u); else save(std::false_type{}, acc, b, a.z1, 0u);
  gemm6q<NS, NP, true, 0, 0, 0>(b, acc, lds, pk2, pk3, false, [] {}, [&](int q, Act (&free_a)[NS]) {
    if (q == 3) {
#pragma unroll
      for (int u = 0; u < NS; ++u) t_load(free_a[u], a.b3, kq);
    }
  });
  if (full) save(std::true_type{}, b, acc, a.z2, 4u); else save(std::false_type{}, b, acc, a.z2, 4u);
  gemm6q<NS, NP, true, 0, 0, NB>(acc, b, lds, pk3, nullptr, false, [] {
  }, [&](int q, Act (&free_b)[NS]) {                // the residual rows arrive while pieces 2 and 3 multiply, in the registers of the
    if (q == 2) { const unsigned ln = lane_now(); t_load32(free_b[0], a.res, row_clamped(0, ln) * 512u + 16u * (ln >> 4)); }      // operand vectors that
    if (q == 3) { const unsigned ln = lane_now(); t_load32(free_b[1], a.res, row_clamped(1, ln) * 512u + 16u * (ln >> 4)); }      // pieces 0 and 1 consumed
  });
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const float mean = row_sum(acc[u]) * (1.f / LAT);
    HGN_FOR_B(fb) acc[u].v[fb] -= mean;
    const float var = row_sum_sq(acc[u]) * (1.f / LAT);
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    HGN_FOR_B(fb) acc[u].v[fb] *= rstd;
    if (full) t_store_rows32<true>(acc[u], a.xhat, row0[u], M, st); else t_store_rows32<false>(acc[u], a.xhat, row0[u], M, st);
    { const unsigned ln = lane_now(); if (row_valid(u, ln) && (ln >> 4) == 0) a.rstd[row_clamped(u, ln)] = rstd; }
    HGN_FOR_B(fb) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.ln_g + 16 * fb + 4 * kq);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(a.ln_b + 16 * fb + 4 * kq);
      acc[u].v[fb] = acc[u].v[fb] * gm + bt;
    }
    HGN_FOR_B(fb) acc[u].v[fb] += b[u].v[fb];
    if (full) t_store_rows32<true>(acc[u], a.out, row0[u], M, st); else t_store_rows32<false>(acc[u], a.out, row0[u], M, st);
  }
  if (a.seg_out) {
#pragma unroll
    for (int u = 0; u < NS; ++u)
      if (tile_row0 + u * TILE_ROWS < M)              // uniform over the workgroup
        tile_segment_sum(acc[u], reinterpret_cast<float*>(lds), a.seg_ids, a.seg_out, a.ld_seg_out, (long)tile_row0 + u * TILE_ROWS, a.M,
                         seg_ids_lds[u]);
  }
}

// ----------------------------------------------------------------------------------------------------------
// Column-split latency form (inference launches of at most 16 x LAT_MAX_TILES rows: the node MLP of a rollout step is 25 tiles of
// 64 rows -- 25 CUs busy, each compute wave running the whole 3-layer chain of its 16 rows one product after the other).  Here a
// workgroup takes 16 rows and its four compute waves SHARE them: wave w owns output blocks 2 w, 2 w + 1 (32 of the 128 features)
// of every layer -- a quarter of the products per wave, four times the workgroups.  In the operand layout of the packed
// weights contraction block c of a layer's input is exactly what lane (n, kq) holds of output blocks 2 c, 2 c + 1 of the
// layer before, so wave w PRODUCES the split operand vectors of contraction block w from its own registers, the four waves
// exchange them through 12 KB of LDS at the barrier that also opens the next weight half (lat_loader), and every wave reads
// all four.  LayerNorm needs whole rows in the summation order of the other kernels: the pre-LayerNorm tile is gathered
// through LDS and every wave normalises it (redundantly), then stores its own 32 columns.  Same products in the same order
// per accumulator, same row sums: bit-identical to the other forms, the saved activations of a training launch included (each wave
// stores its own 32 columns of z1 / z2 / xhat and its byte of the sign words).  No in-kernel segment sums: launches that need them
// take mlp6_fwd_kernel<1, NP, 4 + LAT_LOADERS>.
template <int NP>
__device__ __forceinline__ void cs_split8(const f32x4& v0, const f32x4& v1, bf16x8 (&o)[3], float sc = 1.f) {
  const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  if constexpr (NP == 3) {
    hgn_split::eight16(v, sc, o);
  } else if constexpr (NP == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[0][j] = __builtin_bit_cast(__bf16, (_Float16)v[j]);
  } else {
    hgn_split::eight(v, o);
  }
}

constexpr int CS_LOADERS = 4;
template <int NP>
__global__ __launch_bounds__(64 * (4 + CS_LOADERS), 1) void mlp6_fwd_cs_kernel(const hgn_mlp_fwd_t a) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * HALF_BF16];
  __shared__ __attribute__((aligned(16))) bf16x8 xch[4][3][64];         // [contraction block][split][lane]
  static_assert(sizeof(bf16x8) * 4 * 3 * 64 >= 16 * 132 * 4, "the pre-LayerNorm tile reuses the exchange buffer");
  float* tile = reinterpret_cast<float*>(&xch[0][0][0]);                 // row stride 132 floats; free after the last block's second barrier
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= 4) {                                    // loader waves: the blocks in the order of the code below
    int si = 0, k0 = 0, tail = 0, main_halves = 4;
    for (int i = 0; i < a.n_src; ++i) main_halves += 2 * ((a.src[i].K + 127) >> 7);
    lat_loader<NP, CS_LOADERS>(lds, (unsigned)wave - 4u, [&]() -> const __bf16* {
      if (si < a.n_src) {
        const __bf16* p = reinterpret_cast<const __bf16*>(a.src[si].Wpk) + (long)(k0 >> 7) * BLOCK_BF16;
        k0 += 128;
        if (k0 >= a.src[si].K) { ++si; k0 = 0; }
        return p;
      }
      ++tail;
      if (tail == 1) return reinterpret_cast<const __bf16*>(a.W2pk);
      if (tail == 2) return reinterpret_cast<const __bf16*>(a.W3pk);
      return tail - 3 < a.n_post ? reinterpret_cast<const __bf16*>(a.post_pk[tail - 3]) : nullptr;
    }, a.n_post > 0 ? main_halves : -1, Prod<NP>::SCALED ? 1 : 0, Prod<NP>::SCALED ? 3 : 2);
    return;
  }
  __shared__ float rmax[8][16];                       // scaled mode: the four waves' shares of the row maxima of a block's operand
  const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  const long row = (long)blockIdx.x * 16 + n;
  const bool valid = row < a.M;
  const long rc = valid ? row : a.M - 1;
  const int col0 = 16 * (2 * wave) + 4 * kq, col1 = col0 + 16;          // this lane's two 4-float chunks of a 128-wide row
  int add_row[HGN_MAX_ADD];
#pragma unroll
  for (int i = 0; i < HGN_MAX_ADD; ++i) add_row[i] = i < a.n_add ? a.add[i].idx[rc] : 0;
  auto chunk = [](const float* p) { return *reinterpret_cast<const f32x4*>(p); };

  f32x4 acc[2];
  bf16x8 xs[3][4];
  int slot = 0;
  constexpr int NSP = Prod<NP>::NSPLIT;
  // -> the row's scale exponent (mode 3: whole-row maximum over the four waves' shares, as split_np computes it in the other forms:
  // one barrier of the compute waves' own, which the loaders join -- lat_loader: xbar)
  // `c0`, `c1`, `sw`: what the accumulators hold when the block of this input starts, and the (first) block's exponent -- the row's
  // exponent is capped so that those contents survive the products' scale (split_bf16.h: SCALE_EASY / acc_room; the second
  // maximum travels with the first, no barrier of its own)
  auto max8 = [](const f32x4& v0, const f32x4& v1) -> float {
    return fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))),
                 fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
  };
  auto produce = [&](const f32x4& v0, const f32x4& v1, const f32x4& c0, const f32x4& c1, int sw) -> int {
    bf16x8 o[3];
    int e = 0;
    float sc = 1.f;
    if constexpr (Prod<NP>::SCALED) {
      const float m = rows4_max(max8(v0, v1)), am = rows4_max(max8(c0, c1));
      if (kq == 0) { rmax[wave][n] = m; rmax[4 + wave][n] = am; }
      wg_barrier_lds();
      e = scale_exp_of(fmaxf(fmaxf(rmax[0][n], rmax[1][n]), fmaxf(rmax[2][n], rmax[3][n])));
      if (e + sw > hgn_split::SCALE_EASY)
        e = min(e, hgn_split::acc_room(fmaxf(fmaxf(rmax[4][n], rmax[5][n]), fmaxf(rmax[6][n], rmax[7][n]))) - sw);
      sc = pow2f(e);
    }
    cs_split8<NP>(v0, v1, o, sc);
#pragma unroll
    for (int sp = 0; sp < NSP; ++sp) xch[wave][sp][lane] = o[sp];
    return e;
  };
  auto consume = [&] {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int sp = 0; sp < NSP; ++sp) xs[sp][c] = xch[c][sp][lane];
  };
  auto sweep = [&](auto HALF_) {                      // this wave's two output blocks against one weight half
    constexpr int HALF = decltype(HALF_)::value;
    const __bf16* lp = lds + slot * HALF_BF16 + lane * 8 + (2 * wave) * TILE_BF16;
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const int c = 2 * HALF + cl;
      bf16x8 fr[2][3];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int sp = 0; sp < NSP; ++sp) fr[k][sp] = *reinterpret_cast<const bf16x8*>(lp + ((sp * 2 + cl) * 8 + k) * TILE_BF16);
      f32x4 t0 = acc[0], t1 = acc[1];
      if constexpr (NP == 3) {                        // smallest terms first, the two accumulation chains interleaved
        t0 = mfma_f16(fr[0][1], xs[0][c], t0);
        t1 = mfma_f16(fr[1][1], xs[0][c], t1);
        t0 = mfma_f16(fr[0][0], xs[1][c], t0);
        t1 = mfma_f16(fr[1][0], xs[1][c], t1);
        t0 = mfma_f16(fr[0][0], xs[0][c], t0);
        t1 = mfma_f16(fr[1][0], xs[0][c], t1);
      } else if constexpr (NP != 6) {
        t0 = mfma_one<NP>(fr[0][0], xs[0][c], t0);
        t1 = mfma_one<NP>(fr[1][0], xs[0][c], t1);
      } else {
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][2], xs[0][c], t0, 0, 0, 0);      // smallest terms first
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][2], xs[0][c], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][0], xs[2][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][0], xs[2][c], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][1], xs[1][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][1], xs[1][c], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][1], xs[0][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][1], xs[0][c], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][0], xs[1][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][0], xs[1][c], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][0], xs[0][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][0], xs[0][c], t1, 0, 0, 0);
      }
      acc[0] = t0; acc[1] = t1;
    }
    slot = slot == 2 ? 0 : slot + 1;
  };
  // `T` (scaled mode): exponent of the operand rows' scale + the block's (acc is carried at 2^T through the block, true scale outside)
  auto block = [&](bool opened = false, int T = 0) {  // xch holds the operand vectors of this block's input
    if constexpr (Prod<NP>::SCALED) { scale4(acc[0], T); scale4(acc[1], T); }
    if (!opened) wg_barrier_lds();                    // ... visible to every wave; the block's first weight half has landed
    consume();
    sweep(std::integral_constant<int, 0>{});
    wg_barrier_lds();                                 // second half landed; every wave has read xch
    sweep(std::integral_constant<int, 1>{});
    if constexpr (Prod<NP>::SCALED) { scale4(acc[0], -T); scale4(acc[1], -T); }
  };
  auto sw_of = [&](const void* pk) -> int { return Prod<NP>::SCALED ? pack_scale_exp(reinterpret_cast<const __bf16*>(pk)) : 0; };

  bool first = true;
  for (int si = 0; si < a.n_src; ++si) {
    const hgn_src_t s = a.src[si];
    const bool vec = ((s.ld & 3) == 0) && ((s.K & 3) == 0) && ((reinterpret_cast<uintptr_t>(s.x) & 15) == 0);
    for (int k0 = 0; k0 < s.K; k0 += 128) {
      const int kw = min(128, s.K - k0);
      const long srow = s.idx ? (long)s.idx[rc] : rc;
      const float* xr = s.x + srow * s.ld + k0;
      f32x4 v[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {                   // zero beyond the source's width (the pack is zero padded there)
        const int col = k ? col1 : col0;
        if (vec) v[k] = col < kw ? chunk(xr + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        else {
#pragma unroll
          for (int u = 0; u < 4; ++u) v[k][u] = col + u < kw ? xr[col + u] : 0.f;
        }
      }
      if (first) {                                    // acc = b1 + P0[row] + P1[row], in the order of the other kernels
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int col = k ? col1 : col0;
          if (a.n_add == 0) acc[k] = chunk(a.b1 + col);
          else {
            f32x4 t = chunk(a.add[0].P + (long)add_row[0] * a.add[0].ld + col);
            t = chunk(a.b1 + col) + t;
            if (a.n_add > 1) t += chunk(a.add[1].P + (long)add_row[1] * a.add[1].ld + col);
            acc[k] = t;
          }
        }
        first = false;
      }
      const int sw = sw_of(reinterpret_cast<const __bf16*>(s.Wpk) + (long)(k0 >> 7) * BLOCK_BF16);
      const int e = produce(v[0], v[1], acc[0], acc[1], sw);
      block(false, e + sw);
    }
  }
  const float* bias[2] = {a.b2, a.b3};
  const void* wpk[2] = {a.W2pk, a.W3pk};
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    // relu as the other forms do it (mlp_common.h: relu_int / relu_with_bits: integer maximum of the bit pattern and 0); training
    // launches save z_{l+1} and its sign bits: wave w holds output blocks 2 w, 2 w + 1 of the row = byte w of the lane group's sign
    // word (bit 4 fb + u of word kq <-> unit 16 fb + 4 kq + u), written as a byte of its own
    unsigned sign = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int z = max(__float_as_int(acc[k][u]), 0);
        acc[k][u] = __int_as_float(z);
        sign |= (z != 0 ? 1u : 0u) << (4 * k + u);
      }
    if (valid) {
      float* zs = l == 0 ? a.z1 : a.z2;
      if (zs) {
        *reinterpret_cast<f32x4*>(zs + row * LAT + col0) = acc[0];
        *reinterpret_cast<f32x4*>(zs + row * LAT + col1) = acc[1];
      }
      if (a.relu_bits) reinterpret_cast<unsigned char*>(a.relu_bits + row * 8 + 4 * l + kq)[wave] = (unsigned char)sign;
    }
    const f32x4 b0 = chunk(bias[l] + col0), b1 = chunk(bias[l] + col1);
    const int sw = sw_of(wpk[l]);
    const int e = produce(acc[0], acc[1], b0, b1, sw);
    acc[0] = b0;
    acc[1] = b1;
    block(false, e + sw);
  }
  // ---- the whole pre-LayerNorm tile to every wave (row sums in the order of the other kernels) -----------------------------
  *reinterpret_cast<f32x4*>(tile + n * 132 + col0) = acc[0];
  *reinterpret_cast<f32x4*>(tile + n * 132 + col1) = acc[1];
  wg_barrier_lds();                                   // (no post blocks: the loader waves have left; with them: the barrier that opens the first post half)
  Act y;
  HGN_FOR_B(fb) y.v[fb] = *reinterpret_cast<const f32x4*>(tile + n * 132 + 16 * fb + 4 * kq);
  if (a.ln_g) {
    const float mean = row_sum(y) * (1.f / LAT);
    HGN_FOR_B(fb) y.v[fb] -= mean;
    const float var = row_sum_sq(y) * (1.f / LAT);
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    HGN_FOR_B(fb) y.v[fb] *= rstd;
    if (a.rstd && valid && wave == 0 && kq == 0) a.rstd[row] = rstd;
  }
  f32x4 o2[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int fb = 2 * wave + k, col = k ? col1 : col0;
    f32x4 o = y.v[0];
    HGN_FOR_B(q) if (q == fb) o = y.v[q];             // (wave-uniform select: y is a register array)
    if (a.ln_g && a.xhat && valid) *reinterpret_cast<f32x4*>(a.xhat + row * LAT + col) = o;     // training: the normalised row
    if (a.ln_g) o = o * chunk(a.ln_g + col) + chunk(a.ln_b + col);
    if (a.res) o += chunk(a.res + rc * a.ld_res + col);
    if (valid) *reinterpret_cast<f32x4*>(a.out + row * a.ld_out + col) = o;
    o2[k] = o;
  }
  // ---- post-projection blocks: the output rows are the next edge block's node operand (hgn_mlp_fwd_t.post_*) -------------------
  // The barrier in front of the tile gather was the one that opens the first post half (the loader waves are still streaming);
  // two more, which the loaders join (lat_loader: extra_after), separate the tile reads from the operand vectors that replace
  // the tile in LDS, and those from their readers.
  if (a.n_post > 0) {
    wg_barrier_lds();                                 // every wave has read the tile
    const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const int e_post = produce(o2[0], o2[1], z4, z4, 0);      // (scaled mode: one more barrier inside, lat_loader: extra_n = 3; the blocks run from zero)
    wg_barrier_lds();                                 // the operand vectors of the output rows are visible
    for (int pb = 0; pb < a.n_post; ++pb) {
      acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      block(pb == 0, e_post + sw_of(a.post_pk[pb]));  // (xs is re-read per block: 12 LDS reads against 48 products)
      if (valid) {
        *reinterpret_cast<f32x4*>(a.post_out + row * a.ld_post + 128 * pb + col0) = acc[0];
        *reinterpret_cast<f32x4*>(a.post_out + row * a.ld_post + 128 * pb + col1) = acc[1];
      }
    }
    if (a.post_zero && valid) {
      *reinterpret_cast<f32x4*>(a.post_zero + row * a.ld_post_zero + col0) = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(a.post_zero + row * a.ld_post_zero + col1) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// single Linear over packed 128-wide blocks (node pre-projection of the split edge layer)
struct Lin6Args { const float* x; long ldx; long M; const __bf16* pk[4]; int n_blocks; float* out; long ld_out;
                  float* zero; long ld_zero;        // zero (optional): rows [0, M) x 128 floats set to 0 in the same pass
                  int accumulate; };                // backward only: out += ... (the accumulators start from the rows of `out`)

// LATF: the latency form (see mlp6_fwd_kernel<1, NP, 6>): 4 compute waves + LAT_LOADERS loader waves, ring of three 48 KB slots
template <int NP, bool LATF = false>
__global__ __launch_bounds__(LATF ? 64 * (4 + LAT_LOADERS) : WG, LATF ? 1 : 3) void linear6_fwd_kernel(const Lin6Args a) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[LATF ? 3 * HALF_BF16 : HALF_BF16];
  const int kq = (threadIdx.x & 63) >> 4;
  if constexpr (LATF) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4) {
      int blk = 0;
      lat_loader<NP, LAT_LOADERS>(lds, (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) - 4u,
                                  [&]() -> const __bf16* { return blk < a.n_blocks ? a.pk[blk++] : nullptr; });
      return;
    }
  }
  const Rows<1> R(a.M);
  Act acc[1], b[1];
  int ring_slot = 0;
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    auto between = [&] {
      if (blk == 0) t_load(b[0], a.x + R.rc[0] * a.ldx, kq);
      t_zero(acc[0]);
    };
    if constexpr (LATF) gemm6_lat<NP>(acc, b, lds, ring_slot, a.pk[blk], between, [](Act (&)[1]) {});
    else gemm6<1, NP>(acc, b, lds, a.pk[blk], between);
    if (R.valid[0]) t_store(acc[0], a.out + R.row[0] * a.ld_out + 128 * blk, kq);
  }
  if (a.zero && R.valid[0]) { t_zero(acc[0]); t_store(acc[0], a.zero + R.row[0] * a.ld_zero, kq); }
}

// ----------------------------------------------------------------------------------------------------------
// backward (data gradients): same contract as mlp_bwd_kernel; weights come as TRANSPOSED-form packs.  Eligibility
// guarantees LayerNorm, its workspace and the ReLU sign words: straight-line code without optional parts.
// ----------------------------------------------------------------------------------------------------------
// NWV = 4 + LAT_LOADERS ("latency form", launches of at most one 64-row tile per CU -- the node update's backward of a one-graph
// step is 25 tiles): four compute waves + loader waves that stream the packed transposed blocks through a ring of three 48 KB
// slots in the order the chain multiplies them (mlp6_device.h: lat_loader / gemm6_lat), as in the forward's latency form: a
// workgroup that has its CU to itself otherwise waits ~2 us for each of its ten weight halves.  Same arithmetic, bit for bit.
template <int NS, int NP, bool PARK, int NWV = WG / 64>
__global__ __launch_bounds__(64 * NWV, NWV != WG / 64 ? 1 : (NS == 1 ? 3 : 2)) void mlp6_bwd_kernel(const hgn_mlp_bwd_t a) {
  constexpr bool LATF = NWV != WG / 64;
  static_assert(!LATF || (NS == 1 && !PARK), "latency form: one 16-row sub-tile per compute wave");
  __shared__ __attribute__((aligned(16))) float ldsf[(LATF ? 3 * HALF_BF16 : HALF_BF16) / 2 + (WG / 64) * 256];
  static_assert(HALF_BF16 / 2 >= SEG_LDS_FLOATS, "the weight stage doubles as the segment-sum tile");
  __bf16* lds = reinterpret_cast<__bf16*>(ldsf);
  float* lnl = ldsf + (LATF ? 3 * HALF_BF16 : HALF_BF16) / 2;
  if constexpr (LATF) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4) {       // a loader wave: the blocks in the order of the code below
      int stage = 0, di = 0, k0 = 0;
      lat_loader<NP, NWV - 4>(lds, (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) - 4u, [&]() -> const __bf16* {
        if (stage == 0) { ++stage; return reinterpret_cast<const __bf16*>(a.W3pk_t); }
        if (stage == 1) { ++stage; return reinterpret_cast<const __bf16*>(a.W2pk_t); }
        if (di >= a.n_dx) return nullptr;
        const __bf16* p = reinterpret_cast<const __bf16*>(a.dx[di].Wpk_t) + (long)(k0 >> 7) * BLOCK_BF16;
        k0 += 128;
        if (k0 >= a.dx[di].K) { ++di; k0 = 0; }
        return p;
      });
      return;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const Rows<NS, WG / 64> R(a.M);
  int ring_slot = 0;
  auto block = [&](Act (&acc_)[NS], Act (&b_)[NS], const __bf16* pk_, auto&& between_) {
    if constexpr (LATF) gemm6_lat<NP>(acc_, b_, lds, ring_slot, pk_, between_, [](Act (&)[NS]) {});
    else gemm6<NS, NP>(acc_, b_, lds, pk_, between_);
  };

  Act g[NS], t[NS];
  // PARK (several aggregation ops, i.e. pna: d_out_eff costs ~3.5 KB of gathered reads per row -- four d(agg) slots, two
  // arg-index rows): it is parked in the residual dx buffer once instead of being gathered a second time for the skip
  // connection.  A separate instantiation, so that the single-op kernel keeps its register allocation.
  float* park = nullptr; long park_ld = 0;
  if (PARK)
    for (int di = 0; di < a.n_dx; ++di)
      if (a.dx[di].residual) { park = a.dx[di].dx; park_ld = a.dx[di].ld; }
  unsigned mb1[NS], mb2[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) { mb1[u] = a.relu_bits[R.rc[u] * 8 + kq]; mb2[u] = a.relu_bits[R.rc[u] * 8 + 4 + kq]; }
  int pre_seg[NS];                                  // receiver of the row (aggregation backward): loaded with the sign words
#pragma unroll
  for (int u = 0; u < NS; ++u) pre_seg[u] = a.agg_dout ? a.agg_seg[R.rc[u]] : -1;
  // ---- dz3 (LayerNorm backward, computed while the first half of W3 is in flight), dz2 = relu'(z2) * (W3^T dz3) -------
  block(t, g, reinterpret_cast<const __bf16*>(a.W3pk_t), [&] {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      Act& xh = t[u];
      if (HGN_ABL & 8) t_zero(g[u]); else load_dout<false>(g[u], a, R.rc[u], kq, pre_seg[u]);
      if (PARK && R.valid[u]) t_store(g[u], park + R.row[u] * park_ld, kq);
      if (HGN_ABL & 8) t_zero(xh); else t_load(xh, a.xhat + R.rc[u] * LAT, kq);
      HGN_FOR_B(fb) {                               // LayerNorm-affine gradient partials of this wave's rows
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float pb = R.valid[u] ? g[u].v[fb][w] : 0.f;
          float pg = row16_sum(pb * xh.v[fb][w]);
          pb = row16_sum(pb);
          if (n == 0) {
            float* dst = lnl + wave * 256 + 16 * fb + 4 * kq + w;
            if (u == 0) { dst[0] = pg; dst[128] = pb; } else { dst[0] += pg; dst[128] += pb; }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      HGN_FOR_B(fb) g[u].v[fb] *= *reinterpret_cast<const f32x4*>(a.ln_g + 16 * fb + 4 * kq);
      const float m1 = row_sum(g[u]) * (1.f / LAT);
      const float m2 = row_dot(g[u], xh) * (1.f / LAT);
      const float r = a.rstd[R.rc[u]];
      HGN_FOR_B(fb) g[u].v[fb] = r * (g[u].v[fb] - m1 - xh.v[fb] * m2);
      if (!(HGN_ABL & 4) && a.dz3 && R.valid[u]) t_store(g[u], a.dz3 + R.row[u] * LAT, kq);
      t_zero(t[u]);
    }
  });
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    relu_mask_bits(t[u], mb2[u]);
    if (!(HGN_ABL & 4) && a.dz2 && R.valid[u]) t_store(t[u], a.dz2 + R.row[u] * LAT, kq);
  }
  // ---- dz1 = relu'(z1) * (W2^T dz2) ----------------------------------------------------------------------
  block(g, t, reinterpret_cast<const __bf16*>(a.W2pk_t), [&] {
#pragma unroll
    for (int u = 0; u < NS; ++u) t_zero(g[u]);
  });
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    relu_mask_bits(g[u], mb1[u]);
    if (!(HGN_ABL & 4) && a.dz1 && R.valid[u]) t_store(g[u], a.dz1 + R.row[u] * LAT, kq);
  }
  // receiver sums of dz1 while the tile is still in registers (the stage buffer is free between two blocks; the next block's
  // opening barrier orders the reads below before its weight DMA)
  if (a.seg_dz1) {
#pragma unroll
    for (int u = 0; u < NS; ++u)
      if (R.tile_row0 + u * TILE_ROWS < a.M)
        tile_segment_sum(g[u], ldsf, a.seg_ids, a.seg_dz1, a.ld_seg_dz1, R.tile_row0 + u * TILE_ROWS, a.M);
  }
  // ---- dx_src = dz1 * W1[:, cols]  (+ d_out_eff for the residual source) -----------------------------------
  for (int di = 0; di < a.n_dx; ++di) {
    const hgn_dx_t d = a.dx[di];
    const __bf16* pk = reinterpret_cast<const __bf16*>(d.Wpk_t);
    for (int k0 = 0; k0 < d.K; k0 += 128) {
      block(t, g, pk + (long)(k0 >> 7) * BLOCK_BF16, [&] {
#pragma unroll
        for (int u = 0; u < NS; ++u) t_zero(t[u]);
      });
#pragma unroll
      for (int u = 0; u < NS; ++u)
        if (R.valid[u]) {
          if (d.residual) { if (PARK) t_add(t[u], park + R.row[u] * park_ld, kq); else load_dout<true>(t[u], a, R.rc[u], kq, pre_seg[u]); }
          t_store(t[u], d.dx + R.row[u] * d.ld + k0, kq);
        }
    }
  }
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int w = 0; w < WG / 64; ++w) sum += lnl[w * 256 + threadIdx.x];
  a.ln_ws[(long)blockIdx.x * 256 + threadIdx.x] = sum;
  if (blockIdx.x == 0 && threadIdx.x == 0) { reinterpret_cast<unsigned*>(a.ln_ws)[-256] = 0u; reinterpret_cast<unsigned*>(a.ln_ws)[-255] = gridDim.x; }        // ticket of ln_reduce_kernel (csrc/mlp.hip)
}

// Column-split latency form of the backward (training launches of at most 16 x LAT_MAX_TILES rows WITHOUT a folded aggregation backward
// or in-kernel segment sums: the node updates of a one-graph step, 1 600 rows -- 100 workgroups instead of 25).  The mirror of
// mlp6_fwd_cs_kernel: 16 rows per workgroup, wave w owns columns 32 w .. 32 w + 31 of every layer's gradient, the operand vectors go
// through the same 12 KB exchange buffer.  The LayerNorm backward needs whole rows in the summation order of the other kernels:
// every wave loads the 16 full rows of d_out and x-hat in the row-per-wave layout and computes dz3 redundantly (wave 0 also forms
// the workgroup's LayerNorm-affine partials: one slab per 16 rows).  Same products in the same order per accumulator, same row
// sums: the data gradients are bit-identical to the other forms.
template <int NP>
__global__ __launch_bounds__(64 * (4 + CS_LOADERS), 1) void mlp6_bwd_cs_kernel(const hgn_mlp_bwd_t a) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * HALF_BF16];
  __shared__ __attribute__((aligned(16))) bf16x8 xch[4][3][64];         // [contraction block][split][lane]
  __shared__ float rmax[4][16];
  __shared__ float lnl[256];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= 4) {                                    // loader waves: W3^T, W2^T, then the W1^T blocks of the requested sources
    int stage = 0, di = 0, k0 = 0;
    // (scaled mode: one exchange barrier in front of the blocks of halves 0, 2, 4; the further dx blocks share the operand of the first)
    lat_loader<NP, CS_LOADERS>(lds, (unsigned)wave - 4u, [&]() -> const __bf16* {
      if (stage == 0) { ++stage; return reinterpret_cast<const __bf16*>(a.W3pk_t); }
      if (stage == 1) { ++stage; return reinterpret_cast<const __bf16*>(a.W2pk_t); }
      if (di >= a.n_dx) return nullptr;
      const __bf16* p = reinterpret_cast<const __bf16*>(a.dx[di].Wpk_t) + (long)(k0 >> 7) * BLOCK_BF16;
      k0 += 128;
      if (k0 >= a.dx[di].K) { ++di; k0 = 0; }
      return p;
    }, Prod<NP>::SCALED ? 6 : -1, Prod<NP>::SCALED ? 1 : 0, 0);
    return;
  }
  const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  const long row = (long)blockIdx.x * 16 + n;
  const bool valid = row < a.M;
  const long rc = valid ? row : a.M - 1;
  const int col0 = 16 * (2 * wave) + 4 * kq, col1 = col0 + 16;
  auto chunk = [](const float* p) { return *reinterpret_cast<const f32x4*>(p); };
  constexpr int NSP = Prod<NP>::NSPLIT;
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- dz3: LayerNorm backward on whole rows (as mlp6_bwd_kernel) ------------------------------------------------------------------
  f32x4 acc[2];
  {
    Act g, xh;
    t_load(g, a.d_out + rc * a.ld_dout, kq);
    t_load(xh, a.xhat + rc * LAT, kq);
    if (wave == 0) {                                  // LayerNorm-affine gradient partials of the workgroup's 16 rows
      HGN_FOR_B(fb) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          float pb = valid ? g.v[fb][w] : 0.f;
          const float pg = row16_sum(pb * xh.v[fb][w]);
          pb = row16_sum(pb);
          if (n == 0) { lnl[16 * fb + 4 * kq + w] = pg; lnl[128 + 16 * fb + 4 * kq + w] = pb; }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    HGN_FOR_B(fb) g.v[fb] *= *reinterpret_cast<const f32x4*>(a.ln_g + 16 * fb + 4 * kq);
    const float m1 = row_sum(g) * (1.f / LAT);
    const float m2 = row_dot(g, xh) * (1.f / LAT);
    const float r = a.rstd[rc];
    HGN_FOR_B(fb) g.v[fb] = r * (g.v[fb] - m1 - xh.v[fb] * m2);
    acc[0] = g.v[0]; acc[1] = g.v[1];
    HGN_FOR_B(q) if (q == 2 * wave) acc[0] = g.v[q];  // (wave-uniform select: g is a register array)
    HGN_FOR_B(q) if (q == 2 * wave + 1) acc[1] = g.v[q];
  }
  if (a.dz3 && valid) {
    *reinterpret_cast<f32x4*>(a.dz3 + row * LAT + col0) = acc[0];
    *reinterpret_cast<f32x4*>(a.dz3 + row * LAT + col1) = acc[1];
  }

  bf16x8 xs[3][4];
  int slot = 0;
  auto max8 = [](const f32x4& v0, const f32x4& v1) -> float {
    return fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))),
                 fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
  };
  // this wave's 32 columns of a block's operand rows -> operand vectors of contraction block `wave` (every product below starts from
  // zero accumulators: no cap of the row exponent is needed)
  auto produce = [&](const f32x4& v0, const f32x4& v1) -> int {
    bf16x8 o[3];
    int e = 0;
    float sc = 1.f;
    if constexpr (Prod<NP>::SCALED) {
      const float m = rows4_max(max8(v0, v1));
      if (kq == 0) rmax[wave][n] = m;
      wg_barrier_lds();
      e = scale_exp_of(fmaxf(fmaxf(rmax[0][n], rmax[1][n]), fmaxf(rmax[2][n], rmax[3][n])));
      sc = pow2f(e);
    }
    cs_split8<NP>(v0, v1, o, sc);
#pragma unroll
    for (int sp = 0; sp < NSP; ++sp) xch[wave][sp][lane] = o[sp];
    return e;
  };
  auto sweep = [&](auto HALF_) {                      // (the sweep of mlp6_fwd_cs_kernel)
    constexpr int HALF = decltype(HALF_)::value;
    const __bf16* lp = lds + slot * HALF_BF16 + lane * 8 + (2 * wave) * TILE_BF16;
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const int c = 2 * HALF + cl;
      bf16x8 fr[2][3];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int sp = 0; sp < NSP; ++sp) fr[k][sp] = *reinterpret_cast<const bf16x8*>(lp + ((sp * 2 + cl) * 8 + k) * TILE_BF16);
      f32x4 t0 = acc[0], t1 = acc[1];
      if constexpr (NP == 3) {
        t0 = mfma_f16(fr[0][1], xs[0][c], t0);
        t1 = mfma_f16(fr[1][1], xs[0][c], t1);
        t0 = mfma_f16(fr[0][0], xs[1][c], t0);
        t1 = mfma_f16(fr[1][0], xs[1][c], t1);
        t0 = mfma_f16(fr[0][0], xs[0][c], t0);
        t1 = mfma_f16(fr[1][0], xs[0][c], t1);
      } else if constexpr (NP != 6) {
        t0 = mfma_one<NP>(fr[0][0], xs[0][c], t0);
        t1 = mfma_one<NP>(fr[1][0], xs[0][c], t1);
      } else {
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][2], xs[0][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][2], xs[0][c], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][0], xs[2][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][0], xs[2][c], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][1], xs[1][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][1], xs[1][c], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][1], xs[0][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][1], xs[0][c], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][0], xs[1][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][0], xs[1][c], t1, 0, 0, 0);
        t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][0], xs[0][c], t0, 0, 0, 0);
        t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][0], xs[0][c], t1, 0, 0, 0);
      }
      acc[0] = t0; acc[1] = t1;
    }
    slot = slot == 2 ? 0 : slot + 1;
  };
  // acc = W_block^T x (operand vectors in xch), from zero; T: exponent of the products' scale
  auto block = [&](int T) {
    acc[0] = z4; acc[1] = z4;
    wg_barrier_lds();                                 // the operand vectors are visible; the block's first weight half has landed
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int sp = 0; sp < NSP; ++sp) xs[sp][c] = xch[c][sp][lane];
    sweep(std::integral_constant<int, 0>{});
    wg_barrier_lds();                                 // second half landed; every wave has read xch
    sweep(std::integral_constant<int, 1>{});
    if constexpr (Prod<NP>::SCALED) { scale4(acc[0], -T); scale4(acc[1], -T); }
  };
  auto sw_of = [&](const void* pk) -> int { return Prod<NP>::SCALED ? pack_scale_exp(reinterpret_cast<const __bf16*>(pk)) : 0; };
  auto mask = [&](unsigned word) {                    // relu'(z): this wave's byte of the lane group's sign word (bit 4 k + u)
    const unsigned m = (word >> (8 * wave)) & 0xffu;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        acc[k][u] = __uint_as_float(__float_as_uint(acc[k][u]) & (unsigned)__builtin_amdgcn_sbfe((int)m, 4 * k + u, 1));
  };
  const unsigned mb1 = a.relu_bits[rc * 8 + kq], mb2 = a.relu_bits[rc * 8 + 4 + kq];

  // ---- dz2 = relu'(z2) * (W3^T dz3) ----------------------------------------------------------------------------------------------
  int e = produce(acc[0], acc[1]);
  block(e + sw_of(a.W3pk_t));
  mask(mb2);
  if (a.dz2 && valid) {
    *reinterpret_cast<f32x4*>(a.dz2 + row * LAT + col0) = acc[0];
    *reinterpret_cast<f32x4*>(a.dz2 + row * LAT + col1) = acc[1];
  }
  // ---- dz1 = relu'(z1) * (W2^T dz2) ----------------------------------------------------------------------------------------------
  e = produce(acc[0], acc[1]);
  block(e + sw_of(a.W2pk_t));
  mask(mb1);
  if (a.dz1 && valid) {
    *reinterpret_cast<f32x4*>(a.dz1 + row * LAT + col0) = acc[0];
    *reinterpret_cast<f32x4*>(a.dz1 + row * LAT + col1) = acc[1];
  }
  // ---- dx_src = dz1 W1[:, cols]  (+ d_out for the residual source) -------------------------------------------------------------------
  if (a.n_dx > 0) e = produce(acc[0], acc[1]);        // one operand for all of them
  for (int di = 0; di < a.n_dx; ++di) {
    const hgn_dx_t d = a.dx[di];
    const __bf16* pk = reinterpret_cast<const __bf16*>(d.Wpk_t);
    for (int k0 = 0; k0 < d.K; k0 += 128) {
      const __bf16* pkb = pk + (long)(k0 >> 7) * BLOCK_BF16;
      block(e + sw_of(pkb));
      if (valid) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int col = k ? col1 : col0;
          if (k0 + col < d.K) {
            f32x4 o = acc[k];
            if (d.residual) o += chunk(a.d_out + rc * a.ld_dout + col);
            *reinterpret_cast<f32x4*>(d.dx + row * d.ld + k0 + col) = o;
          }
        }
      }
    }
  }
  wg_barrier_lds();                                   // (the loader waves have left; wave 0's partials are in lnl since the first block)
  a.ln_ws[(long)blockIdx.x * 256 + threadIdx.x] = lnl[threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0) { reinterpret_cast<unsigned*>(a.ln_ws)[-256] = 0u; reinterpret_cast<unsigned*>(a.ln_ws)[-255] = gridDim.x; }
}

template <int NP, bool LATF = false>
__global__ __launch_bounds__(LATF ? 64 * (4 + LAT_LOADERS) : WG, LATF ? 1 : 3) void linear6_bwd_kernel(const Lin6Args a) {
  // here a.x = g [M, 128*n_blocks], a.out = dx [M,128]; packs are transposed-form.  LATF: the latency form (see linear6_fwd_kernel)
  __shared__ __attribute__((aligned(16))) __bf16 lds[LATF ? 3 * HALF_BF16 : HALF_BF16];
  const int kq = (threadIdx.x & 63) >> 4;
  if constexpr (LATF) {
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= 4) {
      int blk = 0;
      lat_loader<NP, LAT_LOADERS>(lds, (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) - 4u,
                                  [&]() -> const __bf16* { return blk < a.n_blocks ? a.pk[blk++] : nullptr; });
      return;
    }
  }
  const Rows<1> R(a.M);
  Act acc[1], b[1];
  int ring_slot = 0;
  if (a.accumulate) t_load(acc[0], a.out + R.rc[0] * a.ld_out, kq); else t_zero(acc[0]);
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    auto between = [&] { t_load(b[0], a.x + R.rc[0] * a.ldx + 128 * blk, kq); };
    if constexpr (LATF) gemm6_lat<NP>(acc, b, lds, ring_slot, a.pk[blk], between, [](Act (&)[1]) {});
    else gemm6<1, NP>(acc, b, lds, a.pk[blk], between);
  }
  if (R.valid[0]) t_store(acc[0], a.out + R.row[0] * a.ld_out, kq);
}


// Column-split latency form of the single-Linear backward (at most 16 x LAT_MAX_TILES rows: the pre-projection gradient of a one-graph
// step): 16 rows per workgroup, wave w owns columns 32 w .. 32 w + 31 of dx and of every 128-wide block of g; the accumulators run
// over the blocks exactly as in linear6_bwd_kernel (row scale capped by what they hold, same products in the same order): same bits.
template <int NP>
__global__ __launch_bounds__(64 * (4 + CS_LOADERS), 1) void linear6_bwd_cs_kernel(const Lin6Args a) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * HALF_BF16];
  __shared__ __attribute__((aligned(16))) bf16x8 xch[4][3][64];
  __shared__ float rmax[8][16];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= 4) {
    int blk = 0;
    lat_loader<NP, CS_LOADERS>(lds, (unsigned)wave - 4u, [&]() -> const __bf16* { return blk < a.n_blocks ? a.pk[blk++] : nullptr; }, -1,
                               Prod<NP>::SCALED ? 1 : 0);
    return;
  }
  const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  const long row = (long)blockIdx.x * 16 + n;
  const bool valid = row < a.M;
  const long rc = valid ? row : a.M - 1;
  const int col0 = 16 * (2 * wave) + 4 * kq, col1 = col0 + 16;
  constexpr int NSP = Prod<NP>::NSPLIT;
  auto chunk = [](const float* p) { return *reinterpret_cast<const f32x4*>(p); };
  auto max8 = [](const f32x4& v0, const f32x4& v1) -> float {
    return fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))),
                 fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
  };
  f32x4 acc[2];
  acc[0] = a.accumulate ? chunk(a.out + rc * a.ld_out + col0) : f32x4{0.f, 0.f, 0.f, 0.f};
  acc[1] = a.accumulate ? chunk(a.out + rc * a.ld_out + col1) : f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xs[3][4];
  int slot = 0;
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    const f32x4 v0 = chunk(a.x + rc * a.ldx + 128 * blk + col0), v1 = chunk(a.x + rc * a.ldx + 128 * blk + col1);
    int T = 0;
    {                                                 // operand vectors of contraction block `wave`; the row's exponent (see mlp6_fwd_cs_kernel: produce)
      bf16x8 o[3];
      float sc = 1.f;
      if constexpr (Prod<NP>::SCALED) {
        const int sw = pack_scale_exp(a.pk[blk]);
        const float m = rows4_max(max8(v0, v1)), am = rows4_max(max8(acc[0], acc[1]));
        if (kq == 0) { rmax[wave][n] = m; rmax[4 + wave][n] = am; }
        wg_barrier_lds();
        int e = scale_exp_of(fmaxf(fmaxf(rmax[0][n], rmax[1][n]), fmaxf(rmax[2][n], rmax[3][n])));
        if (e + sw > hgn_split::SCALE_EASY)
          e = min(e, hgn_split::acc_room(fmaxf(fmaxf(rmax[4][n], rmax[5][n]), fmaxf(rmax[6][n], rmax[7][n]))) - sw);
        sc = pow2f(e);
        T = e + sw;
      }
      cs_split8<NP>(v0, v1, o, sc);
#pragma unroll
      for (int sp = 0; sp < NSP; ++sp) xch[wave][sp][lane] = o[sp];
    }
    if constexpr (Prod<NP>::SCALED) { scale4(acc[0], T); scale4(acc[1], T); }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      wg_barrier_lds();                               // this half has landed (first: the operand vectors are visible; second: every wave has read them)
      if (half == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int sp = 0; sp < NSP; ++sp) xs[sp][c] = xch[c][sp][lane];
      }
      const __bf16* lp = lds + slot * HALF_BF16 + lane * 8 + (2 * wave) * TILE_BF16;
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        bf16x8 fr[2][3];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int sp = 0; sp < NSP; ++sp) fr[k][sp] = *reinterpret_cast<const bf16x8*>(lp + ((sp * 2 + cl) * 8 + k) * TILE_BF16);
        f32x4 t0 = acc[0], t1 = acc[1];
#pragma unroll
        for (int hsel = 0; hsel < 2; ++hsel) {
          if (hsel != half) continue;                 // (xs is a register array: the contraction block is selected at compile time)
          const int c = 2 * hsel + cl;
          if constexpr (NP == 3) {                    // smallest terms first, the two accumulation chains interleaved
            t0 = mfma_f16(fr[0][1], xs[0][c], t0);
            t1 = mfma_f16(fr[1][1], xs[0][c], t1);
            t0 = mfma_f16(fr[0][0], xs[1][c], t0);
            t1 = mfma_f16(fr[1][0], xs[1][c], t1);
            t0 = mfma_f16(fr[0][0], xs[0][c], t0);
            t1 = mfma_f16(fr[1][0], xs[0][c], t1);
          } else if constexpr (NP != 6) {
            t0 = mfma_one<NP>(fr[0][0], xs[0][c], t0);
            t1 = mfma_one<NP>(fr[1][0], xs[0][c], t1);
          } else {
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][2], xs[0][c], t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][2], xs[0][c], t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][0], xs[2][c], t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][0], xs[2][c], t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][1], xs[1][c], t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][1], xs[1][c], t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][1], xs[0][c], t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][1], xs[0][c], t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][0], xs[1][c], t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][0], xs[1][c], t1, 0, 0, 0);
            t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[0][0], xs[0][c], t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[1][0], xs[0][c], t1, 0, 0, 0);
          }
        }
        acc[0] = t0; acc[1] = t1;
      }
      slot = slot == 2 ? 0 : slot + 1;
    }
    if constexpr (Prod<NP>::SCALED) { scale4(acc[0], -T); scale4(acc[1], -T); }
  }
  if (valid) {
    *reinterpret_cast<f32x4*>(a.out + row * a.ld_out + col0) = acc[0];
    *reinterpret_cast<f32x4*>(a.out + row * a.ld_out + col1) = acc[1];
  }
}
}  // namespace hgn

using namespace hgn;

#if HGN_ABL & 16
extern "C" int hgn_debug_mlp6_stamps(unsigned long long* host256, int* n) {
  (void)hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(host256, HIP_SYMBOL(hgn::g_hgn_stamps), 256 * 8) != hipSuccess) return HGN_E_LAUNCH;
  if (hipMemcpyFromSymbol(n, HIP_SYMBOL(hgn::g_hgn_stamp_n), 4) != hipSuccess) return HGN_E_LAUNCH;
  int zero = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(hgn::g_hgn_stamp_n), &zero, 4);
  return HGN_OK;
}
#endif

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int hgn_pack_bf16x3(const hgn_pack_t* blocks, int n, void* stream) {
  if (!blocks || n < 1 || n > HGN_MAX_PACK) return hgn_fail(HGN_E_INVALID, "hgn_pack_bf16x3: 1..HGN_MAX_PACK blocks per call");
  PackArgs a;
  for (int i = 0; i < n; ++i) {
    const hgn_pack_t& d = blocks[i];
    if (!d.W || !d.out || d.ldw < 1 || d.n_out < 1 || d.n_out > 128 || d.n_in < 1 || d.n_in > 128 || !aligned16(d.out))
      return hgn_fail(HGN_E_INVALID, "hgn_pack_bf16x3: bad block (at most 128 x 128, 16-byte aligned output)");
    a.d[i] = d;
  }
  bool scaled = false;
  for (int i = 0; i < n; ++i) scaled = scaled || (blocks[i].transposed & 4);
  if (scaled) hipLaunchKernelGGL(pack_scale_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(pack_bf16x3_kernel, dim3(2 * 2 * 8 * 64 * 8 / 256, n), dim3(256), 0, (hipStream_t)stream, a);
  return hgn_check_launch("hgn_pack_bf16x3");
}

extern "C" int hgn_pack_bf16x3_table(const hgn_pack_t* blocks, const hgn_pack_t* blocks_dev, int n, void* stream) {
  if (!blocks || !blocks_dev || n < 1 || n > 65535) return hgn_fail(HGN_E_INVALID, "hgn_pack_bf16x3_table: 1..65535 blocks, host copy and device table");
  for (int i = 0; i < n; ++i) {
    const hgn_pack_t& d = blocks[i];
    if (!d.W || !d.out || d.ldw < 1 || d.n_out < 1 || d.n_out > 128 || d.n_in < 1 || d.n_in > 128 || !aligned16(d.out))
      return hgn_fail(HGN_E_INVALID, "hgn_pack_bf16x3_table: bad block (at most 128 x 128, 16-byte aligned output)");
  }
  bool scaled = false;
  for (int i = 0; i < n; ++i) scaled = scaled || (blocks[i].transposed & 4);
  if (scaled) hipLaunchKernelGGL(pack_scale_table_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, blocks_dev);
  hipLaunchKernelGGL(pack_bf16x3_table_kernel, dim3(2 * 2 * 8 * 64 * 8 / 256, n), dim3(256), 0, (hipStream_t)stream, blocks_dev);
  return hgn_check_launch("hgn_pack_bf16x3_table");
}

// Eligibility of the split-bf16 forward: every source with a packed image (narrow sources: zero-padded blocks), 128-wide
// output, packed W2 / W3.  (The decoder, 3 outputs wide, keeps the fp32 kernel.)
extern "C" int hgn_mlp_fwd6_eligible(const hgn_mlp_fwd_t* a) {
  if (!a || a->out_w != 128 || !a->W2pk || !a->W3pk || a->n_src < 1) return 0;
  for (int i = 0; i < a->n_src; ++i) {
    const hgn_src_t& s = a->src[i];
    if (!s.Wpk || s.K < 1) return 0;
  }
  if ((a->ld_out & 3) || !aligned16(a->out) || (a->res && ((a->ld_res & 3) || !aligned16(a->res)))) return 0;
  return (a->flags & HGN_F_FP32_MFMA) ? 0 : 1;
}

namespace hgn {
// Forward launches of at least FWD128_MIN_ROWS rows: 128-row workgroups (two sub-tiles per wave, 2 workgroups per CU) -- half the
// weight DMA, LDS operand reads, waits and barriers per row; edge forward 1.156 -> 1.141 ms, whole step 66.2 -> 65.6 ms at 1.19 M
// rows, same bits.  (The backward is 4 % SLOWER that way, small launches lose workgroups: both keep 64-row tiles.)
constexpr long FWD128_MIN_ROWS = 192L * 256 * 2;
#if HGN_LAB
// ---- laboratory build only (tools/lab/build_lab.sh; never in the shipped library) --------------------------------------------
// HGN_TILE128: two sub-tiles per wave in ALL fused MLP launches (measured no faster: forward 1.18 vs 1.21 ms, backward 1.37 vs 1.34
// ms at 1.19 M rows).  HGN_DIAG_LDS_PAD=bytes: extra dynamic LDS per workgroup (2 or 1 workgroups per CU).  HGN_BIG_TILES /
// hgn_set_big_tiles: 12-wave workgroups on 192-row tiles (a third of the weight DMA; bit-identical, 1.241 vs 1.231 ms).
static bool tile128() { static const bool v = getenv("HGN_TILE128") != nullptr; return v; }
static unsigned lds_pad() { static const unsigned v = getenv("HGN_DIAG_LDS_PAD") ? (unsigned)atoi(getenv("HGN_DIAG_LDS_PAD")) : 0u; return v; }
static int g_big_tiles = getenv("HGN_BIG_TILES") ? 1 : 0;
static long big_min_rows() { static const long v = getenv("HGN_BIG_MIN_ROWS") ? atol(getenv("HGN_BIG_MIN_ROWS")) : FWD128_MIN_ROWS; return v; }
static bool tile128_fwd() { static const bool v = getenv("HGN_NO_TILE128_FWD") == nullptr; return v; }
static long lat_max_tiles() { static const long v = getenv("HGN_LAT_MAX_TILES") ? atol(getenv("HGN_LAT_MAX_TILES")) : LAT_MAX_TILES; return v; }
static bool cs_enabled() { static const bool v = getenv("HGN_NO_COLSPLIT") == nullptr; return v; }
#else
static constexpr long lat_max_tiles() { return LAT_MAX_TILES; }
static constexpr bool cs_enabled() { return true; }
static constexpr bool tile128() { return false; }
static constexpr unsigned lds_pad() { return 0u; }
static constexpr long big_min_rows() { return FWD128_MIN_ROWS; }
static constexpr bool tile128_fwd() { return true; }
#endif
}  // namespace hgn
#if HGN_LAB
extern "C" int hgn_set_big_tiles(int on) { hgn::g_big_tiles = on ? 1 : 0; return HGN_OK; }
#endif
namespace hgn {

// at most 16 rows per CU, no in-kernel segment sums: the column-split latency form (inference and, with the saves of a backward pass,
// training: a one-graph node update is 1 600 rows -- 100 workgroups of 16 rows instead of 25 of 64)
bool cs_eligible(const hgn_mlp_fwd_t* a) {
  if (!(a->M <= 16 * lat_max_tiles() && !a->seg_out && cs_enabled())) return false;
  if ((a->z1 && !aligned16(a->z1)) || (a->z2 && !aligned16(a->z2)) || (a->xhat && !aligned16(a->xhat))) return false;
  return true;
}

// the arguments of a training edge block as mlp6_fwd_edge_kernel assumes them (everything else: the general kernel)
static bool edge_block_shape(const hgn_mlp_fwd_t* a) {
  if ((a->flags & HGN_F_GENERAL_FWD) || a->n_src != 1 || a->n_add != 2 || a->n_post != 0) return false;
  const hgn_src_t& s = a->src[0];
  if (s.K != 128 || s.idx || s.ld != 128 || !aligned16(s.x)) return false;
  if (!a->z1 || !a->z2 || !a->xhat || !a->rstd || !a->relu_bits || !a->ln_g || !a->ln_b || !a->res) return false;
  if (a->ld_out != 128 || a->ld_res != 128 || a->out_w != 128) return false;
  if (!aligned16(a->z1) || !aligned16(a->z2) || !aligned16(a->xhat) || !aligned16(a->out) || !aligned16(a->res)) return false;
  for (int i = 0; i < 2; ++i)
    if (!a->add[i].P || !a->add[i].idx || (a->add[i].ld & 3) || !aligned16(a->add[i].P)) return false;
  return (a->M + 2 * TILE_ROWS) * 512 < ((int64_t)1 << 32);      // 32-bit byte offsets into the [M, 128] row arrays
}

int launch_mlp6_fwd(const hgn_mlp_fwd_t* a, void* stream) {
  const int np_ = matmul_products(a->products);
#if HGN_LAB
  if (g_big_tiles && !tile128() && a->M >= big_min_rows()) {
    const long tiles = (a->M + 191) / 192;
    if (np_ == 1) hipLaunchKernelGGL((mlp6_fwd_kernel<1, 1, 12>), dim3((unsigned)tiles), dim3(768), 0, (hipStream_t)stream, *a);
    else if (np_ == 2) hipLaunchKernelGGL((mlp6_fwd_kernel<1, 2, 12>), dim3((unsigned)tiles), dim3(768), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((mlp6_fwd_kernel<1, 6, 12>), dim3((unsigned)tiles), dim3(768), 0, (hipStream_t)stream, *a);
    return hgn_check_launch("hgn_mlp_fwd (split-bf16, 192-row tiles)");
  }
#endif
  if (cs_eligible(a)) {
    const long wgs = (a->M + 15) / 16;               // inference on at most 16 rows per CU: the column-split latency form
    constexpr int T = 64 * (4 + CS_LOADERS);
    if (np_ == 1) hipLaunchKernelGGL((mlp6_fwd_cs_kernel<1>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, *a);
    else if (np_ == 2) hipLaunchKernelGGL((mlp6_fwd_cs_kernel<2>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, *a);
    else if (np_ == 3) hipLaunchKernelGGL((mlp6_fwd_cs_kernel<3>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((mlp6_fwd_cs_kernel<6>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, *a);
    return hgn_check_launch("hgn_mlp_fwd (split-bf16, column-split latency form)");
  }
  if ((a->M + TILE_ROWS - 1) / TILE_ROWS <= lat_max_tiles()) {     // a tile per CU at most: nothing to hide a weight DMA behind but loader waves
    const long tiles = (a->M + TILE_ROWS - 1) / TILE_ROWS;
#if HGN_LAB
    static const int nl = getenv("HGN_LAT_LOADERS") ? atoi(getenv("HGN_LAT_LOADERS")) : LAT_LOADERS;
    if (np_ == 6 && nl == 1) { hipLaunchKernelGGL((mlp6_fwd_kernel<1, 6, 5>), dim3((unsigned)tiles), dim3(320), 0, (hipStream_t)stream, *a); return hgn_check_launch("hgn_mlp_fwd (latency form, 1 loader)"); }
    if (np_ == 6 && nl == 4) { hipLaunchKernelGGL((mlp6_fwd_kernel<1, 6, 8>), dim3((unsigned)tiles), dim3(512), 0, (hipStream_t)stream, *a); return hgn_check_launch("hgn_mlp_fwd (latency form, 4 loaders)"); }
#endif
    constexpr int T = 64 * (4 + LAT_LOADERS);
    if (np_ == 1) hipLaunchKernelGGL((mlp6_fwd_kernel<1, 1, 4 + LAT_LOADERS>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, *a);
    else if (np_ == 2) hipLaunchKernelGGL((mlp6_fwd_kernel<1, 2, 4 + LAT_LOADERS>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, *a);
    else if (np_ == 3) hipLaunchKernelGGL((mlp6_fwd_kernel<1, 3, 4 + LAT_LOADERS>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((mlp6_fwd_kernel<1, 6, 4 + LAT_LOADERS>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, *a);
    return hgn_check_launch("hgn_mlp_fwd (split-bf16, latency form)");
  }
  // 128-row tiles (two sub-tiles per wave, two workgroups per CU) pay for edge-shaped launches -- one source (+ gathered pre-projections) over
  // >= 98 304 rows --; a node update (several 128-wide sources, a sixth of the rows) is faster on 64-row tiles at three workgroups per CU
  // (mode 3, 204 800 rows: 0.190 -> 0.176 ms, tools/exp_tile64.py).  Same bits either way.
  const bool edge_like = a->n_src == 1 || a->n_add > 0;
  if ((tile128() || (tile128_fwd() && a->M >= big_min_rows() && !(a->flags & HGN_F_TILE64_FWD) && (edge_like || np_ == 6))) && (np_ == 6 || np_ == 3) &&
      a->M > TILE_ROWS) {
    const long tiles = (a->M + 2 * TILE_ROWS - 1) / (2 * TILE_ROWS);
    if (np_ == 3) {
      if (edge_block_shape(a)) hipLaunchKernelGGL((mlp6_fwd_edge_kernel<3>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
      else hipLaunchKernelGGL((mlp6_fwd_kernel<2, 3>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
    } else if (edge_block_shape(a)) hipLaunchKernelGGL((mlp6_fwd_edge_kernel<6>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((mlp6_fwd_kernel<2, 6>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
  } else {
    const long tiles = (a->M + TILE_ROWS - 1) / TILE_ROWS;
    if (np_ == 1) hipLaunchKernelGGL((mlp6_fwd_kernel<1, 1>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
    else if (np_ == 2) hipLaunchKernelGGL((mlp6_fwd_kernel<1, 2>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
    else if (np_ == 3) hipLaunchKernelGGL((mlp6_fwd_kernel<1, 3>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((mlp6_fwd_kernel<1, 6>), dim3((unsigned)tiles), dim3(WG), lds_pad(), (hipStream_t)stream, *a);
  }
  return hgn_check_launch("hgn_mlp_fwd (split-bf16)");
}
}  // namespace hgn

namespace hgn {
// Column-split latency form of the pre-projection (see mlp6_fwd_cs_kernel): 16 rows per workgroup, wave w produces the operand
// vectors of contraction block w of the input rows and owns output blocks 2 w, 2 w + 1 of every 128-wide output block.
template <int NP>
__global__ __launch_bounds__(64 * (4 + CS_LOADERS), 1) void linear6_fwd_cs_kernel(const Lin6Args a) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * HALF_BF16];
  __shared__ __attribute__((aligned(16))) bf16x8 xch[4][3][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= 4) {
    int blk = 0;
    lat_loader<NP, CS_LOADERS>(lds, (unsigned)wave - 4u, [&]() -> const __bf16* { return blk < a.n_blocks ? a.pk[blk++] : nullptr; }, -1,
                               Prod<NP>::SCALED ? 2 : 0);
    return;
  }
  __shared__ float rmax[4][16];
  const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
  const long row = (long)blockIdx.x * 16 + n;
  const bool valid = row < a.M;
  const long rc = valid ? row : a.M - 1;
  const int col0 = 16 * (2 * wave) + 4 * kq, col1 = col0 + 16;
  constexpr int NSP = Prod<NP>::NSPLIT;
  int e_row = 0;
  {
    const float* xr = a.x + rc * a.ldx;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(xr + col0), v1 = *reinterpret_cast<const f32x4*>(xr + col1);
    float sc = 1.f;
    if constexpr (Prod<NP>::SCALED) {                 // whole-row maximum over the four waves' shares (one barrier, joined by the loaders)
      float m = fmaxf(fmaxf(fmaxf(fabsf(v0[0]), fabsf(v0[1])), fmaxf(fabsf(v0[2]), fabsf(v0[3]))),
                      fmaxf(fmaxf(fabsf(v1[0]), fabsf(v1[1])), fmaxf(fabsf(v1[2]), fabsf(v1[3]))));
      m = rows4_max(m);
      if (kq == 0) rmax[wave][n] = m;
      wg_barrier_lds();
      e_row = scale_exp_of(fmaxf(fmaxf(rmax[0][n], rmax[1][n]), fmaxf(rmax[2][n], rmax[3][n])));
      sc = pow2f(e_row);
    }
    bf16x8 o[3];
    cs_split8<NP>(v0, v1, o, sc);
#pragma unroll
    for (int sp = 0; sp < NSP; ++sp) xch[wave][sp][lane] = o[sp];
  }
  bf16x8 xs[3][4];
  int slot = 0;
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      wg_barrier_lds();                               // this half has landed (and, the first time, the operand vectors are visible)
      if (blk == 0 && half == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int sp = 0; sp < NSP; ++sp) xs[sp][c] = xch[c][sp][lane];
      }
      const __bf16* lp = lds + slot * HALF_BF16 + lane * 8 + (2 * wave) * TILE_BF16;
#pragma unroll
      for (int cl = 0; cl < 2; ++cl) {
        bf16x8 fr[2][3];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int sp = 0; sp < NSP; ++sp) fr[k][sp] = *reinterpret_cast<const bf16x8*>(lp + ((sp * 2 + cl) * 8 + k) * TILE_BF16);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          f32x4 t = acc[k];
#pragma unroll
          for (int hsel = 0; hsel < 2; ++hsel) {
            if (hsel != half) continue;               // (xs is a register array: the contraction block is selected at compile time)
            const int c = 2 * hsel + cl;
            if constexpr (NP == 3) {
              t = mfma_f16(fr[k][1], xs[0][c], t);      // smallest terms first
              t = mfma_f16(fr[k][0], xs[1][c], t);
              t = mfma_f16(fr[k][0], xs[0][c], t);
            } else if constexpr (NP != 6) t = mfma_one<NP>(fr[k][0], xs[0][c], t);
            else {
              t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[k][2], xs[0][c], t, 0, 0, 0);      // smallest terms first
              t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[k][0], xs[2][c], t, 0, 0, 0);
              t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[k][1], xs[1][c], t, 0, 0, 0);
              t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[k][1], xs[0][c], t, 0, 0, 0);
              t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[k][0], xs[1][c], t, 0, 0, 0);
              t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[k][0], xs[0][c], t, 0, 0, 0);
            }
          }
          acc[k] = t;
        }
      }
      slot = slot == 2 ? 0 : slot + 1;
    }
    if constexpr (Prod<NP>::SCALED) {                 // (the products ran from zero: only the way back to the true scale)
      const int T = -(e_row + pack_scale_exp(a.pk[blk]));
      scale4(acc[0], T); scale4(acc[1], T);
    }
    if (valid) {
      *reinterpret_cast<f32x4*>(a.out + row * a.ld_out + 128 * blk + col0) = acc[0];
      *reinterpret_cast<f32x4*>(a.out + row * a.ld_out + 128 * blk + col1) = acc[1];
    }
  }
  if (a.zero && valid) {
    *reinterpret_cast<f32x4*>(a.zero + row * a.ld_zero + col0) = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(a.zero + row * a.ld_zero + col1) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
}  // namespace hgn

extern "C" int hgn_linear_fwd6z(const float* x, int64_t ldx, int64_t M, const void* const* pk_blocks, int nb, float* out,
                                int64_t ld_out, float* zero_rows, int64_t ld_zero, int products, void* stream);
extern "C" int hgn_linear_fwd6(const float* x, int64_t ldx, int64_t M, const void* const* pk_blocks, int nb, float* out,
                               int64_t ld_out, int products, void* stream) {
  return hgn_linear_fwd6z(x, ldx, M, pk_blocks, nb, out, ld_out, nullptr, 0, products, stream);
}
extern "C" int hgn_linear_fwd6z(const float* x, int64_t ldx, int64_t M, const void* const* pk_blocks, int nb, float* out,
                                int64_t ld_out, float* zero_rows, int64_t ld_zero, int products, void* stream) {
  if (!valid_products(products)) return hgn_fail(HGN_E_INVALID, "hgn_linear_fwd6: products must be 0 (default), 6, 3, 1 or 2");
  const int np_ = matmul_products(products);
  if (M == 0) return HGN_OK;
  if (!x || !pk_blocks || !out || M < 0 || nb < 1 || nb > 4 || (ldx & 3) || (ld_out & 3) || !aligned16(x) || !aligned16(out) ||
      (zero_rows && ((ld_zero & 3) || ld_zero < 128 || !aligned16(zero_rows))))
    return hgn_fail(HGN_E_INVALID, "hgn_linear_fwd6: bad argument");
  Lin6Args a;
  a.x = x; a.ldx = ldx; a.M = M; a.n_blocks = nb; a.out = out; a.ld_out = ld_out; a.zero = zero_rows; a.ld_zero = ld_zero; a.accumulate = 0;
  for (int i = 0; i < 4; ++i) a.pk[i] = i < nb ? reinterpret_cast<const __bf16*>(pk_blocks[i]) : nullptr;
  for (int i = 0; i < nb; ++i)
    if (!a.pk[i]) return hgn_fail(HGN_E_INVALID, "hgn_linear_fwd6: null packed block");
  const long tiles = (M + TILE_ROWS - 1) / TILE_ROWS;
  ProfScope ps(7, (double)M, (hipStream_t)stream);
  if (M <= 16 * hgn::lat_max_tiles() && hgn::cs_enabled()) {
    constexpr int T = 64 * (4 + hgn::CS_LOADERS);
    const long wgs = (M + 15) / 16;
    if (np_ == 1) hipLaunchKernelGGL((hgn::linear6_fwd_cs_kernel<1>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, a);
    else if (np_ == 2) hipLaunchKernelGGL((hgn::linear6_fwd_cs_kernel<2>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, a);
    else if (np_ == 3) hipLaunchKernelGGL((hgn::linear6_fwd_cs_kernel<3>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((hgn::linear6_fwd_cs_kernel<6>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, a);
    return hgn_check_launch("hgn_linear_fwd6 (column-split latency form)");
  }
  if (tiles <= hgn::lat_max_tiles()) {
    constexpr int T = 64 * (4 + hgn::LAT_LOADERS);
    if (np_ == 1) hipLaunchKernelGGL((linear6_fwd_kernel<1, true>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, a);
    else if (np_ == 2) hipLaunchKernelGGL((linear6_fwd_kernel<2, true>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, a);
    else if (np_ == 3) hipLaunchKernelGGL((linear6_fwd_kernel<3, true>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((linear6_fwd_kernel<6, true>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, a);
    return hgn_check_launch("hgn_linear_fwd6 (latency form)");
  }
  if (np_ == 1) hipLaunchKernelGGL(linear6_fwd_kernel<1>, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, a);
  else if (np_ == 2) hipLaunchKernelGGL(linear6_fwd_kernel<2>, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, a);
  else if (np_ == 3) hipLaunchKernelGGL(linear6_fwd_kernel<3>, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(linear6_fwd_kernel<6>, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, a);
  return hgn_check_launch("hgn_linear_fwd6");
}

extern "C" int hgn_mlp_bwd6_eligible(const hgn_mlp_bwd_t* a) {
  if (!a || a->out_w != 128 || !a->W3pk_t || !a->W2pk_t || !a->ln_g || !a->xhat || !a->rstd || !a->ln_ws || !a->relu_bits) return 0;
  for (int i = 0; i < a->n_dx; ++i) {
    const hgn_dx_t& d = a->dx[i];
    if (!d.Wpk_t || (d.K & 127) || (d.ld & 3) || !aligned16(d.dx)) return 0;
  }
  if (a->d_out && ((a->ld_dout & 3) || !aligned16(a->d_out))) return 0;
  return (a->flags & HGN_F_FP32_MFMA) ? 0 : 1;
}

namespace hgn {
// *n_slabs = number of 256-float LayerNorm-gradient partials written to a->ln_ws (one per workgroup)
// the column-split backward: small launches of the plain shape (d_out given, no folded aggregation backward, no in-kernel segment sums)
static bool cs_bwd_eligible(const hgn_mlp_bwd_t* a) {
  if (!(a->M <= 16 * lat_max_tiles() && cs_enabled()) || a->agg_dout || a->seg_dz1 || !a->d_out || !a->relu_bits) return false;
  if ((a->ld_dout & 3) || !aligned16(a->d_out) || !aligned16(a->xhat)) return false;
  if ((a->dz3 && !aligned16(a->dz3)) || (a->dz2 && !aligned16(a->dz2)) || (a->dz1 && !aligned16(a->dz1))) return false;
  for (int i = 0; i < a->n_dx; ++i)
    if (!a->dx[i].Wpk_t || (a->dx[i].K & 3) || (a->dx[i].ld & 3) || !aligned16(a->dx[i].dx)) return false;
  return true;
}

int launch_mlp6_bwd(const hgn_mlp_bwd_t* a, void* stream, long* n_slabs) {
  const int nb_ = bwd_products(a->products);
  if (tile128() && nb_ == 6 && a->M > TILE_ROWS) {
    const long tiles = (a->M + 2 * TILE_ROWS - 1) / (2 * TILE_ROWS);
    hipLaunchKernelGGL((mlp6_bwd_kernel<2, 6, false>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
    *n_slabs = tiles;
  } else {
    const long tiles = (a->M + TILE_ROWS - 1) / TILE_ROWS;
    bool park = false;                      // several aggregation ops feeding a residual source gradient (pna edge blocks)
    if (a->agg_dout && a->n_agg_ops > 1)
      for (int i = 0; i < a->n_dx; ++i) park = park || a->dx[i].residual;
    if (cs_bwd_eligible(a)) {                                   // at most 16 rows per CU: the column-split latency form
      const long wgs = (a->M + 15) / 16;
      constexpr int T = 64 * (4 + CS_LOADERS);
      if (nb_ == 1) hipLaunchKernelGGL((mlp6_bwd_cs_kernel<1>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, *a);
      else if (nb_ == 3) hipLaunchKernelGGL((mlp6_bwd_cs_kernel<3>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, *a);
      else hipLaunchKernelGGL((mlp6_bwd_cs_kernel<6>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, *a);
      *n_slabs = wgs;
      return hgn_check_launch("hgn_mlp_bwd (split products, column-split latency form)");
    }
    if (tiles <= lat_max_tiles() && !park && !a->seg_dz1) {      // a tile per CU at most: the latency form (loader waves + LDS weight ring)
      constexpr int T = 64 * (4 + LAT_LOADERS);
      if (nb_ == 1) hipLaunchKernelGGL((mlp6_bwd_kernel<1, 1, false, 4 + LAT_LOADERS>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, *a);
      else if (nb_ == 3) hipLaunchKernelGGL((mlp6_bwd_kernel<1, 3, false, 4 + LAT_LOADERS>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, *a);
      else hipLaunchKernelGGL((mlp6_bwd_kernel<1, 6, false, 4 + LAT_LOADERS>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, *a);
      *n_slabs = tiles;
      return hgn_check_launch("hgn_mlp_bwd (split products, latency form)");
    }
    if (nb_ == 1) hipLaunchKernelGGL((mlp6_bwd_kernel<1, 1, false>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
    else if (nb_ == 3 && park) hipLaunchKernelGGL((mlp6_bwd_kernel<1, 3, true>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
    else if (nb_ == 3) hipLaunchKernelGGL((mlp6_bwd_kernel<1, 3, false>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
    else if (park) hipLaunchKernelGGL((mlp6_bwd_kernel<1, 6, true>), dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL((mlp6_bwd_kernel<1, 6, false>), dim3((unsigned)tiles), dim3(WG), lds_pad(), (hipStream_t)stream, *a);
    *n_slabs = tiles;
  }
  return hgn_check_launch("hgn_mlp_bwd (split-bf16)");
}
}  // namespace hgn

extern "C" int hgn_linear_bwd6a(const float* g, int64_t ldg, int64_t M, const void* const* pk_blocks, int nb, float* dx,
                                int64_t ld_dx, int accumulate, int products, void* stream);
extern "C" int hgn_linear_bwd6(const float* g, int64_t ldg, int64_t M, const void* const* pk_blocks, int nb, float* dx,
                               int64_t ld_dx, int products, void* stream) {
  return hgn_linear_bwd6a(g, ldg, M, pk_blocks, nb, dx, ld_dx, 0, products, stream);
}
extern "C" int hgn_linear_bwd6a(const float* g, int64_t ldg, int64_t M, const void* const* pk_blocks, int nb, float* dx,
                                int64_t ld_dx, int accumulate, int products, void* stream) {
  if (!valid_products(products)) return hgn_fail(HGN_E_INVALID, "hgn_linear_bwd6: products must be 0 (default), 6, 3, 1 or 2");
  const int nb_ = bwd_products(products);
  if (M == 0) return HGN_OK;
  if (!g || !pk_blocks || !dx || M < 0 || nb < 1 || nb > 4 || (ldg & 3) || (ld_dx & 3) || !aligned16(g) || !aligned16(dx))
    return hgn_fail(HGN_E_INVALID, "hgn_linear_bwd6: bad argument");
  Lin6Args a;
  a.x = g; a.ldx = ldg; a.M = M; a.n_blocks = nb; a.out = dx; a.ld_out = ld_dx; a.zero = nullptr; a.ld_zero = 0; a.accumulate = accumulate ? 1 : 0;
  for (int i = 0; i < 4; ++i) a.pk[i] = i < nb ? reinterpret_cast<const __bf16*>(pk_blocks[i]) : nullptr;
  for (int i = 0; i < nb; ++i)
    if (!a.pk[i]) return hgn_fail(HGN_E_INVALID, "hgn_linear_bwd6: null packed block");
  const long tiles = (M + TILE_ROWS - 1) / TILE_ROWS;
  ProfScope ps(8, (double)M, (hipStream_t)stream);
  if (M <= 16 * hgn::lat_max_tiles() && hgn::cs_enabled()) {        // at most 16 rows per CU: the column-split latency form
    const long wgs = (M + 15) / 16;
    constexpr int T = 64 * (4 + hgn::CS_LOADERS);
    if (nb_ == 1) hipLaunchKernelGGL((linear6_bwd_cs_kernel<1>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, a);
    else if (nb_ == 3) hipLaunchKernelGGL((linear6_bwd_cs_kernel<3>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((linear6_bwd_cs_kernel<6>), dim3((unsigned)wgs), dim3(T), 0, (hipStream_t)stream, a);
    return hgn_check_launch("hgn_linear_bwd6 (column-split latency form)");
  }
  if (tiles <= hgn::lat_max_tiles()) {
    constexpr int T = 64 * (4 + hgn::LAT_LOADERS);
    if (nb_ == 1) hipLaunchKernelGGL((linear6_bwd_kernel<1, true>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, a);
    else if (nb_ == 3) hipLaunchKernelGGL((linear6_bwd_kernel<3, true>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((linear6_bwd_kernel<6, true>), dim3((unsigned)tiles), dim3(T), 0, (hipStream_t)stream, a);
    return hgn_check_launch("hgn_linear_bwd6 (latency form)");
  }
  if (nb_ == 1) hipLaunchKernelGGL(linear6_bwd_kernel<1>, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, a);
  else if (nb_ == 3) hipLaunchKernelGGL(linear6_bwd_kernel<3>, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(linear6_bwd_kernel<6>, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, a);
  return hgn_check_launch("hgn_linear_bwd6");
}
