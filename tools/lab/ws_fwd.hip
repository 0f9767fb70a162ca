// Edge-block forward, WEIGHT-STATIONARY: the three 128 x 128 layers of the edge MLP stay in REGISTERS for the whole launch.
//
// Why.  mlp6_fwd_kernel (csrc/mlp6.hip) stages every layer's packed weights through LDS once per 64-row tile: 288 KB of
// L2 -> LDS traffic per tile, six DMA waits and twelve workgroup barriers, and per wave a strictly serial chain
// load -> wait -> DMA -> wait -> products -> ... whose waits the three resident workgroups hide only in part (measured, 1.19 M
// rows: alone a workgroup needs 27 us per tile, three together 51 us each; profiles/r02_edge_kernel_ablation.log).
// Here the ROLES of the two MFMA operands' homes are swapped: wave w of a persistent 8-wave workgroup owns output units
// [16 w, 16 w + 16) of all three layers and keeps their split-bf16 weight fragments (3 layers x 3 splits x 4 contraction blocks
// x 16 bytes = 144 registers per lane) from kernel entry on; what goes through LDS is the ACTIVATION image of the 64-row tile
// (3 splits x 64 rows x 128 features, 48 KB; written once per layer by the waves that produced it, read by all eight).
// No weight traffic at all in the tile loop, one barrier per layer, and every wave does the same thing.
//
// Per tile:   e rows (prefetched one tile ahead, 16 registers per lane) -> split -> image X          | barrier
//             layer 1 products (image X) ; + b1 + P_s[snd] + P_r[rcv] ; relu ; z1, sign bits ; split -> image Y   | barrier
//             layer 2 products (image Y) ; + b2 ; relu ; z2, sign bits ; split -> image X            | barrier
//             layer 3 products (image X) ; + b3 ; LayerNorm (row sums across the 8 waves through LDS: 2 barriers) ;
//             x-hat, rstd ; affine ; + residual ; e' ; receiver sums of e' (tile_segment_walk: 2 barriers)
// Results: the products of every output are accumulated in the order of mlp6_fwd_kernel (per 32-wide contraction block: lo x
// hi, hi x lo, mid x mid, mid x hi, hi x mid, hi x hi); the first layer adds bias and gathered pre-projections AFTER the products
// (there: before), so the two kernels agree to fp32 rounding, not bit for bit.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include "hgn_device.h"
#include "hgn_host.h"
#include "mlp_common.h"
#include "mlp6_device.h"

namespace hgn {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int WS_T = 512;                                  // 8 waves: 2 per SIMD, 256 registers each
constexpr int IMG_ROWSTRIDE = TILE_ROWS + 1;               // vectors per (split, block, quarter) line: + 1 spreads the producers' writes
constexpr int IMG_VECS = 3 * 4 * 4 * IMG_ROWSTRIDE;        // bf16x8 vectors of one activation image
constexpr int IMG_BYTES = IMG_VECS * 16;                   // 49 920
constexpr int WS_SEG_OFF = 2 * IMG_BYTES;                  // the 64 x 132 float tile of tile_segment_walk (+ its hand-over line)
constexpr int WS_RED_OFF = WS_SEG_OFF + SEG_LDS_FLOATS * 4;      // 2 x [64 rows][8 waves] LayerNorm partials
constexpr int WS_BITS_OFF = WS_RED_OFF + 2 * 64 * 8 * 4;         // [2 layers][64 rows][4] ReLU sign words under construction
constexpr int WS_IDS_OFF = WS_BITS_OFF + 2 * 64 * 4 * 4;         // SegPre ids
constexpr int WS_IDS_STRIDE = (SEG_PRE_INTS + 3) / 4 * 4;         // ints; two buffers (tile parity): a tile's walk may still read its ids
constexpr int WS_GID_OFF = WS_IDS_OFF + 2 * WS_IDS_STRIDE * 4;   // while the next tile's are being parked
constexpr int WS_LDS = WS_GID_OFF + 2 * TILE_ROWS * 4;           // gather rows (snd, rcv) of the tile's 64 rows
static_assert(WS_LDS <= 160 * 1024, "LDS budget of one CU");

__device__ __forceinline__ int img_vec(int s, int c, int q, int row) { return ((s * 4 + c) * 4 + q) * IMG_ROWSTRIDE + row; }

__device__ __forceinline__ void ws_split8(const f32x4& lo4, const f32x4& hi4, bf16x8 (&s)[3]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = j < 4 ? lo4[j] : hi4[j - 4];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    s[0][j] = h; s[1][j] = m; s[2][j] = (__bf16)r2;
  }
}
__device__ __forceinline__ void ws_split4(const f32x4& x, bf16x4 (&s)[3]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float v = x[j];
    const __bf16 h = (__bf16)v;
    const float r1 = v - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    s[0][j] = h; s[1][j] = m; s[2][j] = (__bf16)r2;
  }
}

__device__ __forceinline__ unsigned ws_opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

// acc[r] (+)= W_layer[16 w .. +16, :] x image rows 16 (RT0 + r) .. +16, r < NR.  `img_lane`: LDS address of the lane's vector
// (split 0, block 0, quarter kq, row n) of the image -- ONE register; everything else is an immediate offset below 64 KB.
template <int NR>
__device__ __forceinline__ void ws_products(f32x4 (&acc)[NR], const bf16x8 (&W)[3][4], const unsigned char* __restrict__ img_lane, int rt0) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      bf16x8 b[3];
#pragma unroll
      for (int s = 0; s < 3; ++s)
        b[s] = *reinterpret_cast<const bf16x8*>(img_lane + (((s * 4 + c) * 4) * IMG_ROWSTRIDE + 16 * (rt0 + r)) * 16);
      f32x4 t = acc[r];
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[2][c], b[0], t, 0, 0, 0);      // smallest terms first
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[0][c], b[2], t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[1][c], b[1], t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[1][c], b[0], t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[0][c], b[1], t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[0][c], b[0], t, 0, 0, 0);
      acc[r] = t;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

__global__ __launch_bounds__(WS_T, 2) void ws_fwd_kernel(const hgn_mlp_fwd_t a, const long tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[WS_LDS];
  float* segf = reinterpret_cast<float*>(smem + WS_SEG_OFF);
  float* red = reinterpret_cast<float*>(smem + WS_RED_OFF);
  unsigned* bits = reinterpret_cast<unsigned*>(smem + WS_BITS_OFF);
  int* seg_ids_base = reinterpret_cast<int*>(smem + WS_IDS_OFF);
  int* gid = reinterpret_cast<int*>(smem + WS_GID_OFF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const long M = a.M;
  // this workgroup's tiles: workgroups b, b + 8, ... share an XCD (round-robin dispatch; speed only): XCD-major order
  const long G = gridDim.x, bx = blockIdx.x;
  const long q8 = G >> 3, r8 = G & 7, xc = bx & 7, ix = bx >> 3;
  const long pos = (xc < r8 ? xc * (q8 + 1) : r8 * (q8 + 1) + (xc - r8) * q8) + ix;
  const long t_beg = pos * tiles / G, t_end = (pos + 1) * tiles / G;

  // ---- this wave's weights: output block w of the three packed layers ([half][split][cl][ob][lane][8], csrc/mlp6.hip) ----
  bf16x8 W[3][3][4];
  {
    const __bf16* pk[3] = {reinterpret_cast<const __bf16*>(a.src[0].Wpk), reinterpret_cast<const __bf16*>(a.W2pk),
                           reinterpret_cast<const __bf16*>(a.W3pk)};
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          W[l][s][c] = *reinterpret_cast<const bf16x8*>(pk[l] + (c >> 1) * HALF_BF16 + ((s * 2 + (c & 1)) * 8 + w) * TILE_BF16 + lane * 8);
  }
  if (tid < 2 * 64 * 4) bits[tid] = 0u;

  // ---- lane roles --------------------------------------------------------------------------------------------------------
  // producer of the e image: row tid / 8, features 32 c + 8 qh + {0..7} and 32 c + 16 + 8 qh + {0..7}  (c = (tid & 7) / 2, qh = tid & 1)
  const int prow = tid >> 3, pc = (tid & 7) >> 1, pqh = tid & 1;
  const float* ex = a.src[0].x;
  const long lde = a.src[0].ld;
  // consumer / epilogue: rows 16 rt + n of the tile, units 16 w + 4 kq .. + 3
  const int ucol = 16 * w + 4 * kq;
  // (LDS addresses: one opaque 32-bit register per image and role, immediates below 64 KB -- see csrc/fused_bwd.hip on why)
  const unsigned char* imgr0 = smem + ws_opaque((unsigned)((kq * IMG_ROWSTRIDE + n) * 16));
  const unsigned char* imgr1 = smem + ws_opaque((unsigned)(IMG_BYTES + (kq * IMG_ROWSTRIDE + n) * 16));
  // my half vectors of the NEXT layer's image: block w / 2, quarter kq, bytes 8 (w & 1) .. + 7 of the vector
  unsigned char* imgw0 = smem + ws_opaque((unsigned)((((w >> 1) * 4 + kq) * IMG_ROWSTRIDE + n) * 16 + 8 * (w & 1)));
  unsigned char* imgw1 = smem + ws_opaque((unsigned)(IMG_BYTES + (((w >> 1) * 4 + kq) * IMG_ROWSTRIDE + n) * 16 + 8 * (w & 1)));
  // the e image: my two vectors (split 0) of image 0
  unsigned char* imge = smem + ws_opaque((unsigned)(img_vec(0, pc, 2 * pqh, prow) * 16));

  // Global rows are addressed as UNIFORM base + 32-bit byte offset (eligibility bounds every array to 4 GiB): one register per
  // address instead of a 64-bit pointer per array and row.
  auto ld4 = [](const float* base, unsigned off) { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off); };
  auto ld4w = [](const float* base, size_t off) { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + off); };      // (gathered rows: the node arrays' size is not known here)
  auto st4 = [](float* base, unsigned off, const f32x4& v) { *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(base) + off) = v; };
  const unsigned lde4 = (unsigned)a.src[0].ld * 4u, ldo4 = (unsigned)a.ld_out * 4u, ldr4 = (unsigned)a.ld_res * 4u;
  const unsigned ldp0 = (unsigned)a.add[0].ld * 4u, ldp1 = (unsigned)a.add[1].ld * 4u;
  const unsigned Mm1 = (unsigned)(M - 1);

  f32x4 ep[4];                                               // the e values of the next tile's image (this lane's 16)
  // Gather rows (P_s[snd], P_r[rcv]) of the next tile: loaded by 2 x 16 lanes one tile ahead, parked in LDS ahead of the tile's
  // last barriers and read back by every lane at the next tile's start (8 registers per lane for a whole tile otherwise).
  int nid[4];
  SegPre sp;
  const bool id_loader = w < 2 && kq == 0 && w < a.n_add;
  auto prefetch_rows = [&](long tile) {
    const unsigned r = min((unsigned)(tile * TILE_ROWS) + (unsigned)prow, Mm1);
    const unsigned off = r * lde4 + (unsigned)(32 * pc + 8 * pqh) * 4u;
    ep[0] = ld4(ex, off);
    ep[1] = ld4(ex, off + 16u);
    ep[2] = ld4(ex, off + 64u);
    ep[3] = ld4(ex, off + 80u);
  };
  auto prefetch_ids = [&](long tile) {
    if (id_loader) {
      const int32_t* idx = w == 0 ? a.add[0].idx : a.add[1].idx;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) nid[rt] = idx[min((unsigned)(tile * TILE_ROWS) + (unsigned)(16 * rt + n), Mm1)];
    }
    if (a.seg_out) sp.load(a.seg_ids, tile * TILE_ROWS, M);
  };
  auto park_ids = [&]() {
    if (id_loader) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) gid[w * TILE_ROWS + 16 * rt + n] = nid[rt];
    }
  };
  if (t_beg < t_end) { prefetch_ids(t_beg); prefetch_rows(t_beg); park_ids(); }
  __syncthreads();

  for (long tile = t_beg; tile < t_end; ++tile) {
    const long row0 = tile * TILE_ROWS;
    int* seg_ids_lds = seg_ids_base + (int)((tile - t_beg) & 1) * WS_IDS_STRIDE;
    const unsigned rown = (unsigned)row0 + (unsigned)ws_opaque((unsigned)n);      // my row of row tile 0 (re-derived per tile: not hoisted)
    const unsigned ucol4 = ws_opaque((unsigned)ucol * 4u);
    // ---- e image (buffer 0); gathers of the first two row tiles; prefetch of the next tile's rows --------------------------
    {
      bf16x8 sa[3], sb[3];
      ws_split8(ep[0], ep[2], sa);                           // quarter 2 qh
      ws_split8(ep[1], ep[3], sb);                           // quarter 2 qh + 1
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        *reinterpret_cast<bf16x8*>(imge + (s * 16 * IMG_ROWSTRIDE) * 16) = sa[s];
        *reinterpret_cast<bf16x8*>(imge + (s * 16 * IMG_ROWSTRIDE + IMG_ROWSTRIDE) * 16) = sb[s];
      }
    }
    if (a.seg_out) sp.stash(seg_ids_lds);
    f32x4 pg[2], ph[2];                                      // P_s[snd], P_r[rcv] of two of my rows (my 4 units)
    auto gather = [&](int h) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (a.n_add > 0) pg[r] = ld4w(a.add[0].P, (size_t)(unsigned)gid[16 * (2 * h + r) + n] * ldp0 + ucol4);
        if (a.n_add > 1) ph[r] = ld4w(a.add[1].P, (size_t)(unsigned)gid[TILE_ROWS + 16 * (2 * h + r) + n] * ldp1 + ucol4);
      }
    };
    gather(0);
    wg_barrier_lds();                                        // B1: image 0 = e
    // ---- layer 1 ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      ws_products<2>(acc, W[0], imgr0, 2 * h);
      const f32x4 b1 = ld4(a.b1, ucol4);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int rt = 2 * h + r;
        const unsigned row = rown + 16u * rt;
        f32x4 add = b1;
        if (a.n_add > 0) add = add + pg[r];
        if (a.n_add > 1) add = add + ph[r];
        f32x4 v = acc[r] + add;
        unsigned m = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { m |= (v[j] > 0.f ? 1u : 0u) << j; v[j] = fmaxf(v[j], 0.f); }
        if (a.z1 && row <= Mm1) st4(a.z1, row * (LAT * 4u) + ucol4, v);
        if (a.relu_bits) atomicOr(&bits[(16 * rt + n) * 4 + kq], m << (4 * w));
        bf16x4 sv[3];
        ws_split4(v, sv);
#pragma unroll
        for (int s = 0; s < 3; ++s)
          *reinterpret_cast<bf16x4*>(imgw1 + ((s * 16) * IMG_ROWSTRIDE + 16 * rt) * 16) = sv[s];
      }
      if (h == 0) gather(1);                                 // (arrives while the second pair of row tiles multiplies)
    }
    wg_barrier_lds();                                        // B2: image 1 = relu(z1); everyone is done with image 0
    if (a.relu_bits && tid < 256) {                          // layer-1 sign words are complete: write and clear
      const unsigned row = (unsigned)row0 + (unsigned)(tid >> 2);
      const unsigned m = bits[tid];
      bits[tid] = 0u;
      if (row <= Mm1) reinterpret_cast<unsigned*>(a.relu_bits)[(long)row * 8 + (tid & 3)] = m;
    }
    // ---- layer 2 ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 b2 = ld4(a.b2, ucol4);
      f32x4 acc[2] = {b2, b2};
      ws_products<2>(acc, W[1], imgr1, 2 * h);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int rt = 2 * h + r;
        const unsigned row = rown + 16u * rt;
        f32x4 v = acc[r];
        unsigned m = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { m |= (v[j] > 0.f ? 1u : 0u) << j; v[j] = fmaxf(v[j], 0.f); }
        if (a.z2 && row <= Mm1) st4(a.z2, row * (LAT * 4u) + ucol4, v);
        if (a.relu_bits) atomicOr(&bits[256 + (16 * rt + n) * 4 + kq], m << (4 * w));
        bf16x4 sv[3];
        ws_split4(v, sv);
#pragma unroll
        for (int s = 0; s < 3; ++s)
          *reinterpret_cast<bf16x4*>(imgw0 + ((s * 16) * IMG_ROWSTRIDE + 16 * rt) * 16) = sv[s];
      }
    }
    wg_barrier_lds();                                        // B3: image 0 = relu(z2)
    if (a.relu_bits && tid < 256) {
      const unsigned row = (unsigned)row0 + (unsigned)(tid >> 2);
      const unsigned m = bits[256 + tid];
      bits[256 + tid] = 0u;
      if (row <= Mm1) reinterpret_cast<unsigned*>(a.relu_bits)[(long)row * 8 + 4 + (tid & 3)] = m;
    }
    // ---- layer 3 + LayerNorm ---------------------------------------------------------------------------------------------
    // The next tile's rows and gather indices start their way HERE, not earlier: their 27 registers would otherwise be live
    // through layers 1 and 2, where the budget (144 weight registers + the epilogue's working set) has no room for them.
    if (tile + 1 < t_end) { prefetch_ids(tile + 1); prefetch_rows(tile + 1); }
    f32x4 o[4];
    {
      const f32x4 b3 = ld4(a.b3, ucol4);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) o[rt] = b3;
    }
    ws_products<4>(o, W[2], imgr0, 0);
    f32x4 rs[4];                                             // residual rows
    {                                                        // (LayerNorm is part of the eligible shape: straight-line code)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        float p = (o[rt][0] + o[rt][1]) + (o[rt][2] + o[rt][3]);
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        if (kq == 0) red[(16 * rt + n) * 8 + w] = p;
      }
      wg_barrier_lds();                                      // B4
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(red + (16 * rt + n) * 8);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(red + (16 * rt + n) * 8 + 4);
        const float mean = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / LAT);
        o[rt] = o[rt] - mean;
        float q = (__fmul_rn(o[rt][0], o[rt][0]) + __fmul_rn(o[rt][1], o[rt][1])) + (__fmul_rn(o[rt][2], o[rt][2]) + __fmul_rn(o[rt][3], o[rt][3]));
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        if (kq == 0) red[512 + (16 * rt + n) * 8 + w] = q;
        __builtin_amdgcn_sched_barrier(0);                   // one row tile at a time: the unrolled loop otherwise reads all partials first
      }
      if (tile + 1 < t_end) park_ids();                      // (this tile's gathers are long done; readers are two barriers away)
      wg_barrier_lds();                                      // B5
      if (a.res) {                                           // (in flight while x-hat is formed and stored)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) rs[rt] = ld4(a.res, min(rown + 16u * rt, Mm1) * ldr4 + ucol4);
      }
      const f32x4 gm = ld4(a.ln_g, ucol4);
      const f32x4 bt = ld4(a.ln_b, ucol4);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const unsigned row = rown + 16u * rt;
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(red + 512 + (16 * rt + n) * 8);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(red + 512 + (16 * rt + n) * 8 + 4);
        const float var = (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]))) * (1.f / LAT);
        const float rstd = 1.f / sqrtf(var + 1e-5f);
        o[rt] = o[rt] * rstd;
        if (a.xhat && row <= Mm1) st4(a.xhat, row * (LAT * 4u) + ucol4, o[rt]);
        if (a.rstd && row <= Mm1 && w == 0 && kq == 0) a.rstd[row] = rstd;
        o[rt] = o[rt] * gm + bt;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const unsigned row = rown + 16u * rt;
      if (a.res) o[rt] = o[rt] + rs[rt];
      if (row <= Mm1) st4(a.out, row * ldo4 + ucol4, o[rt]);
      if (a.seg_out) *reinterpret_cast<f32x4*>(segf + (16 * rt + n) * 132 + ucol) = o[rt];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (a.seg_out) {
      wg_barrier_lds();                                      // B6: the e' tile and its ids are in LDS
      tile_segment_walk(segf, seg_ids_lds, seg_ids_lds[TILE_ROWS], seg_ids_lds[TILE_ROWS + 1], a.seg_out, a.ld_seg_out, row0, M, true, tid);
    }
    // (the next tile writes image 0 only after B4, which every wave passes behind its last read of it; the segment tile after
    //  the next tile's B1..B5; the ids are double-buffered)
  }
}

}  // namespace hgn

using namespace hgn;

static bool ws_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// The edge-block shape and nothing else: one 128-wide ungathered source with a packed weight block, at most two gathered
// pre-projections, packed W2 / W3, LayerNorm, 128-wide output, six-product arithmetic.
extern "C" int hgn_mlp_fwd_ws_eligible(const hgn_mlp_fwd_t* a) {
  static const bool off = getenv("HGN_NO_WS_FWD") != nullptr || getenv("HGN_FP32_MFMA") != nullptr;
  if (off || !a || !hgn_mlp_fwd6_eligible(a) || matmul_products(a->products) != 6) return 0;
  if (a->n_src != 1 || a->src[0].K != 128 || a->src[0].idx || (a->src[0].ld & 3) || !ws_aligned16(a->src[0].x)) return 0;
  if (a->n_add < 0 || a->n_add > 2 || !a->ln_g || !a->ln_b || a->out_w != 128) return 0;
  for (int i = 0; i < a->n_add; ++i)
    if (!a->add[i].idx || (a->add[i].ld & 3) || !ws_aligned16(a->add[i].P)) return 0;
  // 32-bit byte offsets inside the kernel: every row array below 4 GiB
  const int64_t lim = (int64_t)1 << 32;
  if (a->M < 1 || a->M * a->src[0].ld * 4 >= lim || a->M * a->ld_out * 4 >= lim || a->M * 512 >= lim || (a->res && a->M * a->ld_res * 4 >= lim)) return 0;
  return 1;
}

namespace hgn {
int launch_ws_fwd(const hgn_mlp_fwd_t* a, void* stream) {
  static const long cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t pr;
      if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount;
    }
    return (long)n;
  }();
  const long tiles = (a->M + TILE_ROWS - 1) / TILE_ROWS;
  const long G = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL(ws_fwd_kernel, dim3((unsigned)G), dim3(WS_T), 0, (hipStream_t)stream, *a, tiles);
  return hgn_check_launch("hgn_mlp_fwd (weight-stationary)");
}
}  // namespace hgn
