// Fused 3-layer MLP kernels of the message-passing path (forward, data-gradient backward, single Linear).
// See hgn_device.h for the register-chained transposed-MFMA formulation and include/hgn_mp.h for the ABI.
#include "hgn_device.h"
#include "hgn_host.h"
#include "mlp_common.h"

namespace hgn {

// Diagnostic builds only (-DHGN_STAMP, tools/stamp_build.sh): per-wave s_memtime stamps into a side buffer that no
// other code reads.  The shipped library is built without it and contains no stamp instruction.
#ifdef HGN_STAMP
__device__ unsigned long long* g_stamps = nullptr;
__device__ int g_flags = 0;     // ablations: 1 skip MFMA, 2 skip stores, 4 skip gathers, 8 skip row loads, 16 skip DMA, 32 skip barriers
#define STAMP(i)                                                                                                   \
  do {                                                                                                             \
    if (g_stamps && (threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                                  \
      g_stamps[((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_readcyclecounter();           \
  } while (0)
#define ABL(bit) (g_flags & (bit))
#else
#define STAMP(i)
#define ABL(bit) false
#endif

__device__ __forceinline__ void relu_inplace(Act& a) {
  HGN_FOR_B(fb)
#pragma unroll
  for (int u = 0; u < 4; ++u) a.v[fb][u] = fmaxf(a.v[fb][u], 0.f);
}

// ---------------------------------------------------------------------------------------------------------
// forward:  out = [res +] [LN]( W3 relu(W2 relu(z1) + b2) + b3 )
// Every 128-wide contraction block is two staged halves (gemm_n); the wave's own global loads are issued between the
// first half's DMA and its wait, so they fly together; stores of saved activations stay in flight across the raw barriers.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG, 3) void mlp_fwd_kernel(const hgn_mlp_fwd_t a) {
  __shared__ __attribute__((aligned(16))) float wlds[WLDS_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const long row = xcd_tile() * TILE_ROWS + wave * WAVE_ROWS + n;
  const bool valid = row < a.M;
  const long rc = valid ? row : a.M - 1;

  Act acc, b;
  STAMP(0);
  // ---- layer 1 ------------------------------------------------------------------------------------------
  bool first = true;
  for (int si = 0; si < a.n_src; ++si) {
    const hgn_src_t s = a.src[si];
    const long srow = s.idx ? (long)s.idx[rc] : rc;
    const bool vec = ((s.ld & 3) == 0) && ((s.K & 3) == 0) && ((reinterpret_cast<uintptr_t>(s.x) & 15) == 0);
    for (int k0 = 0; k0 < s.K; k0 += 128) {
      const int kw = min(128, s.K - k0);
      const float* xr = s.x + srow * s.ld + k0;
      gemm_n(acc, b, wlds, s.W + k0, a.ldw1, 128, kw, [&] {
        if (vec) { if (kw == 128) t_load(b, xr, kq); else t_load_w(b, xr, kq, kw); } else t_load_masked(b, xr, kq, kw);
        if (first) {
          t_load(acc, a.b1, kq);
          for (int i = 0; i < a.n_add; ++i) t_add(acc, a.add[i].P + (long)a.add[i].idx[rc] * a.add[i].ld, kq);
          first = false;
        }
      });
    }
  }
  if (first) t_load(acc, a.b1, kq);
  STAMP(1);
  relu_inplace(acc);
  if (a.z1 && valid) t_store(acc, a.z1 + row * LAT, kq);
  if (a.relu_bits && valid) a.relu_bits[row * 8 + kq] = relu_bits_of(acc);
  // ---- layer 2 ------------------------------------------------------------------------------------------
  gemm_n(b, acc, wlds, a.W2, LAT, 128, 128, [&] { t_load(b, a.b2, kq); });      // b := b2 + W2 * acc
  STAMP(2);
  relu_inplace(b);
  if (a.z2 && valid) t_store(b, a.z2 + row * LAT, kq);
  if (a.relu_bits && valid) a.relu_bits[row * 8 + 4 + kq] = relu_bits_of(b);
  // ---- layer 3 ------------------------------------------------------------------------------------------
  gemm_n(acc, b, wlds, a.W3, LAT, a.out_w, 128, [&] {
    if (a.out_w == LAT) t_load(acc, a.b3, kq); else t_load_masked(acc, a.b3, kq, a.out_w);
  });
  STAMP(3);
  // ---- LayerNorm (eps 1e-5, biased variance: torch.nn.LayerNorm) + residual -------------------------------
  if (a.ln_g) {
    const float mean = row_sum(acc) * (1.f / LAT);
    HGN_FOR_B(fb) {
      acc.v[fb] -= mean;
      b.v[fb] = acc.v[fb] * acc.v[fb];
    }
    const float var = row_sum(b) * (1.f / LAT);
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    HGN_FOR_B(fb) acc.v[fb] *= rstd;
    if (a.xhat && valid) t_store(acc, a.xhat + row * LAT, kq);
    if (a.rstd && valid && kq == 0) a.rstd[row] = rstd;
    HGN_FOR_B(fb) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.ln_g + 16 * fb + 4 * kq);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(a.ln_b + 16 * fb + 4 * kq);
      acc.v[fb] = acc.v[fb] * gm + bt;
    }
  }
  if (valid) {
    if (a.out_w == LAT && (a.ld_out & 3) == 0) {
      if (a.res) t_add(acc, a.res + row * a.ld_res, kq);
      t_store(acc, a.out + row * a.ld_out, kq);
    } else {
      if (a.res) {
        t_load_masked(b, a.res + row * a.ld_res, kq, a.out_w);
        HGN_FOR_B(fb) acc.v[fb] += b.v[fb];
      }
      t_store_masked(acc, a.out + row * a.ld_out, kq, a.out_w);
    }
  }
  STAMP(4);
}

// ---------------------------------------------------------------------------------------------------------
// backward (data gradients)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG, 3) void mlp_bwd_kernel(const hgn_mlp_bwd_t a) {
  __shared__ __attribute__((aligned(16))) float wlds[WLDS_FLOATS + (WG / 64) * 256];
  float* lnl = wlds + WLDS_FLOATS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const long row = xcd_tile() * TILE_ROWS + wave * WAVE_ROWS + n;
  const bool valid = row < a.M;
  const long rc = valid ? row : a.M - 1;

  Act g, t;
  unsigned mb1 = 0, mb2 = 0;
  if (a.relu_bits) { mb1 = a.relu_bits[rc * 8 + kq]; mb2 = a.relu_bits[rc * 8 + 4 + kq]; }
  // ---- dz3 (LayerNorm backward, computed while the first half of W3 is in flight), dz2 = relu'(z2) * (W3^T dz3) -------
  gemm_t(t, g, wlds, a.W3, LAT, a.out_w, 128, [&] {
    load_dout<false>(g, a, rc, kq);
    if (a.ln_g) {
      // y = xhat*gamma + beta ;  dz3 = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat))
      t_load(t, a.xhat + rc * LAT, kq);
      if (a.ln_ws) {
        // LayerNorm-affine gradient partials of this wave's 16 rows: reduce over the row lanes (n), keep per feature
        HGN_FOR_B(fb) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float pb = valid ? g.v[fb][u] : 0.f;
            float pg = row16_sum(pb * t.v[fb][u]);
            pb = row16_sum(pb);
            if (n == 0) { lnl[wave * 256 + 16 * fb + 4 * kq + u] = pg; lnl[wave * 256 + 128 + 16 * fb + 4 * kq + u] = pb; }
          }
          __builtin_amdgcn_sched_barrier(0);      // one feature block at a time: keeps the live set small
        }
      }
      HGN_FOR_B(fb) g.v[fb] *= *reinterpret_cast<const f32x4*>(a.ln_g + 16 * fb + 4 * kq);
      const float m1 = row_sum(g) * (1.f / LAT);
      const float m2 = row_dot(g, t) * (1.f / LAT);
      const float r = a.rstd[rc];
      HGN_FOR_B(fb) g.v[fb] = r * (g.v[fb] - m1 - t.v[fb] * m2);
    }
    if (a.dz3 && valid) t_store(g, a.dz3 + row * LAT, kq);
    t_zero(t);
  });
  if (a.relu_bits) relu_mask_bits(t, mb2); else relu_mask(t, a.z2 + rc * LAT, kq);
  if (a.dz2 && valid) t_store(t, a.dz2 + row * LAT, kq);
  // ---- dz1 = relu'(z1) * (W2^T dz2) ----------------------------------------------------------------------
  gemm_t(g, t, wlds, a.W2, LAT, 128, 128, [&] { t_zero(g); });
  if (a.relu_bits) relu_mask_bits(g, mb1); else relu_mask(g, a.z1 + rc * LAT, kq);
  if (a.dz1 && valid) t_store(g, a.dz1 + row * LAT, kq);
  // ---- dx_src = dz1 * W1[:, cols]  (+ d_out_eff for the residual source) -----------------------------------
  for (int di = 0; di < a.n_dx; ++di) {
    const hgn_dx_t d = a.dx[di];
    for (int k0 = 0; k0 < d.K; k0 += 128) {
      const int kw = min(128, d.K - k0);
      gemm_t(t, g, wlds, d.W + k0, a.ldw1, 128, kw, [&] { t_zero(t); });
      if (valid) {
        float* dst = d.dx + row * d.ld + k0;
        if (kw == 128 && (d.ld & 3) == 0) {
          if (d.residual) load_dout<true>(t, a, rc, kq);   // re-read (L2-resident) rather than kept live through three GEMMs
          t_store(t, dst, kq);
        } else {
          t_store_masked(t, dst, kq, kw);      // residual sources are always 128 wide (latent)
        }
      }
    }
  }
  if (a.ln_ws) {
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < WG / 64; ++w) sum += lnl[w * 256 + threadIdx.x];
    a.ln_ws[(long)blockIdx.x * 256 + threadIdx.x] = sum;
    if (blockIdx.x == 0 && threadIdx.x == 0) { reinterpret_cast<unsigned*>(a.ln_ws)[-256] = 0u; reinterpret_cast<unsigned*>(a.ln_ws)[-255] = gridDim.x; }      // ticket of ln_reduce_kernel
  }
}

// Fixed-order column sum of the per-workgroup LayerNorm slabs [n_wg][256] -> dgamma[128], dbeta[128].
// Two steps, both with whole 1 KiB slabs read by 256 consecutive threads: LN_PARTS blocks add the slabs b, b + LN_PARTS, ...
// (eight loads in flight) into one partial slab each (stored behind the workgroup slabs); the block that finishes LAST (a ticket
// in the workspace's header slab, zeroed by the kernel that wrote the slabs) adds the partials in fixed order -- one launch, and the
// result does not depend on which block that is.  (Two launches before: 64 of them per training step, 9 % of a one-graph step.)
constexpr int LN_PARTS = 128;
__device__ __forceinline__ void ln_reduce_block(const float* __restrict__ ws, long n_wg, float* part, unsigned* ticket,
                                                float* __restrict__ dg, float* __restrict__ db, int acc) {
  __shared__ int is_last;
  const int c = threadIdx.x;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  long w = blockIdx.x;
  for (; w + 7 * LN_PARTS < n_wg; w += 8 * LN_PARTS) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += ws[(w + u * LN_PARTS) * 256 + c];
  }
  for (int u = 0; w < n_wg; w += LN_PARTS, ++u) s[u & 7] += ws[w * 256 + c];
  // The partial goes out as a device-coherent store (sc1: written through to where every XCD sees it) and is complete (vmcnt) before
  // this block draws its ticket; the last block reads the partials with device-coherent loads.  No fence instruction: a device-scope
  // release / acquire fence writes back / invalidates a whole L2 (the merged kernel took 16 us with them, 10 us as two launches).
  __hip_atomic_store(part + blockIdx.x * 256 + c, ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the store has been acknowledged at device scope
  __syncthreads();
  if (c == 0) is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(LN_PARTS - 1);
  __syncthreads();
  if (!is_last) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 4
  for (int b = 0; b < LN_PARTS; b += 4) {
    s0 += __hip_atomic_load(part + (b + 0) * 256 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s1 += __hip_atomic_load(part + (b + 1) * 256 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s2 += __hip_atomic_load(part + (b + 2) * 256 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s3 += __hip_atomic_load(part + (b + 3) * 256 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const float t = (s0 + s1) + (s2 + s3);
  float* dst = c < 128 ? dg + c : db + (c - 128);
  *dst = acc ? *dst + t : t;
}
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* __restrict__ ws, long n_wg, float* part, unsigned* ticket,
                                                        float* __restrict__ dg, float* __restrict__ db, int acc) {
  ln_reduce_block(ws, n_wg, part, ticket, dg, db, acc);
}
// The deferred form (hgn_ln_reduce_batch): blockIdx.y = the backward call whose slabs these are; their number was left beside the
// ticket by the kernel that wrote them (uniform: one scalar load).
struct LnBatchTask { float* ws; float* part; float* dg; float* db; int acc; int pad; };
struct LnBatch { LnBatchTask t[HGN_MAX_LN_TASK]; };
__global__ __launch_bounds__(256) void ln_reduce_batch_kernel(const LnBatch b) {
  const LnBatchTask t = b.t[blockIdx.y];
  unsigned* header = reinterpret_cast<unsigned*>(t.ws) - 256;
  ln_reduce_block(t.ws, (long)header[1], t.part, header, t.dg, t.db, t.acc);
}

// ---------------------------------------------------------------------------------------------------------
// single Linear over 128-wide output blocks (node pre-projection of the split edge layer) and its dgrad
// ---------------------------------------------------------------------------------------------------------
struct LinArgs {
  const float* x; long ldx; long M; const float* W[4]; int n_blocks; long ldw; float* out; long ld_out;
};

__global__ __launch_bounds__(WG, 4) void linear_fwd_kernel(const LinArgs a) {
  __shared__ __attribute__((aligned(16))) float wlds[WLDS_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const long row = xcd_tile() * TILE_ROWS + wave * WAVE_ROWS + n;
  const bool valid = row < a.M;
  const long rc = valid ? row : a.M - 1;
  Act acc, b;
  for (int blk = 0; blk < a.n_blocks; ++blk) {
    gemm_n(acc, b, wlds, a.W[blk], a.ldw, 128, 128, [&] {
      if (blk == 0) t_load(b, a.x + rc * a.ldx, kq);
      t_zero(acc);
    });
    if (valid) t_store(acc, a.out + row * a.ld_out + 128 * blk, kq);
  }
}

__global__ __launch_bounds__(WG, 4) void linear_bwd_kernel(const LinArgs a) {
  // here a.x = g [M, 128*n_blocks], a.out = dx [M,128]
  __shared__ __attribute__((aligned(16))) float wlds[WLDS_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const long row = xcd_tile() * TILE_ROWS + wave * WAVE_ROWS + n;
  const bool valid = row < a.M;
  const long rc = valid ? row : a.M - 1;
  Act acc, b;
  t_zero(acc);
  for (int blk = 0; blk < a.n_blocks; ++blk)
    gemm_t(acc, b, wlds, a.W[blk], a.ldw, 128, 128, [&] { t_load(b, a.x + rc * a.ldx + 128 * blk, kq); });
  if (valid) t_store(acc, a.out + row * a.ld_out, kq);
}

// `ws`: the slabs as the kernels see them, i.e. one slab behind the start of the caller's ln_ws (its first slab is the header)
int launch_ln_reduce(float* ws, long n_slabs, float* part, float* d_gamma, float* d_beta, int accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(ln_reduce_kernel, dim3(LN_PARTS), dim3(256), 0, stream, ws, n_slabs, part, reinterpret_cast<unsigned*>(ws - 256),
                     d_gamma, d_beta, accumulate);
  return hgn_check_launch("LayerNorm gradient reduce");
}

}  // namespace hgn

namespace hgn { bool cs_eligible(const hgn_mlp_fwd_t* a); int launch_ws_fwd(const hgn_mlp_fwd_t* a, void* stream); int launch_mlp6_fwd(const hgn_mlp_fwd_t* a, void* stream); int launch_mlp6_bwd(const hgn_mlp_bwd_t* a, void* stream, long* n_slabs); }
using namespace hgn;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#if HGN_LAB   // laboratory build only (tools/lab): the weight-stationary edge forward, HGN_WS_FWD=1 or hgn_set_ws_fwd(1)
extern "C" int hgn_mlp_fwd_ws_eligible(const hgn_mlp_fwd_t* a);
static int g_ws_fwd = getenv("HGN_WS_FWD") ? 1 : 0;
extern "C" int hgn_set_ws_fwd(int on) { g_ws_fwd = on ? 1 : 0; return HGN_OK; }
#endif

#ifdef HGN_STAMP
extern "C" int hgn_debug_set_flags(int f) {
  return hipMemcpyToSymbol(HIP_SYMBOL(hgn::g_flags), &f, sizeof(f)) == hipSuccess ? 0 : -2;
}
extern "C" int hgn_debug_set_stamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(hgn::g_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int hgn_mlp_fwd_post_eligible(const hgn_mlp_fwd_t* a) {
  if (!a || a->n_post < 1 || a->n_post > 4 || !a->post_out || (a->ld_post & 3) || a->ld_post < 128L * a->n_post || !aligned16(a->post_out))
    return 0;
  for (int i = 0; i < a->n_post; ++i)
    if (!a->post_pk[i]) return 0;
  if (a->post_zero && ((a->ld_post_zero & 3) || a->ld_post_zero < 128 || !aligned16(a->post_zero))) return 0;
  return a->out_w == 128 && hgn_mlp_fwd6_eligible(a) && a->n_add == 0 && !a->seg_out ? 1 : 0;      // any form of the split-product forward (the edge-block shape has no use for it)
}

extern "C" int hgn_mlp_fwd(const hgn_mlp_fwd_t* a, void* stream) {
  if (!a) return hgn_fail(HGN_E_INVALID, "hgn_mlp_fwd: null args");
  if (!valid_products(a->products)) return hgn_fail(HGN_E_INVALID, "hgn_mlp_fwd: products must be 0 (process default), 6, 3, 1 or 2");
  if (a->M == 0) return HGN_OK;
  if (a->M < 0 || a->n_src < 0 || a->n_src > HGN_MAX_SRC || a->n_add < 0 || a->n_add > HGN_MAX_ADD)
    return hgn_fail(HGN_E_INVALID, "hgn_mlp_fwd: bad counts");
  if (a->out_w < 1 || a->out_w > 128 || (a->ln_g && a->out_w != 128) || (!a->ln_g != !a->ln_b))
    return hgn_fail(HGN_E_INVALID, "hgn_mlp_fwd: bad out_w / LayerNorm");
  if (!a->b1 || !a->W2 || !a->b2 || !a->W3 || !a->b3 || !a->out)
    return hgn_fail(HGN_E_INVALID, "hgn_mlp_fwd: null weight/output pointer");
  for (int i = 0; i < a->n_src; ++i)
    if (!a->src[i].x || !a->src[i].W || a->src[i].K < 1) return hgn_fail(HGN_E_INVALID, "hgn_mlp_fwd: bad source");
  for (int i = 0; i < a->n_add; ++i)
    if (!a->add[i].P || !a->add[i].idx || (a->add[i].ld & 3) || !aligned16(a->add[i].P))
      return hgn_fail(HGN_E_INVALID, "hgn_mlp_fwd: bad addend");
  if (a->res && a->out_w == 128 && ((a->ld_res & 3) || !aligned16(a->res)))
    return hgn_fail(HGN_E_INVALID, "hgn_mlp_fwd: residual must be 16-byte aligned");
  const long tiles = (a->M + TILE_ROWS - 1) / TILE_ROWS;
  const int kid = a->n_add ? 0 : 1;
  ProfScope ps(kid, (double)a->M, (hipStream_t)stream);
  if (a->n_post != 0 && !hgn_mlp_fwd_post_eligible(a))
    return hgn_fail(HGN_E_INVALID, "hgn_mlp_fwd: post_* needs the split-product forward without gathered addends (hgn_mlp_fwd_post_eligible)");
  if (a->seg_out && (!a->seg_ids || a->out_w != 128 || a->ld_seg_out < 128 || !hgn_mlp_fwd6_eligible(a)))
    return hgn_fail(HGN_E_INVALID, "hgn_mlp_fwd: seg_out needs seg_ids, a 128-wide output and the split-bf16 kernel");
#if HGN_LAB
  if (g_ws_fwd && hgn_mlp_fwd_ws_eligible(a)) return launch_ws_fwd(a, stream);
#endif
  if (hgn_mlp_fwd6_eligible(a)) return launch_mlp6_fwd(a, stream);
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, *a);
  return hgn_check_launch("hgn_mlp_fwd");
}

extern "C" int hgn_mlp_bwd_ln_workspace_bytes(int64_t M, size_t* bytes) {
  if (!bytes || M < 0) return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd_ln_workspace_bytes: bad argument");
  const long tiles = ln_slab_capacity(M);
  *bytes = ((size_t)tiles + LN_PARTS + 1) * 256 * sizeof(float) + 256;  // header slab (ticket) + workgroup slabs + the partial slabs of ln_reduce
  return HGN_OK;
}

extern "C" int hgn_mlp_bwd(const hgn_mlp_bwd_t* a, void* stream) {
  if (!a) return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd: null args");
  if (!valid_products(a->products)) return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd: products must be 0 (process default), 6, 3, 1 or 2");
  if (a->M == 0) return HGN_OK;
  if (a->M < 0 || a->n_dx < 0 || a->n_dx > HGN_MAX_SRC) return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd: bad counts");
  if (a->out_w < 1 || a->out_w > 128 || (a->ln_g && (a->out_w != 128 || !a->xhat || !a->rstd)))
    return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd: bad out_w / LayerNorm");
  if ((!a->d_out && !a->agg_dout) || (!a->relu_bits && (!a->z1 || !a->z2)) || !a->W2 || !a->W3) return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd: null pointer");
  if (a->agg_dout) {
    if (a->out_w != 128 || a->n_agg_ops < 1 || a->n_agg_ops > 4 || !a->agg_seg || !a->agg_rowptr || (a->ld_agg & 3) ||
        !aligned16(a->agg_dout))
      return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd: bad aggregation-backward descriptor");
    for (int i = 0; i < a->n_agg_ops; ++i) {
      if (a->agg_ops[i] < 0 || a->agg_ops[i] > 3) return hgn_fail(HGN_E_INVALID, "Invalid operation type!");
      if ((a->agg_ops[i] == HGN_OP_MAX && !a->agg_argmax) || (a->agg_ops[i] == HGN_OP_MIN && !a->agg_argmin))
        return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd: max/min need the saved arg index");
    }
  }
  if (a->ln_ws && (!a->ln_g || !a->d_gamma || !a->d_beta)) return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd: LayerNorm gradient outputs missing");
  for (int i = 0; i < a->n_dx; ++i)
    if (!a->dx[i].W || !a->dx[i].dx || a->dx[i].K < 1 || (a->dx[i].residual && (a->dx[i].K != 128 || a->out_w != 128)))
      return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd: bad dx request");
  long tiles = (a->M + TILE_ROWS - 1) / TILE_ROWS;      // = LayerNorm-gradient partial slabs in ln_ws (one per workgroup)
  const int kid = (a->n_dx == 1 && a->dx[0].residual && a->dz1) ? 2 : 3;
  ProfScope ps(kid, (double)a->M, (hipStream_t)stream);
  if (a->seg_dz1 && (!a->seg_ids || a->ld_seg_dz1 < 128 || !hgn_mlp_bwd6_eligible(a)))
    return hgn_fail(HGN_E_INVALID, "hgn_mlp_bwd: seg_dz1 needs seg_ids and the split-bf16 kernel");
  hgn_mlp_bwd_t b = *a;
  if (b.ln_ws) b.ln_ws += 256;                          // the kernels' slab 0 lies behind the header slab (ticket of ln_reduce_kernel)
  if (hgn_mlp_bwd6_eligible(a)) {
    if (launch_mlp6_bwd(&b, stream, &tiles) != HGN_OK) return HGN_E_LAUNCH;
  } else {
    hipLaunchKernelGGL(mlp_bwd_kernel, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, b);
  }
  if (b.ln_ws && !(a->flags & HGN_F_DEFER_LN)) {
    // partial slabs live behind the slabs of a 64-row tiling (hgn_mlp_bwd_ln_workspace_bytes), whatever tiling ran
    float* part = b.ln_ws + ln_slab_capacity(a->M) * 256;
    if (launch_ln_reduce(b.ln_ws, tiles, part, a->d_gamma, a->d_beta, a->ln_accumulate, (hipStream_t)stream) != HGN_OK) return HGN_E_LAUNCH;
  }
  return hgn_check_launch("hgn_mlp_bwd");
}

extern "C" int hgn_ln_reduce_batch(const hgn_ln_task_t* tasks, int n_tasks, void* stream) {
  if (n_tasks == 0) return HGN_OK;
  if (!tasks || n_tasks < 0 || n_tasks > HGN_MAX_LN_TASK) return hgn_fail(HGN_E_INVALID, "hgn_ln_reduce_batch: bad task list");
  LnBatch b;
  for (int i = 0; i < n_tasks; ++i) {
    const hgn_ln_task_t& t = tasks[i];
    if (!t.ln_ws || !t.d_gamma || !t.d_beta || t.M < 1) return hgn_fail(HGN_E_INVALID, "hgn_ln_reduce_batch: bad task");
    for (int j = 0; j < i; ++j)                       // (the sums are added to their targets without atomics)
      if (tasks[j].d_gamma == t.d_gamma || tasks[j].d_beta == t.d_beta)
        return hgn_fail(HGN_E_INVALID, "hgn_ln_reduce_batch: two tasks of one batch share a target");
    float* ws = t.ln_ws + 256;                        // slab 0 lies behind the header slab (ticket, slab count)
    b.t[i] = {ws, ws + ln_slab_capacity(t.M) * 256, t.d_gamma, t.d_beta, t.accumulate ? 1 : 0, 0};
  }
  hipLaunchKernelGGL(ln_reduce_batch_kernel, dim3(LN_PARTS, (unsigned)n_tasks), dim3(256), 0, (hipStream_t)stream, b);
  return hgn_check_launch("hgn_ln_reduce_batch");
}

static int linear_common(bool fwd, const float* x, int64_t ldx, int64_t M, const float* const* Wb, int nb, int64_t ldw,
                         float* out, int64_t ld_out, void* stream) {
  if (M == 0) return HGN_OK;
  if (!x || !Wb || !out || M < 0 || nb < 1 || nb > 4 || (ldx & 3) || (ld_out & 3) || !aligned16(x) || !aligned16(out))
    return hgn_fail(HGN_E_INVALID, "hgn_linear: bad argument");
  LinArgs a;
  a.x = x; a.ldx = ldx; a.M = M; a.n_blocks = nb; a.ldw = ldw; a.out = out; a.ld_out = ld_out;
  for (int i = 0; i < 4; ++i) a.W[i] = i < nb ? Wb[i] : nullptr;
  const long tiles = (M + TILE_ROWS - 1) / TILE_ROWS;
  ProfScope ps(fwd ? 7 : 8, (double)M, (hipStream_t)stream);
  if (fwd) hipLaunchKernelGGL(linear_fwd_kernel, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(linear_bwd_kernel, dim3((unsigned)tiles), dim3(WG), 0, (hipStream_t)stream, a);
  return hgn_check_launch("hgn_linear");
}

extern "C" int hgn_linear_fwd(const float* x, int64_t ldx, int64_t M, const float* const* Wb, int nb, int64_t ldw,
                              float* out, int64_t ld_out, void* stream) {
  return linear_common(true, x, ldx, M, Wb, nb, ldw, out, ld_out, stream);
}
extern "C" int hgn_linear_bwd(const float* g, int64_t ldg, int64_t M, const float* const* Wb, int nb, int64_t ldw,
                              float* dx, int64_t ld_dx, void* stream) {
  return linear_common(false, g, ldg, M, Wb, nb, ldw, dx, ld_dx, stream);
}
