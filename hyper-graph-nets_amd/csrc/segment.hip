// CSR topology build and the segment (scatter) reductions of the message-passing path.
// HBM-bound: one pass over the edge latents, half a wavefront (32 lanes x float4 = 128 floats = one latent row)
// per receiver row, all requested aggregates (sum / mean / max / min) produced in that single pass.
#include <hipcub/hipcub.hpp>
#include "hgn_host.h"

namespace hgn {

constexpr int MAXOPS = 4;
struct Ops { int n; int op[MAXOPS]; };

// ----------------------------------------------------------------------------------------------------------
// topology
// ----------------------------------------------------------------------------------------------------------
__global__ void csr_prepare_kernel(const int64_t* __restrict__ ids, long E, long N, int* __restrict__ keys,
                                   int* __restrict__ vals, int* __restrict__ flag) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const int64_t v = ids[i];
  if (v < 0 || v >= N) { atomicOr(flag, 1); keys[i] = 0; }
  else keys[i] = (int)v;
  vals[i] = (int)i;
}

__global__ void csr_rowptr_kernel(const int* __restrict__ seg, long E, long N, int* __restrict__ rowptr) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n > N) return;
  long lo = 0, hi = E;                       // first position with seg[pos] >= n
  while (lo < hi) {
    const long mid = (lo + hi) >> 1;
    if (seg[mid] < (int)n) lo = mid + 1; else hi = mid;
  }
  rowptr[n] = (int)lo;
}

// longest CSR row (the fused in-kernel segment sums need it, include/hgn_mp.h: seg_out) -> flag[1], read back with the range flag
__global__ void csr_max_rows_kernel(const int* __restrict__ rowptr, long N, int* __restrict__ flag) {
  const long n = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int len = n < N ? rowptr[n + 1] - rowptr[n] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) len = max(len, __shfl_xor(len, o));
  if ((threadIdx.x & 63) == 0 && len > 0) atomicMax(flag + 1, len);
}

// Order-independent 64-bit fingerprints of two int64 index arrays (position-salted splitmix64, summed mod 2^64 with integer
// atomics): the key under which a topology built from EQUAL index content is found again (topology cache of the host side).
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void index_fingerprint_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b, long n,
                                                                unsigned long long* __restrict__ out) {
  unsigned long long ha = 0, hb = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned long long salt = (unsigned long long)i * 0xd6e8feb86659fd93ull;
    ha += mix64((unsigned long long)a[i] ^ salt);
    if (b) hb += mix64((unsigned long long)b[i] + salt);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ha += __shfl_xor(ha, o);
    hb += __shfl_xor(hb, o);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out, ha);
    atomicAdd(out + 1, hb);
  }
}

__global__ void narrow_gather_kernel(const int64_t* __restrict__ src, const int* __restrict__ perm, long n,
                                     int* __restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (int)src[perm ? perm[i] : i];
}

// ----------------------------------------------------------------------------------------------------------
// forward, D == 128: half-wave per row
// ----------------------------------------------------------------------------------------------------------
// Rows that are read once (the edge rows of a segment sum) come in as non-temporal loads: they do not push the row pointers, the
// permutation and what the NEXT kernel wants to find out of the caches (seg_fwd128 -4 ... -8 %, the step -0.3 ms; the same hint on the row
// loads of the two big edge kernels costs more downstream than it gains: profiles/HISTORY.md).
typedef float f4v_stream __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_load4(const float* p) {
  const f4v_stream v = __builtin_nontemporal_load(reinterpret_cast<const f4v_stream*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
template <bool FEW>      // FEW: few rows, possibly long ones (launch): 64-thread workgroups, 16 rows in flight; same additions in the same order
__global__ __launch_bounds__(256) void seg_fwd128_kernel(const float* __restrict__ data, long ld,
                                                         const int* __restrict__ perm,
                                                         const int* __restrict__ rowptr, long N, Ops ops,
                                                         float* __restrict__ out, long ld_out,
                                                         int* __restrict__ argmax, int* __restrict__ argmin) {
  const long n = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 5;      // (64-thread workgroups when there are few rows: see the launch)
  if (n >= N) return;
  const int c = (threadIdx.x & 31) * 4;
  const int beg = rowptr[n], end = rowptr[n + 1];
  bool want_max = false, want_min = false;
#pragma unroll
  for (int s = 0; s < MAXOPS; ++s)
    if (s < ops.n) { want_max |= ops.op[s] == HGN_OP_MAX; want_min |= ops.op[s] == HGN_OP_MIN; }
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  float mx[4], mn[4];
  int amx[4], amn[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { mx[u] = -INFINITY; mn[u] = INFINITY; amx[u] = -1; amn[u] = -1; }
  int j = beg;
  // Long segments (the rows that arrive at a hyper node: ~100 per cluster): 16 rows in flight, consumed in the same order as below --
  // the walk of one segment is a chain of memory round trips, and the few segments of such a set leave most of the chip idle.
  for (; FEW && j + 16 <= end; j += 16) {
    float4 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long src = perm ? perm[j + q] : (j + q);
      v[q] = stream_load4(data + src * ld + c);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      sum.x += v[q].x; sum.y += v[q].y; sum.z += v[q].z; sum.w += v[q].w;
      const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
      if (want_max) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (e[u] > mx[u] || amx[u] < 0) { mx[u] = e[u]; amx[u] = j + q; }
      }
      if (want_min) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (e[u] < mn[u] || amn[u] < 0) { mn[u] = e[u]; amn[u] = j + q; }
      }
    }
  }
  // 4 rows in flight per half-wave
  for (; j + 4 <= end; j += 4) {
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long src = perm ? perm[j + q] : (j + q);
      v[q] = stream_load4(data + src * ld + c);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sum.x += v[q].x; sum.y += v[q].y; sum.z += v[q].z; sum.w += v[q].w;
      const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
      if (want_max) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (e[u] > mx[u] || amx[u] < 0) { mx[u] = e[u]; amx[u] = j + q; }
      }
      if (want_min) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (e[u] < mn[u] || amn[u] < 0) { mn[u] = e[u]; amn[u] = j + q; }
      }
    }
  }
  for (; j < end; ++j) {
    const long src = perm ? perm[j] : j;
    const float4 v = stream_load4(data + src * ld + c);
    sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    const float e[4] = {v.x, v.y, v.z, v.w};
    if (want_max) {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (e[u] > mx[u] || amx[u] < 0) { mx[u] = e[u]; amx[u] = j; }
    }
    if (want_min) {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (e[u] < mn[u] || amn[u] < 0) { mn[u] = e[u]; amn[u] = j; }
    }
  }
  const int cnt = end - beg;
  const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
#pragma unroll
  for (int s = 0; s < MAXOPS; ++s) {
    if (s < ops.n) {
      float4 o;
      switch (ops.op[s]) {
        case HGN_OP_SUM: o = sum; break;
        case HGN_OP_MEAN: o = make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv); break;
        case HGN_OP_MAX: o = cnt ? make_float4(mx[0], mx[1], mx[2], mx[3]) : make_float4(0.f, 0.f, 0.f, 0.f); break;
        default: o = cnt ? make_float4(mn[0], mn[1], mn[2], mn[3]) : make_float4(0.f, 0.f, 0.f, 0.f); break;
      }
      *reinterpret_cast<float4*>(out + n * ld_out + (long)s * 128 + c) = o;
    }
  }
  if (argmax && want_max) *reinterpret_cast<int4*>(argmax + n * 128 + c) = make_int4(amx[0], amx[1], amx[2], amx[3]);
  if (argmin && want_min) *reinterpret_cast<int4*>(argmin + n * 128 + c) = make_int4(amn[0], amn[1], amn[2], amn[3]);
}

// Two segment sums of the SAME rows in one pass (D = 128): out_a[n] = sum of the rows rowptr_a[n] .. rowptr_a[n+1] (in place: the
// rows are sorted by that key), out_b[n] = sum of the rows perm_b[rowptr_b[n] ..] (a second key, through its permutation).  The backward
// of a split edge layer needs both over dz1 -- by receiver and by sender (graphnet.py:22-32: h[receivers], h[senders]) -- and in a mesh
// the rows a node SENDS are rows its neighbours RECEIVE: with workgroups of one XCD on a contiguous range of nodes the second
// visit of a row finds it in that XCD's L2 (or in the Infinity Cache) instead of reading the array from HBM a second time.  Same
// additions in the same order as two seg_fwd128 launches.
__global__ __launch_bounds__(256) void seg_sum_pair128_kernel(const float* __restrict__ data, long ld, const int* __restrict__ rowptr_a,
                                                              const int* __restrict__ perm_b, const int* __restrict__ rowptr_b, long N,
                                                              float* __restrict__ out_a, long ld_a, float* __restrict__ out_b, long ld_b) {
  // XCD b & 7 gets one contiguous range of workgroup slots (cf. hgn_device.h: xcd_tile)
  const long nt = gridDim.x, b = blockIdx.x;
  const long q = nt >> 3, r = nt & 7, x = b & 7, i = b >> 3;
  const long slot = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  const long n = (slot * 256 + threadIdx.x) >> 5;
  if (n >= N) return;
  const int c = (threadIdx.x & 31) * 4;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int* __restrict__ rowptr = pass ? rowptr_b : rowptr_a;
    const int beg = rowptr[n], end = rowptr[n + 1];
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    int j = beg;
    for (; j + 4 <= end; j += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long src = pass ? perm_b[j + u] : (j + u);
        // (the second visit is the last one: a streaming load leaves the cache to the rows still waiting for theirs: 0.186 -> 0.167 ms)
        v[u] = pass ? stream_load4(data + src * ld + c) : *reinterpret_cast<const float4*>(data + src * ld + c);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
    }
    for (; j < end; ++j) {
      const long src = pass ? perm_b[j] : j;
      const float4 v = pass ? stream_load4(data + src * ld + c) : *reinterpret_cast<const float4*>(data + src * ld + c);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    *reinterpret_cast<float4*>((pass ? out_b + n * ld_b : out_a + n * ld_a) + c) = sum;
  }
}

// forward, any D: thread per (row, d)
__global__ void seg_fwd_generic_kernel(const float* __restrict__ data, long ld, int D, const int* __restrict__ perm,
                                       const int* __restrict__ rowptr, long N, Ops ops, float* __restrict__ out,
                                       long ld_out, int* __restrict__ argmax, int* __restrict__ argmin) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N * D) return;
  const long n = gid / D;
  const int d = (int)(gid - n * D);
  const int beg = rowptr[n], end = rowptr[n + 1];
  float sum = 0.f, mx = -INFINITY, mn = INFINITY;
  int amx = -1, amn = -1;
  for (int j = beg; j < end; ++j) {
    const long src = perm ? perm[j] : j;
    const float v = data[src * ld + d];
    sum += v;
    if (v > mx || amx < 0) { mx = v; amx = j; }
    if (v < mn || amn < 0) { mn = v; amn = j; }
  }
  const int cnt = end - beg;
  for (int s = 0; s < ops.n; ++s) {
    float o;
    switch (ops.op[s]) {
      case HGN_OP_SUM: o = sum; break;
      case HGN_OP_MEAN: o = sum / (float)(cnt > 0 ? cnt : 1); break;
      case HGN_OP_MAX: o = cnt ? mx : 0.f; break;
      default: o = cnt ? mn : 0.f; break;
    }
    out[n * ld_out + (long)s * D + d] = o;
  }
  if (argmax) argmax[n * D + d] = amx;
  if (argmin) argmin[n * D + d] = amn;
}

// ----------------------------------------------------------------------------------------------------------
// backward: gather the row's output gradient back to each of its edges (edge-parallel, coalesced in sorted order)
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_bwd128_kernel(const float* __restrict__ d_out, long ld_out,
                                                         const int* __restrict__ perm, const int* __restrict__ seg,
                                                         const int* __restrict__ rowptr, long E, Ops ops,
                                                         const int* __restrict__ argmax,
                                                         const int* __restrict__ argmin,
                                                         const float* __restrict__ base, float* __restrict__ d_data,
                                                         long ld) {
  const long j = ((long)blockIdx.x * 256 + threadIdx.x) >> 5;
  if (j >= E) return;
  const int c = (threadIdx.x & 31) * 4;
  const long r = seg[j];
  const long pos = perm ? perm[j] : j;
  const int cnt = rowptr[r + 1] - rowptr[r];
  const float inv = 1.f / (float)(cnt > 0 ? cnt : 1);
  float4 g = base ? *reinterpret_cast<const float4*>(base + pos * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int s = 0; s < MAXOPS; ++s) {
    if (s < ops.n) {
      const float4 d = *reinterpret_cast<const float4*>(d_out + r * ld_out + (long)s * 128 + c);
      switch (ops.op[s]) {
        case HGN_OP_SUM: g.x += d.x; g.y += d.y; g.z += d.z; g.w += d.w; break;
        case HGN_OP_MEAN: g.x += d.x * inv; g.y += d.y * inv; g.z += d.z * inv; g.w += d.w * inv; break;
        case HGN_OP_MAX: {
          const int4 a = *reinterpret_cast<const int4*>(argmax + r * 128 + c);
          g.x += a.x == (int)j ? d.x : 0.f; g.y += a.y == (int)j ? d.y : 0.f;
          g.z += a.z == (int)j ? d.z : 0.f; g.w += a.w == (int)j ? d.w : 0.f;
        } break;
        default: {
          const int4 a = *reinterpret_cast<const int4*>(argmin + r * 128 + c);
          g.x += a.x == (int)j ? d.x : 0.f; g.y += a.y == (int)j ? d.y : 0.f;
          g.z += a.z == (int)j ? d.z : 0.f; g.w += a.w == (int)j ? d.w : 0.f;
        } break;
      }
    }
  }
  *reinterpret_cast<float4*>(d_data + pos * ld + c) = g;
}

// The same for rows in segment order (perm == null), one half-wave per SEGMENT: the segment's gradient rows (one per aggregate) and its arg rows
// are loaded once and applied to its rows in turn -- two memory instructions per row instead of seven (the edge-parallel form above re-reads
// them for every row; they hit the caches, but the requests are what the kernel is made of).  Same additions in the same order.
__global__ __launch_bounds__(256) void seg_bwd128_sorted_kernel(const float* __restrict__ d_out, long ld_out, const int* __restrict__ rowptr,
                                                                long N, Ops ops, const int* __restrict__ argmax, const int* __restrict__ argmin,
                                                                const float* __restrict__ base, float* __restrict__ d_data, long ld) {
  const long r = ((long)blockIdx.x * 256 + threadIdx.x) >> 5;
  if (r >= N) return;
  const int c = (threadIdx.x & 31) * 4;
  const int beg = rowptr[r], end = rowptr[r + 1];
  if (beg == end) return;
  const float inv = 1.f / (float)(end - beg);
  float4 d[MAXOPS];
  int4 amx = make_int4(-1, -1, -1, -1), amn = make_int4(-1, -1, -1, -1);
#pragma unroll
  for (int s = 0; s < MAXOPS; ++s)
    if (s < ops.n) {
      d[s] = *reinterpret_cast<const float4*>(d_out + r * ld_out + (long)s * 128 + c);
      if (ops.op[s] == HGN_OP_MAX) amx = *reinterpret_cast<const int4*>(argmax + r * 128 + c);
      if (ops.op[s] == HGN_OP_MIN) amn = *reinterpret_cast<const int4*>(argmin + r * 128 + c);
    }
  for (int j0 = beg; j0 < end; j0 += 4) {
    float4 g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j0 + u < end) g[u] = base ? stream_load4(base + (long)(j0 + u) * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (j0 + u < end) {
        const int j = j0 + u;
#pragma unroll
        for (int s = 0; s < MAXOPS; ++s) {
          if (s < ops.n) {
            const float4 ds = d[s];
            switch (ops.op[s]) {
              case HGN_OP_SUM: g[u].x += ds.x; g[u].y += ds.y; g[u].z += ds.z; g[u].w += ds.w; break;
              case HGN_OP_MEAN: g[u].x += ds.x * inv; g[u].y += ds.y * inv; g[u].z += ds.z * inv; g[u].w += ds.w * inv; break;
              case HGN_OP_MAX:
                g[u].x += amx.x == j ? ds.x : 0.f; g[u].y += amx.y == j ? ds.y : 0.f; g[u].z += amx.z == j ? ds.z : 0.f; g[u].w += amx.w == j ? ds.w : 0.f;
                break;
              default:
                g[u].x += amn.x == j ? ds.x : 0.f; g[u].y += amn.y == j ? ds.y : 0.f; g[u].z += amn.z == j ? ds.z : 0.f; g[u].w += amn.w == j ? ds.w : 0.f;
                break;
            }
          }
        }
        *reinterpret_cast<float4*>(d_data + (long)j * ld + c) = g[u];
      }
  }
}

__global__ void seg_bwd_generic_kernel(const float* __restrict__ d_out, long ld_out, int D,
                                       const int* __restrict__ perm, const int* __restrict__ seg,
                                       const int* __restrict__ rowptr, long E, Ops ops,
                                       const int* __restrict__ argmax, const int* __restrict__ argmin,
                                       const float* __restrict__ base, float* __restrict__ d_data, long ld) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= E * D) return;
  const long j = gid / D;
  const int d = (int)(gid - j * D);
  const long r = seg[j];
  const long pos = perm ? perm[j] : j;
  const int cnt = rowptr[r + 1] - rowptr[r];
  float g = base ? base[pos * ld + d] : 0.f;
  for (int s = 0; s < ops.n; ++s) {
    const float v = d_out[r * ld_out + (long)s * D + d];
    switch (ops.op[s]) {
      case HGN_OP_SUM: g += v; break;
      case HGN_OP_MEAN: g += v / (float)(cnt > 0 ? cnt : 1); break;
      case HGN_OP_MAX: g += argmax[r * D + d] == (int)j ? v : 0.f; break;
      default: g += argmin[r * D + d] == (int)j ? v : 0.f; break;
    }
  }
  d_data[pos * ld + d] = g;
}

// ----------------------------------------------------------------------------------------------------------
// 'std' (src/util.py:129-130 -> torch_scatter.scatter_std, unbiased): thread per (segment, column), two sweeps over the segment's
// rows -- the mean first, then the squared deviations (the second sweep is served by the caches).  Off the hot path.
// ----------------------------------------------------------------------------------------------------------
__global__ void seg_std_fwd_kernel(const float* __restrict__ data, long ld, int D, const int* __restrict__ perm,
                                   const int* __restrict__ rowptr, long N, float* __restrict__ out, long ld_out,
                                   float* __restrict__ mean_out) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N * D) return;
  const long n = gid / D;
  const int d = (int)(gid - n * D);
  const int beg = rowptr[n], end = rowptr[n + 1];
  float sum = 0.f;
  for (int j = beg; j < end; ++j) sum += data[(perm ? (long)perm[j] : (long)j) * ld + d];
  const int cnt = end - beg;
  const float mean = sum / (float)(cnt > 0 ? cnt : 1);
  float ssq = 0.f;
  for (int j = beg; j < end; ++j) {
    const float dev = data[(perm ? (long)perm[j] : (long)j) * ld + d] - mean;
    ssq += dev * dev;
  }
  const float cu = (float)(cnt > 2 ? cnt - 1 : 1);          // clamp(clamp(count, 1) - 1, 1)
  out[n * ld_out + d] = sqrtf(ssq / (cu + 1e-6f));
  if (mean_out) mean_out[n * ld_out + d] = mean;
}

// autograd of the composite: d std / d x = (x - mean) / (std (count_u + 1e-6)); 0 / 0 = NaN for a segment without variance, like the wheel
__global__ void seg_std_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ out, const float* __restrict__ mean,
                                   long ld_out, const float* __restrict__ data, long ld, int D, const int* __restrict__ perm,
                                   const int* __restrict__ seg, const int* __restrict__ rowptr, long E,
                                   float* __restrict__ d_data, long ld_d) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= E * D) return;
  const long j = gid / D;
  const int d = (int)(gid - j * D);
  const long r = seg[j];
  const long pos = perm ? perm[j] : j;
  const int cnt = rowptr[r + 1] - rowptr[r];
  const float cu = (float)(cnt > 2 ? cnt - 1 : 1);
  const float x = data[pos * ld + d];
  d_data[pos * ld_d + d] = d_out[r * ld_out + d] * (x - mean[r * ld_out + d]) / (out[r * ld_out + d] * (cu + 1e-6f));
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static int sort_bits(int64_t N) {
  int b = 1;
  while (b < 31 && ((int64_t)1 << b) < N) ++b;
  return b;
}

}  // namespace hgn

using namespace hgn;

static size_t cub_temp_bytes(int64_t E, int64_t N) {
  size_t t = 0;
  (void)hipcub::DeviceRadixSort::SortPairs<int, int>(nullptr, t, nullptr, nullptr, nullptr, nullptr, (int)E, 0, sort_bits(N));
  return t;
}

extern "C" int hgn_csr_workspace_bytes(int64_t E, int64_t N, size_t* bytes) {
  if (!bytes || E < 0 || N < 0 || E > 0x7fffffff || N > 0x7ffffffe) return hgn_fail(HGN_E_INVALID, "hgn_csr_workspace_bytes: bad size");
  *bytes = 256 + 2 * align_up((size_t)E * 4, 256) + align_up(cub_temp_bytes(E, N), 256) + 256;
  return HGN_OK;
}

extern "C" int hgn_csr_build(const int64_t* ids, int64_t E, int64_t N, int32_t* perm, int32_t* seg, int32_t* rowptr,
                             void* workspace, size_t ws_bytes, int32_t* max_rows, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  size_t need = 0;
  if (hgn_csr_workspace_bytes(E, N, &need) != HGN_OK) return HGN_E_INVALID;
  if (!rowptr || (E > 0 && (!ids || !perm || !seg)) || !workspace || ws_bytes < need)
    return hgn_fail(HGN_E_INVALID, "hgn_csr_build: null pointer or workspace too small");
  ProfScope ps(10, (double)E, stream);
  char* w = (char*)workspace;
  int* flag = (int*)w;
  int* keys = (int*)(w + 256);
  int* vals = (int*)(w + 256 + align_up((size_t)E * 4, 256));
  void* temp = w + 256 + 2 * align_up((size_t)E * 4, 256);
  size_t temp_bytes = cub_temp_bytes(E, N);
  if (hipMemsetAsync(flag, 0, 16, stream) != hipSuccess) return hgn_check_launch("hgn_csr_build memset");
  if (E > 0) {
    hipLaunchKernelGGL(csr_prepare_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, ids, (long)E, (long)N,
                       keys, vals, flag);
    if (hipcub::DeviceRadixSort::SortPairs<int, int>(temp, temp_bytes, keys, seg, vals, perm, (int)E, 0, sort_bits(N),
                                                     stream) != hipSuccess)
      return hgn_check_launch("hgn_csr_build sort");
  }
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, stream, seg, (long)E, (long)N,
                     rowptr);
  if (max_rows && N > 0)
    hipLaunchKernelGGL(csr_max_rows_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, rowptr, (long)N, flag);
  int host_flag[2] = {0, 0};
  if (hipMemcpyAsync(host_flag, flag, 2 * sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return hgn_check_launch("hgn_csr_build readback");
  if (host_flag[0]) return hgn_fail(HGN_E_RANGE, "hgn_csr_build: segment id outside [0, num_segments)");
  if (max_rows) *max_rows = host_flag[1];
  return hgn_check_launch("hgn_csr_build");
}

extern "C" int hgn_index_fingerprint(const int64_t* a, const int64_t* b, int64_t n, uint64_t* out_dev, uint64_t* out_host,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!out_dev || n < 0 || (n > 0 && !a)) return hgn_fail(HGN_E_INVALID, "hgn_index_fingerprint: bad argument");
  if (hipMemsetAsync(out_dev, 0, 16, stream) != hipSuccess) return hgn_check_launch("hgn_index_fingerprint memset");
  if (n > 0) {
    long blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(index_fingerprint_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, b, (long)n,
                       reinterpret_cast<unsigned long long*>(out_dev));
  }
  if (out_host) {
    if (hipMemcpyAsync(out_host, out_dev, 16, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
      return hgn_check_launch("hgn_index_fingerprint readback");
  }
  return hgn_check_launch("hgn_index_fingerprint");
}

extern "C" int hgn_narrow_gather_i64(const int64_t* src, const int32_t* perm, int64_t n, int32_t* dst, void* stream) {
  if (n == 0) return HGN_OK;
  if (!src || !dst || n < 0) return hgn_fail(HGN_E_INVALID, "hgn_narrow_gather_i64: bad argument");
  hipLaunchKernelGGL(narrow_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, perm,
                     (long)n, dst);
  return hgn_check_launch("hgn_narrow_gather_i64");
}

static int make_ops(const int32_t* ops, int n_ops, Ops* o) {
  if (!ops || n_ops < 1 || n_ops > MAXOPS) return hgn_fail(HGN_E_INVALID, "segment_reduce: need 1..4 ops");
  o->n = n_ops;
  for (int i = 0; i < MAXOPS; ++i) {
    o->op[i] = i < n_ops ? ops[i] : 0;
    if (o->op[i] < 0 || o->op[i] > 3) return hgn_fail(HGN_E_INVALID, "Invalid operation type!");
  }
  return HGN_OK;
}

extern "C" int hgn_segment_reduce_fwd(const float* data, int64_t ld, int D, const int32_t* perm, const int32_t* rowptr,
                                      int64_t N, const int32_t* ops, int n_ops, float* out, int64_t ld_out, int32_t* argmax,
                                      int32_t* argmin, void* stream) {
  Ops o;
  if (make_ops(ops, n_ops, &o) != HGN_OK) return HGN_E_INVALID;
  if (N == 0) return HGN_OK;
  if (!rowptr || !out || N < 0 || D < 1 || ld < D || ld_out < (int64_t)n_ops * D)
    return hgn_fail(HGN_E_INVALID, "hgn_segment_reduce_fwd: bad argument");
  bool need_arg = false;
  for (int i = 0; i < n_ops; ++i) need_arg |= ops[i] >= HGN_OP_MAX;
  (void)need_arg;
  ProfScope ps(g_prof_tag == 2 ? 12 : 5, (double)N, (hipStream_t)stream);
  const bool fast = D == 128 && (ld & 3) == 0 && (ld_out & 3) == 0 && ((uintptr_t)data & 15) == 0 && ((uintptr_t)out & 15) == 0;
  if (fast) {
    // few rows (a hyper part: 2 048 rows = 64 workgroups of 256 threads on 256 CUs): two rows per workgroup instead of eight
    if (N * 32 <= 256L * 256)
      hipLaunchKernelGGL(seg_fwd128_kernel<true>, dim3((unsigned)((N * 32 + 63) / 64)), dim3(64), 0, (hipStream_t)stream, data,
                         (long)ld, perm, rowptr, (long)N, o, out, (long)ld_out, argmax, argmin);
    else
      hipLaunchKernelGGL(seg_fwd128_kernel<false>, dim3((unsigned)((N * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, data,
                         (long)ld, perm, rowptr, (long)N, o, out, (long)ld_out, argmax, argmin);
  } else {
    hipLaunchKernelGGL(seg_fwd_generic_kernel, dim3((unsigned)((N * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, data,
                       (long)ld, D, perm, rowptr, (long)N, o, out, (long)ld_out, argmax, argmin);
  }
  return hgn_check_launch("hgn_segment_reduce_fwd");
}

extern "C" int hgn_segment_reduce_bwd_sorted(const float* d_out, int64_t ld_out, const int32_t* rowptr, int64_t N, const int32_t* ops, int n_ops,
                                             const int32_t* argmax, const int32_t* argmin, const float* base, float* d_data, int64_t ld,
                                             void* stream) {
  Ops o;
  if (make_ops(ops, n_ops, &o) != HGN_OK) return HGN_E_INVALID;
  if (N == 0) return HGN_OK;
  if (!d_out || !rowptr || !d_data || N < 0 || ld < 128 || ld_out < (int64_t)n_ops * 128 || (ld & 3) || (ld_out & 3) || ((uintptr_t)d_data & 15) ||
      ((uintptr_t)d_out & 15) || (base && ((uintptr_t)base & 15)))
    return hgn_fail(HGN_E_INVALID, "hgn_segment_reduce_bwd_sorted: 128-wide rows, 16-byte aligned, leading dimensions multiples of 4");
  for (int i = 0; i < n_ops; ++i)
    if ((ops[i] == HGN_OP_MAX && !argmax) || (ops[i] == HGN_OP_MIN && !argmin))
      return hgn_fail(HGN_E_INVALID, "hgn_segment_reduce_bwd_sorted: max/min need the saved arg index");
  ProfScope ps(6, (double)N, (hipStream_t)stream);
  hipLaunchKernelGGL(seg_bwd128_sorted_kernel, dim3((unsigned)((N * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_out, (long)ld_out, rowptr,
                     (long)N, o, argmax, argmin, base, d_data, (long)ld);
  return hgn_check_launch("hgn_segment_reduce_bwd_sorted");
}

extern "C" int hgn_segment_sum_pair(const float* data, int64_t ld, const int32_t* rowptr_a, const int32_t* perm_b, const int32_t* rowptr_b,
                                    int64_t N, float* out_a, int64_t ld_a, float* out_b, int64_t ld_b, void* stream) {
  if (N == 0) return HGN_OK;
  // (data / perm_b may be null when there are no rows: every segment is then empty and neither is dereferenced)
  if (!rowptr_a || !rowptr_b || !out_a || !out_b || N < 0 || ld < 128 || ld_a < 128 || ld_b < 128 || (ld & 3) || (ld_a & 3) ||
      (ld_b & 3) || ((uintptr_t)data & 15) || ((uintptr_t)out_a & 15) || ((uintptr_t)out_b & 15))
    return hgn_fail(HGN_E_INVALID, "hgn_segment_sum_pair: 128-wide rows, 16-byte aligned, leading dimensions multiples of 4");
  ProfScope ps(5, (double)N, (hipStream_t)stream);
  hipLaunchKernelGGL(seg_sum_pair128_kernel, dim3((unsigned)((N * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, data, (long)ld, rowptr_a,
                     perm_b, rowptr_b, (long)N, out_a, (long)ld_a, out_b, (long)ld_b);
  return hgn_check_launch("hgn_segment_sum_pair");
}

extern "C" int hgn_segment_reduce_bwd(const float* d_out, int64_t ld_out, int D, const int32_t* perm, const int32_t* seg,
                                      const int32_t* rowptr, int64_t E, const int32_t* ops, int n_ops, const int32_t* argmax,
                                      const int32_t* argmin, const float* base, float* d_data, int64_t ld, void* stream) {
  Ops o;
  if (make_ops(ops, n_ops, &o) != HGN_OK) return HGN_E_INVALID;
  if (E == 0) return HGN_OK;
  if (!d_out || !seg || !rowptr || !d_data || E < 0 || D < 1) return hgn_fail(HGN_E_INVALID, "hgn_segment_reduce_bwd: bad argument");
  for (int i = 0; i < n_ops; ++i)
    if ((ops[i] == HGN_OP_MAX && !argmax) || (ops[i] == HGN_OP_MIN && !argmin))
      return hgn_fail(HGN_E_INVALID, "hgn_segment_reduce_bwd: max/min need the saved arg index");
  ProfScope ps(6, (double)E, (hipStream_t)stream);
  const bool fast = D == 128 && (ld & 3) == 0 && (ld_out & 3) == 0 && ((uintptr_t)d_data & 15) == 0 &&
                    ((uintptr_t)d_out & 15) == 0 && (!base || ((uintptr_t)base & 15) == 0);
  if (fast) {
    hipLaunchKernelGGL(seg_bwd128_kernel, dim3((unsigned)((E * 32 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_out,
                       (long)ld_out, perm, seg, rowptr, (long)E, o, argmax, argmin, base, d_data, (long)ld);
  } else {
    hipLaunchKernelGGL(seg_bwd_generic_kernel, dim3((unsigned)((E * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_out,
                       (long)ld_out, D, perm, seg, rowptr, (long)E, o, argmax, argmin, base, d_data, (long)ld);
  }
  return hgn_check_launch("hgn_segment_reduce_bwd");
}

extern "C" int hgn_segment_std_fwd(const float* data, int64_t ld, int D, const int32_t* perm, const int32_t* rowptr, int64_t N,
                                   float* out, int64_t ld_out, float* mean, void* stream) {
  if (N == 0) return HGN_OK;
  if (!data || !rowptr || !out || N < 0 || D < 1 || ld < D || ld_out < D) return hgn_fail(HGN_E_INVALID, "hgn_segment_std_fwd: bad argument");
  ProfScope ps(5, (double)N, (hipStream_t)stream);
  hipLaunchKernelGGL(seg_std_fwd_kernel, dim3((unsigned)((N * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, data, (long)ld, D,
                     perm, rowptr, (long)N, out, (long)ld_out, mean);
  return hgn_check_launch("hgn_segment_std_fwd");
}

extern "C" int hgn_segment_std_bwd(const float* d_out, const float* out, const float* mean, int64_t ld_out, const float* data,
                                   int64_t ld, int D, const int32_t* perm, const int32_t* seg, const int32_t* rowptr, int64_t E,
                                   float* d_data, int64_t ld_d, void* stream) {
  if (E == 0) return HGN_OK;
  if (!d_out || !out || !mean || !data || !seg || !rowptr || !d_data || E < 0 || D < 1 || ld < D || ld_d < D || ld_out < D)
    return hgn_fail(HGN_E_INVALID, "hgn_segment_std_bwd: bad argument");
  ProfScope ps(6, (double)E, (hipStream_t)stream);
  hipLaunchKernelGGL(seg_std_bwd_kernel, dim3((unsigned)((E * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_out, out, mean,
                     (long)ld_out, data, (long)ld, D, perm, seg, rowptr, (long)E, d_data, (long)ld_d);
  return hgn_check_launch("hgn_segment_std_bwd");
}
