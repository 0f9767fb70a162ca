// Frame -> graph features (include/hgn_features.h): cell edges, relative-position edge features, node features and
// the online normaliser.  All of it is HBM-bound byte / index work over [E, <=8] and [N, <=12] rows: flat,
// fully coalesced passes, fp64 accumulation for the statistics, fixed-order reductions (deterministic).
#include <hipcub/hipcub.hpp>
#include "hgn_host.h"
#include "../../include/hgn_features.h"

// The reference evaluates these features op by op in fp32 (every product and sum rounded): no fma contraction in
// this translation unit, so that e.g. E[x^2] - mean^2 cancels exactly the way normalizer.py:70 does.
#pragma clang fp contract(off)

namespace hgn {

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// ----------------------------------------------------------------------------------------------------------
// util.py:50-89 -- unique undirected cell edges, two-way
// ----------------------------------------------------------------------------------------------------------
__global__ void cell_keys_kernel(const int64_t* __restrict__ cells, long n_cells, int verts,
                                 unsigned long long* __restrict__ keys, int* __restrict__ flag) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cells * verts) return;
  const long c = i / verts;
  const int k = (int)(i - c * verts);
  const int64_t u = cells[c * verts + k];
  const int64_t v = cells[c * verts + (k + 1 == verts ? 0 : k + 1)];
  if (u < 0 || v < 0 || u > 0x7fffffffLL || v > 0x7fffffffLL) { atomicOr(flag, 1); keys[i] = 0; return; }
  const unsigned long long hi = (unsigned long long)(u > v ? u : v), lo = (unsigned long long)(u > v ? v : u);
  keys[i] = (hi << 32) | lo;
}

__global__ void two_way_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ n_unique,
                               int64_t* __restrict__ senders, int64_t* __restrict__ receivers) {
  const long n = *n_unique;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  const int64_t hi = (int64_t)(k >> 32), lo = (int64_t)(k & 0xffffffffULL);
  senders[i] = hi; receivers[i] = lo;
  senders[n + i] = lo; receivers[n + i] = hi;
}

// ----------------------------------------------------------------------------------------------------------
// relative-position edge features: one thread per edge
// ----------------------------------------------------------------------------------------------------------
__global__ void rel_edge_kernel(const float* __restrict__ a, long lda, int da, const float* __restrict__ b, long ldb,
                                int db, long n_rows, const int64_t* __restrict__ snd, const int64_t* __restrict__ rcv,
                                long E, float* __restrict__ feat, long ldf, float* __restrict__ len_a) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t s = snd[e], r = rcv[e];
  if (s < 0 || r < 0 || s >= n_rows || r >= n_rows) return;
  float ra[3], rb[3];
  float qa = 0.f, qb = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ra[i] = i < da ? a[s * lda + i] - a[r * lda + i] : 0.f;
    qa += ra[i] * ra[i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    rb[i] = i < db ? b[s * ldb + i] - b[r * ldb + i] : 0.f;
    qb += rb[i] * rb[i];
  }
  const float na = sqrtf(qa);
  if (len_a) len_a[e] = na;
  if (feat) {
    float* o = feat + e * ldf;
    for (int i = 0; i < da; ++i) o[i] = ra[i];
    o[da] = na;
    if (db > 0) {
      for (int i = 0; i < db; ++i) o[da + 1 + i] = rb[i];
      o[da + 1 + db] = sqrtf(qb);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------
// node features: flat over [N, d + n_classes]
// ----------------------------------------------------------------------------------------------------------
__global__ void node_feat_kernel(const float* __restrict__ cur, const float* __restrict__ prev, long ld, int d,
                                 const int64_t* __restrict__ node_type, long ldt, const int* __restrict__ map,
                                 int map_len, int n_classes, int vel_first, int vel_mask_type, long N,
                                 float* __restrict__ out, long ldo) {
  const int W = d + n_classes;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * W) return;
  const long n = i / W;
  const int c = (int)(i - n * W);
  const int64_t t = node_type[n * ldt];
  const int vc = vel_first ? c : c - n_classes;      // velocity column or <0 / >=d
  float v;
  if (vc >= 0 && vc < d) {
    v = cur ? cur[n * ld + vc] - (prev ? prev[n * ld + vc] : 0.f) : 0.f;
    if (vel_mask_type >= 0 && t != vel_mask_type) v = 0.f;
  } else {
    const int oc = vel_first ? c - d : c;
    int64_t cls = t;
    if (map) cls = (t >= 0 && t < map_len) ? map[t] : -1;
    v = cls == oc ? 1.f : 0.f;
  }
  out[n * ldo + c] = v;
}

// ----------------------------------------------------------------------------------------------------------
// normaliser
// ----------------------------------------------------------------------------------------------------------
constexpr int ST = 256;          // threads per block of the statistics pass
constexpr int MAXBLK = 1024;

// Flat pass: the active threads of the grid are a multiple of F, so a thread always sees one column.
__global__ void col_stats_kernel(const float* __restrict__ x, long n_elem, int F, int active_per_block,
                                 double* __restrict__ part /*[gridDim.x][2F]*/) {
  __shared__ double s1[ST], s2[ST];
  const int t = threadIdx.x;
  double a = 0.0, q = 0.0;
  if (t < active_per_block) {
    const long stride = (long)gridDim.x * active_per_block;
    for (long i = (long)blockIdx.x * active_per_block + t; i < n_elem; i += stride) {
      const double v = (double)x[i];
      a += v; q += v * v;
    }
  }
  s1[t] = a; s2[t] = q;
  __syncthreads();
  if (t < F) {                  // column of thread u is (blockIdx.x*active + u) % F == u % F (active % F == 0)
    double A = 0.0, Q = 0.0;
    for (int u = t; u < active_per_block; u += F) { A += s1[u]; Q += s2[u]; }
    part[(long)blockIdx.x * 2 * F + t] = A;
    part[(long)blockIdx.x * 2 * F + F + t] = Q;
  }
}

// One block: 256 threads = G groups x 2F columns; a group strides over the block partials, groups are folded in
// fixed order (deterministic).
__global__ void col_stats_final_kernel(const double* __restrict__ part, int nblk, int F, float* __restrict__ batch) {
  __shared__ double sh[ST];
  const int t = threadIdx.x, W = 2 * F, G = ST / W;
  const int c = t % W, g = t / W;
  double A = 0.0;
  if (g < G)
    for (int b = g; b < nblk; b += G) A += part[(long)b * W + c];
  sh[t] = A;
  __syncthreads();
  if (t < W) {
    double S = 0.0;
    for (int k = 0; k < G; ++k) S += sh[k * W + t];
    batch[t] = (float)S;
  }
}

__global__ void normalizer_update_kernel(float* acc_sum, float* acc_sumsq, float* acc_count, float* num_acc,
                                         const float* __restrict__ batch, const float* __restrict__ count, int F,
                                         float max_acc) {
  const int t = threadIdx.x;
  const bool go = *num_acc < max_acc;
  __syncthreads();
  if (!go) return;
  if (t < F) { acc_sum[t] += batch[t]; acc_sumsq[t] += batch[F + t]; }
  if (t == 0) { *acc_count += *count; *num_acc += 1.f; }
}

__global__ void normalize_kernel(const float* __restrict__ x, long n_elem, int F, const float* __restrict__ acc_sum,
                                 const float* __restrict__ acc_sumsq, const float* __restrict__ acc_count, float eps,
                                 int inverse, float* __restrict__ out) {
  __shared__ float mean[HGN_MAX_FEATURE_WIDTH], sd[HGN_MAX_FEATURE_WIDTH];
  if ((int)threadIdx.x < F) {
    const float safe = fmaxf(*acc_count, 1.f);
    const float m = acc_sum[threadIdx.x] / safe;
    const float s = sqrtf(fabsf(acc_sumsq[threadIdx.x] / safe - m * m));   // no fma (file-wide): mean^2 is rounded
    mean[threadIdx.x] = m;
    sd[threadIdx.x] = fmaxf(s, eps);
  }
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_elem) return;
  const int c = (int)(i % F);
  out[i] = inverse ? x[i] * sd[c] + mean[c] : (x[i] - mean[c]) / sd[c];
}

__global__ void lincomb3_kernel(const float* __restrict__ a, float ca, const float* __restrict__ b, float cb,
                                const float* __restrict__ c, float cc, long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = ca * a[i] + cb * b[i];          // contraction is off in this file: three roundings
  if (c) v = v + cc * c[i];
  out[i] = v;
}

// ----------------------------------------------------------------------------------------------------------
// world edges by radius: one wavefront per sender row, receivers in chunks of 64 (ballot keeps ascending order)
// ----------------------------------------------------------------------------------------------------------
template <bool FILL>
__global__ void radius_edges_kernel(const float* __restrict__ pos, long ld, int d, const int64_t* __restrict__ node_type,
                                    long ldt, long N, float radius, int sender_type, int receiver_type,
                                    const int* __restrict__ nbr_rowptr, const int* __restrict__ nbr,
                                    int* __restrict__ counts, const int* __restrict__ offsets,
                                    int64_t* __restrict__ senders, int64_t* __restrict__ receivers) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (s >= N) return;
  const bool active = sender_type < 0 || node_type[s * ldt] == sender_type;      // uniform per wave
  if (!active) { if (!FILL && lane == 0) counts[s] = 0; return; }
  float ps[3];
  for (int i = 0; i < 3; ++i) ps[i] = i < d ? pos[s * ld + i] : 0.f;
  const int nb0 = nbr_rowptr ? nbr_rowptr[s] : 0, nb1 = nbr_rowptr ? nbr_rowptr[s + 1] : 0;
  long base = FILL ? offsets[s] : 0;
  int total = 0;
  for (long r0 = 0; r0 < N; r0 += 64) {
    const long r = r0 + lane;
    bool hit = false;
    if (r < N && r != s && (receiver_type < 0 || node_type[r * ldt] == receiver_type)) {
      float q = 0.f;
      for (int i = 0; i < 3; ++i) {
        const float df = i < d ? ps[i] - pos[r * ld + i] : 0.f;
        q = q + df * df;
      }
      hit = sqrtf(q) < radius;
      if (hit)
        for (int k = nb0; k < nb1; ++k)
          if (nbr[k] == (int)r) { hit = false; break; }
    }
    const unsigned long long m = __ballot(hit);
    if (FILL && hit) {
      const int before = __popcll(m & ((1ULL << lane) - 1ULL));
      senders[base + before] = s;
      receivers[base + before] = r;
    }
    const int c = __popcll(m);
    base += c;
    total += c;
  }
  if (!FILL && lane == 0) counts[s] = total;
}

// ----------------------------------------------------------------------------------------------------------
// balanced Forman curvature (SDRF): one wavefront per edge / per candidate pair
// ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_count_max(int& cnt, float& mx) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    cnt += __shfl_xor(cnt, o);
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
}

// closed form of ricci.py:185-191 / 256-262 in the reference kernel's typing: fp64 expression, fp32 stores
__device__ __forceinline__ float forman_value(double dmax, double dmin, double a2, double a, int sharp, double lam) {
  float c = (float)(((2 / dmax) + (2 / dmin) - 2) + (2 / dmax + 1 / dmin) * a2 * a);
  if (lam > 0) c = (float)((double)c + (double)sharp / (dmax * lam));
  return c;
}

__global__ void forman_curvature_kernel(const float* __restrict__ A, const float* __restrict__ A2,
                                        const float* __restrict__ d_in, const float* __restrict__ d_out, long N,
                                        const int* __restrict__ ei, const int* __restrict__ ej, long nnz,
                                        float* __restrict__ C) {
  const int lane = threadIdx.x & 63;
  const long e = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (e >= nnz) return;
  const long i = ei[e], j = ej[e];
  const float aij = A[i * N + j];
  if (aij == 0.f) { if (lane == 0) C[i * N + j] = 0.f; return; }
  const float di = d_in[i], dj = d_out[j];
  const float dmax = di > dj ? di : dj, dmin = di > dj ? dj : di;
  if (dmax * dmin == 0.f) { if (lane == 0) C[i * N + j] = 0.f; return; }
  int cnt = 0;
  float mx = 0.f;
  for (long k = lane; k < N; k += 64) {
    const float aik = A[i * N + k], akj = A[k * N + j];
    const float t1 = akj * (A2[i * N + k] - aik) * aij;
    const float t2 = aik * (A2[k * N + j] - akj) * aij;
    if (t1 > 0.f) { ++cnt; mx = fmaxf(mx, t1); }
    if (t2 > 0.f) { ++cnt; mx = fmaxf(mx, t2); }
  }
  wave_count_max(cnt, mx);
  if (lane == 0) C[i * N + j] = forman_value(dmax, dmin, A2[i * N + j], aij, cnt, mx);
}

__global__ void forman_post_delta_kernel(const float* __restrict__ A, const float* __restrict__ A2, float d_in_x0,
                                         float d_out_y0, long N, int x, int y, const int* __restrict__ i_nb, int dim_i,
                                         const int* __restrict__ j_nb, int dim_j, float* __restrict__ D) {
  const int lane = threadIdx.x & 63;
  const long p = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= (long)dim_i * dim_j) return;
  const int I = (int)(p / dim_j), J = (int)(p % dim_j);
  const long i = i_nb[I], j = j_nb[J];
  if (i == j || A[i * N + j] != 0.f) { if (lane == 0) D[p] = -1000.f; return; }
  double dx = d_in_x0, dy = d_out_y0;
  if (j == x) dx += 1; else if (i == y) dy += 1;
  if (dx * dy == 0) { if (lane == 0) D[p] = 0.f; return; }
  const double dmax = dx > dy ? dx : dy, dmin = dx > dy ? dy : dx;
  const float axy = A[(long)x * N + y];
  double a2xy = A2[(long)x * N + y];
  if (x == i && A[j * N + y] != 0.f) a2xy += A[j * N + y];
  else if (y == j && A[(long)x * N + i] != 0.f) a2xy += A[(long)x * N + i];
  const float ajy = A[j * N + y], axi = A[(long)x * N + i];
  int cnt = 0;
  float mx = 0.f;
  for (long z = lane; z < N; z += 64) {
    float azy = A[z * N + y], axz = A[(long)x * N + z];
    float a2zy = A2[z * N + y], a2xz = A2[(long)x * N + z];
    if (z == i && y == j) azy += 1.f;
    if (x == i && z == j) axz += 1.f;
    if (z == i && ajy != 0.f) a2zy += ajy;
    if (x == i && A[j * N + z] != 0.f) a2xz += A[j * N + z];
    if (y == j && A[z * N + i] != 0.f) a2zy += A[z * N + i];
    if (z == j && axi != 0.f) a2xz += axi;
    const float t1 = azy * (a2xz - axz) * axy;          // small integers: exact in fp32
    const float t2 = axz * (a2zy - azy) * axy;
    if (t1 > 0.f) { ++cnt; mx = fmaxf(mx, t1); }
    if (t2 > 0.f) { ++cnt; mx = fmaxf(mx, t2); }
  }
  wave_count_max(cnt, mx);
  if (lane == 0) D[p] = forman_value(dmax, dmin, a2xy, axy, cnt, mx);
}

static int stats_blocks(int64_t rows, int F, int* active) {
  *active = (ST / F) * F;
  const int64_t n = rows * F;
  int64_t nb = (n + (int64_t)(*active) * 16 - 1) / ((int64_t)(*active) * 16);
  if (nb < 1) nb = 1;
  if (nb > MAXBLK) nb = MAXBLK;
  return (int)nb;
}

}  // namespace hgn

using namespace hgn;

static size_t cells_cub_bytes(int64_t n) {
  size_t t1 = 0, t2 = 0;
  (void)hipcub::DeviceRadixSort::SortKeys<unsigned long long>(nullptr, t1, nullptr, nullptr, (int)n, 0, 64);
  (void)hipcub::DeviceSelect::Unique<unsigned long long*, unsigned long long*, int*>(nullptr, t2, nullptr, nullptr, nullptr, (int)n);
  return t1 > t2 ? t1 : t2;
}

extern "C" int hgn_cells_to_edges_workspace_bytes(int64_t n_cells, int verts, size_t* bytes) {
  if (!bytes || n_cells < 0 || (verts != 3 && verts != 4) || n_cells * verts > 0x7fffffff)
    return hgn_fail(HGN_E_INVALID, "hgn_cells_to_edges_workspace_bytes: bad size (verts must be 3 or 4)");
  const int64_t n = n_cells * verts;
  *bytes = 256 + 3 * up256((size_t)n * 8) + up256(cells_cub_bytes(n)) + 256;
  return HGN_OK;
}

extern "C" int hgn_cells_to_edges(const int64_t* cells, int64_t n_cells, int verts, int64_t* senders, int64_t* receivers,
                                  int64_t* n_unique, void* workspace, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  size_t need = 0;
  if (hgn_cells_to_edges_workspace_bytes(n_cells, verts, &need) != HGN_OK) return HGN_E_INVALID;
  if (!n_unique || !workspace || ws_bytes < need || (n_cells > 0 && (!cells || !senders || !receivers)))
    return hgn_fail(HGN_E_INVALID, "hgn_cells_to_edges: null pointer or workspace too small");
  *n_unique = 0;
  if (n_cells == 0) return HGN_OK;
  ProfScope ps(13, (double)n_cells, stream);
  const int64_t n = n_cells * verts;
  char* w = (char*)workspace;
  int* flag = (int*)w;                   // [0] range flag, [1] number of unique keys
  unsigned long long* k0 = (unsigned long long*)(w + 256);
  unsigned long long* k1 = (unsigned long long*)(w + 256 + up256((size_t)n * 8));
  unsigned long long* k2 = (unsigned long long*)(w + 256 + 2 * up256((size_t)n * 8));
  void* temp = w + 256 + 3 * up256((size_t)n * 8);
  size_t temp_bytes = cells_cub_bytes(n);
  if (hipMemsetAsync(flag, 0, 16, stream) != hipSuccess) return hgn_check_launch("hgn_cells_to_edges memset");
  hipLaunchKernelGGL(cell_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cells, (long)n_cells, verts,
                     k0, flag);
  size_t tb = temp_bytes;
  if (hipcub::DeviceRadixSort::SortKeys<unsigned long long>(temp, tb, k0, k1, (int)n, 0, 64, stream) != hipSuccess)
    return hgn_check_launch("hgn_cells_to_edges sort");
  tb = temp_bytes;
  if (hipcub::DeviceSelect::Unique(temp, tb, k1, k2, flag + 1, (int)n, stream) != hipSuccess)
    return hgn_check_launch("hgn_cells_to_edges unique");
  hipLaunchKernelGGL(two_way_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, k2, flag + 1, senders,
                     receivers);
  int host[2] = {0, 0};
  if (hipMemcpyAsync(host, flag, 2 * sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return hgn_check_launch("hgn_cells_to_edges readback");
  if (host[0]) return hgn_fail(HGN_E_RANGE, "hgn_cells_to_edges: vertex id outside [0, 2^31)");
  *n_unique = host[1];
  return hgn_check_launch("hgn_cells_to_edges");
}

extern "C" int hgn_rel_edge_features(const float* a, int64_t lda, int da, const float* b, int64_t ldb, int db,
                                     int64_t n_rows, const int64_t* senders, const int64_t* receivers, int64_t E,
                                     float* feat, int64_t ldf, float* len_a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (E < 0 || n_rows < 0 || da < 1 || da > 3 || db < 0 || db > 3 || lda < da || (db > 0 && ldb < db))
    return hgn_fail(HGN_E_INVALID, "hgn_rel_edge_features: widths must satisfy 1<=da<=3, 0<=db<=3, ld >= width");
  const int W = da + 1 + (db > 0 ? db + 1 : 0);
  if (feat && ldf < W) return hgn_fail(HGN_E_INVALID, "hgn_rel_edge_features: ldf smaller than the feature row");
  if (E == 0) return HGN_OK;
  if (!a || (db > 0 && !b) || !senders || !receivers || (!feat && !len_a))
    return hgn_fail(HGN_E_INVALID, "hgn_rel_edge_features: null pointer");
  ProfScope ps(13, (double)E, stream);
  hipLaunchKernelGGL(rel_edge_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, stream, a, (long)lda, da, b,
                     (long)ldb, db, (long)n_rows, senders, receivers, (long)E, feat, (long)ldf, len_a);
  return hgn_check_launch("hgn_rel_edge_features");
}

extern "C" int hgn_node_features(const float* cur, const float* prev, int64_t ld, int d, const int64_t* node_type,
                                 int64_t ldt, const int32_t* map, int map_len, int n_classes, int vel_first,
                                 int vel_mask_type, int64_t N, float* out, int64_t ldo, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N < 0 || d < 0 || n_classes < 0 || d + n_classes < 1 || d + n_classes > HGN_MAX_FEATURE_WIDTH || ldo < d + n_classes ||
      (cur && ld < d) || ldt < 1 || (map && map_len < 1))
    return hgn_fail(HGN_E_INVALID, "hgn_node_features: bad widths / strides");
  if (N == 0) return HGN_OK;
  if (!node_type || !out || (prev && !cur)) return hgn_fail(HGN_E_INVALID, "hgn_node_features: null pointer");
  ProfScope ps(13, (double)N, stream);
  const int64_t n = N * (d + n_classes);
  hipLaunchKernelGGL(node_feat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cur, prev, (long)ld, d,
                     node_type, (long)ldt, map, map_len, n_classes, vel_first, vel_mask_type, (long)N, out, (long)ldo);
  return hgn_check_launch("hgn_node_features");
}

extern "C" int hgn_col_stats_workspace_bytes(int64_t rows, int F, size_t* bytes) {
  if (!bytes || rows < 0 || F < 1 || F > HGN_MAX_FEATURE_WIDTH)
    return hgn_fail(HGN_E_INVALID, "hgn_col_stats_workspace_bytes: bad size (1 <= F <= 32)");
  *bytes = (size_t)MAXBLK * 2 * F * sizeof(double);
  return HGN_OK;
}

extern "C" int hgn_col_stats(const float* x, int64_t rows, int F, float* batch, void* workspace, size_t ws_bytes,
                             void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  size_t need = 0;
  if (hgn_col_stats_workspace_bytes(rows, F, &need) != HGN_OK) return HGN_E_INVALID;
  if (!batch || !workspace || ws_bytes < need || (rows > 0 && !x))
    return hgn_fail(HGN_E_INVALID, "hgn_col_stats: null pointer or workspace too small");
  ProfScope ps(13, (double)rows, stream);
  int active = 0;
  const int nblk = stats_blocks(rows, F, &active);
  hipLaunchKernelGGL(col_stats_kernel, dim3(nblk), dim3(ST), 0, stream, x, (long)(rows * F), F, active, (double*)workspace);
  hipLaunchKernelGGL(col_stats_final_kernel, dim3(1), dim3(ST), 0, stream, (const double*)workspace, nblk, F, batch);
  return hgn_check_launch("hgn_col_stats");
}

extern "C" int hgn_normalizer_update(float* acc_sum, float* acc_sumsq, float* acc_count, float* num_acc,
                                     const float* batch, const float* count, int F, float max_acc, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (F < 1 || F > HGN_MAX_FEATURE_WIDTH || !acc_sum || !acc_sumsq || !acc_count || !num_acc || !batch || !count)
    return hgn_fail(HGN_E_INVALID, "hgn_normalizer_update: null pointer or bad width");
  hipLaunchKernelGGL(normalizer_update_kernel, dim3(1), dim3(64), 0, stream, acc_sum, acc_sumsq, acc_count, num_acc, batch,
                     count, F, max_acc);
  return hgn_check_launch("hgn_normalizer_update");
}

extern "C" int hgn_normalize(const float* x, int64_t rows, int F, const float* acc_sum, const float* acc_sumsq,
                             const float* acc_count, float eps, int inverse, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows < 0 || F < 1 || F > HGN_MAX_FEATURE_WIDTH || !acc_sum || !acc_sumsq || !acc_count)
    return hgn_fail(HGN_E_INVALID, "hgn_normalize: null pointer or bad width");
  if (rows == 0) return HGN_OK;
  if (!x || !out) return hgn_fail(HGN_E_INVALID, "hgn_normalize: null pointer");
  ProfScope ps(13, (double)rows, stream);
  const int64_t n = rows * F;
  hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, (long)n, F, acc_sum,
                     acc_sumsq, acc_count, eps, inverse, out);
  return hgn_check_launch("hgn_normalize");
}

extern "C" int hgn_lincomb3(const float* a, float ca, const float* b, float cb, const float* c, float cc, int64_t n,
                            float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0) return hgn_fail(HGN_E_INVALID, "hgn_lincomb3: negative size");
  if (n == 0) return HGN_OK;
  if (!a || !b || !out) return hgn_fail(HGN_E_INVALID, "hgn_lincomb3: null pointer");
  hipLaunchKernelGGL(lincomb3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, ca, b, cb, c, cc, (long)n,
                     out);
  return hgn_check_launch("hgn_lincomb3");
}

extern "C" int hgn_radius_edges_workspace_bytes(int64_t N, size_t* bytes) {
  if (!bytes || N < 0 || N > 0x7ffffffe) return hgn_fail(HGN_E_INVALID, "hgn_radius_edges_workspace_bytes: bad size");
  size_t t = 0;
  (void)hipcub::DeviceScan::ExclusiveSum<int*, int*>(nullptr, t, nullptr, nullptr, (int)(N + 1));
  *bytes = up256((size_t)(N + 1) * 4) + up256(t) + 256;
  return HGN_OK;
}

static int radius_args_ok(const float* pos, int64_t ld, int d, const int64_t* node_type, int64_t ldt, int64_t N,
                          float radius, const int32_t* nbr_rowptr, const int32_t* nbr) {
  if (N < 0 || d < 1 || d > 3 || ld < d || ldt < 1 || !(radius >= 0.f)) return 0;
  if (N > 0 && (!pos || !node_type)) return 0;
  if ((nbr_rowptr == nullptr) != (nbr == nullptr)) return 0;
  return 1;
}

extern "C" int hgn_radius_edges_count(const float* pos, int64_t ld, int d, const int64_t* node_type, int64_t ldt, int64_t N,
                                      float radius, int sender_type, int receiver_type, const int32_t* nbr_rowptr,
                                      const int32_t* nbr, int32_t* offsets, int64_t* total, void* workspace,
                                      size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  size_t need = 0;
  if (hgn_radius_edges_workspace_bytes(N, &need) != HGN_OK) return HGN_E_INVALID;
  if (!radius_args_ok(pos, ld, d, node_type, ldt, N, radius, nbr_rowptr, nbr) || !offsets || !total || !workspace ||
      ws_bytes < need)
    return hgn_fail(HGN_E_INVALID, "hgn_radius_edges_count: bad argument or workspace too small");
  *total = 0;
  ProfScope ps(13, (double)N, stream);
  int* counts = (int*)workspace;
  void* temp = (char*)workspace + up256((size_t)(N + 1) * 4);
  size_t tb = need - up256((size_t)(N + 1) * 4);
  if (hipMemsetAsync(counts, 0, (size_t)(N + 1) * 4, stream) != hipSuccess) return hgn_check_launch("hgn_radius_edges_count memset");
  if (N > 0)
    hipLaunchKernelGGL(radius_edges_kernel<false>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, pos, (long)ld, d,
                       node_type, (long)ldt, (long)N, radius, sender_type, receiver_type, nbr_rowptr, nbr, counts,
                       (const int*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr);
  if (hipcub::DeviceScan::ExclusiveSum(temp, tb, counts, offsets, (int)(N + 1), stream) != hipSuccess)
    return hgn_check_launch("hgn_radius_edges_count scan");
  int host_total = 0;
  if (hipMemcpyAsync(&host_total, offsets + N, sizeof(int), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return hgn_check_launch("hgn_radius_edges_count readback");
  *total = host_total;
  return hgn_check_launch("hgn_radius_edges_count");
}

extern "C" int hgn_radius_edges_fill(const float* pos, int64_t ld, int d, const int64_t* node_type, int64_t ldt, int64_t N,
                                     float radius, int sender_type, int receiver_type, const int32_t* nbr_rowptr,
                                     const int32_t* nbr, const int32_t* offsets, int64_t* senders, int64_t* receivers,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!radius_args_ok(pos, ld, d, node_type, ldt, N, radius, nbr_rowptr, nbr) || !offsets)
    return hgn_fail(HGN_E_INVALID, "hgn_radius_edges_fill: bad argument");
  if (N == 0) return HGN_OK;
  if (!senders || !receivers) return hgn_fail(HGN_E_INVALID, "hgn_radius_edges_fill: null output");
  ProfScope ps(13, (double)N, stream);
  hipLaunchKernelGGL(radius_edges_kernel<true>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, pos, (long)ld, d,
                     node_type, (long)ldt, (long)N, radius, sender_type, receiver_type, nbr_rowptr, nbr, (int*)nullptr,
                     offsets, senders, receivers);
  return hgn_check_launch("hgn_radius_edges_fill");
}

extern "C" int hgn_forman_curvature(const float* A, const float* A2, const float* d_in, const float* d_out, int64_t N,
                                    const int32_t* ei, const int32_t* ej, int64_t nnz, float* C, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N < 0 || nnz < 0 || N > 46340) return hgn_fail(HGN_E_INVALID, "hgn_forman_curvature: bad size (dense N x N, N <= 46340)");
  if (nnz == 0) return HGN_OK;
  if (!A || !A2 || !d_in || !d_out || !ei || !ej || !C) return hgn_fail(HGN_E_INVALID, "hgn_forman_curvature: null pointer");
  ProfScope ps(13, (double)nnz, stream);
  hipLaunchKernelGGL(forman_curvature_kernel, dim3((unsigned)((nnz + 3) / 4)), dim3(256), 0, stream, A, A2, d_in, d_out, (long)N,
                     ei, ej, (long)nnz, C);
  return hgn_check_launch("hgn_forman_curvature");
}

extern "C" int hgn_forman_post_delta(const float* A, const float* A2, float d_in_x, float d_out_y, int64_t N, int32_t x,
                                     int32_t y, const int32_t* i_nb, int32_t dim_i, const int32_t* j_nb, int32_t dim_j, float* D,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N < 1 || N > 46340 || x < 0 || y < 0 || x >= N || y >= N || dim_i < 0 || dim_j < 0)
    return hgn_fail(HGN_E_INVALID, "hgn_forman_post_delta: bad size / index");
  if (dim_i == 0 || dim_j == 0) return HGN_OK;
  if (!A || !A2 || !i_nb || !j_nb || !D) return hgn_fail(HGN_E_INVALID, "hgn_forman_post_delta: null pointer");
  ProfScope ps(13, (double)dim_i * dim_j, stream);
  const long pairs = (long)dim_i * dim_j;
  hipLaunchKernelGGL(forman_post_delta_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, stream, A, A2, d_in_x, d_out_y,
                     (long)N, x, y, i_nb, dim_i, j_nb, dim_j, D);
  return hgn_check_launch("hgn_forman_post_delta");
}
