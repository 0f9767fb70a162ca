// Exploratory micro-benchmark (NOT part of the product or the tests): the edge-mode fused MLP forward with every fp32
// 128x128 product replaced by six bf16 MFMAs on a 3-way bf16 split of both operands (x = x1 + x2 + x3 captures all 24
// significand bits; products with i + j <= 4; fp32 accumulation in the MFMA).  Same data flow as csrc/mlp.hip's edge
// kernel (row loads, two gathered addends, three saved activations, LayerNorm, residual), so the time is comparable with
// the fp32 kernel's 0.755 ms at 594 048 rows.  Prints the time and the error against an fp64 host evaluation of a few rows.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/bf16x6_mlp.hip -o tools/_build/bf16x6_mlp && tools/_build/bf16x6_mlp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LAT = 128, NB = 8, WG = 256, TILE_ROWS = 64;
constexpr int TILE_BF16 = 512;                    // one 16x32 bf16 operand tile in lane order: 64 lanes x 8
constexpr int HALF_TILES = 3 * 2 * 8;             // splits x c_local x ob
constexpr int HALF_BF16 = HALF_TILES * TILE_BF16; // 48 KB

struct Act { f32x4 v[NB]; };

__device__ __forceinline__ void split3(const Act& x, bf16x8 (&s)[3][4]) {
  // lane (n, q) owns features 16fb + 4q + r; k-block c takes its 8 values from fb = 2c (j < 4) and fb = 2c+1 (j >= 4)
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = x.v[2 * c + (j >> 2)][j & 3];
      const __bf16 h = (__bf16)v;
      const float r1 = v - (float)h;
      const __bf16 m = (__bf16)r1;
      const float r2 = r1 - (float)m;
      s[0][c][j] = h; s[1][c][j] = m; s[2][c][j] = (__bf16)r2;
    }
}

__device__ __forceinline__ void stage_half(__bf16* lds, const __bf16* gsrc) {
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
#ifndef STAGE_TILES
#define STAGE_TILES HALF_TILES
#endif
  for (unsigned i = wave; i < STAGE_TILES; i += WG / 64)          // one tile (1 KiB) per wave instruction, straight copy
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + i * TILE_BF16 + lane * 8),
                                     (__attribute__((address_space(3))) void*)(lds + i * TILE_BF16), 16, 0, 0);
}

// acc[ob] += W(layer) * x, x given as its three bf16 splits
__device__ __forceinline__ void gemm6(Act& acc, const bf16x8 (&xs)[3][4], __bf16* lds, const __bf16* wpk) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage_half(lds, wpk + (long)half * HALF_BF16);
    __syncthreads();
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const int c = 2 * half + cl;
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
        const bf16x8 a_hi = *reinterpret_cast<const bf16x8*>(lds + ((0 * 2 + cl) * 8 + ob) * TILE_BF16 + lane * 8);
        const bf16x8 a_mi = *reinterpret_cast<const bf16x8*>(lds + ((1 * 2 + cl) * 8 + ob) * TILE_BF16 + lane * 8);
        const bf16x8 a_lo = *reinterpret_cast<const bf16x8*>(lds + ((2 * 2 + cl) * 8 + ob) * TILE_BF16 + lane * 8);
        f32x4 t = acc.v[ob];
#ifndef NPROD
#define NPROD 6
#endif
        if (NPROD >= 6) t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, xs[0][c], t, 0, 0, 0);      // small terms first
        if (NPROD >= 5) t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, xs[2][c], t, 0, 0, 0);
        if (NPROD >= 4) t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_mi, xs[1][c], t, 0, 0, 0);
        if (NPROD >= 3) t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_mi, xs[0][c], t, 0, 0, 0);
        if (NPROD >= 2) t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, xs[1][c], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, xs[0][c], t, 0, 0, 0);
        acc.v[ob] = t;
      }
    }
  }
}

#define FOR_B(fb) _Pragma("unroll") for (int fb = 0; fb < NB; ++fb)
__device__ __forceinline__ void t_load(Act& a, const float* row, int kq) { FOR_B(fb) a.v[fb] = *reinterpret_cast<const f32x4*>(row + 16 * fb + 4 * kq); }
__device__ __forceinline__ void t_add(Act& a, const float* row, int kq) { FOR_B(fb) a.v[fb] += *reinterpret_cast<const f32x4*>(row + 16 * fb + 4 * kq); }
__device__ __forceinline__ void t_store(const Act& a, float* row, int kq) { FOR_B(fb) *reinterpret_cast<f32x4*>(row + 16 * fb + 4 * kq) = a.v[fb]; }
__device__ __forceinline__ float row_sum(const Act& a) {
  float s = 0.f;
  FOR_B(fb) s += (a.v[fb][0] + a.v[fb][1]) + (a.v[fb][2] + a.v[fb][3]);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  return s;
}

struct Args {
  long M; const float* e; const float* P; const int* snd; const int* rcv;
  const __bf16* w1; const __bf16* w2; const __bf16* w3; const float* b1; const float* b2; const float* b3;
  const float* gam; const float* bet; float* z1; float* z2; float* xhat; float* out;
};

__global__ __launch_bounds__(WG, 3) void mlp6_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[HALF_BF16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kq = lane >> 4;
  const long row = (long)blockIdx.x * TILE_ROWS + wave * 16 + n;
  const bool valid = row < a.M;
  const long rc = valid ? row : a.M - 1;
  Act acc, b;
  bf16x8 xs[3][4];
  t_load(b, a.e + rc * LAT, kq);
  t_load(acc, a.b1, kq);
  t_add(acc, a.P + (long)a.snd[rc] * 256, kq);
  t_add(acc, a.P + (long)a.rcv[rc] * 256 + 128, kq);
  split3(b, xs);
  gemm6(acc, xs, lds, a.w1);
  FOR_B(fb) for (int u = 0; u < 4; ++u) acc.v[fb][u] = fmaxf(acc.v[fb][u], 0.f);
  if (valid) t_store(acc, a.z1 + row * LAT, kq);
  split3(acc, xs);
  t_load(b, a.b2, kq);
  gemm6(b, xs, lds, a.w2);
  FOR_B(fb) for (int u = 0; u < 4; ++u) b.v[fb][u] = fmaxf(b.v[fb][u], 0.f);
  if (valid) t_store(b, a.z2 + row * LAT, kq);
  split3(b, xs);
  t_load(acc, a.b3, kq);
  gemm6(acc, xs, lds, a.w3);
  const float mean = row_sum(acc) * (1.f / LAT);
  FOR_B(fb) { acc.v[fb] -= mean; b.v[fb] = acc.v[fb] * acc.v[fb]; }
  const float rstd = 1.f / sqrtf(row_sum(b) * (1.f / LAT) + 1e-5f);
  FOR_B(fb) acc.v[fb] *= rstd;
  if (valid) t_store(acc, a.xhat + row * LAT, kq);
  FOR_B(fb) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(a.gam + 16 * fb + 4 * kq), be = *reinterpret_cast<const f32x4*>(a.bet + 16 * fb + 4 * kq);
    acc.v[fb] = acc.v[fb] * g + be;
  }
  if (valid) { t_add(acc, a.e + row * LAT, kq); t_store(acc, a.out + row * LAT, kq); }
}

// ---- variant: every wave owns TWO 16-row sub-tiles (128-row workgroup tiles); each staged half block and each operand
// fragment read from LDS serves both (half the weight DMA and LDS reads per row; 2 waves / SIMD) -------------------------
__device__ __forceinline__ void gemm6x2(Act (&acc)[2], const bf16x8 (&xs)[2][3][4], __bf16* lds, const __bf16* wpk) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage_half(lds, wpk + (long)half * HALF_BF16);
    __syncthreads();
#pragma unroll
    for (int cl = 0; cl < 2; ++cl) {
      const int c = 2 * half + cl;
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
        const bf16x8 a_hi = *reinterpret_cast<const bf16x8*>(lds + ((0 * 2 + cl) * 8 + ob) * TILE_BF16 + lane * 8);
        const bf16x8 a_mi = *reinterpret_cast<const bf16x8*>(lds + ((1 * 2 + cl) * 8 + ob) * TILE_BF16 + lane * 8);
        const bf16x8 a_lo = *reinterpret_cast<const bf16x8*>(lds + ((2 * 2 + cl) * 8 + ob) * TILE_BF16 + lane * 8);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f32x4 t = acc[u].v[ob];
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, xs[u][0][c], t, 0, 0, 0);
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
}

__global__ __launch_bounds__(WG, 2) void mlp6x2_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[HALF_BF16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kq = lane >> 4;
  long row[2], rc[2]; bool valid[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    row[u] = (long)blockIdx.x * 128 + wave * 32 + u * 16 + n;
    valid[u] = row[u] < a.M; rc[u] = valid[u] ? row[u] : a.M - 1;
  }
  Act acc[2], b[2];
  bf16x8 xs[2][3][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    t_load(b[u], a.e + rc[u] * LAT, kq);
    t_load(acc[u], a.b1, kq);
    t_add(acc[u], a.P + (long)a.snd[rc[u]] * 256, kq);
    t_add(acc[u], a.P + (long)a.rcv[rc[u]] * 256 + 128, kq);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) split3(b[u], xs[u]);
  gemm6x2(acc, xs, lds, a.w1);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    FOR_B(fb) for (int w = 0; w < 4; ++w) acc[u].v[fb][w] = fmaxf(acc[u].v[fb][w], 0.f);
    if (valid[u]) t_store(acc[u], a.z1 + row[u] * LAT, kq);
    split3(acc[u], xs[u]);
    t_load(b[u], a.b2, kq);
  }
  gemm6x2(b, xs, lds, a.w2);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    FOR_B(fb) for (int w = 0; w < 4; ++w) b[u].v[fb][w] = fmaxf(b[u].v[fb][w], 0.f);
    if (valid[u]) t_store(b[u], a.z2 + row[u] * LAT, kq);
    split3(b[u], xs[u]);
    t_load(acc[u], a.b3, kq);
  }
  gemm6x2(acc, xs, lds, a.w3);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float mean = row_sum(acc[u]) * (1.f / LAT);
    FOR_B(fb) { acc[u].v[fb] -= mean; b[u].v[fb] = acc[u].v[fb] * acc[u].v[fb]; }
    const float rstd = 1.f / sqrtf(row_sum(b[u]) * (1.f / LAT) + 1e-5f);
    FOR_B(fb) acc[u].v[fb] *= rstd;
    if (valid[u]) t_store(acc[u], a.xhat + row[u] * LAT, kq);
    FOR_B(fb) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(a.gam + 16 * fb + 4 * kq), be = *reinterpret_cast<const f32x4*>(a.bet + 16 * fb + 4 * kq);
      acc[u].v[fb] = acc[u].v[fb] * g + be;
    }
    if (valid[u]) { t_add(acc[u], a.e + row[u] * LAT, kq); t_store(acc[u], a.out + row[u] * LAT, kq); }
  }
}

// ---- host ------------------------------------------------------------------------------------------------------------
static uint16_t f2bf(float f) {           // round to nearest even
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static std::vector<uint16_t> pack(const std::vector<float>& W) {       // W [128 out][128 in] -> [half][split][cl][ob][lane][8]
  std::vector<uint16_t> o(2 * HALF_BF16);
  for (int half = 0; half < 2; ++half)
    for (int sp = 0; sp < 3; ++sp)
      for (int cl = 0; cl < 2; ++cl)
        for (int ob = 0; ob < 8; ++ob)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
              const int m = l & 15, q = l >> 4, c = 2 * half + cl;
              const int feat = 32 * c + (j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4));
              const float w = W[(16 * ob + m) * 128 + feat];
              const uint16_t h = f2bf(w); const float r1 = w - bf2f(h);
              const uint16_t mi = f2bf(r1); const float r2 = r1 - bf2f(mi);
              const uint16_t lo = f2bf(r2);
              o[(long)half * HALF_BF16 + (((sp * 2 + cl) * 8 + ob) * 64 + l) * 8 + j] = sp == 0 ? h : sp == 1 ? mi : lo;
            }
  return o;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  const long E = 594048, N = 102400;
  srand(1);
  auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2 - 1); };
  std::vector<float> e(E * 128), P(N * 256), W[3], B[3], gam(128), bet(128);
  for (auto& v : e) v = rnd();
  for (auto& v : P) v = rnd();
  for (int l = 0; l < 3; ++l) { W[l].resize(128 * 128); B[l].resize(128); for (auto& v : W[l]) v = rnd() * 0.088f; for (auto& v : B[l]) v = rnd() * 0.1f; }
  for (int i = 0; i < 128; ++i) { gam[i] = 1.f + 0.1f * rnd(); bet[i] = 0.1f * rnd(); }
  std::vector<int> snd(E), rcv(E);
  for (long i = 0; i < E; ++i) { rcv[i] = (int)(i * N / E); snd[i] = (int)((rcv[i] + 1 + rand() % 60) % N); }
  float *de, *dP, *db[3], *dg, *dbt, *z1, *z2, *xh, *out; int *ds, *dr; __bf16* dw[3];
  CK(hipMalloc(&de, E * 512)); CK(hipMalloc(&dP, N * 1024)); CK(hipMalloc(&z1, E * 512)); CK(hipMalloc(&z2, E * 512));
  CK(hipMalloc(&xh, E * 512)); CK(hipMalloc(&out, E * 512)); CK(hipMalloc(&ds, E * 4)); CK(hipMalloc(&dr, E * 4));
  CK(hipMalloc(&dg, 512)); CK(hipMalloc(&dbt, 512));
  CK(hipMemcpy(de, e.data(), E * 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dP, P.data(), N * 1024, hipMemcpyHostToDevice));
  CK(hipMemcpy(ds, snd.data(), E * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, rcv.data(), E * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, gam.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dbt, bet.data(), 512, hipMemcpyHostToDevice));
  for (int l = 0; l < 3; ++l) {
    auto pk = pack(W[l]);
    CK(hipMalloc(&dw[l], pk.size() * 2)); CK(hipMemcpy(dw[l], pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&db[l], 512)); CK(hipMemcpy(db[l], B[l].data(), 512, hipMemcpyHostToDevice));
  }
  Args a{E, de, dP, ds, dr, dw[0], dw[1], dw[2], db[0], db[1], db[2], dg, dbt, z1, z2, xh, out};
  const unsigned tiles = (unsigned)((E + 63) / 64);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mlp6_kernel, dim3(tiles), dim3(WG), 0, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(mlp6_kernel, dim3(tiles), dim3(WG), 0, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("bf16x6 edge MLP forward: %.3f ms per launch (%ld rows); fp32-equivalent %.1f TFLOP/s\n", ms / 10, E, 98304.0 * E / (ms / 10) / 1e9);
  {
    const unsigned tiles2 = (unsigned)((E + 127) / 128);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mlp6x2_kernel, dim3(tiles2), dim3(WG), 0, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(mlp6x2_kernel, dim3(tiles2), dim3(WG), 0, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("two sub-tiles per wave (128-row tiles, 2 waves/SIMD): %.3f ms per launch\n", ms / 10);
  }
  // ---- error of a few rows against fp64 ------------------------------------------------------------------------------
  std::vector<float> ho(E * 128);
  CK(hipMemcpy(ho.data(), out, E * 512, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (long r : {0L, 1L, 77L, 12345L, 594047L}) {
    double x[128], y[128];
    for (int i = 0; i < 128; ++i) {
      double s = B[0][i] + P[(long)snd[r] * 256 + i] + P[(long)rcv[r] * 256 + 128 + i];
      for (int k = 0; k < 128; ++k) s += (double)W[0][i * 128 + k] * e[r * 128 + k];
      x[i] = s > 0 ? s : 0;
    }
    for (int l = 1; l < 3; ++l) {
      for (int i = 0; i < 128; ++i) { double s = B[l][i]; for (int k = 0; k < 128; ++k) s += (double)W[l][i * 128 + k] * x[k]; y[i] = s; }
      for (int i = 0; i < 128; ++i) x[i] = l == 1 ? (y[i] > 0 ? y[i] : 0) : y[i];
    }
    double mean = 0, var = 0;
    for (int i = 0; i < 128; ++i) mean += x[i] / 128;
    for (int i = 0; i < 128; ++i) var += (x[i] - mean) * (x[i] - mean) / 128;
    for (int i = 0; i < 128; ++i) {
      const double ref = e[r * 128 + i] + (x[i] - mean) / sqrt(var + 1e-5) * gam[i] + bet[i];
      worst = fmax(worst, fabs(ref - ho[r * 128 + i])); scale = fmax(scale, fabs(ref));
    }
  }
  printf("max |err| / max |ref| over 5 rows = %.2e\n", worst / scale);
  return 0;
}
