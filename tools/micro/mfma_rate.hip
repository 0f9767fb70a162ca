// Cycles per v_mfma_f32_16x16x32_bf16 for one wave per SIMD under different operand arrangements (why a 96-MFMA sweep of the
// latency-form forward takes 1.2 us instead of 0.65):  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
  bf16x8 a[24], b[3];
  f32x4 t[8];
  for (int i = 0; i < 24; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(0.001f * (threadIdx.x + i + j));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(0.002f * (threadIdx.x + 3 * i + j));
  for (int i = 0; i < 8; ++i) t[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (MODE == 0) {          // 4 chains, 6 products each, distinct A registers per (chain, split) -- the sweep's arrangement
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            t[4 * g + kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(3 * kk + (r % 3)) + 12 * g], b[r % 3], t[4 * g + kk], 0, 0, 0);
      } else if (MODE == 1) {   // 4 chains, same A / B registers every time
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            t[4 * g + kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], t[4 * g + kk], 0, 0, 0);
      } else if (MODE == 3) {   // 2 chains (the pair-ahead sweep)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            t[4 * g + 2 * pr + kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(3 * kk + (r % 3)) + 6 * pr + 12 * g], b[r % 3], t[4 * g + 2 * pr + kk], 0, 0, 0);
      } else if (MODE == 4) {   // 3 chains
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int kk = 0; kk < 3; ++kk)
            t[kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(3 * kk + (r % 3)) + 12 * g], b[r % 3], t[kk], 0, 0, 0);
      } else {                  // 8 chains
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int kk = 0; kk < 8; ++kk)
            t[kk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(3 * kk + r) % 24], b[r], t[kk], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += t[i][0] + t[i][1] + t[i][2] + t[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
  const int iters = 200;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
  }
  long long h[5]; hipMemcpy(h, cyc, 40, hipMemcpyDeviceToHost);
  for (int m = 0; m < 5; ++m) printf("mode %d: %.2f clock64 ticks per MFMA (48 per iteration, %d iterations)\n", m, (double)h[m] / (48.0 * iters), iters);
  // one short burst, as in a single-tile kernel: 96 MFMAs
  for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, 2); hipDeviceSynchronize(); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost); printf("burst of 96: %.2f ticks per MFMA\n", (double)h[0] / 96.0); }
  // total ticks against the number of passes over the SAME 48-MFMA loop body (pass 1 fetches its instructions cold)
  for (int it = 1; it <= 6; ++it) { hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, it); hipDeviceSynchronize(); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost); printf("passes %d: %lld ticks\n", it, h[0]); }
  return 0;
}
