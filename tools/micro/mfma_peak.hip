// Micro-benchmark (diagnostic, not part of the product): sustained v_mfma_f32_16x16x4_f32 / 32x32x2 rate from registers,
// no memory traffic, for 1..4 waves per SIMD.  Gives the practical fp32 MFMA ceiling next to the 157.3 TFLOP/s datasheet peak.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float seed) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  float a = seed + threadIdx.x * 1e-6f, b = seed - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void k32(float* out, int iters, float seed) {
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = seed;
  float a = seed + threadIdx.x * 1e-6f, b = seed - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 4096 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int wps = 1; wps <= 4; ++wps) {            // waves per SIMD: blocks of 4 waves, wps blocks per CU
    const int blocks = 256 * wps;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)blocks * 4 * iters * 16 * 2048.0;
      if (rep) printf("16x16x4  4 acc  %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", wps, ms, flop / ms / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)blocks * 4 * iters * 32 * 2048.0;
      if (rep) printf("16x16x4  8 acc  %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", wps, ms, flop / ms / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flop = (double)blocks * 4 * iters * 16 * 4096.0;
      if (rep) printf("32x32x2  2 acc  %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", wps, ms, flop / ms / 1e9);
    }
  }
  return 0;
}
