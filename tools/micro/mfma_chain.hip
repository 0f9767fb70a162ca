// How many cycles does v_mfma_f32_16x16x32_bf16 cost per instruction in the dependency patterns of the fused backward?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
// pattern 0: 8 independent accumulators round-robin; 1: two interleaved chains of 6 dependent products (wgrad_block);
// pattern 2: one chain of 6 dependent products at a time (chain sweep); each with 1, 2, 4, 8 waves per workgroup (1 WG per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, float* sink, int iters, int burst, int sleep_units) {
  bf16x8 a[3], b[3];
  for (int s = 0; s < 3; ++s)
    for (int j = 0; j < 8; ++j) { a[s][j] = (__bf16)(0.001f * (threadIdx.x + s + j)); b[s][j] = (__bf16)(0.002f * (threadIdx.x + 2 * s + j)); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  unsigned long long busy = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (burst && (it % burst) == 0) {              // bursty use of the matrix pipe: `burst` iterations, then an idle gap
      const unsigned long long a0 = __builtin_readcyclecounter();
      for (int z = 0; z < sleep_units; ++z) __builtin_amdgcn_s_sleep(16);
      busy -= __builtin_readcyclecounter() - a0;
    }
    if (PAT == 0) {
#pragma unroll
      for (int r = 0; r < 12; ++r) acc[r & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r % 3], b[(r + 1) % 3], acc[r & 7], 0, 0, 0);
    } else if (PAT == 1) {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r % 3], b[(r + 1) % 3], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(r + 2) % 3], b[(r + 1) % 3], acc[1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 6; ++r) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r % 3], b[(r + 1) % 3], acc[0], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 6; ++r) acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(r + 2) % 3], b[(r + 1) % 3], acc[1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0 + busy;
}

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 64 * 8); hipMalloc(&sink, 256 * 512 * 4);
  const int iters = 2000;
  for (int pat = 0; pat < 3; ++pat)
    for (int waves = 1; waves <= 8; waves *= 2) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, out, sink, iters, 0, 0);
      if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, sink, iters, 0, 0);
      if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, out, sink, iters, 0, 0);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
      const double n = 12.0 * iters;
      printf("pattern %d waves/WG %d: %.1f ticks per MFMA (wave 0), kernel %.3f ms => %.1f ns per MFMA per wave; %.1f TFLOP/s\n", pat, waves,
             h[0] / n, ms, ms * 1e6 / n, 256.0 * waves * n * 16384 / (ms * 1e-3) / 1e12);
    }
  // bursts: 8 iterations (96 MFMAs) of pattern 1, then an idle gap of `g` x 16 x 64 cycles; ticks per MFMA exclude the gaps
  for (int waves = 4; waves <= 8; waves *= 2)
    for (int g = 0; g <= 8; g = g ? g * 2 : 1) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, out, sink, 2000, 8, g);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
      printf("bursty: waves/WG %d, gap %d x 1024 cycles after every 96 MFMAs: %.1f ticks per MFMA inside the bursts; kernel %.3f ms = %.2f us per (burst + gap)\n", waves, g, h[0] / (12.0 * 2000), ms, ms * 1e3 / 250);
    }
  return 0;
}
