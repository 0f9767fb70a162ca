// wgrad_block of csrc/fused_bwd.hip in isolation: 96 MFMAs fed from LDS operand vectors, 4 or 8 waves, 1 workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -I hyper-graph-nets_amd/csrc -I include tools/micro/wgrad_block.hip -o /tmp/wb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int OPS64 = 3 * 8 * 128;

template <int VARIANT>
__device__ __forceinline__ void wgrad_block(f32x4 (&acc)[2][8], const bf16x8* __restrict__ gp, const bf16x8* __restrict__ ap, int blk) {
  bf16x8 gs[2][3];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int s = 0; s < 3; ++s) gs[mb][s] = gp[(s * 8 + blk * 4) * 128 + 16 * mb];
  bf16x8 as[2][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) as[0][s] = ap[(s * 8 + blk * 4) * 128];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    if (nb + 1 < 8) {
#pragma unroll
      for (int s = 0; s < 3; ++s) as[(nb + 1) & 1][s] = ap[(s * 8 + blk * 4) * 128 + 16 * (nb + 1)];
    }
    if (VARIANT == 0) __builtin_amdgcn_sched_barrier(0);
    const bf16x8 (&a)[3] = as[nb & 1];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      f32x4 c = acc[mb][nb];
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][2], a[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], a[2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], a[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][1], a[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], a[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gs[mb][0], a[0], c, 0, 0, 0);
      acc[mb][nb] = c;
    }
    if (VARIANT == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

template <int VARIANT>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, float* sink, int iters, int active_waves, int resident, int random_data = 0) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OPS64 * 16 + 49152];
  bf16x8* ops = reinterpret_cast<bf16x8*>(smem);
  for (int i = threadIdx.x; i < 2 * OPS64; i += blockDim.x) {
    bf16x8 v;
    for (int j = 0; j < 8; ++j) {
      unsigned h = (unsigned)(i * 8 + j) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      const float r = ((int)(h & 0xffff) - 32768) * (1.f / 32768.f);            // uniform in [-1, 1)
      v[j] = random_data == 0 ? (__bf16)(0.001f * ((i + j) % 97)) : random_data == 1 ? (__bf16)r : (__bf16)0.f;
    }
    ops[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, kg = lane >> 4, ww = wave & 3;
  const bf16x8* gp = ops + kg * 128 + 32 * ww + m;
  const bf16x8* ap = ops + OPS64 + kg * 128 + m;
  f32x4 acc[2][8];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned long long t0 = 0, t1 = 0;
  if (wave < active_waves) {
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      wgrad_block<VARIANT>(acc, gp, ap, it & 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    t1 = __builtin_readcyclecounter();
  }
  if (resident) __syncthreads();       // the inactive waves stay RESIDENT (waiting at this barrier) while the active ones work
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 8; ++b) s += acc[a][b][0] + acc[a][b][2];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && lane == 0) out[wave] = t1 - t0;
}

int main() {
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&sink, 256 * 512 * 4);
  const int iters = 500;
  for (int variant = 0; variant < 2; ++variant)
    for (int aw = 1; aw <= 8; aw *= 2) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, sink, iters, aw, 0);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, sink, iters, aw, 0);
      (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[8]; (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
      printf("variant %d (sched barriers %s) active waves %d: %.0f ticks per block of 96 MFMAs (wave 0), kernel %.3f ms => %.2f us per block\n", variant,
             variant == 0 ? "on" : "off", aw, (double)h[0] / iters, ms, ms * 1e3 / iters);
    }
  for (int rd = 0; rd < 3; ++rd) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, sink, iters, 4, 1, rd);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("operand data %s, 4 active waves: kernel %.3f ms => %.2f us per block\n", rd == 0 ? "small regular" : rd == 1 ? "random in [-1,1)" : "all zero", ms, ms * 1e3 / iters);
  }
  for (int aw = 1; aw <= 4; aw *= 2) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, sink, iters, aw, 1);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("inactive waves RESIDENT at a barrier, active waves %d: kernel %.3f ms => %.2f us per block\n", aw, ms, ms * 1e3 / iters);
  }
  return 0;
}
