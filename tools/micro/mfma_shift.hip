// Does a v_mfma whose vdst partially overlaps its srcC (the compiler "slides" accumulators: v[8:11] <- v[10:13]) cost more than
// one with vdst == srcC?   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shift.hip -o tools/_build/mfma_shift
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define M2(d0, c0, d1, c1) "v_mfma_f32_16x16x32_bf16 " d0 ", %0, %1, " c0 "\n v_mfma_f32_16x16x32_bf16 " d1 ", %0, %1, " c1 "\n"
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x + 3 + j)); }
  asm volatile("v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n"
               "v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n" ::: "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {        // same registers: chain X in v[8:11], chain Y in v[14:17]
      asm volatile(M2("v[8:11]", "v[8:11]", "v[14:17]", "v[14:17]") M2("v[8:11]", "v[8:11]", "v[14:17]", "v[14:17]")
                   M2("v[8:11]", "v[8:11]", "v[14:17]", "v[14:17]") M2("v[8:11]", "v[8:11]", "v[14:17]", "v[14:17]")
                   M2("v[8:11]", "v[8:11]", "v[14:17]", "v[14:17]") M2("v[8:11]", "v[8:11]", "v[14:17]", "v[14:17]")
                   :: "v"(a), "v"(b) : "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
    } else if (MODE == 1) { // sliding by two registers: X: 10:13 -> 8:11 -> 10:13 ..., Y: 16:19 -> 14:17 -> 16:19 ...
      asm volatile(M2("v[8:11]", "v[10:13]", "v[14:17]", "v[16:19]") M2("v[10:13]", "v[8:11]", "v[16:19]", "v[14:17]")
                   M2("v[8:11]", "v[10:13]", "v[14:17]", "v[16:19]") M2("v[10:13]", "v[8:11]", "v[16:19]", "v[14:17]")
                   M2("v[8:11]", "v[10:13]", "v[14:17]", "v[16:19]") M2("v[10:13]", "v[8:11]", "v[16:19]", "v[14:17]")
                   :: "v"(a), "v"(b) : "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
    } else {                // disjoint move: X: 8:11 -> 12:15 -> 8:11 (no overlap), Y: 16:19 -> 20:23 -> 16:19
      asm volatile(M2("v[12:15]", "v[8:11]", "v[20:23]", "v[16:19]") M2("v[8:11]", "v[12:15]", "v[16:19]", "v[20:23]")
                   M2("v[12:15]", "v[8:11]", "v[20:23]", "v[16:19]") M2("v[8:11]", "v[12:15]", "v[16:19]", "v[20:23]")
                   M2("v[12:15]", "v[8:11]", "v[20:23]", "v[16:19]") M2("v[8:11]", "v[12:15]", "v[16:19]", "v[20:23]")
                   :: "v"(a), "v"(b) : "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23");
    }
  }
  long long t1 = clock64();
  float s;
  asm volatile("v_add_f32 %0, v8, v14" : "=v"(s) :: );
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 64);
  const int iters = 400;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  long long h[3]; (void)hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
  const char* names[3] = {"vdst == srcC", "vdst overlaps srcC shifted by two registers", "vdst disjoint from srcC"};
  for (int m = 0; m < 3; ++m) printf("%-48s %.2f cycles per MFMA (two chains)\n", names[m], (double)h[m] / (12.0 * iters));
  return 0;
}
