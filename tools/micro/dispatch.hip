// micro-benchmark: what does a launch of 4641 x 256-thread workgroups with 64 KB LDS cost when the body is empty,
// and what does touching scratch add?
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { long a[80]; };
__global__ __launch_bounds__(256, 2) void k_lds(float* out, int n) {
  __shared__ float l[16384];
  if (n < 0) { l[threadIdx.x] = 1.f; __syncthreads(); out[threadIdx.x] = l[(threadIdx.x * 7) & 16383]; }
}
__global__ __launch_bounds__(256, 2) void k_scratch(float* out, int n, const int* idx) {
  __shared__ float l[16384];
  float arr[64];
  if (n < 0) {
    for (int i = 0; i < 64; ++i) arr[i] = out[i];
    l[threadIdx.x] = arr[idx[threadIdx.x] & 63]; __syncthreads(); out[threadIdx.x] = l[(threadIdx.x * 7) & 16383];
  }
}
__global__ __launch_bounds__(256, 2) void k_args(float* out, int n, Big b) {
  __shared__ float l[16384];
  if (n < 0) { l[threadIdx.x] = (float)b.a[threadIdx.x % 80]; __syncthreads(); out[threadIdx.x] = l[(threadIdx.x * 7) & 16383]; }
}
__global__ __launch_bounds__(512, 4) void k_lds512(float* out, int n) {
  __shared__ float l[16384];
  if (n < 0) { l[threadIdx.x] = 1.f; __syncthreads(); out[threadIdx.x] = l[(threadIdx.x * 7) & 16383]; }
}
template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
int main() {
  float* out; hipMalloc(&out, 1 << 20); int* idx; hipMalloc(&idx, 4096); hipMemset(idx, 0, 4096);
  Big big{};
  printf("empty 64KB-LDS kernel, 4641x256 : %.4f ms\n", timeit([&] { hipLaunchKernelGGL(k_lds, dim3(4641), dim3(256), 0, 0, out, 1); }));
  printf("same + scratch array           : %.4f ms\n", timeit([&] { hipLaunchKernelGGL(k_scratch, dim3(4641), dim3(256), 0, 0, out, 1, idx); }));
  printf("same + 640-byte kernarg        : %.4f ms\n", timeit([&] { hipLaunchKernelGGL(k_args, dim3(4641), dim3(256), 0, 0, out, 1, big); }));
  printf("empty 64KB-LDS kernel, 4641x512 : %.4f ms\n", timeit([&] { hipLaunchKernelGGL(k_lds512, dim3(4641), dim3(512), 0, 0, out, 1); }));
  printf("empty 64KB-LDS kernel, 512x256  : %.4f ms\n", timeit([&] { hipLaunchKernelGGL(k_lds, dim3(512), dim3(256), 0, 0, out, 1); }));
  return 0;
}
