// Diagnostic: HBM rate of the row-traffic pattern of the fused MLP kernels (1 read stream + 4 write streams of [M,128] fp32)
//   P0  lane (n = lane & 15, q = lane >> 4) moves 16 bytes of row n at feature 16 b + 4 q   (the MFMA register layout)
//   P1  the same, four outputs interleaved in one [M][4][128] array
//   P2  lane-linear: one wave instruction moves 1 KiB contiguous (two whole rows)
//   P3  P0 with reads only / P4  P0 with writes only / P5  P0 with non-temporal stores / P6  non-temporal loads and stores
// hipcc --offload-arch=gfx950 -O3 tools/micro/stream_pattern.hip -o /tmp/stream_pattern && /tmp/stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int P>
__global__ __launch_bounds__(256) void k(const float* __restrict__ in, float* __restrict__ o0, float* __restrict__ o1,
                                         float* __restrict__ o2, float* __restrict__ o3, long M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row0 = (long)blockIdx.x * 64 + wave * 16;
  f32x4 v[8];
  if (P == 2) {
    const float* p = in + row0 * 128 + lane * 4;
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = *reinterpret_cast<const f32x4*>(p + b * 256);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      *reinterpret_cast<f32x4*>(o0 + row0 * 128 + lane * 4 + b * 256) = v[b];
      *reinterpret_cast<f32x4*>(o1 + row0 * 128 + lane * 4 + b * 256) = v[b] * 2.f;
      *reinterpret_cast<f32x4*>(o2 + row0 * 128 + lane * 4 + b * 256) = v[b] * 3.f;
      *reinterpret_cast<f32x4*>(o3 + row0 * 128 + lane * 4 + b * 256) = v[b] * 4.f;
    }
    return;
  }
  const int n = lane & 15, q = lane >> 4;
  const long row = row0 + n;
  if (P == 6) {
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in + row * 128 + 16 * b + 4 * q));
  } else if (P != 4) {
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = *reinterpret_cast<const f32x4*>(in + row * 128 + 16 * b + 4 * q);
  } else {
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = f32x4{(float)row, 1.f, 2.f, (float)b};
  }
  if (P == 3) {
    f32x4 s = v[0];
#pragma unroll
    for (int b = 1; b < 8; ++b) s += v[b];
    if (s[0] == 123.456f) o0[row] = s[1];
    return;
  }
  if (P == 5 || P == 6) {
#pragma unroll
    for (int b = 0; b < 8; ++b) __builtin_nontemporal_store(v[b], reinterpret_cast<f32x4*>(o0 + row * 128 + 16 * b + 4 * q));
#pragma unroll
    for (int b = 0; b < 8; ++b) __builtin_nontemporal_store(v[b] * 2.f, reinterpret_cast<f32x4*>(o1 + row * 128 + 16 * b + 4 * q));
#pragma unroll
    for (int b = 0; b < 8; ++b) __builtin_nontemporal_store(v[b] * 3.f, reinterpret_cast<f32x4*>(o2 + row * 128 + 16 * b + 4 * q));
#pragma unroll
    for (int b = 0; b < 8; ++b) __builtin_nontemporal_store(v[b] * 4.f, reinterpret_cast<f32x4*>(o3 + row * 128 + 16 * b + 4 * q));
    return;
  }
  if (P == 1) {
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      float* d = o0 + row * 512 + 16 * b + 4 * q;
      *reinterpret_cast<f32x4*>(d) = v[b];
      *reinterpret_cast<f32x4*>(d + 128) = v[b] * 2.f;
      *reinterpret_cast<f32x4*>(d + 256) = v[b] * 3.f;
      *reinterpret_cast<f32x4*>(d + 384) = v[b] * 4.f;
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < 8; ++b) *reinterpret_cast<f32x4*>(o0 + row * 128 + 16 * b + 4 * q) = v[b];
#pragma unroll
  for (int b = 0; b < 8; ++b) *reinterpret_cast<f32x4*>(o1 + row * 128 + 16 * b + 4 * q) = v[b] * 2.f;
#pragma unroll
  for (int b = 0; b < 8; ++b) *reinterpret_cast<f32x4*>(o2 + row * 128 + 16 * b + 4 * q) = v[b] * 3.f;
#pragma unroll
  for (int b = 0; b < 8; ++b) *reinterpret_cast<f32x4*>(o3 + row * 128 + 16 * b + 4 * q) = v[b] * 4.f;
}
template <int P>
void run(const char* name, float* in, float* o, long M, double bytes_per_row) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float* o0 = o; float* o1 = o + M * 128; float* o2 = o + 2 * M * 128; float* o3 = o + 3 * M * 128;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<P>, dim3(M / 64), dim3(256), 0, 0, in, o0, o1, o2, o3, M);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<P>, dim3(M / 64), dim3(256), 0, 0, in, o0, o1, o2, o3, M);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
  printf("%-46s %.3f ms  %.2f TB/s\n", name, ms, bytes_per_row * M / ms / 1e9);
}
int main() {
  const long M = 1188096;
  float *in, *o;
  hipMalloc(&in, M * 128 * 4); hipMalloc(&o, 4 * M * 128 * 4);
  hipMemset(in, 0, M * 128 * 4);
  run<0>("P0 MFMA layout, 1 read + 4 write streams", in, o, M, 2560);
  run<1>("P1 MFMA layout, writes interleaved per row", in, o, M, 2560);
  run<2>("P2 lane-linear, 1 read + 4 write streams", in, o, M, 2560);
  run<3>("P3 MFMA layout, read only", in, o, M, 512);
  run<4>("P4 MFMA layout, 4 write streams only", in, o, M, 2048);
  run<5>("P5 P0 with non-temporal stores", in, o, M, 2560);
  run<6>("P6 P0 with non-temporal loads and stores", in, o, M, 2560);
  run<0>("P0 again", in, o, M, 2560);
  return 0;
}
