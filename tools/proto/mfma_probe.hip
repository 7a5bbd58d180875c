// Cycles per MFMA for the 32x32x16 and the legacy 32x32x8 f16 forms, one wave per SIMD and two (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  f16v acc[4];
  for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) acc[i][v] = 0.f;
  h8 a8, b8; h4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = _Float16(threadIdx.x * 0.001f + i); b8[i] = _Float16(1.0f + i); }
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (KIND == 16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[i], 0, 0, 0);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int v = 0; v < 16; ++v) s += acc[i][v];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int threads : {256, 512}) {
    for (int kind : {16, 8}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        else hipLaunchKernelGGL(k<8>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("K=%d, %d waves per SIMD: %.1f memtime ticks per MFMA per wave (100 MHz ticks x ?), %.3f ms for %d MFMAs per wave -> %.1f ns per MFMA per SIMD\n",
             kind, threads / 256, double(c) / (iters * 4), ms, iters * 4, ms * 1e6 / (iters * 4) / (threads / 256));
    }
  }
  return 0;
}
