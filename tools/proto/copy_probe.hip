// Streaming-copy sweep: which grid / loads in flight / cache policy streams fastest (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, long n4) {
  const long stride = long(gridDim.x) * blockDim.x;
  long i = blockIdx.x * long(blockDim.x) + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
      else dst[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}
template <int U, bool NT>
float run(const f32x4* s, f32x4* d, long n4, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_copy<U, NT>), dim3(blocks), dim3(256), 0, 0, s, d, n4);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_copy<U, NT>), dim3(blocks), dim3(256), 0, 0, s, d, n4);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return 2.0f * n4 * 16 * 10 / (ms * 1e-3f) / 1e9f;
}
int main() {
  for (long mib : {512L, 2048L, 8192L}) {
    const long n4 = mib << 16;
    f32x4 *s, *d;
    if (hipMalloc(&s, n4 * 16) != hipSuccess || hipMalloc(&d, n4 * 16) != hipSuccess) return 1;
    hipMemset(s, 1, n4 * 16);
    for (int blocks : {1024, 2048, 4096, 8192, 16384, 65536}) {
      printf("%ld MiB, %d blocks: U1 %.0f  U2 %.0f  U4 %.0f  U8 %.0f | nt U1 %.0f U4 %.0f U8 %.0f GB/s\n", mib, blocks,
             run<1, false>(s, d, n4, blocks), run<2, false>(s, d, n4, blocks), run<4, false>(s, d, n4, blocks),
             run<8, false>(s, d, n4, blocks), run<1, true>(s, d, n4, blocks), run<4, true>(s, d, n4, blocks),
             run<8, true>(s, d, n4, blocks));
      fflush(stdout);
    }
    hipFree(s); hipFree(d);
  }
  return 0;
}
