// Throwaway measurement: what does one same-address atomicAdd (with return) per workgroup cost a
// small kernel, against the same kernel without it?   hipcc --offload-arch=gfx950 -O3 atomic_cost.hip -o atomic_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool ATOMIC>
__global__ __launch_bounds__(256) void k(unsigned long long* state, float* out) {
  const unsigned long long c = state[1];
  out[blockIdx.x * 256 + threadIdx.x] = float(c) + threadIdx.x;
  __syncthreads();
  if (ATOMIC && threadIdx.x == 0) {
    const unsigned long long done = atomicAdd(&state[2], 1ull);
    if (done == gridDim.x - 1ull) { state[2] = 0ull; state[1] = c + 1ull; }
  }
}
template <bool ATOMIC>
float run(int blocks, unsigned long long* state, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<ATOMIC>, dim3(blocks), dim3(256), 0, 0, state, out);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k<ATOMIC>, dim3(blocks), dim3(256), 0, 0, state, out);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 2000 * 1000;
}
int main() {
  unsigned long long* state; float* out;
  hipMalloc(&state, 64); hipMemset(state, 0, 64); hipMalloc(&out, 4096 * 256 * 4);
  for (int blocks : {1, 64, 128, 512, 2048}) {
    const float a = run<false>(blocks, state, out), b = run<true>(blocks, state, out);
    printf("%5d workgroups: %.2f us per launch without, %.2f us with the ticket atomic\n", blocks, a, b);
  }
  unsigned long long h[3]; hipMemcpy(h, state, 24, hipMemcpyDeviceToHost);
  printf("counter %llu ticket %llu\n", h[1], h[2]);
  return 0;
}
