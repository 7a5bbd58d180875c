// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs, and how does it round the
// fp32 accumulation?  Also checks ds_read_b64_tr_b16's lane map with per-lane row addresses.
// hipcc --offload-arch=gfx950 -O2 tools/proto/mfma_denorm_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f16v __attribute__((ext_vector_type(16)));

// B[k][n] (16 x 32 halfs) sits in LDS row-major with 64-byte rows at row slots given by slot[k];
// A[m][k] = (m + k) % 3 == 0 ? 1 : 0.
__global__ void probe(const _Float16* Bg, const int* slot, float* D) {
  __shared__ __attribute__((aligned(1024))) _Float16 lds[64 * 32];
  const int l = threadIdx.x;
  for (int i = l; i < 64 * 32; i += 64) lds[i] = Bg[i];
  __syncthreads();
  const int h = l >> 5, g1 = (l >> 4) & 1, q = (l >> 2) & 3, p = l & 3;
  h8 b, a;
  for (int r = 0; r < 2; ++r) {
    const int kk = 8 * h + 4 * r + q;
    const char* base = reinterpret_cast<const char*>(lds) + slot[kk] * 64 + 32 * g1 + 8 * p;
    hv4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hv4*)(base));
    for (int j = 0; j < 4; ++j) b[4 * r + j] = (_Float16)v[j];
  }
  const int m = l & 31;
  for (int j = 0; j < 8; ++j) a[j] = ((m + 8 * h + j) % 3 == 0) ? (_Float16)1.0f : (_Float16)0.0f;
  f16v c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int v = 0; v < 16; ++v) {
    const int row = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5);
    D[row * 32 + (l & 31)] = c[v];
  }
}

int main() {
  std::vector<_Float16> B(64 * 32);
  std::vector<int> slot(16);
  for (int k = 0; k < 16; ++k) slot[k] = (k * 7 + 3) % 64;   // arbitrary distinct rows
  std::vector<float> Bf(16 * 32);
  for (int i = 0; i < 64 * 32; ++i) B[i] = (_Float16)0.0f;
  for (int k = 0; k < 16; ++k)
    for (int n = 0; n < 32; ++n) {
      // n < 16: ordinary values; n >= 16: fp16 subnormals (2^-15 .. 2^-24 scale)
      float v = n < 16 ? (float)((k * 37 + n * 11) % 97) / 64.0f - 0.7f : ldexpf((float)((k * 5 + n) % 31 + 1), -24);
      _Float16 hv = (_Float16)v;
      B[slot[k] * 32 + n] = hv;
      Bf[k * 32 + n] = (float)hv;
    }
  _Float16* dB; int* ds; float* dD;
  hipMalloc(&dB, B.size() * 2); hipMalloc(&ds, 64); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(ds, slot.data(), 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dB, ds, dD);
  std::vector<float> D(32 * 32);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  double max_norm = 0, max_sub = 0;
  int bad = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double ref = 0;
      for (int k = 0; k < 16; ++k) if ((m + k) % 3 == 0) ref += Bf[k * 32 + n];
      double err = std::fabs(D[m * 32 + n] - ref);
      double rel = ref != 0 ? err / std::fabs(ref) : err;
      if (n < 16) max_norm = std::fmax(max_norm, rel); else max_sub = std::fmax(max_sub, rel);
      if (rel > 1e-6) ++bad;
    }
  printf("layout+values: %d mismatches; max rel err normal cols %.3g, subnormal cols %.3g (1.0 there = flushed)\n", bad,
         max_norm, max_sub);
  return 0;
}
