// Dense sub-adjacency sampling for the auto-encoder layers: out = A[idx, :][:, idx]
// read straight from the CSR that lives on the GPU (model/SGNN.py:167-170 does it
// with two scipy fancy-index passes, .todense() and a host-to-device copy per Adam
// iteration).
//
// One workgroup per sampled row i; its 256 lanes walk the b columns of the output
// row, each lane looking idx[j] up in row idx[i]'s sorted column list by binary
// search.  b*b searches of ~log2(degree) steps over a few KB that stay in L2: the
// kernel is bound by the b*b*4-byte output it writes, coalesced.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void k_sample_block(int64_t n_rows, const int* __restrict__ rowptr,
                                                      const int* __restrict__ colidx,
                                                      const float* __restrict__ vals, int64_t b,
                                                      const int64_t* __restrict__ idx, int skip_self,
                                                      float* __restrict__ out, int64_t ldo) {
  const int64_t i = blockIdx.x;
  const int64_t node = idx[i];
  int lo0 = 0, hi0 = 0;
  if (node >= 0 && node < n_rows) {
    lo0 = rowptr[node];
    hi0 = rowptr[node + 1];
  }
  for (int64_t j = threadIdx.x; j < b; j += blockDim.x) {
    const int64_t want = idx[j];
    float v = 0.f;
    if (!(skip_self && want == node)) {
      int lo = lo0, hi = hi0;          // first position with colidx >= want
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (int64_t(colidx[mid]) < want) lo = mid + 1; else hi = mid;
      }
      if (lo < hi0 && int64_t(colidx[lo]) == want) v = vals ? vals[lo] : 1.f;
    }
    out[i * ldo + j] = v;
  }
}

}  // namespace

extern "C" int sgnn_sample_block_f32(int64_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                                     const float* vals, int64_t b, const int64_t* idx, int flags, float* out,
                                     int64_t ldo, void* stream) {
  using sgnn::fail;
  if (n_rows < 0 || nnz < 0 || b < 0) return fail(SGNN_ERR_INVALID, "negative size");
  if (flags & ~SGNN_SAMPLE_SKIP_SELF) return fail(SGNN_ERR_INVALID, "unknown sampling flag");
  if (b == 0) return SGNN_OK;
  if (ldo < b) return fail(SGNN_ERR_INVALID, "ldo smaller than the block width");
  if (b > 0x7fffffff) return fail(SGNN_ERR_INVALID, "block larger than 2^31-1 rows");
  if (!rowptr || !idx || !out || (nnz > 0 && !colidx)) return fail(SGNN_ERR_INVALID, "null pointer");
  hipLaunchKernelGGL(k_sample_block, dim3(unsigned(b)), dim3(256), 0, static_cast<hipStream_t>(stream), n_rows,
                     rowptr, colidx, vals, b, idx, (flags & SGNN_SAMPLE_SKIP_SELF) ? 1 : 0, out, ldo);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

// ---------------------------------------------------------------------------------
// Batch indices: the first b entries of a pseudo-random permutation of [0, n).
// The layers draw torch.randperm(n)[:batch] per optimiser iteration
// (model/SGNN.py:115-117); sorting n keys for 512 indices costs more than the rest of
// the iteration.  Here the permutation is a keyed bijection — a 4-round Feistel
// network on 2*h bits (2^(2h) >= n) with cycle walking back into [0, n) — evaluated
// only at positions 0..b-1, so the indices are distinct by construction.  The key is
// (seed, counter) in device memory; the kernel advances the counter, so replaying a
// captured HIP graph draws a new batch every time.
namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // murmur3 finaliser
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void k_batch_indices(int64_t n, int64_t b, uint64_t* __restrict__ state,
                                                       int64_t* __restrict__ out) {
  const uint64_t seed = state[0], counter = state[1];
  int h = 1;
  while ((int64_t(1) << (2 * h)) < n) ++h;
  const uint32_t hm = (1u << h) - 1u;
  uint32_t key[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    key[r] = mix32(uint32_t(seed) + 0x9e3779b9u * uint32_t(r + 1)) ^ mix32(uint32_t(seed >> 32) + uint32_t(counter)) ^
             mix32(uint32_t(counter >> 32) + 0x7f4a7c15u * uint32_t(r + 1));
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < b; i += int64_t(gridDim.x) * blockDim.x) {
    uint64_t x = uint64_t(i);
    do {   // cycle walking: a bijection on [0, 2^(2h)) restricted to [0, n)
      uint32_t L = uint32_t(x >> h) & hm, Rr = uint32_t(x) & hm;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t t = L ^ (mix32(Rr ^ key[r]) & hm);
        L = Rr;
        Rr = t;
      }
      x = (uint64_t(L) << h) | Rr;
    } while (int64_t(x) >= n);
    out[i] = int64_t(x);
  }
}

__global__ void k_batch_advance(uint64_t* state) { state[1] += 1; }

}  // namespace

extern "C" int sgnn_batch_indices(int64_t n, int64_t b, uint64_t* state, int64_t* out, void* stream) {
  using sgnn::fail;
  if (n <= 0 || b < 0 || b > n) return fail(SGNN_ERR_INVALID, "need 0 <= b <= n, n > 0");
  if (n > (int64_t(1) << 62)) return fail(SGNN_ERR_INVALID, "n too large");
  if (!state || (b > 0 && !out)) return fail(SGNN_ERR_INVALID, "null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (b > 0) {
    const unsigned blocks = unsigned(b < 65536 * 256 ? (b + 255) / 256 : 65536);
    hipLaunchKernelGGL(k_batch_indices, dim3(blocks), dim3(256), 0, st, n, b, state, out);
    HIP_TRY(hipGetLastError());
  }
  hipLaunchKernelGGL(k_batch_advance, dim3(1), dim3(1), 0, st, state);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}
