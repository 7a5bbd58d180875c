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
