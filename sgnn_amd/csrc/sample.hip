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
// (seed, counter) in device memory; the launch advances the counter, so replaying a
// captured HIP graph draws a new batch every time.  state[2] counts the workgroups that
// have finished: each reads the key before it counts itself in, and the last one advances
// the counter, so no workgroup of a launch sees the key change under it (nothing else
// passes between workgroups, so no fence is involved).
namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // murmur3 finaliser
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void k_batch_indices(int64_t n, int64_t b, unsigned long long* state,
                                                       const int64_t* __restrict__ map, int64_t* __restrict__ out) {
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
    out[i] = map ? map[x] : int64_t(x);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long done = atomicAdd(&state[2], 1ull);
    if (done == gridDim.x - 1ull) {
      state[2] = 0ull;
      state[1] = counter + 1ull;
    }
  }
}

}  // namespace

extern "C" int sgnn_batch_indices(int64_t n, int64_t b, uint64_t* state, const int64_t* map, int64_t* out,
                                  void* stream) {
  using sgnn::fail;
  if (n <= 0 || b < 0 || b > n) return fail(SGNN_ERR_INVALID, "need 0 <= b <= n, n > 0");
  if (n > (int64_t(1) << 62)) return fail(SGNN_ERR_INVALID, "n too large");
  if (!state || (b > 0 && !out)) return fail(SGNN_ERR_INVALID, "null pointer");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned blocks = unsigned(b < 65536 * 256 ? (b + 255) / 256 : 65536);
  hipLaunchKernelGGL(k_batch_indices, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, n, b,
                     reinterpret_cast<unsigned long long*>(state), map, out);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

// ---------------------------------------------------------------------------------
// Reconstruction loss of the graph auto-encoder layer on one batch, fused.
//
// Reference (model/SGNN.py:138-164, 209-226, 682-702), per optimiser iteration:
// dense target block A_s = A[idx][:, idx]; ignore mask M = rand(b, b) <= rate made
// symmetric, diagonal cleared, merged with the overlook block; retain = 1 - M;
// P = sigmoid(clamp(Z Z^T, +-40));
//   loss = mean(-ratio * A_s * log(max(P, 1e-6)) * retain)  [only if num_pos > 0]
//        + mean(-(1 - A_s) * log(max(1 - P, 1e-6)) * retain),
//   num_pos = sum(A_s * retain), ratio = sum(retain) / num_pos - 1.
// That is ~60 elementwise / reduction kernels over b x b plus their backward.  Here:
//   k_gae_targets  one workgroup per batch row: looks every A_s entry up in the CSR
//                  (binary search, as k_sample_block), draws the ignore mask from a
//                  counter-based hash of the unordered pair (one uniform per pair
//                  against 2*rate - rate^2, the probability that either of the
//                  reference's two draws masks it), applies the overlook entries, and
//                  writes A_s, retain and the row's sums of A_s*retain and retain;
//   k_gae_grad     elementwise over S = Z Z^T (computed by the caller's GEMM): the loss
//                  terms and G = dloss/dS, with per-row loss sums.
// The caller finishes with dZ = (G + G^T) Z.  Sums are per row, then added in row
// order by every reader: results do not depend on scheduling.
namespace {

__device__ __forceinline__ float csr_lookup(const int* __restrict__ colidx, const float* __restrict__ vals, int lo0,
                                            int hi0, int64_t want) {
  int lo = lo0, hi = hi0;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (int64_t(colidx[mid]) < want) lo = mid + 1; else hi = mid;
  }
  if (lo < hi0 && int64_t(colidx[lo]) == want) return vals ? vals[lo] : 1.f;
  return 0.f;
}

constexpr int kStageCols = 6144, kStageOverlook = 2048;      // 32 KB of LDS per workgroup
constexpr int kLongPivots = 1024;

// piv[p] = src[p * stride] for p < n, by the whole workgroup, eight loads in flight per thread
// (issued one at a time, each would cost a trip to HBM: the rows are cold).
__device__ __forceinline__ void stage_pivots(int* piv, int n, const int* __restrict__ src, int stride) {
  for (int base = threadIdx.x; base < n; base += 8 * blockDim.x) {
    int v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = base + e * blockDim.x;
      v[e] = p < n ? src[int64_t(p) * stride] : 0;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int p = base + e * blockDim.x;
      if (p < n) piv[p] = v[e];
    }
  }
}

// Position in colidx[lo0, hi0) (ascending) of `want`, or -1.  piv[p] = colidx[lo0 + p * stride]
// for p < n_piv sits in LDS: the search runs over the pivots first and finishes inside one
// stride of the global list (stride 1: the whole list is staged and no global step is left).
__device__ __forceinline__ int staged_find(const int* piv, int n_piv, int stride, const int* __restrict__ colidx,
                                           int lo0, int hi0, int64_t want) {
  int lo = 0, hi = n_piv;                         // first pivot > want
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (int64_t(piv[mid]) <= want) lo = mid + 1; else hi = mid;
  }
  if (lo == 0) return -1;                         // smaller than the first entry (or empty row)
  const int p = lo - 1;
  if (int64_t(piv[p]) == want) return lo0 + p * stride;
  if (stride == 1) return -1;
  int a = lo0 + p * stride + 1, b = lo0 + (p + 1) * stride < hi0 ? lo0 + (p + 1) * stride : hi0;
  while (a < b) {
    const int mid = a + ((b - a) >> 1);
    const int64_t c = colidx[mid];
    if (c == want) return mid;
    if (c < want) a = mid + 1; else b = mid;
  }
  return -1;
}

__global__ __launch_bounds__(256) void k_gae_targets(int64_t n_rows, const int* __restrict__ rowptr,
                                                     const int* __restrict__ colidx, const float* __restrict__ vals,
                                                     const int* __restrict__ o_rowptr, const int* __restrict__ o_colidx,
                                                     int64_t b, const int64_t* __restrict__ idx, float p_mask,
                                                     const uint64_t* __restrict__ state, float* __restrict__ target,
                                                     float* __restrict__ retain, float* __restrict__ row_pos,
                                                     float* __restrict__ row_ret) {
  const int64_t i = blockIdx.x;
  // this thread's columns of the block, fetched ahead of the chain idx[i] -> rowptr -> colidx
  // below so that their latency hides behind it (the first kPre trips of the j loop)
  constexpr int kPre = 4;
  int64_t pre[kPre];
#pragma unroll
  for (int e = 0; e < kPre; ++e) {
    const int64_t j = threadIdx.x + int64_t(e) * blockDim.x;
    pre[e] = j < b ? idx[j] : -1;
  }
  const int64_t node = idx[i];
  const bool ok = node >= 0 && node < n_rows;
  const int lo0 = ok ? rowptr[node] : 0, hi0 = ok ? rowptr[node + 1] : 0;
  const int olo = (ok && o_rowptr) ? o_rowptr[node] : 0, ohi = (ok && o_rowptr) ? o_rowptr[node + 1] : 0;
  // The b binary searches of this row all walk the same column list: stage it in LDS with one
  // coalesced pass, so a search step costs an LDS read instead of a dependent global load.
  // Of a row longer than the buffer every stride-th entry is staged; the search then ends
  // with a few global steps inside one stride (a hub row of 10^5 entries: 5 instead of 17).
  __shared__ int s_cols[kStageCols];
  __shared__ int s_ocols[kStageOverlook];
  const int len = hi0 - lo0, olen = ohi - olo;
  // a row that does not fit is sampled at kLongPivots points only: one CU streaming a hub row of
  // 10^5 entries through its pivots would take longer than the searches it saves
  const int stride = len <= kStageCols ? 1 : (len + kLongPivots - 1) / kLongPivots;
  const int ostride = olen <= kStageOverlook ? 1 : (olen + kLongPivots - 1) / kLongPivots;
  const int n_piv = (len + stride - 1) / stride, n_opiv = (olen + ostride - 1) / ostride;
  stage_pivots(s_cols, n_piv, colidx + lo0, stride);
  stage_pivots(s_ocols, n_opiv, o_colidx + olo, ostride);
  __syncthreads();
  const uint64_t seed = state[0], counter = state[1];
  const uint32_t k0 = mix32(uint32_t(seed) ^ mix32(uint32_t(counter) + 0x9e3779b9u));
  const uint32_t k1 = mix32(uint32_t(seed >> 32) + mix32(uint32_t(counter >> 32) ^ 0x85ebca6bu));
  const uint32_t thresh = p_mask >= 1.f ? 0xffffffffu : uint32_t(double(p_mask) * 4294967296.0);
  float s_pos = 0.f, s_ret = 0.f;
  int trip = 0;
  for (int64_t j = threadIdx.x; j < b; j += blockDim.x, ++trip) {
    int64_t want;
    switch (trip) {                               // constant indices keep pre[] in registers
      case 0: want = pre[0]; break;
      case 1: want = pre[1]; break;
      case 2: want = pre[2]; break;
      case 3: want = pre[3]; break;
      default: want = idx[j];
    }
    const bool self = want == node;
    float a = 0.f;
    if (!self) {
      const int pos = staged_find(s_cols, n_piv, stride, colidx, lo0, hi0, want);
      if (pos >= 0) a = vals ? vals[pos] : 1.f;
    }
    // one uniform per unordered pair of batch positions
    const uint32_t lo_ij = uint32_t(i < j ? i : j), hi_ij = uint32_t(i < j ? j : i);
    const uint32_t u = mix32(mix32(lo_ij ^ k0) + hi_ij * 0x9e3779b9u + k1);
    bool ignore = (i != j) && (p_mask > 0.f) && (u < thresh || p_mask >= 1.f);
    if (self) ignore = true;                                            // the overlook matrix has a unit diagonal
    else if (o_rowptr && staged_find(s_ocols, n_opiv, ostride, o_colidx, olo, ohi, want) >= 0) ignore = true;
    const float r = ignore ? 0.f : 1.f;
    target[i * b + j] = a;
    retain[i * b + j] = r;
    s_pos += a * r;
    s_ret += r;
  }
  // block sum in a fixed order: lanes fold in halves, then the 4 wavefronts in order
  __shared__ float sh[2][4];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    s_pos += __shfl_down(s_pos, off);
    s_ret += __shfl_down(s_ret, off);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = s_pos;
    sh[1][threadIdx.x >> 6] = s_ret;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    row_pos[i] = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
    row_ret[i] = ((sh[1][0] + sh[1][1]) + sh[1][2]) + sh[1][3];
  }
}

__global__ __launch_bounds__(256) void k_gae_grad(int64_t b, const float* __restrict__ S, int64_t lds,
                                                  const float* __restrict__ target, const float* __restrict__ retain,
                                                  const float* __restrict__ row_pos, const float* __restrict__ row_ret,
                                                  float* __restrict__ G, float* __restrict__ row_loss, uint64_t* state) {
  const int64_t i = blockIdx.x;
  // The mask of this batch has been drawn by the k_gae_targets launch that precedes this one on
  // the stream, so the counter can move on here without any workgroup counting (a ticket of one
  // atomic per workgroup costs ~12 ns apiece on one word: 5 us for 512 of them).
  if (state != nullptr && i == 0 && threadIdx.x == 0) state[1] += 1;
  __shared__ float tsh[2][4];
  __shared__ float sh[4];
  {   // every workgroup adds the b row sums in the same order: thread t takes t, t + 256, ...,
      // the lanes fold in halves, then the 4 wavefronts in order
    float tp = 0.f, tr = 0.f;
    for (int64_t r = threadIdx.x; r < b; r += blockDim.x) {
      tp += row_pos[r];
      tr += row_ret[r];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      tp += __shfl_down(tp, off);
      tr += __shfl_down(tr, off);
    }
    if ((threadIdx.x & 63) == 0) {
      tsh[0][threadIdx.x >> 6] = tp;
      tsh[1][threadIdx.x >> 6] = tr;
    }
  }
  __syncthreads();
  const float num_pos = ((tsh[0][0] + tsh[0][1]) + tsh[0][2]) + tsh[0][3];
  const float tot_ret = ((tsh[1][0] + tsh[1][1]) + tsh[1][2]) + tsh[1][3];
  const float ratio = num_pos > 0.f ? tot_ret / num_pos - 1.f : 0.f;
  const float inv = 1.f / (float(b) * float(b));
  float loss = 0.f;
  for (int64_t j = threadIdx.x; j < b; j += blockDim.x) {
    const float s = S[i * lds + j];
    const float sc = fminf(fmaxf(s, -40.f), 40.f);
    const float p = 1.f / (1.f + expf(-sc));
    const float a = target[i * b + j], r = retain[i * b + j];
    const float q = 1.f - p;
    float g = 0.f;
    if (num_pos > 0.f) {
      loss += -ratio * a * logf(fmaxf(p, 1e-6f)) * r;
      if (p >= 1e-6f) g += -ratio * a * r * q;
    }
    loss += -(1.f - a) * logf(fmaxf(q, 1e-6f)) * r;
    if (q >= 1e-6f) g += (1.f - a) * r * p;
    if (!(s >= -40.f && s <= 40.f)) g = 0.f;   // outside the clamp nothing flows back
    G[i * b + j] = g * inv;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) loss += __shfl_down(loss, off);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = loss;
  __syncthreads();
  if (threadIdx.x == 0) row_loss[i] = (((sh[0] + sh[1]) + sh[2]) + sh[3]) * inv;
}

}  // namespace

extern "C" int sgnn_gae_targets_f32(int64_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                                    const float* vals, const int32_t* overlook_rowptr,
                                    const int32_t* overlook_colidx, int64_t b, const int64_t* idx, float mask_rate,
                                    uint64_t* state, float* target, float* retain, float* row_pos, float* row_ret,
                                    void* stream) {
  using sgnn::fail;
  if (n_rows < 0 || nnz < 0 || b < 0) return fail(SGNN_ERR_INVALID, "negative size");
  if (!(mask_rate >= 0.f && mask_rate <= 1.f)) return fail(SGNN_ERR_INVALID, "mask_rate must lie in [0, 1]");
  if (b == 0) return SGNN_OK;
  if (b > 0x7fffffff) return fail(SGNN_ERR_INVALID, "batch larger than 2^31-1");
  if (!rowptr || !idx || !state || !target || !retain || !row_pos || !row_ret || (nnz > 0 && !colidx))
    return fail(SGNN_ERR_INVALID, "null pointer");
  if ((overlook_rowptr == nullptr) != (overlook_colidx == nullptr))
    return fail(SGNN_ERR_INVALID, "overlook rowptr and colidx go together");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float p = 2.f * mask_rate - mask_rate * mask_rate;   // either of the two draws masks the pair
  hipLaunchKernelGGL(k_gae_targets, dim3(unsigned(b)), dim3(256), 0, st, n_rows, rowptr, colidx, vals, overlook_rowptr,
                     overlook_colidx, b, idx, p, state, target, retain, row_pos, row_ret);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

extern "C" int sgnn_gae_loss_grad_f32(int64_t b, const float* S, int64_t lds, const float* target,
                                      const float* retain, const float* row_pos, const float* row_ret, float* G,
                                      float* row_loss, uint64_t* state, void* stream) {
  using sgnn::fail;
  if (b < 0) return fail(SGNN_ERR_INVALID, "negative size");
  if (b == 0) return SGNN_OK;
  if (lds < b) return fail(SGNN_ERR_INVALID, "lds smaller than the block width");
  if (!S || !target || !retain || !row_pos || !row_ret || !G || !row_loss)
    return fail(SGNN_ERR_INVALID, "null pointer");
  hipLaunchKernelGGL(k_gae_grad, dim3(unsigned(b)), dim3(256), 0, static_cast<hipStream_t>(stream), b, S, lds, target,
                     retain, row_pos, row_ret, G, row_loss, state);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}
