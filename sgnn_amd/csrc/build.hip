// Adjacency build: COO -> canonical CSR of A (+ I), weighted row degree and
// dinv = 1/sqrt(deg); CSR transpose.  Replaces utils.py:54-62 (scipy -> torch COO
// upload), utils.py:82-89 (self-loop COO, A + I, row degree, sqrt, reciprocal —
// redone by the reference on every call) and model/SGNN.py:497-501 (diagonal strip).
//
// Sorting and prefix sums use rocPRIM's device-wide primitives; everything that
// carries the reference's semantics (key packing and diagonal strip, duplicate
// merge in input order, self-loop insertion modes, row pointers, sequential fp32
// degree, two-rounding dinv) is the kernels below.
#include <cstring>  // rocprim's texture iterator calls memset from host code

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace {
using namespace sgnn;

struct BuildWs {
  uint64_t *keys_a, *keys_b;
  uint32_t *idx_a, *idx_b;
  float* rsum;
  uint32_t *keep, *pos;
  int32_t* orow;
  int32_t* flag;  // [0] = invalid index seen
  void* prim;
  size_t prim_bytes;
  size_t total;
};

size_t prim_bytes_for(int64_t m) {
  size_t s1 = 0, s2 = 0;
  uint64_t* k = nullptr;
  uint32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, s1, k, k, v, v, size_t(m), 0u, 64u, hipStream_t(0));
  (void)rocprim::exclusive_scan(nullptr, s2, v, v, 0u, size_t(m), rocprim::plus<uint32_t>(), hipStream_t(0));
  return align_up(s1 > s2 ? s1 : s2);
}

BuildWs carve_build(void* base, int64_t m) {
  BuildWs w{};
  size_t off = 0;
  char* p = static_cast<char*>(base);
  auto take = [&](size_t bytes) {
    char* q = p ? p + off : nullptr;
    off += align_up(bytes);
    return q;
  };
  w.keys_a = reinterpret_cast<uint64_t*>(take(size_t(m) * 8));
  w.keys_b = reinterpret_cast<uint64_t*>(take(size_t(m) * 8));
  w.idx_a = reinterpret_cast<uint32_t*>(take(size_t(m) * 4));
  w.idx_b = reinterpret_cast<uint32_t*>(take(size_t(m) * 4));
  w.rsum = reinterpret_cast<float*>(take(size_t(m) * 4));
  w.keep = reinterpret_cast<uint32_t*>(take(size_t(m) * 4));
  w.pos = reinterpret_cast<uint32_t*>(take(size_t(m) * 4));
  w.orow = reinterpret_cast<int32_t*>(take(size_t(m) * 4));
  w.flag = reinterpret_cast<int32_t*>(take(256));
  w.prim_bytes = prim_bytes_for(m);
  w.prim = take(w.prim_bytes);
  w.total = off;
  return w;
}

// Entries [0, nnz_src) are the input triples, [nnz_src, nnz_in) their mirror images
// (c, r) when the matrix is being symmetrised (nnz_in == 2 * nnz_src then, else
// nnz_in == nnz_src), [nnz_in, m) the appended self-loop markers.
__global__ void k_make_keys(int64_t n, int64_t nnz_src, int64_t nnz_in, int64_t m, const int64_t* __restrict__ row,
                            const int64_t* __restrict__ col, int strip, int cbits, uint64_t sentinel,
                            uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, int32_t* __restrict__ flag) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= m) return;
  uint64_t key;
  if (i < nnz_in) {
    const bool mirror = i >= nnz_src;
    const int64_t e = mirror ? i - nnz_src : i;
    int64_t r = mirror ? col[e] : row[e], c = mirror ? row[e] : col[e];
    if (r < 0 || r >= n || c < 0 || c >= n) {
      flag[0] = 1;
      key = sentinel;
    } else if (strip && r == c) {
      key = sentinel;
    } else {
      key = (uint64_t(r) << cbits) | uint64_t(c);
    }
  } else {
    uint64_t r = uint64_t(i - nnz_in);  // appended self-loop marker of row r
    key = (r << cbits) | r;
  }
  keys[i] = key;
  idx[i] = uint32_t(i);
}

// One thread per sorted entry; the first entry of a run of equal keys sums the
// run in sorted (= input, the sort is stable) order.
__global__ void k_merge_runs(int64_t m, int64_t nnz_src, int64_t nnz_in, const uint64_t* __restrict__ keys,
                             const uint32_t* __restrict__ idx, const float* __restrict__ val, uint64_t sentinel,
                             int cbits, int self_loops, int drop_zeros, int clamp_one, float* __restrict__ rsum,
                             uint32_t* __restrict__ keep) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= m) return;
  uint64_t key = keys[i];
  bool head = key != sentinel && (i == 0 || keys[i - 1] != key);
  if (!head) {
    keep[i] = 0;
    return;
  }
  float s = 0.0f, s_mirror = 0.0f;
  bool any = false, open = true;  // open: the input part of the run has not been closed yet
  bool loop_added = false;
  // REMAINING: a diagonal run keeps the weight of its last input entry (torch_geometric
  // assigns loop weights by index, so later entries overwrite earlier ones), 1 if it has none
  const bool assign = self_loops == SGNN_LOOPS_REMAINING && (key >> cbits) == (key & ((uint64_t(1) << cbits) - 1));
  // The entries as given and their mirror images are summed separately and the two sums
  // added once, as (adj.T + adj) adds two matrices whose duplicates are already merged:
  // the result is bitwise symmetric for any weights.
  auto close = [&]() {
    s += s_mirror;
    if (clamp_one) s = fminf(s, 1.0f);
    open = false;
  };
  for (int64_t j = i; j < m && keys[j] == key; ++j) {
    uint32_t id = idx[j];
    if (id < nnz_in) {
      const bool mirror = id >= nnz_src;
      float v = val ? val[mirror ? id - nnz_src : id] : 1.0f;
      if (assign) s = v; else if (mirror) s_mirror += v; else s += v;
      any = true;
    } else {
      close();  // the marker is the last entry of its run
      if (self_loops == SGNN_LOOPS_ADD || !any) {
        s += 1.0f;
        loop_added = true;
      }
    }
  }
  if (open) close();
  rsum[i] = s;
  // zero-dropping applies to the merged input (what scipy's sparse arithmetic removes);
  // an entry the self-loop was added to stays, even if -1 + 1 made it 0
  keep[i] = (drop_zeros && s == 0.0f && !loop_added) ? 0u : 1u;
}

__global__ void k_scatter(int64_t m, const uint64_t* __restrict__ keys, const float* __restrict__ rsum,
                          const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos, int cbits,
                          int32_t* __restrict__ colidx, float* __restrict__ vals, int32_t* __restrict__ orow) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= m || !keep[i]) return;
  uint32_t p = pos[i];
  uint64_t key = keys[i];
  colidx[p] = int32_t(key & ((uint64_t(1) << cbits) - 1));
  vals[p] = rsum[i];
  orow[p] = int32_t(key >> cbits);
}

// rowptr from the (sorted) row id of every output entry: entry j opens every
// row in (row[j-1], row[j]]; the thread past the end closes the remaining rows.
__global__ void k_rowptr(int64_t n, int64_t m, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos,
                         const int32_t* __restrict__ orow, const int32_t* __restrict__ flag,
                         int32_t* __restrict__ rowptr, int64_t* __restrict__ nnz_out) {
  int64_t j = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  int64_t total = m > 0 ? int64_t(pos[m - 1]) + keep[m - 1] : 0;
  if (j > total) return;
  int64_t lo = j == 0 ? 0 : int64_t(orow[j - 1]) + 1;
  int64_t hi = j == total ? n : int64_t(orow[j]);
  for (int64_t q = lo; q <= hi; ++q) rowptr[q] = int32_t(j);
  if (j == total) nnz_out[0] = flag[0] ? -1 : total;
}

// Weighted row degree and dinv, one wavefront per row.  Summation order (fixed, so
// the result is reproducible and a CPU restatement can match it bit for bit for
// any weights): lane l adds entries l, l+64, l+128, ... of the row in that order,
// then the 64 lane sums are folded in halves (lane i += lane i+32, then +16, ... +1).
__global__ __launch_bounds__(256) void k_degree(int64_t n, const int32_t* __restrict__ rowptr,
                                                const float* __restrict__ vals, int zero_inf,
                                                float* __restrict__ deg, float* __restrict__ dinv) {
  const int lane = threadIdx.x & 63;
  const int64_t r = blockIdx.x * int64_t(blockDim.x / 64) + threadIdx.x / 64;
  if (r >= n) return;
  const int32_t e0 = rowptr[r], e1 = rowptr[r + 1];
  float s = 0.0f;
  for (int32_t e = e0 + lane; e < e1; e += 64) s += vals[e];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) {
    deg[r] = s;
    float d = 1.0f / sqrtf(s);
    if (zero_inf && isinf(d)) d = 0.0f;
    dinv[r] = d;
  }
}

// ---------------------------------------------------------------------------
// transpose
// ---------------------------------------------------------------------------
__global__ void k_transpose_keys(int64_t n, int64_t nnz, const int32_t* __restrict__ rowptr,
                                 const int32_t* __restrict__ colidx, int cbits, uint64_t* __restrict__ keys,
                                 uint32_t* __restrict__ idx) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= nnz) return;
  int64_t lo = 0, hi = n;  // last row whose rowptr <= i
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (int64_t(rowptr[mid]) <= i) lo = mid; else hi = mid;
  }
  keys[i] = (uint64_t(colidx[i]) << cbits) | uint64_t(lo);
  idx[i] = uint32_t(i);
}

__global__ void k_transpose_out(int64_t n, int64_t nnz, const uint64_t* __restrict__ keys,
                                const uint32_t* __restrict__ idx, const float* __restrict__ vals, int cbits,
                                int32_t* __restrict__ t_rowptr, int32_t* __restrict__ t_colidx,
                                float* __restrict__ t_vals) {
  int64_t j = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (j > nnz) return;
  if (j < nnz) {
    uint64_t key = keys[j];
    t_colidx[j] = int32_t(key & ((uint64_t(1) << cbits) - 1));
    if (vals) t_vals[j] = vals[idx[j]];
  }
  int64_t lo = j == 0 ? 0 : int64_t(keys[j - 1] >> cbits) + 1;
  int64_t hi = j == nnz ? n : int64_t(keys[j] >> cbits);
  for (int64_t q = lo; q <= hi; ++q) t_rowptr[q] = int32_t(j);
}

}  // namespace

extern "C" {

size_t sgnn_build_workspace_bytes(int64_t n, int64_t nnz_in) {
  if (n < 0 || nnz_in < 0) return 0;
  return carve_build(nullptr, nnz_in + n).total;
}

int sgnn_build_norm_csr(int64_t n, int64_t nnz_in, const int64_t* row, const int64_t* col, const float* val,
                        int flags, int self_loops, int zero_inf, int32_t* rowptr, int32_t* colidx, float* vals,
                        float* deg, float* dinv, int64_t* nnz_out, void* workspace, size_t workspace_bytes,
                        void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0 || nnz_in < 0) return fail(SGNN_ERR_INVALID, "n must be positive and nnz_in non-negative");
  if (flags & ~(SGNN_STRIP_DIAG | SGNN_DROP_ZEROS | SGNN_SYMMETRIZE | SGNN_CLAMP_ONE))
    return fail(SGNN_ERR_INVALID, "unknown build flag");
  const int64_t nnz_src = nnz_in;
  if (flags & SGNN_SYMMETRIZE) nnz_in *= 2;
  if (n >= (int64_t(1) << 31) || nnz_in + n >= (int64_t(1) << 31))
    return fail(SGNN_ERR_INVALID, "n and the entry count (doubled when symmetrising) + n must stay below 2^31");
  if ((nnz_src > 0 && (!row || !col)) || !rowptr || !colidx || !vals || !deg || !dinv || !nnz_out)
    return fail(SGNN_ERR_INVALID, "null pointer");
  if (self_loops < SGNN_LOOPS_NONE || self_loops > SGNN_LOOPS_REMAINING)
    return fail(SGNN_ERR_INVALID, "unknown self-loop mode");
  const int64_t m = nnz_in + (self_loops != SGNN_LOOPS_NONE ? n : 0);
  BuildWs w = carve_build(workspace, nnz_in + n);
  if (!workspace || workspace_bytes < w.total)
    return fail(SGNN_ERR_WORKSPACE, "build workspace too small: need " + std::to_string(w.total) + " bytes");
  const int cbits = ceil_log2(n);
  const uint64_t sentinel = uint64_t(1) << (2 * cbits);
  const unsigned T = 256;
  HIP_TRY(hipMemsetAsync(w.flag, 0, 256, st));
  if (m == 0) {
    HIP_TRY(hipMemsetAsync(rowptr, 0, size_t(n + 1) * 4, st));
    HIP_TRY(hipMemsetAsync(nnz_out, 0, 8, st));
    hipLaunchKernelGGL(k_degree, dim3(unsigned((n + 3) / 4)), dim3(256), 0, st, n, rowptr, vals, zero_inf, deg, dinv);
    HIP_TRY(hipGetLastError());
    return SGNN_OK;
  }
  const unsigned gm = unsigned((m + T - 1) / T);
  hipLaunchKernelGGL(k_make_keys, dim3(gm), dim3(T), 0, st, n, nnz_src, nnz_in, m, row, col, (flags & SGNN_STRIP_DIAG) ? 1 : 0,
                     cbits, sentinel, w.keys_a, w.idx_a, w.flag);
  HIP_TRY(hipGetLastError());
  size_t pb = w.prim_bytes;
  HIP_TRY(rocprim::radix_sort_pairs(w.prim, pb, w.keys_a, w.keys_b, w.idx_a, w.idx_b, size_t(m), 0u,
                                    unsigned(2 * cbits + 1), st));
  hipLaunchKernelGGL(k_merge_runs, dim3(gm), dim3(T), 0, st, m, nnz_src, nnz_in, w.keys_b, w.idx_b, val, sentinel, cbits,
                     self_loops, (flags & SGNN_DROP_ZEROS) ? 1 : 0, (flags & SGNN_CLAMP_ONE) ? 1 : 0, w.rsum, w.keep);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.keep, w.pos, 0u, size_t(m), rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_scatter, dim3(gm), dim3(T), 0, st, m, w.keys_b, w.rsum, w.keep, w.pos, cbits, colidx, vals,
                     w.orow);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_rowptr, dim3(unsigned((m + 1 + T - 1) / T)), dim3(T), 0, st, n, m, w.keep, w.pos, w.orow,
                     w.flag, rowptr, nnz_out);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_degree, dim3(unsigned((n + 3) / 4)), dim3(256), 0, st, n, rowptr, vals, zero_inf, deg, dinv);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

size_t sgnn_transpose_workspace_bytes(int64_t n, int64_t nnz) {
  if (n < 0 || nnz < 0) return 0;
  return carve_build(nullptr, nnz > 0 ? nnz : 1).total;
}

int sgnn_csr_transpose(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                       int32_t* t_rowptr, int32_t* t_colidx, float* t_vals, void* workspace, size_t workspace_bytes,
                       void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0 || nnz < 0 || n >= (int64_t(1) << 31) || nnz >= (int64_t(1) << 31))
    return fail(SGNN_ERR_INVALID, "bad size");
  if (!rowptr || !t_rowptr || (nnz > 0 && (!colidx || !t_colidx)) || (vals && !t_vals))
    return fail(SGNN_ERR_INVALID, "null pointer");
  if (nnz == 0) {
    HIP_TRY(hipMemsetAsync(t_rowptr, 0, size_t(n + 1) * 4, st));
    return SGNN_OK;
  }
  BuildWs w = carve_build(workspace, nnz);
  if (!workspace || workspace_bytes < w.total)
    return fail(SGNN_ERR_WORKSPACE, "transpose workspace too small: need " + std::to_string(w.total) + " bytes");
  const int cbits = ceil_log2(n);
  const unsigned T = 256;
  hipLaunchKernelGGL(k_transpose_keys, dim3(unsigned((nnz + T - 1) / T)), dim3(T), 0, st, n, nnz, rowptr, colidx,
                     cbits, w.keys_a, w.idx_a);
  HIP_TRY(hipGetLastError());
  size_t pb = w.prim_bytes;
  HIP_TRY(rocprim::radix_sort_pairs(w.prim, pb, w.keys_a, w.keys_b, w.idx_a, w.idx_b, size_t(nnz), 0u,
                                    unsigned(2 * cbits), st));
  hipLaunchKernelGGL(k_transpose_out, dim3(unsigned((nnz + 1 + T - 1) / T)), dim3(T), 0, st, n, nnz, w.keys_b,
                     w.idx_b, vals, cbits, t_rowptr, t_colidx, t_vals);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

}  // extern "C"
