// libsgnn_hip.so — gfx950 kernels and C ABI for SGNN's sparse propagation path.
//
//   A_hat = D^-1/2 (A + I) D^-1/2  : sgnn_build_norm_csr   (utils.py:54-62,82-89; model/SGNN.py:497-501)
//   Y     = A_hat^K . X            : sgnn_spmm_csr_f32 / sgnn_spmm_chain_f32   (utils.py:91-96)
//
// The propagation is a sparse gather-reduce: per stored entry one row of X is
// fetched and added into one row of Y.  There is no dense contraction, so no
// MFMA; the kernels are organised around the memory system instead
// (64-lane wavefronts laid along the feature dimension, 8/16-byte loads per
// lane, several independent row fetches in flight per wavefront, long rows cut
// into fixed-size segments so that no wavefront owns a 10^5-entry hub).
#include <hip/hip_runtime.h>

#include <cstring>  // rocprim's texture iterator calls memset from host code

#include <rocprim/rocprim.hpp>

#include <cstdint>
#include <cstdio>
#include <string>

#include "sgnn_hip.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(SGNN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

inline int ceil_log2(int64_t n) {
  int b = 1;
  while ((int64_t(1) << b) < n) ++b;
  return b;
}

// ---------------------------------------------------------------------------
// adjacency build
// ---------------------------------------------------------------------------
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

__global__ void k_make_keys(int64_t n, int64_t nnz_in, int64_t m, const int64_t* __restrict__ row,
                            const int64_t* __restrict__ col, int strip, int cbits, uint64_t sentinel,
                            uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, int32_t* __restrict__ flag) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= m) return;
  uint64_t key;
  if (i < nnz_in) {
    int64_t r = row[i], c = col[i];
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
__global__ void k_merge_runs(int64_t m, int64_t nnz_in, const uint64_t* __restrict__ keys,
                             const uint32_t* __restrict__ idx, const float* __restrict__ val, uint64_t sentinel,
                             int cbits, int self_loops, int drop_zeros, float* __restrict__ rsum,
                             uint32_t* __restrict__ keep) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= m) return;
  uint64_t key = keys[i];
  bool head = key != sentinel && (i == 0 || keys[i - 1] != key);
  if (!head) {
    keep[i] = 0;
    return;
  }
  float s = 0.0f;
  bool any = false;
  // REMAINING: a diagonal run keeps the weight of its last input entry (torch_geometric
  // assigns loop weights by index, so later entries overwrite earlier ones), 1 if it has none
  const bool assign = self_loops == SGNN_LOOPS_REMAINING && (key >> cbits) == (key & ((uint64_t(1) << cbits) - 1));
  for (int64_t j = i; j < m && keys[j] == key; ++j) {
    uint32_t id = idx[j];
    if (id < nnz_in) {
      float v = val ? val[id] : 1.0f;
      s = assign ? v : s + v;
      any = true;
    } else if (self_loops == SGNN_LOOPS_ADD || !any) {
      s += 1.0f;  // the marker is the last entry of its run
    }
  }
  rsum[i] = s;
  keep[i] = (drop_zeros && s == 0.0f) ? 0u : 1u;
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

__global__ void k_degree(int64_t n, const int32_t* __restrict__ rowptr, const float* __restrict__ vals, int zero_inf,
                         float* __restrict__ deg, float* __restrict__ dinv) {
  int64_t r = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (r >= n) return;
  float s = 0.0f;
  for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) s += vals[e];
  deg[r] = s;
  float d = 1.0f / sqrtf(s);
  if (zero_inf && isinf(d)) d = 0.0f;
  dinv[r] = d;
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

// ---------------------------------------------------------------------------
// SpMM
// ---------------------------------------------------------------------------
constexpr int kWave = 64;
constexpr int kSeg = 1024;          // most entries one wavefront reduces; longer rows are cut into segments
constexpr int kWavesPerBlock = 4;

template <int V, int AL>
__device__ __forceinline__ void load_vec(const float* p, float (&r)[V]) {
  if constexpr (V == 1) {
    r[0] = *p;
  } else {
    typedef float vt __attribute__((ext_vector_type(V), aligned(AL)));
    vt v = *reinterpret_cast<const vt*>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = v[i];
  }
}

template <int V, int AL>
__device__ __forceinline__ void store_vec(float* p, const float (&r)[V]) {
  if constexpr (V == 1) {
    *p = r[0];
  } else {
    typedef float vt __attribute__((ext_vector_type(V), aligned(AL)));
    vt v;
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = r[i];
    *reinterpret_cast<vt*>(p) = v;
  }
}

struct SpmmArgs {
  const int32_t* rowptr;
  const int32_t* colidx;
  const float* vals;
  const float* dinv_row;
  const float* dinv_col;
  const float* X;
  float* Y;
  int64_t ldx, ldy;
  int32_t n_rows;
  int32_t F;
  int epilogue;
  // long-row machinery (device scratch)
  int32_t* counters;   // [0] segments appended, [1] long rows appended
  int32_t* seg_row;    // [max_segs] row of segment
  int32_t* seg_k;      // [max_segs] index of the segment inside its row
  int32_t* long_row;   // [max_long] row
  int32_t* long_slot;  // [max_long] first partial slot of the row
  float* partial;      // [max_segs, F]
  int32_t max_segs, max_long;
};

// Reduce entries [e0, e1) of one row into acc.  The wavefront is split into
// G = 64/LPR groups of LPR lanes; group g takes entries g, g+G, ... and lane l
// of a group owns features f0 + c*LPR*V + l*V .. +V of every fetched row
// (c < C).  The groups are combined by a shuffle tree at the end, so the
// summation order is a function of the row's entries alone.
template <int V, int AL, int LPR, int C, int U>
__device__ __forceinline__ void reduce_entries(const SpmmArgs& a, int32_t e0, int32_t e1, int f0, float (&acc)[C][V]) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1);
  const int g = lane / LPR;
  const int l = lane % LPR;
  bool fok[C];
#pragma unroll
  for (int c = 0; c < C; ++c) fok[c] = (f0 + (c * LPR + l) * V) < a.F;
  const float* xl = a.X + f0 + l * V;

  for (int32_t base = e0; base < e1; base += kWave) {
    int32_t e = base + lane;
    int32_t cj = 0;
    float wj = 0.0f;
    if (e < e1) {
      cj = a.colidx[e];
      wj = a.vals ? a.vals[e] : 1.0f;
      if (a.dinv_col) wj *= a.dinv_col[cj];
    }
    const int cnt = min(kWave, e1 - base);
    const int steps = (cnt + G - 1) / G;
    for (int k0 = 0; k0 < steps; k0 += U) {
      float xv[U][C][V];
      float w[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int src = (k0 + u) * G + g;
        int32_t cc;
        if constexpr (G == 1) {
          int s = min(src, kWave - 1);
          cc = __builtin_amdgcn_readlane(cj, s);
          w[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wj), s));
        } else {
          cc = __shfl(cj, src & (kWave - 1));
          w[u] = __shfl(wj, src & (kWave - 1));
        }
        ok[u] = src < cnt;
        const float* xr = xl + int64_t(cc) * a.ldx;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          if (ok[u] && fok[c]) {
            load_vec<V, AL>(xr + c * LPR * V, xv[u][c]);
          } else {
#pragma unroll
            for (int v = 0; v < V; ++v) xv[u][c][v] = 0.0f;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ok[u]) {
#pragma unroll
          for (int c = 0; c < C; ++c)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[c][v] = fmaf(w[u], xv[u][c][v], acc[c][v]);
        }
      }
    }
  }
  if constexpr (G > 1) {
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
#pragma unroll
      for (int c = 0; c < C; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[c][v] += __shfl_xor(acc[c][v], off);
    }
  }
}

template <int V, int AL, int LPR, int C>
__device__ __forceinline__ void write_row(const SpmmArgs& a, float* dst, int32_t row, int f0, float (&acc)[C][V],
                                          bool final_row) {
  const int lane = threadIdx.x & (kWave - 1);
  if (lane >= LPR) return;
  float dr = 1.0f;
  const bool scale = final_row && a.dinv_row != nullptr;
  if (scale) dr = a.dinv_row[row];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    int f = f0 + (c * LPR + lane) * V;
    if (f < a.F) {
      float o[V];
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float y = acc[c][v];
        if (scale) {
          y = y * dr;
          if (a.epilogue == SGNN_EPI_SCALE_ROW2) y = y * dr;
        }
        o[v] = y;
      }
      store_vec<V, AL>(dst + f, o);
    }
  }
}

// list the rows longer than kSeg and give each of their segments a partial slot
__global__ void k_collect_long(SpmmArgs a) {
  int64_t row = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (row >= a.n_rows) return;
  int32_t len = a.rowptr[row + 1] - a.rowptr[row];
  if (len <= kSeg) return;
  int32_t nseg = (len + kSeg - 1) / kSeg;
  int32_t slot = atomicAdd(&a.counters[0], nseg);
  int32_t li = atomicAdd(&a.counters[1], 1);
  if (slot + nseg > a.max_segs || li >= a.max_long) return;  // cannot happen with the sized workspace
  a.long_row[li] = int32_t(row);
  a.long_slot[li] = slot;
  for (int32_t k = 0; k < nseg; ++k) {
    a.seg_row[slot + k] = int32_t(row);
    a.seg_k[slot + k] = k;
  }
}

// The SpMM kernel.  One wavefront per (work item, feature tile).  Blocks
// [0, row_blocks) take the rows of at most kSeg entries, one row per wavefront,
// and write Y directly; the blocks after them take the segments of the longer
// rows and write unscaled partial sums (combined by k_spmm_combine).
template <int V, int AL, int LPR, int C, int U>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_spmm(SpmmArgs a, unsigned row_blocks) {
  const int wave = threadIdx.x / kWave;
  const int f0 = blockIdx.y * (LPR * V * C);
  float acc[C][V];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int v = 0; v < V; ++v) acc[c][v] = 0.0f;
  if (blockIdx.x < row_blocks) {
    const int64_t row = int64_t(blockIdx.x) * kWavesPerBlock + wave;
    if (row >= a.n_rows) return;
    const int32_t e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
    if (e1 - e0 > kSeg) return;  // cut into segments below
    reduce_entries<V, AL, LPR, C, U>(a, e0, e1, f0, acc);
    write_row<V, AL, LPR, C>(a, a.Y + row * a.ldy, int32_t(row), f0, acc, true);
    return;
  }
  const int64_t s = int64_t(blockIdx.x - row_blocks) * kWavesPerBlock + wave;
  const int32_t nseg = min(a.counters[0], a.max_segs);
  if (s >= nseg) return;
  const int32_t row = a.seg_row[s];
  const int32_t k = a.seg_k[s];
  const int32_t r0 = a.rowptr[row], r1 = a.rowptr[row + 1];
  const int32_t e0 = r0 + k * kSeg;
  const int32_t e1 = min(e0 + kSeg, r1);
  reduce_entries<V, AL, LPR, C, U>(a, e0, e1, f0, acc);
  // partial rows are dense [F]; their starts are 16-byte aligned only when F % 4 == 0: scalar stores
  const int lane = threadIdx.x & (kWave - 1);
  if (lane < LPR) {
    float* dst = a.partial + s * int64_t(a.F);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      int f = f0 + (c * LPR + lane) * V;
      if (f < a.F) {
#pragma unroll
        for (int v = 0; v < V; ++v) dst[f + v] = acc[c][v];
      }
    }
  }
}

// sum a long row's partials in segment order, then the row epilogue
__global__ void k_spmm_combine(SpmmArgs a) {
  const int32_t nlong = min(a.counters[1], a.max_long);
  const int32_t li = blockIdx.x;
  if (li >= nlong) return;
  const int32_t row = a.long_row[li];
  const int32_t slot = a.long_slot[li];
  const int32_t len = a.rowptr[row + 1] - a.rowptr[row];
  const int32_t nseg = (len + kSeg - 1) / kSeg;
  const bool scale = a.dinv_row != nullptr;
  const float dr = scale ? a.dinv_row[row] : 1.0f;
  for (int f = blockIdx.y * blockDim.x + threadIdx.x; f < a.F; f += gridDim.y * blockDim.x) {
    float s = 0.0f;
    for (int32_t k = 0; k < nseg; ++k) s += a.partial[(slot + k) * int64_t(a.F) + f];
    if (scale) {
      s = s * dr;
      if (a.epilogue == SGNN_EPI_SCALE_ROW2) s = s * dr;
    }
    a.Y[row * a.ldy + f] = s;
  }
}

__global__ void k_row_scale(int64_t n, int64_t F, const float* __restrict__ scale, const float* __restrict__ in,
                            int64_t ldin, float* __restrict__ out, int64_t ldout) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n * F) return;
  int64_t r = i / F, f = i - r * F;
  out[r * ldout + f] = in[r * ldin + f] * scale[r];
}

struct SpmmWs {
  int32_t *counters, *seg_row, *seg_k, *long_row, *long_slot;
  float* partial;
  int32_t max_segs, max_long;
  size_t total;
};

SpmmWs carve_spmm(void* base, int64_t nnz, int64_t F) {
  SpmmWs w{};
  int64_t max_long = nnz / kSeg + 1;       // rows with more than kSeg entries
  int64_t max_segs = nnz / kSeg + max_long + 1;
  size_t off = 0;
  char* p = static_cast<char*>(base);
  auto take = [&](size_t bytes) {
    char* q = p ? p + off : nullptr;
    off += align_up(bytes);
    return q;
  };
  w.counters = reinterpret_cast<int32_t*>(take(256));
  w.seg_row = reinterpret_cast<int32_t*>(take(size_t(max_segs) * 4));
  w.seg_k = reinterpret_cast<int32_t*>(take(size_t(max_segs) * 4));
  w.long_row = reinterpret_cast<int32_t*>(take(size_t(max_long) * 4));
  w.long_slot = reinterpret_cast<int32_t*>(take(size_t(max_long) * 4));
  w.partial = reinterpret_cast<float*>(take(size_t(max_segs) * size_t(F) * 4));
  w.max_segs = int32_t(max_segs);
  w.max_long = int32_t(max_long);
  w.total = off;
  return w;
}

template <int V, int AL, int LPR, int C, int U>
int launch_spmm(const SpmmArgs& a, hipStream_t st) {
  const int tile = LPR * V * C;
  const unsigned ftiles = unsigned((a.F + tile - 1) / tile);
  const unsigned row_blocks = unsigned((int64_t(a.n_rows) + kWavesPerBlock - 1) / kWavesPerBlock);
  const unsigned seg_blocks = unsigned((int64_t(a.max_segs) + kWavesPerBlock - 1) / kWavesPerBlock);
  hipLaunchKernelGGL((k_spmm<V, AL, LPR, C, U>), dim3(row_blocks + seg_blocks, ftiles),
                     dim3(kWave * kWavesPerBlock), 0, st, a, row_blocks);
  HIP_TRY(hipGetLastError());
  const unsigned fy = unsigned((a.F + 255) / 256 > 8 ? 8 : (a.F + 255) / 256);
  hipLaunchKernelGGL(k_spmm_combine, dim3(unsigned(a.max_long), fy), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

// pick the lane layout from F and the alignment of X / Y
int dispatch_spmm(const SpmmArgs& a, hipStream_t st) {
  const uintptr_t px = reinterpret_cast<uintptr_t>(a.X), py = reinterpret_cast<uintptr_t>(a.Y);
  const bool al16 = (px % 16 == 0) && (py % 16 == 0) && (a.ldx % 4 == 0) && (a.ldy % 4 == 0) && (a.F % 4 == 0);
  const bool al8 = (px % 8 == 0) && (py % 8 == 0) && (a.ldx % 2 == 0) && (a.ldy % 2 == 0) && (a.F % 2 == 0);
  const int F = a.F;
  if (al16) {
    if (F <= 16) return launch_spmm<4, 16, 4, 1, 4>(a, st);
    if (F <= 32) return launch_spmm<4, 16, 8, 1, 4>(a, st);
    if (F <= 64) return launch_spmm<4, 16, 16, 1, 4>(a, st);
    if (F <= 128) return launch_spmm<4, 16, 32, 1, 4>(a, st);
    if (F <= 256) return launch_spmm<4, 16, 64, 1, 4>(a, st);
    if (F <= 512) return launch_spmm<4, 16, 64, 2, 4>(a, st);
    return launch_spmm<4, 16, 64, 3, 2>(a, st);
  }
  if (al8) {
    if (F <= 8) return launch_spmm<2, 8, 4, 1, 4>(a, st);
    if (F <= 32) return launch_spmm<2, 8, 16, 1, 4>(a, st);
    if (F <= 64) return launch_spmm<2, 8, 32, 1, 4>(a, st);
    if (F <= 128) return launch_spmm<2, 8, 64, 1, 4>(a, st);
    if (F <= 256) return launch_spmm<2, 8, 64, 2, 4>(a, st);
    if (F <= 384) return launch_spmm<2, 8, 64, 3, 4>(a, st);
    return launch_spmm<2, 8, 64, 5, 2>(a, st);
  }
  if (F <= 1) return launch_spmm<1, 4, 1, 1, 4>(a, st);
  if (F <= 8) return launch_spmm<1, 4, 8, 1, 4>(a, st);
  if (F <= 32) return launch_spmm<1, 4, 32, 1, 4>(a, st);
  if (F <= 64) return launch_spmm<1, 4, 64, 1, 4>(a, st);
  return launch_spmm<1, 4, 64, 4, 2>(a, st);
}

int spmm_impl(int64_t n_rows, int64_t n_cols, int64_t F, const int32_t* rowptr, const int32_t* colidx,
              const float* vals, const float* dinv_row, const float* dinv_col, const float* X, int64_t ldx, float* Y,
              int64_t ldy, int epilogue, int64_t nnz_bound, void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (n_rows < 0 || n_cols < 0 || F < 0) return fail(SGNN_ERR_INVALID, "negative size");
  if (n_rows == 0 || F == 0) return SGNN_OK;
  if (!rowptr || !colidx || !X || !Y) return fail(SGNN_ERR_INVALID, "null pointer");
  if (n_rows >= (int64_t(1) << 31) || n_cols >= (int64_t(1) << 31) || F >= (int64_t(1) << 24))
    return fail(SGNN_ERR_INVALID, "size exceeds int32 index range");
  if (ldx < F || ldy < F) return fail(SGNN_ERR_INVALID, "leading dimension smaller than F");
  if (epilogue != SGNN_EPI_SCALE_ROW && epilogue != SGNN_EPI_SCALE_ROW2)
    return fail(SGNN_ERR_INVALID, "unknown epilogue mode");
  SpmmArgs a{};
  a.rowptr = rowptr; a.colidx = colidx; a.vals = vals; a.dinv_row = dinv_row; a.dinv_col = dinv_col;
  a.X = X; a.Y = Y; a.ldx = ldx; a.ldy = ldy; a.n_rows = int32_t(n_rows); a.F = int32_t(F); a.epilogue = epilogue;
  SpmmWs w = carve_spmm(workspace, nnz_bound, F);
  if (workspace == nullptr || workspace_bytes < w.total)
    return fail(SGNN_ERR_WORKSPACE, "spmm workspace too small: need " + std::to_string(w.total) + " bytes");
  a.counters = w.counters; a.seg_row = w.seg_row; a.seg_k = w.seg_k; a.long_row = w.long_row;
  a.long_slot = w.long_slot; a.partial = w.partial; a.max_segs = w.max_segs; a.max_long = w.max_long;
  HIP_TRY(hipMemsetAsync(a.counters, 0, 256, st));
  hipLaunchKernelGGL(k_collect_long, dim3(unsigned((n_rows + 255) / 256)), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return dispatch_spmm(a, st);
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int sgnn_version(void) { return 100; }

const char* sgnn_last_error(void) { return g_err.c_str(); }

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
  if (n >= (int64_t(1) << 31) || nnz_in + n >= (int64_t(1) << 31))
    return fail(SGNN_ERR_INVALID, "n and nnz_in + n must stay below 2^31");
  if ((nnz_in > 0 && (!row || !col)) || !rowptr || !colidx || !vals || !deg || !dinv || !nnz_out)
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
    hipLaunchKernelGGL(k_degree, dim3(unsigned((n + T - 1) / T)), dim3(T), 0, st, n, rowptr, vals, zero_inf, deg, dinv);
    HIP_TRY(hipGetLastError());
    return SGNN_OK;
  }
  const unsigned gm = unsigned((m + T - 1) / T);
  hipLaunchKernelGGL(k_make_keys, dim3(gm), dim3(T), 0, st, n, nnz_in, m, row, col, (flags & SGNN_STRIP_DIAG) ? 1 : 0,
                     cbits, sentinel, w.keys_a, w.idx_a, w.flag);
  HIP_TRY(hipGetLastError());
  size_t pb = w.prim_bytes;
  HIP_TRY(rocprim::radix_sort_pairs(w.prim, pb, w.keys_a, w.keys_b, w.idx_a, w.idx_b, size_t(m), 0u,
                                    unsigned(2 * cbits + 1), st));
  hipLaunchKernelGGL(k_merge_runs, dim3(gm), dim3(T), 0, st, m, nnz_in, w.keys_b, w.idx_b, val, sentinel, cbits,
                     self_loops, (flags & SGNN_DROP_ZEROS) ? 1 : 0, w.rsum, w.keep);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.keep, w.pos, 0u, size_t(m), rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_scatter, dim3(gm), dim3(T), 0, st, m, w.keys_b, w.rsum, w.keep, w.pos, cbits, colidx, vals,
                     w.orow);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_rowptr, dim3(unsigned((m + 1 + T - 1) / T)), dim3(T), 0, st, n, m, w.keep, w.pos, w.orow,
                     w.flag, rowptr, nnz_out);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_degree, dim3(unsigned((n + T - 1) / T)), dim3(T), 0, st, n, rowptr, vals, zero_inf, deg, dinv);
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

size_t sgnn_spmm_workspace_bytes(int64_t n_rows, int64_t nnz, int64_t F) {
  (void)n_rows;
  if (nnz < 0 || F < 0) return 0;
  return carve_spmm(nullptr, nnz, F).total;
}

int sgnn_spmm_csr_f32(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t F, const int32_t* rowptr,
                      const int32_t* colidx, const float* vals, const float* dinv_row, const float* dinv_col,
                      const float* X, int64_t ldx, float* Y, int64_t ldy, int epilogue, void* workspace,
                      size_t workspace_bytes, void* stream) {
  if (nnz < 0) return fail(SGNN_ERR_INVALID, "negative nnz");
  return spmm_impl(n_rows, n_cols, F, rowptr, colidx, vals, dinv_row, dinv_col, X, ldx, Y, ldy, epilogue, nnz,
                   workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int sgnn_spmm_chain_f32(int K, int64_t n, int64_t nnz, int64_t F, const int32_t* rowptr, const int32_t* colidx,
                        const float* vals, const float* dinv, const float* X, int64_t ldx, float* Y, int64_t ldy, float* tmp,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (K < 1) return fail(SGNN_ERR_INVALID, "K must be at least 1");
  if (K > 1 && !tmp) return fail(SGNN_ERR_INVALID, "tmp is required when K > 1");
  if (!dinv) return fail(SGNN_ERR_INVALID, "null dinv");
  // ping-pong so that the last hop lands in Y: hop h writes Y when (K-1-h) is even
  const float* src = X;
  int64_t lds = ldx;
  for (int h = 0; h < K; ++h) {
    const bool last = h == K - 1;
    float* dst = ((K - 1 - h) % 2 == 0) ? Y : tmp;
    const int64_t ldd = ((K - 1 - h) % 2 == 0) ? ldy : F;
    int rc = sgnn_spmm_csr_f32(n, n, nnz, F, rowptr, colidx, vals, dinv, h == 0 ? dinv : nullptr, src, lds, dst, ldd,
                               last ? SGNN_EPI_SCALE_ROW : SGNN_EPI_SCALE_ROW2, workspace, workspace_bytes, stream);
    if (rc != SGNN_OK) return rc;
    src = dst;
    lds = ldd;
  }
  return SGNN_OK;
}

int sgnn_row_scale_f32(int64_t n, int64_t F, const float* scale, const float* in, int64_t ldin, float* out,
                       int64_t ldout, void* stream) {
  if (n < 0 || F < 0) return fail(SGNN_ERR_INVALID, "negative size");
  if (n == 0 || F == 0) return SGNN_OK;
  if (!scale || !in || !out) return fail(SGNN_ERR_INVALID, "null pointer");
  const int64_t total = n * F;
  hipLaunchKernelGGL(k_row_scale, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     n, F, scale, in, ldin, out, ldout);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

}  // extern "C"
