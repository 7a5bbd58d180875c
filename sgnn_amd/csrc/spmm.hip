// CSR SpMM: Y = diag(dinv_row) . CSR . diag(dinv_col) . X   (fp32), one trip of
// the loop at utils.py:92-95, and the K-hop chain (utils.py:91-96).
//
// The propagation is a sparse gather-reduce: per stored entry one row of X is
// fetched and added into one row of Y.  There is no dense contraction, so no
// MFMA; the kernel is organised around the memory system instead: 64-lane
// wavefronts laid along the feature dimension, 8/16-byte buffer loads per lane
// with the row address in scalar registers, several independent row fetches in
// flight per wavefront, and long rows cut into fixed-size segments so that no
// wavefront owns a 10^5-entry hub.
#include "common.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
using namespace sgnn;

constexpr int kSeg = 1024;          // most entries one wavefront reduces; longer rows are cut into segments
constexpr int kWavesPerBlock = 4;

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
  int32_t row_bytes;   // ldx * 4 (valid when x_bytes < 4 GiB)
  int64_t x_bytes;     // bytes from X to the end of its last row
  int y_vec;           // widest vector (floats) the rows of Y are aligned for
  // long-row machinery (device scratch)
  int32_t* counters;   // [0] segments appended, [1] long rows appended
  int32_t* seg_row;    // [max_segs] row of segment
  int32_t* seg_k;      // [max_segs] index of the segment inside its row
  int32_t* long_row;   // [max_long] row
  int32_t* long_slot;  // [max_long] first partial slot of the row
  float* partial;      // [max_segs, F]
  int32_t max_segs, max_long;
  const uint32_t* guard;   // nullable device word: the launch does nothing while it is 0
};

// ---- full-wavefront layout (LPR == 64) --------------------------------------
// All 64 lanes lie along the feature dimension: lane l owns features
// f0 + (c*64 + l)*V .. +V, c < C.  The column ids (and weights) of a row are then
// wave-uniform, so they are fetched with scalar loads; the fetched row of X is
// addressed through a buffer descriptor over X with the row's byte offset in a
// scalar register (soffset) and the lane's byte offset inside the row in one
// loop-invariant VGPR: a row fetch costs no vector ALU work.  Buffer loads need
// only dword alignment.  The hardware range check compares the lane's voffset alone
// with the descriptor's size (soffset is not part of it), so a lane whose features
// lie past the end of the row is given a voffset beyond any descriptor
// (kOobVoffset): it reads 0 and touches no memory, also on the last row of X.
// The loop is software-pipelined: the ids of the next U entries are requested
// while the U*C vector loads of the current ones are in flight.  Entries are
// added in column order.
// BIGX: X spans 4 GiB or more, so the descriptor is re-based on the fetched row
// (64-bit scalar arithmetic) instead of carrying the row offset in soffset.
template <int V, int C, int UU, bool WEIGHTED, bool BIGX>
__device__ __forceinline__ void uniform_step(const SpmmArgs& a, i32x4 rsrc, int32_t cols, float ws, int k,
                                             const int (&boff)[C], float (&acc)[C][V]) {
  float xv[UU][C][V];
  float w[UU];
#pragma unroll
  for (int u = 0; u < UU; ++u) {
    const int32_t col = __builtin_amdgcn_readlane(cols, k + u);
    if constexpr (WEIGHTED)
      w[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ws), k + u));
    if constexpr (BIGX) {
      const i32x4 rr = make_rsrc(a.X + int64_t(col) * a.ldx, uint32_t(a.F) * 4u);
#pragma unroll
      for (int c = 0; c < C; ++c) buffer_load_vec<V>(rr, boff[c], 0, xv[u][c]);
    } else {
      const int soff = int(uint32_t(col) * uint32_t(a.row_bytes));   // up to 4 GiB: unsigned
#pragma unroll
      for (int c = 0; c < C; ++c) buffer_load_vec<V>(rsrc, boff[c], soff, xv[u][c]);
    }
  }
#pragma unroll
  for (int u = 0; u < UU; ++u)
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if constexpr (WEIGHTED) acc[c][v] = fmaf(w[u], xv[u][c][v], acc[c][v]);
        else acc[c][v] += xv[u][c][v];
      }
}

// Entries are taken 64 at a time: one coalesced vector load brings the next 64
// column ids (and values) into a VGPR while the current 64 are being worked
// through, each entry's id moving to a scalar register with v_readlane.  The
// dinv[col] gather of the next batch depends on its ids, so it is issued after
// the first step of the current batch, by which time the ids have landed.
template <int V, int C, int U, bool HASV, bool HASD, bool BIGX>
__device__ __forceinline__ void uniform_loop(const SpmmArgs& a, int32_t e0, int32_t e1, const int (&boff)[C],
                                             float (&acc)[C][V]) {
  constexpr bool WEIGHTED = HASV || HASD;
  const i32x4 rsrc = make_rsrc(a.X, BIGX ? 0u : uint32_t(a.x_bytes));
  const int lane = threadIdx.x & (kWave - 1);
  int32_t ncols = 0;
  float nws = 1.0f;
  if (e0 + lane < e1) {
    ncols = a.colidx[e0 + lane];
    if constexpr (HASV) nws = a.vals[e0 + lane];
    if constexpr (HASD) nws *= a.dinv_col[ncols];
  }
  for (int32_t e = e0; e < e1; e += kWave) {
    const int32_t cols = ncols;
    const float ws = nws;
    const int cnt = min(kWave, e1 - e);
    const int32_t en = e + kWave;
    const bool more = en < e1;            // wave-uniform
    const bool mine = en + lane < e1;
    if (more) {
      ncols = mine ? a.colidx[en + lane] : 0;
      if constexpr (HASV) nws = mine ? a.vals[en + lane] : 1.0f;
    }
    int k = 0;
    if (cnt >= U) {
      uniform_step<V, C, U, WEIGHTED, BIGX>(a, rsrc, cols, ws, 0, boff, acc);
      k = U;
    }
    if constexpr (HASD) {
      if (more) nws = mine ? (HASV ? nws * a.dinv_col[ncols] : a.dinv_col[ncols]) : 1.0f;
    }
    for (; k + U <= cnt; k += U) uniform_step<V, C, U, WEIGHTED, BIGX>(a, rsrc, cols, ws, k, boff, acc);
    for (; k < cnt; ++k) uniform_step<V, C, 1, WEIGHTED, BIGX>(a, rsrc, cols, ws, k, boff, acc);
  }
}

template <int V, int C, int U, bool BIGX>
__device__ __forceinline__ void uniform_reduce(const SpmmArgs& a, int32_t e0, int32_t e1, const int (&boff)[C],
                                               float (&acc)[C][V]) {
  const bool hv = a.vals != nullptr, hd = a.dinv_col != nullptr;
  if (!hv && !hd) uniform_loop<V, C, U, false, false, BIGX>(a, e0, e1, boff, acc);
  else if (!hv) uniform_loop<V, C, U, false, true, BIGX>(a, e0, e1, boff, acc);
  else if (!hd) uniform_loop<V, C, U, true, false, BIGX>(a, e0, e1, boff, acc);
  else uniform_loop<V, C, U, true, true, BIGX>(a, e0, e1, boff, acc);
}

// ---- split layout (narrow F: LPR < 64, V == 1) --------------------------------
// The wavefront is split into G = 64/LPR groups of LPR lanes; the wavefront
// loads 64 column ids (and weights) at a time, group g takes entries g, g+G, ...
// of them through lane shuffles, and the groups are combined by a shuffle tree
// at the end.
template <int LPR, int UU, bool PRED>
__device__ __forceinline__ void group_step(const SpmmArgs& a, int32_t cj, float wj, int k0, int cnt, int foff,
                                           float& acc) {
  constexpr int G = kWave / LPR;
  const int g = (threadIdx.x & (kWave - 1)) / LPR;
  float xv[UU];
  float w[UU];
  bool ok[UU];
#pragma unroll
  for (int u = 0; u < UU; ++u) {
    const int src = (k0 + u) * G + g;
    const int32_t cc = __shfl(cj, src & (kWave - 1));
    w[u] = __shfl(wj, src & (kWave - 1));
    ok[u] = !PRED || src < cnt;
    xv[u] = ok[u] ? a.X[int64_t(cc) * a.ldx + foff] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < UU; ++u)
    if (ok[u]) acc = fmaf(w[u], xv[u], acc);
}

template <int LPR, int U>
__device__ __forceinline__ void group_reduce(const SpmmArgs& a, int32_t e0, int32_t e1, int f0, float& acc) {
  constexpr int G = kWave / LPR;
  const int lane = threadIdx.x & (kWave - 1);
  const int foff = min(f0 + lane % LPR, a.F - 1);   // lanes past the row end re-read its last element
  for (int32_t base = e0; base < e1; base += kWave) {
    const int32_t e = base + lane;
    int32_t cj = 0;
    float wj = 0.0f;
    if (e < e1) {
      cj = a.colidx[e];
      wj = a.vals ? a.vals[e] : 1.0f;
      if (a.dinv_col) wj *= a.dinv_col[cj];
    }
    const int cnt = min(kWave, e1 - base);
    const int full = cnt / G;      // steps in which every group has an entry
    int k0 = 0;
    for (; k0 + U <= full; k0 += U) group_step<LPR, U, false>(a, cj, wj, k0, cnt, foff, acc);
    for (; k0 < full; ++k0) group_step<LPR, 1, false>(a, cj, wj, k0, cnt, foff, acc);
    if (full * G < cnt) group_step<LPR, 1, true>(a, cj, wj, full, cnt, foff, acc);
  }
#pragma unroll
  for (int off = LPR; off < kWave; off <<= 1) acc += __shfl_xor(acc, off);
}

// Reduce entries [e0, e1) of one row into acc; the summation order is a function
// of the row's entries alone in both layouts.
template <int V, int LPR, int C, int U>
__device__ __forceinline__ void reduce_entries(const SpmmArgs& a, int32_t e0, int32_t e1, int f0, float (&acc)[C][V]) {
  if constexpr (LPR == kWave) {
    e0 = __builtin_amdgcn_readfirstlane(e0);
    e1 = __builtin_amdgcn_readfirstlane(e1);
    const int lane = threadIdx.x & (kWave - 1);
    int boff[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int f = f0 + (c * kWave + lane) * V;   // V divides F: a lane is inside the row or past it
      boff[c] = f < a.F ? f * 4 : kOobVoffset;
    }
    if (a.x_bytes < kBigX) uniform_reduce<V, C, U, false>(a, e0, e1, boff, acc);
    else uniform_reduce<V, C, U, true>(a, e0, e1, boff, acc);
  } else {
    static_assert(LPR == kWave || (V == 1 && C == 1), "the split layout is scalar");
    group_reduce<LPR, U>(a, e0, e1, f0, acc[0][0]);
  }
}

template <int V, int LPR, int C>
__device__ __forceinline__ void write_row(const SpmmArgs& a, float* dst, int32_t row, int f0, float (&acc)[C][V],
                                          bool final_row, int aligned_for) {
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
      store_vec<V>(dst + f, o, aligned_for);
    }
  }
}

// list the rows longer than kSeg and give each of their segments a partial slot
__global__ void k_collect_long(SpmmArgs a) {
  if (a.guard != nullptr && *a.guard == 0) return;
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
template <int V, int LPR, int C, int U>
__device__ __forceinline__ void spmm_block(const SpmmArgs& a, unsigned row_blocks, unsigned bx) {
  const int wave = threadIdx.x / kWave;
  const int f0 = blockIdx.y * (LPR * V * C);
  float acc[C][V];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int v = 0; v < V; ++v) acc[c][v] = 0.0f;
  if (bx < row_blocks) {
    const int64_t row = int64_t(bx) * kWavesPerBlock + wave;
    if (row >= a.n_rows) return;
    const int32_t e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
    if (e1 - e0 > kSeg) return;  // cut into segments below
    reduce_entries<V, LPR, C, U>(a, e0, e1, f0, acc);
    write_row<V, LPR, C>(a, a.Y + row * a.ldy, int32_t(row), f0, acc, true, a.y_vec);
    return;
  }
  const int64_t s = int64_t(bx - row_blocks) * kWavesPerBlock + wave;
  const int32_t nseg = min(a.counters[0], a.max_segs);
  if (s >= nseg) return;
  const int32_t row = a.seg_row[s];
  const int32_t k = a.seg_k[s];
  const int32_t r0 = a.rowptr[row], r1 = a.rowptr[row + 1];
  const int32_t e0 = r0 + k * kSeg;
  const int32_t e1 = min(e0 + kSeg, r1);
  reduce_entries<V, LPR, C, U>(a, e0, e1, f0, acc);
  // partial rows are dense [F]: vector stores only when F keeps every row start aligned
  write_row<V, LPR, C>(a, a.partial + s * int64_t(a.F), row, f0, acc, false, (a.F % 4 == 0) ? 4 : (a.F % 2 == 0 ? 2 : 1));
}

// Block bx of the grid described above; an unguarded launch has one workgroup per block.  A
// guarded launch (the fall-back behind the tile-group kernel, which nearly always finds its
// guard at 0) strides a bounded grid over the blocks instead: workgroups that only
// read the guard and leave cost 0.15 ms per hop on the ogbn-products shape (612 000 of them) and
// 0.57 ms on R-MAT with 10^7 rows.
template <int V, int LPR, int C, int U>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_spmm(SpmmArgs a, unsigned row_blocks, unsigned total_blocks) {
  if (a.guard != nullptr && *a.guard == 0) return;
  for (unsigned bx = blockIdx.x; bx < total_blocks; bx += gridDim.x) spmm_block<V, LPR, C, U>(a, row_blocks, bx);
}

// sum a long row's partials in segment order, then the row epilogue
__global__ void k_spmm_combine(SpmmArgs a) {
  if (a.guard != nullptr && *a.guard == 0) return;
  // The count of long rows is known on the device only; the grid is sized for a typical
  // count and strides over the rest (a grid of max_long workgroups, nearly all of which
  // find nothing to do, costs 70 us on the Reddit shape).
  const int32_t nlong = min(a.counters[1], a.max_long);
  for (int32_t li = blockIdx.x; li < nlong; li += gridDim.x) {
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
}

// out[r, :] = scale[r] * in[r, :]; one wavefront per row, lanes along the row
__global__ __launch_bounds__(256) void k_row_scale(int64_t n, int64_t F, const float* __restrict__ scale,
                                                   const float* __restrict__ in, int64_t ldin,
                                                   float* __restrict__ out, int64_t ldout) {
  const int64_t r = blockIdx.x * int64_t(4) + threadIdx.x / kWave;
  if (r >= n) return;
  const float s = scale[r];
  const float* src = in + r * ldin;
  float* dst = out + r * ldout;
  for (int64_t f = threadIdx.x & (kWave - 1); f < F; f += kWave) dst[f] = src[f] * s;
}

// out[r, :] = ((a[r, :] + b[r, :]) * scale[r]) (* scale[r]): the raw sums of a row block's two
// column blocks (own columns, remote columns) added in that fixed order, then the row scaling of
// utils.py:95 (and, between hops, utils.py:93 of the next hop)
__global__ __launch_bounds__(256) void k_add_scale(int64_t n, int64_t F, const float* __restrict__ a, int64_t lda,
                                                   const float* __restrict__ b, int64_t ldb,
                                                   const float* __restrict__ scale, int twice,
                                                   float* __restrict__ out, int64_t ldout) {
  const int64_t r = blockIdx.x * int64_t(4) + threadIdx.x / kWave;
  if (r >= n) return;
  const float s = scale[r];
  const float* pa = a + r * lda;
  const float* pb = b + r * ldb;
  float* dst = out + r * ldout;
  for (int64_t f = threadIdx.x & (kWave - 1); f < F; f += kWave) {
    float y = (pa[f] + pb[f]) * s;
    if (twice) y = y * s;
    dst[f] = y;
  }
}

// dst <- src, 16 bytes per lane, non-temporal loads and stores, four workgroups per CU walking the
// buffers with a grid stride: the fastest of the forms swept in tools/proto/copy_probe.hip (more
// loads in flight per lane or a larger grid stream 10-25 % slower on this card).  A yardstick for
// the roofline: what the card streams, read plus write.
__global__ __launch_bounds__(256) void k_stream_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n4) {
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n4; i += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
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

template <int V, int LPR, int C, int U>
int launch_spmm(const SpmmArgs& a, hipStream_t st) {
  const int tile = LPR * V * C;
  const unsigned ftiles = unsigned((a.F + tile - 1) / tile);
  const unsigned row_blocks = unsigned((int64_t(a.n_rows) + kWavesPerBlock - 1) / kWavesPerBlock);
  const unsigned seg_blocks = unsigned((int64_t(a.max_segs) + kWavesPerBlock - 1) / kWavesPerBlock);
  if (!a.guard) sgnn::kernel_timer_begin(st);
  const unsigned total_blocks = row_blocks + seg_blocks;
  const unsigned grid_x = (a.guard && total_blocks > 4096u) ? 4096u : total_blocks;
  hipLaunchKernelGGL((k_spmm<V, LPR, C, U>), dim3(grid_x, ftiles), dim3(kWave * kWavesPerBlock), 0,
                     st, a, row_blocks, total_blocks);
  HIP_TRY(hipGetLastError());
  if (!a.guard) sgnn::kernel_timer_end(st);
  const unsigned fy = unsigned((a.F + 255) / 256 > 8 ? 8 : (a.F + 255) / 256);
  hipLaunchKernelGGL(k_spmm_combine, dim3(unsigned(a.max_long < 4096 ? a.max_long : 4096), fy), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

// Tuning knobs (sgnn_tune_set).  tile = width in floats of the feature slice one
// wavefront owns (0 = policy below); unroll = row fetches issued back to back.
struct Tune {
  int tile = 0;
  int unroll = 0;
};
Tune g_tune;

template <int V, int LPR, int C>
int launch_u(const SpmmArgs& a, hipStream_t st, int U) {
  if (U >= 4) return launch_spmm<V, LPR, C, 4>(a, st);
  return launch_spmm<V, LPR, C, 2>(a, st);
}

template <int V>
int dispatch_wide(const SpmmArgs& a, hipStream_t st, int W) {
  const int C = (W + 64 * V - 1) / (64 * V);
  int U = g_tune.unroll;
  if (U == 0) U = C <= 2 ? 4 : 2;
  if (C <= 1) return launch_u<V, 64, 1>(a, st, U);
  if (C <= 2) return launch_u<V, 64, 2>(a, st, U);
  if (C <= 3) return launch_u<V, 64, 3>(a, st, U);
  if (C <= 4) return launch_u<V, 64, 4>(a, st, U);
  return launch_u<V, 64, 5>(a, st, U);   // wider rows are covered by more slices (grid.y)
}

// Lane layout from F.  Rows wider than 32 floats use the full-wavefront layout
// with the widest vector that divides F and still keeps all 64 lanes busy; a
// wavefront owns up to 5 vectors per lane (the whole row for F <= 1280 at V=4),
// wider rows are cut into slices swept one after the other.  Narrow rows use the
// split layout.
int dispatch_spmm(const SpmmArgs& a, hipStream_t st) {
  const int F = a.F;
  int W = g_tune.tile > 0 ? (g_tune.tile < F ? g_tune.tile : F) : F;
  if (W > 32) {
    if (F % 4 == 0 && W > 128) return dispatch_wide<4>(a, st, W);
    if (F % 2 == 0 && W > 64) return dispatch_wide<2>(a, st, W);
    return dispatch_wide<1>(a, st, W);
  }
  int U = g_tune.unroll ? g_tune.unroll : 4;
  if (W <= 1) return launch_u<1, 1, 1>(a, st, U);
  if (W <= 2) return launch_u<1, 2, 1>(a, st, U);
  if (W <= 4) return launch_u<1, 4, 1>(a, st, U);
  if (W <= 8) return launch_u<1, 8, 1>(a, st, U);
  if (W <= 16) return launch_u<1, 16, 1>(a, st, U);
  return launch_u<1, 32, 1>(a, st, U);
}

int spmm_impl(int64_t n_rows, int64_t n_cols, int64_t F, const int32_t* rowptr, const int32_t* colidx,
              const float* vals, const float* dinv_row, const float* dinv_col, const float* X, int64_t ldx, float* Y,
              int64_t ldy, int epilogue, int64_t nnz_bound, void* workspace, size_t workspace_bytes, hipStream_t st,
              const uint32_t* guard = nullptr) {
  if (n_rows < 0 || n_cols < 0 || F < 0) return fail(SGNN_ERR_INVALID, "negative size");
  if (n_rows == 0 || F == 0) return SGNN_OK;
  if (!rowptr || !colidx || !X || !Y) return fail(SGNN_ERR_INVALID, "null pointer");
  if (n_cols == 0) return fail(SGNN_ERR_INVALID, "X has no rows");
  if (n_rows >= (int64_t(1) << 31) || n_cols >= (int64_t(1) << 31) || F >= (int64_t(1) << 24))
    return fail(SGNN_ERR_INVALID, "size exceeds int32 index range");
  if (ldx < F || ldy < F) return fail(SGNN_ERR_INVALID, "leading dimension smaller than F");
  if (reinterpret_cast<uintptr_t>(X) % 4 || reinterpret_cast<uintptr_t>(Y) % 4)
    return fail(SGNN_ERR_INVALID, "X and Y must be 4-byte aligned");
  if (epilogue != SGNN_EPI_SCALE_ROW && epilogue != SGNN_EPI_SCALE_ROW2)
    return fail(SGNN_ERR_INVALID, "unknown epilogue mode");
  SpmmArgs a{};
  a.rowptr = rowptr; a.colidx = colidx; a.vals = vals; a.dinv_row = dinv_row; a.dinv_col = dinv_col;
  a.X = X; a.Y = Y; a.ldx = ldx; a.ldy = ldy; a.n_rows = int32_t(n_rows); a.F = int32_t(F); a.epilogue = epilogue;
  a.x_bytes = ((n_cols - 1) * ldx + F) * 4;
  a.row_bytes = a.x_bytes < kBigX ? int32_t(ldx * 4) : 0;
  const uintptr_t py = reinterpret_cast<uintptr_t>(Y);
  a.y_vec = (py % 16 == 0 && ldy % 4 == 0) ? 4 : ((py % 8 == 0 && ldy % 2 == 0) ? 2 : 1);
  SpmmWs w = carve_spmm(workspace, nnz_bound, F);
  if (workspace == nullptr || workspace_bytes < w.total)
    return fail(SGNN_ERR_WORKSPACE, "spmm workspace too small: need " + std::to_string(w.total) + " bytes");
  a.counters = w.counters; a.seg_row = w.seg_row; a.seg_k = w.seg_k; a.long_row = w.long_row;
  a.long_slot = w.long_slot; a.partial = w.partial; a.max_segs = w.max_segs; a.max_long = w.max_long;
  a.guard = guard;
  HIP_TRY(hipMemsetAsync(a.counters, 0, 256, st));
  hipLaunchKernelGGL(k_collect_long, dim3(unsigned((n_rows + 255) / 256)), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  return dispatch_spmm(a, st);
}

// ---- scratch for the entry point that allocates nothing --------------------------
// sgnn_spmm_csr_noalloc_f32 keeps the long-row scratch of the SpMM (segment lists and
// partial rows) in this array instead of a caller-provided workspace.  Every GPU the
// library runs on holds its own copy, at its own address, until the process ends.
// The array's name is a host-side stand-in for those copies, not a device address:
// host code uses the pointer hipGetSymbolAddress returns for the current device.
constexpr size_t kScratchBytes = size_t(32) << 20;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
std::atomic<void*> g_scratch_ptr[kMaxDevices];

}  // namespace

namespace sgnn {
int spmm_csr_guarded(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t F, const int32_t* rowptr,
                     const int32_t* colidx, const float* vals, const float* dinv_row, const float* dinv_col,
                     const float* X, int64_t ldx, float* Y, int64_t ldy, int epilogue, void* workspace,
                     size_t workspace_bytes, const uint32_t* guard, hipStream_t st) {
  return spmm_impl(n_rows, n_cols, F, rowptr, colidx, vals, dinv_row, dinv_col, X, ldx, Y, ldy, epilogue, nnz, workspace,
                   workspace_bytes, st, guard);
}
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

namespace {
// Kernel timer.  One pool of event pairs for the process, guarded by a mutex (entry points may
// be called from several host threads); a pair belongs to the device it was created on and is
// created anew when its slot is reused from another device.  A pair left open by a launch path
// that returned early is abandoned by the next begin.
struct TimerPair { hipEvent_t e0, e1; int dev; };
constexpr size_t kMaxTimed = 8192;      // launches kept between two reads; later ones go untimed
std::mutex g_timer_mu;
bool g_timer_on = false;
std::vector<TimerPair> g_timer_pool;    // events are created on first use and kept
size_t g_timer_used = 0;
bool g_timer_open = false;
}  // namespace

void kernel_timer_begin(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_timer_mu);
  g_timer_open = false;
  if (!g_timer_on || g_timer_used >= kMaxTimed) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  if (g_timer_used < g_timer_pool.size() && g_timer_pool[g_timer_used].dev != dev) {
    for (size_t i = g_timer_used; i < g_timer_pool.size(); ++i) {   // this pair and the ones behind it are recreated as they come up
      (void)hipEventDestroy(g_timer_pool[i].e0);
      (void)hipEventDestroy(g_timer_pool[i].e1);
    }
    g_timer_pool.resize(g_timer_used);
  }
  if (g_timer_used == g_timer_pool.size()) {
    TimerPair p;
    p.dev = dev;
    if (hipEventCreate(&p.e0) != hipSuccess) return;
    if (hipEventCreate(&p.e1) != hipSuccess) { (void)hipEventDestroy(p.e0); return; }
    g_timer_pool.push_back(p);
  }
  if (hipEventRecord(g_timer_pool[g_timer_used].e0, st) == hipSuccess) g_timer_open = true;
}

void kernel_timer_end(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_timer_mu);
  if (!g_timer_open) return;
  g_timer_open = false;
  if (hipEventRecord(g_timer_pool[g_timer_used].e1, st) == hipSuccess) ++g_timer_used;
}
}  // namespace sgnn

extern "C" {

int sgnn_version(void) { return 100; }

int sgnn_tune_set(const char* key, int value) {
  if (!key) return fail(SGNN_ERR_INVALID, "null key");
  if (!std::strcmp(key, "tile")) g_tune.tile = value;
  else if (!std::strcmp(key, "unroll")) g_tune.unroll = value;
  else if (!std::strcmp(key, "rg_seg")) sgnn::set_rowgroup_segment(value);
  else if (!std::strcmp(key, "rg_R")) sgnn::set_rowgroup_R(value);
  else if (!std::strcmp(key, "rg_order")) sgnn::set_rowgroup_order(value);
  else if (!std::strcmp(key, "tg_seg")) sgnn::set_tilegroup_segment(value);
  else if (!std::strcmp(key, "tg_mode")) sgnn::set_tilegroup_mode(value);
  else if (!std::strcmp(key, "tg_order")) sgnn::set_tilegroup_order(value);
  else if (!std::strcmp(key, "tg_combine_gy")) sgnn::set_tilegroup_combine_rows(value);
  else if (!std::strcmp(key, "tg_no_sample")) sgnn::set_tilegroup_no_sample(value);
  else if (!std::strcmp(key, "tg_split_blocks")) sgnn::set_tilegroup_split_shape(value, -1);
  else if (!std::strcmp(key, "tg_split_threads")) sgnn::set_tilegroup_split_shape(-1, value);
  else if (!std::strcmp(key, "time_kernel")) {
    std::lock_guard<std::mutex> lock(sgnn::g_timer_mu);
    sgnn::g_timer_on = value != 0;
    sgnn::g_timer_used = 0;
    sgnn::g_timer_open = false;
  }
  else return fail(SGNN_ERR_INVALID, std::string("unknown tuning key ") + key);
  return SGNN_OK;
}

const char* sgnn_last_error(void) { return sgnn::g_err.c_str(); }

int sgnn_kernel_time_ms(double* total_ms, int64_t* launches) {
  if (!total_ms || !launches) return fail(SGNN_ERR_INVALID, "null pointer");
  double sum = 0;
  std::lock_guard<std::mutex> lock(sgnn::g_timer_mu);
  for (size_t i = 0; i < sgnn::g_timer_used; ++i) {
    float ms = 0;
    HIP_TRY(hipEventSynchronize(sgnn::g_timer_pool[i].e1));
    HIP_TRY(hipEventElapsedTime(&ms, sgnn::g_timer_pool[i].e0, sgnn::g_timer_pool[i].e1));
    sum += ms;
  }
  *total_ms = sum;
  *launches = int64_t(sgnn::g_timer_used);
  sgnn::g_timer_used = 0;
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

int sgnn_spmm_csr_noalloc_f32(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t F, const int32_t* rowptr,
                              const int32_t* colidx, const float* vals, const float* dinv_row,
                              const float* dinv_col, const float* X, int64_t ldx, float* Y, int64_t ldy,
                              int epilogue, void* stream) {
  if (nnz < 0 || F < 0) return fail(SGNN_ERR_INVALID, "negative size");
  const size_t need = carve_spmm(nullptr, nnz, F).total;
  if (need > kScratchBytes)
    return fail(SGNN_ERR_WORKSPACE, "problem too large for the built-in scratch: needs " + std::to_string(need) +
                                        " bytes, the library holds " + std::to_string(kScratchBytes) +
                                        " per GPU (use sgnn_spmm_csr_f32 with a workspace)");
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(SGNN_ERR_INVALID, "device ordinal out of range");
  void* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (ptr == nullptr) {   // first call on this device: look its copy of the array up, once
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (ptr == nullptr) return fail(SGNN_ERR_HIP, "hipGetSymbolAddress returned a null address");
    g_scratch_ptr[dev].store(ptr, std::memory_order_release);
  }
  // spmm_impl clears the counters at the head of the scratch on the stream before it
  // launches; everything else in it is written before it is read
  return spmm_impl(n_rows, n_cols, F, rowptr, colidx, vals, dinv_row, dinv_col, X, ldx, Y, ldy, epilogue, nnz, ptr,
                   kScratchBytes, static_cast<hipStream_t>(stream));
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
  hipLaunchKernelGGL(k_row_scale, dim3(unsigned((n + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     n, F, scale, in, ldin, out, ldout);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

int sgnn_stream_copy_f32(const float* src, float* dst, int64_t count, void* stream) {
  if (count < 0) return fail(SGNN_ERR_INVALID, "negative size");
  if (count == 0) return SGNN_OK;
  if (!src || !dst) return fail(SGNN_ERR_INVALID, "null pointer");
  if (count % 4 || reinterpret_cast<uintptr_t>(src) % 16 || reinterpret_cast<uintptr_t>(dst) % 16)
    return fail(SGNN_ERR_INVALID, "count must be a multiple of 4 and both buffers 16-byte aligned");
  hipLaunchKernelGGL(k_stream_copy, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst), count / 4);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

int sgnn_add_scale_f32(int64_t n, int64_t F, const float* a, int64_t lda, const float* b, int64_t ldb,
                       const float* scale, int twice, float* out, int64_t ldout, void* stream) {
  if (n < 0 || F < 0) return fail(SGNN_ERR_INVALID, "negative size");
  if (n == 0 || F == 0) return SGNN_OK;
  if (!a || !b || !scale || !out) return fail(SGNN_ERR_INVALID, "null pointer");
  if (lda < F || ldb < F || ldout < F) return fail(SGNN_ERR_INVALID, "leading dimension smaller than F");
  hipLaunchKernelGGL(k_add_scale, dim3(unsigned((n + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     n, F, a, lda, b, ldb, scale, twice, out, ldout);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

}  // extern "C"
