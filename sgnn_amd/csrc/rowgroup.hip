// Row-group SpMM: R output rows share every fetched row of X.
//
// The plain CSR kernel fetches one row of X per stored entry.  On graphs of
// Reddit's shape that is ~277 GB of cache-level traffic per hop at F = 602,
// served mostly from the Infinity Cache at its random-row rate, and it bounds the
// kernel.  Rows of similar degree share many of their columns, so the rows are
// ranked by degree and cut into groups of R; each group keeps ONE merged, sorted
// column list with an R-bit mask per column saying which of its rows hold that
// entry.  A wavefront walks the merged list, fetches each listed row of X once
// and adds it into the accumulators of the rows whose bit is set (R x F/64
// floats per lane, all in registers).  On the Reddit-shape benchmark graph the
// merged lists hold 49 % (R = 16) of the entries, and the kernel time follows.
//
// Reference path served: the adj.mm of utils.py:94 for adjacencies whose
// off-diagonal entries all carry one weight (the 0/1 matrices of the reference's
// loaders, data_loader.py:545-549, and Reddit's weight-2 A + A^T, data_loader.py:75):
// the masks carry no values, the epilogue applies the weight and the diagonal
// correction.  Other weighted graphs take the plain CSR kernel.
#include <cstring>  // rocprim's texture iterator calls memset from host code
#include <type_traits>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace {
using namespace sgnn;

constexpr int kWavesPerBlock = 4;
constexpr int kXcds = 8;   // workgroup b of a 1-D grid runs on XCD b % 8

// Merged columns one wavefront walks; longer lists are cut into segments of this
// length.  A wavefront works through its list serially, so small graphs get short
// segments (enough wavefronts to hide latency) and large ones long segments
// (fewer partial rows to combine).
int g_seg_override = 0;   // sgnn_tune_set("rg_seg", v)
int g_R_override = 0;     // sgnn_tune_set("rg_R", v)

int seg_cols_for(int64_t nnz) {
  if (g_seg_override > 0) return g_seg_override;
  int seg = 64;
  while (seg < 1024 && nnz / seg > 32768) seg *= 2;
  return seg;
}

struct Plan {
  int32_t* grows;       // [G*R] row of (group, slot); -1 = empty slot
  int32_t* gptr;        // [G+1] merged-list offsets
  int32_t* gcol;        // [nnz] merged columns (capacity)
  uint32_t* gmask;      // [nnz] which rows of the group hold the column
  int32_t* item_group;  // [cap_items] work items: (group, segment of its merged list)
  int32_t* item_m0;
  int32_t* item_m1;
  int32_t* item_out;    // -1: the item is its group's only one and writes Y; else partial slot
  int32_t* long_group;  // [cap_long] groups with several items
  int32_t* long_slot;   // [cap_long] first partial slot of the group
  int64_t* counts;      // [8] merged, launch slots, partial slots, long groups, segment length, items
  int64_t G, cap_items, cap_long, cap_slots;
  size_t total;
};

Plan carve_plan(void* base, int64_t n_rows, int64_t nnz, int R) {
  Plan p{};
  Carver c(base);
  p.G = (n_rows + R - 1) / R;
  p.cap_long = nnz / seg_cols_for(nnz) + 1;
  p.cap_items = p.G + p.cap_long;
  p.counts = c.take<int64_t>(8);
  p.grows = c.take<int32_t>(size_t(p.G) * R);
  p.gptr = c.take<int32_t>(size_t(p.G) + 1);
  p.gcol = c.take<int32_t>(size_t(nnz));
  p.gmask = c.take<uint32_t>(size_t(nnz));
  // launch slots: the items are dealt out to the 8 XCDs in work-balanced runs, padded
  // to the longest run (see k_item_deal), so up to 8x the item count
  p.cap_slots = kXcds * (p.cap_items + kWavesPerBlock);
  p.item_group = c.take<int32_t>(size_t(p.cap_slots));
  p.item_m0 = c.take<int32_t>(size_t(p.cap_slots));
  p.item_m1 = c.take<int32_t>(size_t(p.cap_slots));
  p.item_out = c.take<int32_t>(size_t(p.cap_slots));
  p.long_group = c.take<int32_t>(size_t(p.cap_long));
  p.long_slot = c.take<int32_t>(size_t(p.cap_long));
  p.total = c.off;
  return p;
}

struct BuildWs {
  uint32_t *dkey_a, *dkey_b, *drow_a, *drow_b;  // degree sort
  int32_t* rank;
  uint64_t *keys_a, *keys_b;
  uint32_t *head, *pos;
  int32_t* ogroup;
  uint32_t *nseg, *first, *lseg, *slot0, *lflag, *lidx;
  uint64_t *ikey_a, *ikey_b;   // item ordering
  int32_t* itmp;               // [4 * cap_items]
  uint32_t *ilen, *icum;       // [cap_items] merged columns per item in launch order, running sum
  int32_t* xbound;             // [kXcds + 1] first item of every XCD's run, [kXcds + 1] = dealing on
  void* prim;
  size_t prim_bytes;
  size_t total;
};

BuildWs carve_build(void* base, int64_t n_rows, int64_t nnz, int R) {
  BuildWs w{};
  Carver c(base);
  const int64_t G = (n_rows + R - 1) / R;
  w.dkey_a = c.take<uint32_t>(n_rows);
  w.dkey_b = c.take<uint32_t>(n_rows);
  w.drow_a = c.take<uint32_t>(n_rows);
  w.drow_b = c.take<uint32_t>(n_rows);
  w.rank = c.take<int32_t>(n_rows);
  w.keys_a = c.take<uint64_t>(nnz);
  w.keys_b = c.take<uint64_t>(nnz);
  w.head = c.take<uint32_t>(nnz);
  w.pos = c.take<uint32_t>(nnz);
  w.ogroup = c.take<int32_t>(nnz);
  w.nseg = c.take<uint32_t>(G);
  w.first = c.take<uint32_t>(G);
  w.lseg = c.take<uint32_t>(G);
  w.slot0 = c.take<uint32_t>(G);
  w.lflag = c.take<uint32_t>(G);
  w.lidx = c.take<uint32_t>(G);
  const int64_t cap_items = G + nnz / seg_cols_for(nnz) + 1;
  w.ikey_a = c.take<uint64_t>(cap_items);
  w.ikey_b = c.take<uint64_t>(cap_items);
  w.itmp = c.take<int32_t>(4 * cap_items);
  w.ilen = c.take<uint32_t>(cap_items);
  w.icum = c.take<uint32_t>(cap_items);
  w.xbound = c.take<int32_t>(kXcds + 2);
  size_t s1 = 0, s2 = 0, s3 = 0;
  uint64_t* k64 = nullptr;
  uint32_t* k32 = nullptr;
  (void)rocprim::radix_sort_keys(nullptr, s1, k64, k64, size_t(nnz > cap_items ? nnz : cap_items), 0u, 64u,
                                 hipStream_t(0));
  (void)rocprim::radix_sort_pairs(nullptr, s2, k32, k32, k32, k32, size_t(n_rows), 0u, 32u, hipStream_t(0));
  (void)rocprim::exclusive_scan(nullptr, s3, k32, k32, 0u, size_t(nnz > n_rows ? nnz : n_rows),
                                rocprim::plus<uint32_t>(), hipStream_t(0));
  w.prim_bytes = align_up(s1 > s2 ? (s1 > s3 ? s1 : s3) : (s2 > s3 ? s2 : s3));
  w.prim = c.take<char>(w.prim_bytes);
  w.total = c.off;
  return w;
}

// ---- plan construction -------------------------------------------------------
__global__ void k_deg_keys(int64_t n, const int32_t* __restrict__ rowptr, uint32_t* __restrict__ key,
                           uint32_t* __restrict__ row) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  key[i] = 0xFFFFFFFFu - uint32_t(rowptr[i + 1] - rowptr[i]);   // ascending key = descending degree
  row[i] = uint32_t(i);
}

__global__ void k_rank(int64_t n, int64_t slots, const uint32_t* __restrict__ order, int32_t* __restrict__ rank,
                       int32_t* __restrict__ grows) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= slots) return;
  if (i < n) {
    rank[order[i]] = int32_t(i);
    grows[i] = int32_t(order[i]);
  } else {
    grows[i] = -1;
  }
}

// key of an entry = (group of its row, column, slot of its row in the group)
__global__ void k_entry_keys(int64_t n, int64_t nnz, const int32_t* __restrict__ rowptr,
                             const int32_t* __restrict__ colidx, const int32_t* __restrict__ rank, int R, int rbits,
                             int cbits, uint64_t* __restrict__ keys) {
  int64_t e = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (e >= nnz) return;
  int64_t lo = 0, hi = n;  // last row whose rowptr <= e
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (int64_t(rowptr[mid]) <= e) lo = mid; else hi = mid;
  }
  const uint32_t rk = uint32_t(rank[lo]);
  keys[e] = (((uint64_t(rk / R) << cbits) | uint64_t(colidx[e])) << rbits) | uint64_t(rk % R);
}

__global__ void k_heads(int64_t nnz, const uint64_t* __restrict__ keys, int rbits, uint32_t* __restrict__ head) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= nnz) return;
  head[i] = (i == 0 || (keys[i] >> rbits) != (keys[i - 1] >> rbits)) ? 1u : 0u;
}

__global__ void k_emit(int64_t nnz, const uint64_t* __restrict__ keys, const uint32_t* __restrict__ head,
                       const uint32_t* __restrict__ pos, int rbits, int cbits, int32_t* __restrict__ gcol,
                       uint32_t* __restrict__ gmask, int32_t* __restrict__ ogroup) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= nnz || !head[i]) return;
  const uint64_t gc = keys[i] >> rbits;
  uint32_t mask = 0;
  for (int64_t j = i; j < nnz && (keys[j] >> rbits) == gc; ++j) mask |= 1u << uint32_t(keys[j] & ((1u << rbits) - 1));
  const uint32_t p = pos[i];
  gcol[p] = int32_t(gc & ((uint64_t(1) << cbits) - 1));
  gmask[p] = mask;
  ogroup[p] = int32_t(gc >> cbits);
}

__global__ void k_gptr(int64_t G, int64_t nnz, const uint32_t* __restrict__ head, const uint32_t* __restrict__ pos,
                       const int32_t* __restrict__ ogroup, int32_t* __restrict__ gptr, int64_t* __restrict__ counts) {
  int64_t j = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  const int64_t merged = nnz > 0 ? int64_t(pos[nnz - 1]) + head[nnz - 1] : 0;
  if (j > merged) return;
  int64_t lo = j == 0 ? 0 : int64_t(ogroup[j - 1]) + 1;
  int64_t hi = j == merged ? G : int64_t(ogroup[j]);
  for (int64_t q = lo; q <= hi; ++q) gptr[q] = int32_t(j);
  if (j == merged) counts[0] = merged;
}

__global__ void k_nseg(int64_t G, int kSegCols, const int32_t* __restrict__ gptr, uint32_t* __restrict__ nseg,
                       uint32_t* __restrict__ lseg, uint32_t* __restrict__ lflag) {
  int64_t g = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (g >= G) return;
  const int32_t len = gptr[g + 1] - gptr[g];
  const uint32_t ns = len > kSegCols ? uint32_t((len + kSegCols - 1) / kSegCols) : 1u;
  nseg[g] = ns;
  lseg[g] = ns > 1 ? ns : 0u;
  lflag[g] = ns > 1 ? 1u : 0u;
}

__global__ void k_items(int64_t G, int kSegCols, const int32_t* __restrict__ gptr, const uint32_t* __restrict__ nseg,
                        const uint32_t* __restrict__ first, const uint32_t* __restrict__ slot0,
                        const uint32_t* __restrict__ lidx, int32_t* __restrict__ item_group,
                        int32_t* __restrict__ item_m0, int32_t* __restrict__ item_m1, int32_t* __restrict__ item_out,
                        int32_t* __restrict__ long_group, int32_t* __restrict__ long_slot,
                        int64_t* __restrict__ counts) {
  int64_t g = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (g >= G) return;
  const uint32_t ns = nseg[g];
  const int32_t b = gptr[g], e = gptr[g + 1];
  for (uint32_t k = 0; k < ns; ++k) {
    const uint32_t it = first[g] + k;
    item_group[it] = int32_t(g);
    item_m0[it] = b + int32_t(k) * kSegCols;
    item_m1[it] = min(b + int32_t(k + 1) * kSegCols, e);
    item_out[it] = ns > 1 ? int32_t(slot0[g] + k) : -1;
  }
  if (ns > 1) {
    long_group[lidx[g]] = int32_t(g);
    long_slot[lidx[g]] = int32_t(slot0[g]);
  }
  if (g == G - 1) {
    counts[5] = int64_t(first[g]) + ns;
    counts[2] = int64_t(slot0[g]) + (ns > 1 ? ns : 0);
    counts[3] = int64_t(lidx[g]) + (ns > 1 ? 1 : 0);
    counts[4] = kSegCols;
  }
}

// Order the work items by the column block their merged list starts in (ties: build
// order).  Wavefronts that run at the same time then walk the same region of X, and
// most of their row fetches hit the XCD's L2 instead of going to the Infinity Cache.
int g_order_cols = 4096;   // sgnn_tune_set("rg_order", v)

__global__ void k_item_keys(int64_t cap_items, const int64_t* __restrict__ counts, const int32_t* __restrict__ item_m0,
                            const int32_t* __restrict__ gcol, int order_cols, uint64_t* __restrict__ keys) {
  int64_t it = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (it >= cap_items) return;
  keys[it] = it < counts[5] ? ((uint64_t(gcol[item_m0[it]] / order_cols) << 32) | uint64_t(it)) : ~uint64_t(0);
}

__global__ void k_item_copy(int64_t cap_items, const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                            const int32_t* __restrict__ c, const int32_t* __restrict__ d, int32_t* __restrict__ tmp) {
  int64_t it = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (it >= cap_items) return;
  tmp[it] = a[it];
  tmp[cap_items + it] = b[it];
  tmp[2 * cap_items + it] = c[it];
  tmp[3 * cap_items + it] = d[it];
}

__global__ void k_item_len(int64_t cap_items, const int64_t* __restrict__ counts, const uint64_t* __restrict__ keys,
                           const int32_t* __restrict__ tmp, uint32_t* __restrict__ len) {
  int64_t it = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (it >= cap_items) return;
  uint32_t l = 0;
  if (it < counts[5]) {
    const int64_t src = int64_t(keys[it] & 0xffffffffu);
    l = uint32_t(tmp[2 * cap_items + src] - tmp[cap_items + src]);
  }
  len[it] = l;
}

// Cut the ordered item list into kXcds consecutive runs of equal merged-column work
// (run starts rounded up to whole workgroups).  One thread per boundary.
// Dealing pays where the gather bounds the kernel and rows share columns (measured on
// the Reddit shape, R = 16: 12.5 -> 11.3 ms); on sparse graphs whose merged lists are
// barely shorter than the rows (merged > 3/4 of the entries) and for the narrow-row
// R = 8 plans it only adds padding slots, so those keep one run: bound[1..] = items.
__global__ void k_xcd_bounds(const uint32_t* __restrict__ cum, int64_t* __restrict__ counts,
                             int32_t* __restrict__ bound, int R, int64_t nnz) {
  __shared__ int32_t b[kXcds + 1];
  const int k = threadIdx.x;
  const int64_t items = counts[5];
  const bool deal = R == 16 && counts[0] * 4 < nnz * 3;
  if (k <= kXcds) {
    int64_t pos = items;
    if (k == 0) pos = 0;
    else if (deal && k < kXcds && items > 0) {
      const uint64_t want = uint64_t(cum[items - 1]) * uint64_t(k) / kXcds;
      int64_t lo = 0, hi = items;   // first item whose running sum reaches `want`
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (uint64_t(cum[mid]) < want) lo = mid + 1; else hi = mid;
      }
      pos = min(items, (lo + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock);
    }
    b[k] = int32_t(pos);
    bound[k] = int32_t(pos);
  }
  __syncthreads();
  if (k == 0) {
    int32_t mx = 0;
    for (int q = 0; q < kXcds; ++q) mx = max(mx, b[q + 1] - b[q]);
    mx = (mx + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock;
    counts[1] = deal ? int64_t(mx) * kXcds : int64_t(mx);   // launch slots
    bound[kXcds + 1] = deal ? 1 : 0;
  }
}

__global__ void k_item_fill(int64_t cap_slots, int32_t* __restrict__ a, int32_t* __restrict__ b,
                            int32_t* __restrict__ c, int32_t* __restrict__ d) {
  int64_t it = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (it >= cap_slots) return;
  a[it] = 0;
  b[it] = 0;
  c[it] = 0;
  d[it] = -2;   // padding slot: nothing to do
}

// Deal the ordered items out so that workgroup b (4 consecutive launch slots) takes
// its items from run b % kXcds: each XCD then works through one contiguous stretch of
// the column-ordered list, and its L2 is not shared with seven other stretches.
__global__ void k_item_deal(int64_t cap_items, const int64_t* __restrict__ counts, const uint64_t* __restrict__ keys,
                            const int32_t* __restrict__ tmp, const int32_t* __restrict__ bound,
                            int32_t* __restrict__ a, int32_t* __restrict__ b, int32_t* __restrict__ c,
                            int32_t* __restrict__ d) {
  int64_t it = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (it >= counts[5]) return;
  int k = 0;
  while (k + 1 < kXcds && it >= bound[k + 1]) ++k;
  const int64_t j = it - bound[k];
  const int64_t pos = bound[kXcds + 1] ? ((j / kWavesPerBlock) * kXcds + k) * kWavesPerBlock + j % kWavesPerBlock : it;
  const int64_t src = int64_t(keys[it] & 0xffffffffu);
  a[pos] = tmp[src];
  b[pos] = tmp[cap_items + src];
  c[pos] = tmp[2 * cap_items + src];
  d[pos] = tmp[3 * cap_items + src];
}

// Dense per-column weights for graphs whose stored values differ: gw[m][r] = value of the
// entry (row r of the group, merged column m), 0 where the row has no such entry.  One
// wavefront per (group, row of the group): its lanes take the row's CSR entries and look
// each column up in the group's sorted merged list.
__global__ __launch_bounds__(256) void k_fill_weights(int64_t slots, int R, const int32_t* __restrict__ grows,
                                                      const int32_t* __restrict__ gptr, const int32_t* __restrict__ gcol,
                                                      const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ colidx, const float* __restrict__ vals,
                                                      float* __restrict__ gw) {
  const int64_t s = blockIdx.x * int64_t(blockDim.x / kWave) + threadIdx.x / kWave;
  if (s >= slots) return;
  const int row = grows[s];
  if (row < 0) return;
  const int64_t g = s / R;
  const int r = int(s % R);
  const int lo0 = gptr[g], hi0 = gptr[g + 1];
  for (int e = rowptr[row] + int(threadIdx.x & (kWave - 1)); e < rowptr[row + 1]; e += kWave) {
    const int c = colidx[e];
    int lo = lo0, hi = hi0;
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (gcol[mid] < c) lo = mid + 1; else hi = mid;
    }
    gw[int64_t(lo) * R + r] = vals[e];   // the column is in the list by construction
  }
}

// ---- the kernel ------------------------------------------------------------------
struct RgArgs {
  const int32_t *grows, *gptr, *gcol;
  const uint32_t* gmask;
  const int32_t *item_group, *item_m0, *item_m1, *item_out, *long_group, *long_slot;
  const float *dinv_row, *X, *corr;
  const float* dinv_col;   // scale of every fetched row (utils.py:93) applied in the kernel; NULL: X arrives scaled
  const float* gw;   // [merged][R] weight of (merged column, row of the group), NULL: pattern only
  float *Y, *partial;
  int64_t ldx, ldy, x_bytes, x_row_offset;
  float w_off;
  int32_t items, n_long, F, row_bytes, y_vec, epilogue, seg;
};

// ---- feature layout ------------------------------------------------------------------
// A lane owns NF = 4*C4 + 2*C2 + C1 features of every accumulated row: C4 chunks of
// 256 features read with one 16-byte load per lane, then C2 chunks of 128 (8-byte
// loads), then C1 chunks of 64 (4-byte loads).  Measured on the Reddit shape: a row
// fetch costs ~2.5 ms per 16-byte chunk (1 KiB per wavefront instruction) against
// ~1.7 ms per 8-byte chunk (512 B), so the wide loads carry as much of the row as the
// register budget allows and the narrower ones only its tail.  Rows of X must start on
// 16-byte boundaries (X 16-byte aligned, ldx a multiple of 4).
template <int C4, int C2, int C1>
struct Lay {
  static constexpr int NF = 4 * C4 + 2 * C2 + C1;
  static constexpr int kOff2 = C4 * 256;             // first feature of the 8-byte chunks
  static constexpr int kOff1 = C4 * 256 + C2 * 128;  // first feature of the 4-byte chunks
};

// Byte offset of a lane's piece of a fetched row.  The buffer load's range check sees
// this voffset only (the row offset travels in soffset, which is not checked), so the
// lanes of the row's LAST chunk whose first feature lies at or past F get an offset no
// descriptor covers and read 0 without touching memory.  A lane that starts below F may
// run up to 3 floats past it: those lie inside the row's stride (ldx is a multiple of
// 4), which is why X must be readable for n_cols * ldx floats.
template <bool LAST_KIND>
__device__ __forceinline__ int rg_voff(const RgArgs& a, int c, int chunks, int feature) {
  if (LAST_KIND && c == chunks - 1) return feature < a.F ? feature * 4 : kOobVoffset;
  return feature * 4;
}

// Weight modes of the kernel: pattern sum (the fetched rows are added as they are), one
// weight per (merged column, row of the group) pair, one scale per fetched row.
constexpr int kPattern = 0, kPairWeights = 1, kColScale = 2;

// Issue the fetches of UU consecutive merged columns (lanes k .. k+UU-1 of `cols`).
// With pair weights the R weights of each of the UU columns follow: they are stored
// [merged column][row of the group], so one coalesced load by the first UU*R lanes brings
// the whole step's weights (`mabs` = index of the step's first column in the merged list).
template <class L, int R, int C4, int C2, int C1, int UU, bool BIGX, int WM>
__device__ __forceinline__ void rg_load(const RgArgs& a, i32x4 rsrc, int32_t cols, int k, int lane, int64_t mabs,
                                        float (&xv)[UU][L::NF], float& wv) {
  if constexpr (WM == kPairWeights) {
    static_assert(UU * R <= kWave, "a step's weights must fit one wavefront-wide load");
    wv = lane < UU * R ? a.gw[mabs * R + lane] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < UU; ++u) {
    const int32_t col = __builtin_amdgcn_readlane(cols, k + u);
    i32x4 rr = rsrc;
    int soff = 0;
    if constexpr (BIGX) {   // X spans 4 GiB or more: re-base the descriptor on the fetched row
      rr = make_rsrc(a.X + int64_t(col) * a.ldx, uint32_t(a.F) * 4u);
    } else {
      soff = int(uint32_t(col) * uint32_t(a.row_bytes));   // up to 4 GiB: unsigned
    }
#pragma unroll
    for (int c = 0; c < C4; ++c) {
      const f32x4 x = raw_buffer_load_f32x4(rr, rg_voff<(C2 == 0 && C1 == 0)>(a, c, C4, lane * 4 + c * 256), soff, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[u][4 * c + j] = x[j];
    }
#pragma unroll
    for (int c = 0; c < C2; ++c) {
      const f32x2 x = raw_buffer_load_f32x2(rr, rg_voff<(C1 == 0)>(a, c, C2, lane * 2 + L::kOff2 + c * 128), soff, 0);
      xv[u][4 * C4 + 2 * c] = x[0];
      xv[u][4 * C4 + 2 * c + 1] = x[1];
    }
#pragma unroll
    for (int c = 0; c < C1; ++c)
      xv[u][4 * C4 + 2 * C2 + c] = raw_buffer_load_f32(rr, rg_voff<true>(a, c, C1, lane + L::kOff1 + c * 64), soff, 0);
  }
}

// Add the fetched rows into the accumulators of the rows whose mask bit is set.
// Two thirds of the merged columns belong to a single row of their group, so the R
// bit tests are guarded four at a time: an empty nibble costs one test, not four.
// (Walking the set bits with find-first-set and a compare tree down to the statically
// indexed accumulator was tried: the compiler folds the tree's leaves into one
// dynamically indexed access and moves the accumulators to scratch memory; with the
// fold blocked the kernel needs 386 registers, one wavefront per SIMD.)
// (`wv`: the step's pair weights as rg_load left them; `dc`: the scales of the batch's 64
// columns, one per lane.)
template <int R, int NF, int UU, int WM>
__device__ __forceinline__ void rg_accum(uint32_t masks, int k, const float (&xv)[UU][NF], float wv, float dc,
                                         float (&acc)[R][NF]) {
#pragma unroll
  for (int u = 0; u < UU; ++u) {
    const uint32_t mk = uint32_t(__builtin_amdgcn_readlane(int(masks), k + u));
    // Column scale: the fetched row is scaled once, rounded, and then added like a row that
    // arrived scaled, so that the result does not depend on who applied the scaling.
    [[maybe_unused]] float xs[NF];
    if constexpr (WM == kColScale) {
#pragma clang fp contract(off)
      const float wc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dc), k + u));
#pragma unroll
      for (int i = 0; i < NF; ++i) xs[i] = wc * xv[u][i];
    }
#pragma unroll
    for (int q = 0; q < R; q += 4)
      if (mk & (0xFu << q)) {   // wave-uniform: scalar tests and branches throughout
#pragma unroll
        for (int r = q; r < q + 4; ++r)
          if (mk & (1u << r)) {
            if constexpr (WM == kPairWeights) {
              const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv), u * R + r));
#pragma unroll
              for (int i = 0; i < NF; ++i) acc[r][i] = fmaf(w, xv[u][i], acc[r][i]);
            } else if constexpr (WM == kColScale) {
#pragma unroll
              for (int i = 0; i < NF; ++i) acc[r][i] += xs[i];
            } else {
#pragma unroll
              for (int i = 0; i < NF; ++i) acc[r][i] += xv[u][i];
            }
          }
      }
  }
}

// Row epilogue.  `s` is the pattern sum (every stored entry counted once).  For an
// adjacency whose off-diagonal entries all carry the weight w_off and whose
// diagonal entry of row i is d_i, the weighted sum is w_off*s + (d_i - w_off)*x'_i
// (corr[i] = d_i - w_off, x' = the row of X the kernel saw); unit-weight graphs have
// w_off = 1 and corr = NULL.  Then the row scaling of utils.py:95.
__device__ __forceinline__ float rg_epilogue(const RgArgs& a, float s, int row, int f, bool scale, float dr) {
  float y = s;
  if (a.gw != nullptr) {
    // the weights were applied entry by entry
  } else if (a.corr != nullptr) {
    const int64_t xr = int64_t(row) + a.x_row_offset;
    float xs = a.X[xr * a.ldx + f];
    if (a.dinv_col != nullptr) xs *= a.dinv_col[xr];
    y = fmaf(a.w_off, y, a.corr[row] * xs);
  } else if (a.w_off != 1.0f) {
    y *= a.w_off;
  }
  if (scale) {
    y = y * dr;
    if (a.epilogue == SGNN_EPI_SCALE_ROW2) y = y * dr;
  }
  return y;
}

// store V consecutive features f .. f+V-1 of one output row (those below F)
template <int V>
__device__ __forceinline__ void rg_store(const RgArgs& a, float* dst, const float* o, int f, int aligned_for) {
  if (f + V <= a.F) {
    if constexpr (V == 4) {
      if (aligned_for >= 4) {
        *reinterpret_cast<f32x4*>(dst + f) = f32x4{o[0], o[1], o[2], o[3]};
        return;
      }
      if (aligned_for >= 2) {
        *reinterpret_cast<f32x2*>(dst + f) = f32x2{o[0], o[1]};
        *reinterpret_cast<f32x2*>(dst + f + 2) = f32x2{o[2], o[3]};
        return;
      }
    } else if constexpr (V == 2) {
      if (aligned_for >= 2) {
        *reinterpret_cast<f32x2*>(dst + f) = f32x2{o[0], o[1]};
        return;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j)
    if (f + j < a.F) dst[f + j] = o[j];
}

// Write a wavefront's R accumulated rows: straight to Y through the epilogue when the
// group has a single work item (out < 0), else as partial rows into slot `out`.
template <int R, int C4, int C2, int C1>
__device__ __forceinline__ void rg_write_rows(const RgArgs& a, const float (&acc)[R][Lay<C4, C2, C1>::NF], int g,
                                              int out, int lane) {
  using L = Lay<C4, C2, C1>;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = a.grows[g * R + r];
    if (row < 0) continue;
    float* dst;
    float dr = 1.0f;
    bool scale = false;
    int aligned_for;
    if (out < 0) {
      dst = a.Y + int64_t(row) * a.ldy;
      scale = a.dinv_row != nullptr;
      if (scale) dr = a.dinv_row[row];
      aligned_for = a.y_vec;
    } else {
      dst = a.partial + (int64_t(out) * R + r) * a.F;
      aligned_for = (a.F % 4 == 0) ? 4 : (a.F % 2 == 0 ? 2 : 1);
    }
    auto put = [&](auto vtag, int f, const float* s) {
      constexpr int V = decltype(vtag)::value;
      if (f >= a.F) return;
      float o[V];
#pragma unroll
      for (int j = 0; j < V; ++j)
        o[j] = (out < 0 && f + j < a.F) ? rg_epilogue(a, s[j], row, f + j, scale, dr) : s[j];
      rg_store<V>(a, dst, o, f, aligned_for);
    };
#pragma unroll
    for (int c = 0; c < C4; ++c) put(std::integral_constant<int, 4>{}, c * 256 + lane * 4, &acc[r][4 * c]);
#pragma unroll
    for (int c = 0; c < C2; ++c)
      put(std::integral_constant<int, 2>{}, L::kOff2 + c * 128 + lane * 2, &acc[r][4 * C4 + 2 * c]);
#pragma unroll
    for (int c = 0; c < C1; ++c)
      put(std::integral_constant<int, 1>{}, L::kOff1 + c * 64 + lane, &acc[r][4 * C4 + 2 * C2 + c]);
  }
}

// One wavefront per work item; a lane owns NF features of all R accumulated rows.
// Merged columns are taken 64 at a time through a VGPR (ids and masks) and handed to
// the scalar unit by v_readlane; every row of a group is summed in ascending column
// order.
template <int R, int C4, int C2, int C1, int U, bool BIGX, int WM>
__global__ __launch_bounds__(kWave* kWavesPerBlock) void k_rg_spmm(RgArgs a) {
  using L = Lay<C4, C2, C1>;
  constexpr int NF = L::NF;
  const int lane = threadIdx.x & (kWave - 1);
  // Launch slots are laid out so that the workgroups of one XCD walk one contiguous,
  // work-balanced stretch of the column-ordered item list (k_item_deal).
  const int item = __builtin_amdgcn_readfirstlane(int(blockIdx.x) * kWavesPerBlock + int(threadIdx.x) / kWave);
  if (item >= a.items) return;
  const int out = a.item_out[item];
  if (out < -1) return;   // padding slot
  const int g = a.item_group[item];
  const int m0 = a.item_m0[item], m1 = a.item_m1[item];
  const i32x4 rsrc = make_rsrc(a.X, BIGX ? 0u : uint32_t(a.x_bytes));
  float acc[R][NF];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[r][i] = 0.0f;

  int32_t ncols = 0;
  uint32_t nmask = 0;
  float ndc = 0.0f;   // column-scale mode: the scales of the batch's columns
  if (m0 + lane < m1) {
    ncols = a.gcol[m0 + lane];
    nmask = a.gmask[m0 + lane];
    if constexpr (WM == kColScale) ndc = a.dinv_col[ncols];
  }
  // Two fetch buffers: while the rows of one step are added, the fetches of the
  // following step are already in flight (also across the 64-column batches).
  // Invariant at the top of a batch: xa holds the batch's step 0 iff the batch has
  // at least U columns.  Every batch but an item's last is full and 64 / U is even,
  // so an odd number of full steps can only occur in the last batch.
  float xa[U][NF], xb[U][NF];
  float wa = 0.0f, wb = 0.0f;
  if (min(kWave, m1 - m0) >= U) rg_load<L, R, C4, C2, C1, U, BIGX, WM>(a, rsrc, ncols, 0, lane, m0, xa, wa);
  for (int m = m0; m < m1; m += kWave) {
    const int32_t cols = ncols;
    const uint32_t masks = nmask;
    const float dc = ndc;
    const int cnt = min(kWave, m1 - m);
    const bool more = m + kWave < m1;
    const bool mine = m + kWave + lane < m1;
    if (more) {
      ncols = mine ? a.gcol[m + kWave + lane] : 0;
      nmask = mine ? a.gmask[m + kWave + lane] : 0u;
    }
    const int nfull = cnt / U * U;
    const bool next_full_step = more && (m1 - (m + kWave)) >= U;
    int k = 0;
    for (; k + 2 * U <= nfull; k += 2 * U) {
      rg_load<L, R, C4, C2, C1, U, BIGX, WM>(a, rsrc, cols, k + U, lane, m + k + U, xb, wb);
      rg_accum<R, NF, U, WM>(masks, k, xa, wa, dc, acc);
      if (k + 2 * U < nfull) rg_load<L, R, C4, C2, C1, U, BIGX, WM>(a, rsrc, cols, k + 2 * U, lane, m + k + 2 * U, xa, wa);
      else if (next_full_step) rg_load<L, R, C4, C2, C1, U, BIGX, WM>(a, rsrc, ncols, 0, lane, m + kWave, xa, wa);
      rg_accum<R, NF, U, WM>(masks, k + U, xb, wb, dc, acc);
    }
    if (k < nfull) {   // odd count of full steps: last batch only, xa holds the step
      rg_accum<R, NF, U, WM>(masks, k, xa, wa, dc, acc);
      k += U;
    }
    for (; k < cnt; ++k) {
      float x1[1][NF];
      float w1 = 0.0f;
      rg_load<L, R, C4, C2, C1, 1, BIGX, WM>(a, rsrc, cols, k, lane, m + k, x1, w1);
      rg_accum<R, NF, 1, WM>(masks, k, x1, w1, dc, acc);
    }
    // the next batch's scales: its column ids were asked for at the top of this batch
    if constexpr (WM == kColScale)
      if (more) ndc = mine ? a.dinv_col[ncols] : 0.0f;
  }
  rg_write_rows<R, C4, C2, C1>(a, acc, g, out, lane);
}

// rows of a group cut into several items: add the items' partial rows in order
template <int R>
__global__ void k_rg_combine(RgArgs a) {
  const int li = blockIdx.x;
  if (li >= a.n_long) return;
  const int g = a.long_group[li];
  const int slot = a.long_slot[li];
  const int len = a.gptr[g + 1] - a.gptr[g];
  const int nseg = (len + a.seg - 1) / a.seg;
  const int r = blockIdx.y;
  const int row = a.grows[g * R + r];
  if (row < 0) return;
  const bool scale = a.dinv_row != nullptr;
  const float dr = scale ? a.dinv_row[row] : 1.0f;
  const int64_t stride = int64_t(R) * a.F;                       // between consecutive segments' partial rows
  for (int f = threadIdx.x; f < a.F; f += blockDim.x) {
    // four interleaved running sums (segments k, k+4, ...) keep four loads in flight;
    // they are folded in a fixed order, so the result stays reproducible
    const float* p = a.partial + (int64_t(slot) * R + r) * a.F + f;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int k = 0;
    for (; k + 4 <= nseg; k += 4) {
      s0 += p[(k + 0) * stride];
      s1 += p[(k + 1) * stride];
      s2 += p[(k + 2) * stride];
      s3 += p[(k + 3) * stride];
    }
    for (; k < nseg; ++k) s0 += p[k * stride];
    float s = (s0 + s1) + (s2 + s3);
    s = rg_epilogue(a, s, row, f, scale, dr);
    a.Y[int64_t(row) * a.ldy + f] = s;
  }
}

template <int R, int C4, int C2, int C1>
int launch_rg(const RgArgs& a, hipStream_t st) {
  // row fetches in flight per wavefront and buffer: about 8-10 registers' worth of loads
  constexpr int NF = Lay<C4, C2, C1>::NF;
  constexpr int U = NF <= 2 ? 8 : (NF <= 4 ? 4 : 2);
  constexpr int UW = U * R <= kWave ? U : kWave / R;   // with weights: a step's U*R weights in one load
  const dim3 grid(unsigned((a.items + kWavesPerBlock - 1) / kWavesPerBlock)), block(kWave * kWavesPerBlock);
  const bool bigx = a.x_bytes >= kBigX;
  sgnn::kernel_timer_begin(st);
  if (a.gw != nullptr) {
    if (bigx) hipLaunchKernelGGL((k_rg_spmm<R, C4, C2, C1, UW, true, kPairWeights>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_rg_spmm<R, C4, C2, C1, UW, false, kPairWeights>), grid, block, 0, st, a);
  } else if (a.dinv_col != nullptr) {
    // built for the groups of 8 of small graphs only (the entry point has checked R and the size of X)
    if constexpr (R == 8) hipLaunchKernelGGL((k_rg_spmm<R, C4, C2, C1, U, false, kColScale>), grid, block, 0, st, a);
    else return fail(SGNN_ERR_UNSUPPORTED, "column scaling inside the row-group kernel needs R = 8");
  } else {
    if (bigx) hipLaunchKernelGGL((k_rg_spmm<R, C4, C2, C1, U, true, kPattern>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_rg_spmm<R, C4, C2, C1, U, false, kPattern>), grid, block, 0, st, a);
  }
  HIP_TRY(hipGetLastError());
  sgnn::kernel_timer_end(st);
  if (a.n_long > 0) {
    hipLaunchKernelGGL(k_rg_combine<R>, dim3(unsigned(a.n_long), R), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
  }
  return SGNN_OK;
}

// Feature layout for width F: as many 16-byte chunks as fit, the remainder in one
// narrower chunk (or one more 16-byte chunk when it is longer than an 8-byte chunk holds).
struct Layout {
  int c4, c2, c1;
  int nf() const { return 4 * c4 + 2 * c2 + c1; }
};

Layout rg_layout(int64_t F) {
  Layout l{int(F / 256), 0, 0};
  const int64_t rem = F - int64_t(l.c4) * 256;
  if (rem > 128) l.c4 += 1;
  else if (rem > 64) l.c2 = 1;
  else if (rem > 0) l.c1 = 1;
  return l;
}

// Widths beyond what 16 rows x 10 registers hold are cut into equal feature slices of at
// most 640 columns, one launch per slice (measured on the Reddit shape: F = 1024 as one
// R = 8 launch 26.7 ms, as two R = 16 slices 15.8 ms).
constexpr int64_t kMaxSlice = 640;

int64_t rg_slice_width(int64_t F) {
  if (F <= kMaxSlice) return F;
  const int64_t ns = (F + kMaxSlice - 1) / kMaxSlice;
  return ((F + ns - 1) / ns + 3) / 4 * 4;
}

// rows per group for a layout: the accumulators (R * NF registers per lane) must leave
// room for two wavefronts per SIMD; narrow rows are bound by the mask tests, not the
// gather, and keep R = 8
int rg_rows_for(int64_t F) {
  if (F < 33) return 0;
  const int64_t width = rg_slice_width(F);
  if (F > width && F - (F - 1) / width * width < 33) return 0;   // last slice too narrow: plain CSR kernel
  F = width;
  const Layout l = rg_layout(F);
  if (l.nf() > 10) return 0;   // cannot happen for a slice of at most 640 columns
  if (g_R_override == 8 || g_R_override == 16) return g_R_override;
  return F > 128 ? 16 : 8;
}

template <int R, int C4>
int dispatch_rg_tail(const RgArgs& a, const Layout& l, hipStream_t st) {
  if (l.c2 == 0 && l.c1 == 0) {
    if constexpr (C4 > 0) return launch_rg<R, C4, 0, 0>(a, st);
  } else if (l.c2 == 1 && l.c1 == 0) {
    if constexpr (4 * C4 + 2 <= 10) return launch_rg<R, C4, 1, 0>(a, st);
  } else if (l.c2 == 0 && l.c1 == 1) {
    if constexpr (4 * C4 + 1 <= 10) return launch_rg<R, C4, 0, 1>(a, st);
  }
  return fail(SGNN_ERR_UNSUPPORTED, "no row-group kernel for this (R, F)");
}

int dispatch_rg(const RgArgs& a, int R, const Layout& l, hipStream_t st) {
  if (R == 16 && l.nf() <= 10) {
    if (l.c4 == 0) return dispatch_rg_tail<16, 0>(a, l, st);
    if (l.c4 == 1) return dispatch_rg_tail<16, 1>(a, l, st);
    if (l.c4 == 2) return dispatch_rg_tail<16, 2>(a, l, st);
  } else if (R == 8 && l.nf() <= 10) {
    if (l.c4 == 0) return dispatch_rg_tail<8, 0>(a, l, st);
    if (l.c4 == 1) return dispatch_rg_tail<8, 1>(a, l, st);
    if (l.c4 == 2) return dispatch_rg_tail<8, 2>(a, l, st);
  }
  return fail(SGNN_ERR_UNSUPPORTED, "no row-group kernel for this (R, F)");
}

}  // namespace

namespace sgnn {
void set_rowgroup_segment(int cols) { g_seg_override = cols; }
void set_rowgroup_R(int R) { g_R_override = R; }
void set_rowgroup_order(int cols) { g_order_cols = cols > 0 ? cols : 4096; }
}  // namespace sgnn

extern "C" {

int sgnn_rowgroup_config(int64_t F) { return rg_rows_for(F); }

// A graph with few rows AND few entries makes too few work items at R = 16 (one per group
// plus one per further 64-column segment of a long merged list) for the chip's 1024 SIMDs to
// hide the gather latency: it takes groups of 8 at every width (PubMed shape, F = 500:
// 2152 items at R = 16, main kernel 54 us; 3296 items at R = 8, 36 us).
int sgnn_rowgroup_config_for(int64_t n_rows, int64_t nnz, int64_t F) {
  const int R = rg_rows_for(F);
  if (R == 16 && g_R_override == 0 && n_rows / 16 + nnz / 64 < 8192) return 8;
  return R;
}

size_t sgnn_rowgroup_plan_bytes(int64_t n_rows, int64_t nnz, int R) {
  if (n_rows <= 0 || nnz < 0 || (R != 8 && R != 16)) return 0;
  return carve_plan(nullptr, n_rows, nnz, R).total;
}

size_t sgnn_rowgroup_workspace_bytes(int64_t n_rows, int64_t nnz, int R) {
  if (n_rows <= 0 || nnz < 0 || (R != 8 && R != 16)) return 0;
  return carve_build(nullptr, n_rows, nnz > 0 ? nnz : 1, R).total;
}

int sgnn_rowgroup_build(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                        int R, void* plan, size_t plan_bytes, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_rows <= 0 || n_cols <= 0 || nnz <= 0) return fail(SGNN_ERR_INVALID, "empty matrix");
  if (R != 8 && R != 16) return fail(SGNN_ERR_INVALID, "R must be 8 or 16");
  if (n_rows >= (int64_t(1) << 31) || n_cols >= (int64_t(1) << 31) || nnz >= (int64_t(1) << 31))
    return fail(SGNN_ERR_INVALID, "size exceeds int32 index range");
  if (!rowptr || !colidx || !plan || !workspace) return fail(SGNN_ERR_INVALID, "null pointer");
  Plan p = carve_plan(plan, n_rows, nnz, R);
  if (plan_bytes < p.total) return fail(SGNN_ERR_WORKSPACE, "row-group plan buffer too small");
  BuildWs w = carve_build(workspace, n_rows, nnz, R);
  if (workspace_bytes < w.total) return fail(SGNN_ERR_WORKSPACE, "row-group build workspace too small");
  const int rbits = R == 16 ? 4 : 3;
  const int cbits = ceil_log2(n_cols);
  const int gbits = ceil_log2(p.G + 1);
  if (rbits + cbits + gbits > 64) return fail(SGNN_ERR_INVALID, "key does not fit 64 bits");
  const unsigned T = 256;
  auto blocks = [&](int64_t items) { return dim3(unsigned((items + T - 1) / T)); };
  HIP_TRY(hipMemsetAsync(p.counts, 0, 64, st));
  // rank the rows by degree (stable: ties keep row order)
  hipLaunchKernelGGL(k_deg_keys, blocks(n_rows), dim3(T), 0, st, n_rows, rowptr, w.dkey_a, w.drow_a);
  HIP_TRY(hipGetLastError());
  size_t pb = w.prim_bytes;
  HIP_TRY(rocprim::radix_sort_pairs(w.prim, pb, w.dkey_a, w.dkey_b, w.drow_a, w.drow_b, size_t(n_rows), 0u, 32u, st));
  hipLaunchKernelGGL(k_rank, blocks(p.G * R), dim3(T), 0, st, n_rows, p.G * R, w.drow_b, w.rank, p.grows);
  HIP_TRY(hipGetLastError());
  // sort the entries by (group, column, slot) and merge equal (group, column)
  hipLaunchKernelGGL(k_entry_keys, blocks(nnz), dim3(T), 0, st, n_rows, nnz, rowptr, colidx, w.rank, R, rbits, cbits,
                     w.keys_a);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::radix_sort_keys(w.prim, pb, w.keys_a, w.keys_b, size_t(nnz), 0u, unsigned(rbits + cbits + gbits), st));
  hipLaunchKernelGGL(k_heads, blocks(nnz), dim3(T), 0, st, nnz, w.keys_b, rbits, w.head);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.head, w.pos, 0u, size_t(nnz), rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_emit, blocks(nnz), dim3(T), 0, st, nnz, w.keys_b, w.head, w.pos, rbits, cbits, p.gcol, p.gmask,
                     w.ogroup);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_gptr, blocks(nnz + 1), dim3(T), 0, st, p.G, nnz, w.head, w.pos, w.ogroup, p.gptr, p.counts);
  HIP_TRY(hipGetLastError());
  // work items
  const int seg = seg_cols_for(nnz);
  hipLaunchKernelGGL(k_nseg, blocks(p.G), dim3(T), 0, st, p.G, seg, p.gptr, w.nseg, w.lseg, w.lflag);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.nseg, w.first, 0u, size_t(p.G), rocprim::plus<uint32_t>(), st));
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.lseg, w.slot0, 0u, size_t(p.G), rocprim::plus<uint32_t>(), st));
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.lflag, w.lidx, 0u, size_t(p.G), rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_items, blocks(p.G), dim3(T), 0, st, p.G, seg, p.gptr, w.nseg, w.first, w.slot0, w.lidx,
                     p.item_group, p.item_m0, p.item_m1, p.item_out, p.long_group, p.long_slot, p.counts);
  HIP_TRY(hipGetLastError());
  // launch order of the items
  hipLaunchKernelGGL(k_item_keys, blocks(p.cap_items), dim3(T), 0, st, p.cap_items, p.counts, p.item_m0, p.gcol,
                     g_order_cols, w.ikey_a);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::radix_sort_keys(w.prim, pb, w.ikey_a, w.ikey_b, size_t(p.cap_items), 0u, 64u, st));
  hipLaunchKernelGGL(k_item_copy, blocks(p.cap_items), dim3(T), 0, st, p.cap_items, p.item_group, p.item_m0,
                     p.item_m1, p.item_out, w.itmp);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_item_len, blocks(p.cap_items), dim3(T), 0, st, p.cap_items, p.counts, w.ikey_b, w.itmp, w.ilen);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::inclusive_scan(w.prim, pb, w.ilen, w.icum, size_t(p.cap_items), rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_xcd_bounds, dim3(1), dim3(64), 0, st, w.icum, p.counts, w.xbound, R, nnz);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_item_fill, blocks(p.cap_slots), dim3(T), 0, st, p.cap_slots, p.item_group, p.item_m0, p.item_m1,
                     p.item_out);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_item_deal, blocks(p.cap_items), dim3(T), 0, st, p.cap_items, p.counts, w.ikey_b, w.itmp, w.xbound,
                     p.item_group, p.item_m0, p.item_m1, p.item_out);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

size_t sgnn_spmm_rowgroup_workspace_bytes(int64_t slots, int R, int64_t F) {
  if (slots < 0 || F < 0 || R <= 0) return 0;
  return align_up(size_t(slots) * size_t(R) * size_t(F) * 4 + 256);
}

size_t sgnn_rowgroup_weights_bytes(int64_t merged, int R) {
  if (merged < 0 || (R != 8 && R != 16)) return 0;
  return align_up(size_t(merged) * size_t(R) * sizeof(float) + 256);
}

int sgnn_rowgroup_weights(int64_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                          const float* vals, int R, const void* plan, size_t plan_bytes, int64_t merged,
                          float* group_weights, size_t group_weights_bytes, void* stream) {
  if (n_rows <= 0 || nnz <= 0 || merged <= 0) return fail(SGNN_ERR_INVALID, "empty matrix");
  if (R != 8 && R != 16) return fail(SGNN_ERR_INVALID, "R must be 8 or 16");
  if (!rowptr || !colidx || !vals || !plan || !group_weights) return fail(SGNN_ERR_INVALID, "null pointer");
  Plan p = carve_plan(const_cast<void*>(plan), n_rows, nnz, R);
  if (plan_bytes < p.total) return fail(SGNN_ERR_WORKSPACE, "row-group plan buffer too small");
  if (merged > nnz || group_weights_bytes < size_t(merged) * size_t(R) * sizeof(float))
    return fail(SGNN_ERR_WORKSPACE, "group weight buffer too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemsetAsync(group_weights, 0, size_t(merged) * size_t(R) * sizeof(float), st));
  const int64_t slots = p.G * R;
  hipLaunchKernelGGL(k_fill_weights, dim3(unsigned((slots + 3) / 4)), dim3(256), 0, st, slots, R, p.grows, p.gptr,
                     p.gcol, rowptr, colidx, vals, group_weights);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

int sgnn_spmm_rowgroup_f32(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t F, int R, const void* plan,
                           size_t plan_bytes, const float* group_weights, int64_t items, int64_t slots,
                           int64_t n_long, const float* dinv_row,
                           const float* dinv_col, float w_off, const float* corr, int64_t x_row_offset,
                           const float* X, int64_t ldx, float* Y, int64_t ldy, int epilogue,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (n_rows <= 0 || n_cols <= 0 || nnz <= 0 || F <= 0) return fail(SGNN_ERR_INVALID, "empty problem");
  if (!plan || !X || !Y) return fail(SGNN_ERR_INVALID, "null pointer");
  if (ldx < F || ldy < F) return fail(SGNN_ERR_INVALID, "leading dimension smaller than F");
  if (reinterpret_cast<uintptr_t>(X) % 4 || reinterpret_cast<uintptr_t>(Y) % 4)
    return fail(SGNN_ERR_INVALID, "X and Y must be 4-byte aligned");
  if (epilogue != SGNN_EPI_SCALE_ROW && epilogue != SGNN_EPI_SCALE_ROW2)
    return fail(SGNN_ERR_INVALID, "unknown epilogue mode");
  // Column scaling inside the kernel costs a gather and a multiply per fetched row that the
  // large graphs do better without (they take X scaled once by sgnn_row_scale_f32); it is
  // built for R = 8, pattern plans and X below 4 GiB, where it saves a small graph a launch
  // and a pass over X.
  if (dinv_col != nullptr && (R != 8 || group_weights != nullptr || ((n_cols - 1) * ldx + F) * 4 >= kBigX))
    return fail(SGNN_ERR_UNSUPPORTED,
                "column scaling inside the row-group kernel: only R = 8, no group weights, X below 4 GiB "
                "(else scale X first: sgnn_row_scale_f32)");
  if (reinterpret_cast<uintptr_t>(X) % 16 || ldx % 4)
    return fail(SGNN_ERR_UNSUPPORTED, "the row-group kernel needs X 16-byte aligned with ldx a multiple of 4");
  if (F < 33) return fail(SGNN_ERR_UNSUPPORTED, "F too narrow for the row-group kernel");
  Plan p = carve_plan(const_cast<void*>(plan), n_rows, nnz, R);
  if (plan_bytes < p.total) return fail(SGNN_ERR_WORKSPACE, "row-group plan buffer too small");
  if (items <= 0 || items > p.cap_slots || slots < 0 || n_long < 0 || n_long > p.cap_long)
    return fail(SGNN_ERR_INVALID, "item counts do not match the plan");
  if (slots > 0 && (!workspace || workspace_bytes < sgnn_spmm_rowgroup_workspace_bytes(slots, R, F)))
    return fail(SGNN_ERR_WORKSPACE, "row-group spmm workspace too small");
  RgArgs a{};
  a.grows = p.grows; a.gptr = p.gptr; a.gcol = p.gcol; a.gmask = p.gmask;
  a.item_group = p.item_group; a.item_m0 = p.item_m0; a.item_m1 = p.item_m1; a.item_out = p.item_out;
  a.long_group = p.long_group; a.long_slot = p.long_slot;
  a.dinv_row = dinv_row; a.partial = static_cast<float*>(workspace);
  a.gw = group_weights;
  a.dinv_col = dinv_col;
  a.w_off = w_off; a.corr = corr; a.x_row_offset = x_row_offset;
  if (corr && (x_row_offset < 0 || x_row_offset + n_rows > n_cols))
    return fail(SGNN_ERR_INVALID, "x_row_offset places the block's own rows outside X");
  a.ldx = ldx; a.ldy = ldy;
  a.items = int32_t(items); a.n_long = int32_t(n_long);
  a.epilogue = epilogue;
  a.seg = seg_cols_for(nnz);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // one launch per feature slice (a single one up to 640 columns); the slices share the
  // partial-row workspace, which is safe because they run in order on one stream
  const int64_t width = rg_slice_width(F);
  // every slice is checked before the first launch, so an unsupported width leaves Y untouched
  for (int64_t f0 = 0; f0 < F; f0 += width) {
    const int64_t w = F - f0 < width ? F - f0 : width;
    if (w < 33) return fail(SGNN_ERR_UNSUPPORTED, "feature slice too narrow for the row-group kernel");
    if (rg_layout(w).nf() > 10) return fail(SGNN_ERR_UNSUPPORTED, "no row-group kernel for this (R, F)");
  }
  if (R != 8 && R != 16) return fail(SGNN_ERR_UNSUPPORTED, "no row-group kernel for this (R, F)");
  for (int64_t f0 = 0; f0 < F; f0 += width) {
    const int64_t w = F - f0 < width ? F - f0 : width;
    a.X = X + f0;
    a.Y = Y + f0;
    a.F = int32_t(w);
    a.x_bytes = ((n_cols - 1) * ldx + w) * 4;
    a.row_bytes = a.x_bytes < kBigX ? int32_t(ldx * 4) : 0;
    const uintptr_t py = reinterpret_cast<uintptr_t>(a.Y);
    a.y_vec = (py % 16 == 0 && ldy % 4 == 0) ? 4 : ((py % 8 == 0 && ldy % 2 == 0) ? 2 : 1);
    const int rc = dispatch_rg(a, R, rg_layout(w), st);
    if (rc != SGNN_OK) return rc;
  }
  return SGNN_OK;
}

}  // extern "C"
