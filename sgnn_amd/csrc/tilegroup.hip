// Tile-group SpMM: 512 output rows share every fetched row of X, and the sums run on
// the matrix cores.
//
// Where it comes from.  On graphs of Reddit's shape the row-group kernel (rowgroup.hip)
// is bound by the rows of X it pulls from L2: with 16 output rows per wavefront the
// benchmark graph needs 56 M row fetches of 2.4 KB (135 GB per hop).  The number of
// distinct (group, column) pairs falls steadily with the group size (merged(R) on that
// graph: R = 16 56.0 M, 128 24.3 M, 512 12.1 M), but a wavefront cannot hold more than
// 16 rows of accumulators, and sharing fetched rows between the wavefronts of a
// workgroup through LDS was tried with vector adds and lost to the mask arithmetic and
// to barriers that wait for the busiest wavefront.
//
// What this kernel does instead.  A workgroup owns a group of 512 degree-ranked rows
// (16 sub-tiles of 32 rows) and one 128-column slice of the features.  It walks the
// group's merged column list in stages of 128 columns; a stage's rows of X arrive in LDS
// by LDS-DMA (global_load_lds_dwordx4) while the stage before it is being multiplied.  Every sub-tile has, per stage,
// its own packed list of the columns it actually uses, cut into steps of 16; a step is
// one 32x16 0/1 pattern tile times a 16x32 tile of X per 32 features:
// v_mfma_f32_32x32x16_f16, fp32 accumulation.  All wavefronts do the same kind of work
// (no masks, no divergence), so a barrier per stage costs little, and the arithmetic a
// pattern of zeros wastes is cheap at the matrix cores' rate.
//
// fp32 in, fp32 out, fp32 accumulation.  The matrix cores take 16-bit inputs, so X is
// split beforehand (k_tg_split) into two fp16 planes, x * scale = hi + lo / 2048 with
// scale a power of two that brings the largest magnitude just below 2^15; the pattern
// tile is 1.0 for the hi plane and 2^-11 for the lo plane, so both land in the same
// accumulator.  hi + lo / 2048 reproduces an fp32 input to 2^-24 relative, or to 2^-50 of
// the largest magnitude in X where that is larger (elements more than 2^26 below the
// largest); products with 0 / 1 / 2^-11 are exact and the sums are fp32, so the result
// differs from the vector kernels' by summation order and at most one extra rounding of
// each input.  The split pass checks every element, counts the ones that keep fewer than
// 17 of their bits (more than 2^33 below the largest) and raises a flag when X is not
// finite or holds more than one such element in 2^14; the SpMM kernel then does
// nothing and the fall-back launched behind it (the plain CSR kernel on the original
// fp32 operand, guarded by the same flag) computes Y.
//
// Reference path served: the adj.mm of utils.py:94 with its two row scalings
// (utils.py:93, 95) for adjacencies whose off-diagonal entries all carry one weight.
#include <atomic>
#include <cstddef>
#include <cstring>
#include <type_traits>

#include <rocprim/rocprim.hpp>

#include "common.h"

// the DMA helpers below set M0 inside their inline assembly and say so in the clobber list
#pragma clang diagnostic ignored "-Winline-asm"

namespace {
using namespace sgnn;

constexpr int kSub = 32;                         // rows of a sub-tile: M of one MFMA
constexpr int kSubs = 16;                        // sub-tiles of a group
constexpr int kRows = kSub * kSubs;              // 512 rows of a group
constexpr int kStageCols = 128;                  // merged columns per stage
constexpr int kSlotBits = 7;                     // log2(kStageCols)
constexpr int kStepCols = 16;                    // K of one MFMA
constexpr int kSlice = 128;                      // features of a slice (4 MFMA N-tiles)
constexpr int kWaves = 8;
#ifndef SGNN_TG_ISSUERS
#define SGNN_TG_ISSUERS 8
#endif
constexpr int kIssuers = SGNN_TG_ISSUERS;        // wavefronts (the first of the workgroup) that issue a stage's DMA instructions
constexpr int kIssuerRows = kStageCols / kIssuers;
constexpr int kColRegs = kIssuerRows / 4, kRowInstr = 2 * kColRegs;
constexpr int kEarlyRows = 2;                    // row DMA instructions of a wavefront issued ahead of the stage's first tile reads (4 and 8 measured 1 % slower)
constexpr int kRing = 2;                         // stages resident in LDS: one being read, one arriving
constexpr int kRowBytes = 512;                   // one node's slice: hi 256 B | lo 256 B
constexpr int kStageRowsBytes = kStageCols * kRowBytes;
constexpr int kMaxStageSteps = kSubs * (kStageCols / kStepCols);   // 128
constexpr int kSlotsBytes = kMaxStageSteps * 32;
constexpr int kBitsBytes = kMaxStageSteps * 64;
constexpr int kHdrBytes = 512;                   // first steps of this stage's and the next stages' sub-tiles
constexpr int kColsBytes = 1024;                 // column ids of the next two stages
constexpr int kStageBytes = kStageRowsBytes + kSlotsBytes + kBitsBytes + kHdrBytes + kColsBytes;   // 79360 = 155 * 512
constexpr int kTableBytes = 1024;                // LDS offset 0: pattern table (16 entries of 8 bytes), the work ticket at 496, at 512 the 128 per-feature maxima of a unit's written operand; the stage buffers follow
constexpr int kTicketOff = 496, kAmaxOff = 512;
constexpr int kStageOutBytes = 16384;            // write-out staging per wavefront: 32 rows x 512 bytes
constexpr int kRowTabBytes = 768;                // write-out: a wavefront's 64 row ids, row scales and diagonal factors, behind the staging areas
constexpr int kLdsBytes = kTableBytes + kRing * kStageBytes;
constexpr int kPadSteps = 2 * kMaxStageSteps;                   // readable slack behind the step arrays (a stage's DMA over-reads)
static_assert((1 << kSlotBits) == kStageCols && kStageCols == 16 * kWaves && (kIssuers == 8 || kIssuers == 4), "a wavefront loads 16 (or 32) rows of a stage");
static_assert(kStageBytes % 512 == 0, "the slot swizzle needs 512-byte aligned stage buffers");
static_assert(kTableBytes + kWaves * (kStageOutBytes + kRowTabBytes) <= kLdsBytes, "the write-out staging must fit the ring");

std::atomic<int> g_seg_stages{0};     // stages per work item, longer merged lists are cut; 0: chosen from the plan's size (sgnn_tune_set("tg_seg", v))
std::atomic<int> g_no_sample{0};      // sgnn_tune_set("tg_no_sample", 1): the split pass always measures the maxima of all rows first
std::atomic<int> g_split_blocks{0}, g_split_threads{0};   // launch shape of the split pass (sgnn_tune_set("tg_split_blocks" / "tg_split_threads", v)); 0 = built-in
std::atomic<int> g_combine_gy{0};     // workgroups per split group of the combine pass (sgnn_tune_set("tg_combine_gy", v)); 0 = the built-in choice
std::atomic<int> g_order_cols{0};     // work item order (sgnn_tune_set("tg_order", v)): 0 = chosen by the graph; 1 = longest first; < 0 = neighbours together (length buckets, then column block, then group); > 1 = by column blocks of this width
std::atomic<int> g_tg_mode{0};         // diagnostics (sgnn_tune_set("tg_mode", m)): 1 = no multiplication, 2 = no row DMA, 32 = no rows written, 64 = rows fetched from the first 4096 nodes only (results are then wrong); 128 = one workgroup per unit, 256 / 1024 = one work counter per XCD / one for the launch whatever the plan says

struct TgState {          // device words that carry an operand's scales and its fall-back flag to the SpMM that reads it
  uint32_t rsv0;
  uint32_t flag;          // != 0: the operand cannot go through the fp16 planes (1: an element, 2: too many tiny ones, 4: plan mismatch, 8: inherited from the hop before)
  uint32_t redo;          // the split pass met an element its sampled bound cannot hold: the exact passes behind it run
  uint32_t sampled;       // the maxima are those of a sample of the rows (the bounds carry a margin): not a bound for the next operand
  uint32_t tiny;          // nonzero elements that keep fewer than 17 of their bits in the planes
  uint32_t done;          // workgroups of the running SpMM launch that have left the work loop (the last one zeroes the work counters)
  uint32_t tiny_limit;    // more tiny elements than this: fall back
  uint32_t queue;         // next work index of the running SpMM launch (this word and the eight behind it are zero between launches)
  uint32_t xcd_queue[8];  // the same, one per XCD (see k_tg_spmm)
  // [Fpad] per feature, the bit pattern of its largest |element| as measured while the operand
  // was written, followed by [Fpad] the bound its scale was derived from.  An operand split
  // from fp32 is measured first, so the two agree; one written by the epilogue of the hop
  // before uses what was known when that hop started (bound_of) and is measured on the way out.
  uint32_t amax[1];
};
constexpr size_t kStateHead = 64;
static_assert(offsetof(TgState, amax) == kStateHead, "state header is sixteen words");
__device__ __forceinline__ uint32_t* bound_of(TgState* s, int Fpad) { return s->amax + Fpad; }
__device__ __forceinline__ const uint32_t* bound_of(const TgState* s, int Fpad) { return s->amax + Fpad; }
// behind the two arrays: kAmaxBlocks x Fpad words in which the workgroups of the measuring pass
// leave their maxima (one atomic per feature and workgroup on the same few hundred words took a
// third of that pass)
constexpr int kAmaxBlocks = 512;
__device__ __forceinline__ uint32_t* scratch_of(TgState* s, int Fpad) { return s->amax + 2 * Fpad; }
inline size_t tg_state_live_bytes(int64_t Fpad) { return kStateHead + size_t(2 * Fpad) * sizeof(uint32_t); }

// ---- plan ------------------------------------------------------------------------
struct TgPlan {
  int64_t* counts;        // [16] merged, stages, steps, items, partial slots, long groups, stages per item, rows, neighbour order (0 / 1)
  int32_t* grows;         // [G * 512] row of (group, slot), -1 = none
  int32_t* gstage0;       // [G + 1] first stage of every group
  int32_t* stage_col;     // [stages * 64] columns of a stage (tail of a group's last stage: its first column again)
  int32_t* step0;         // [stages * 16 + 1 (+ slack)] first step of (stage, sub-tile)
  uint16_t* step_slots;   // [(steps + slack) * 16] per column of a step: byte offset of its row in the stage's LDS image
                          // ((slot >> 2) * 2048 + (slot & 3) * 64: blocks of 4 rows, stored segment-major)
  uint16_t* step_bits;    // [(steps + slack) * 32] per row of the sub-tile: which of the 16 columns it has
  int32_t *item_group, *item_s0, *item_s1, *item_out;   // [items] in launch order
  int32_t *long_group, *long_slot, *long_nseg;          // [long groups]
  int64_t G;
  size_t total;
};

TgPlan carve_tg_plan(void* base, int64_t n_rows, int64_t stages, int64_t steps, int64_t items, int64_t n_long) {
  TgPlan p{};
  Carver c(base);
  p.G = (n_rows + kRows - 1) / kRows;
  p.counts = c.take<int64_t>(16);
  p.grows = c.take<int32_t>(size_t(p.G) * kRows);
  p.gstage0 = c.take<int32_t>(size_t(p.G) + 1);
  p.stage_col = c.take<int32_t>(size_t(stages) * kStageCols + 3 * kStageCols);   // a stage's DMA also brings the next two stages' ids
  p.step0 = c.take<int32_t>(size_t(stages) * kSubs + 1 + 128);   // the header DMA of the last stage reads 64 entries
  p.step_slots = c.take<uint16_t>(size_t(steps + kPadSteps) * 16);
  p.step_bits = c.take<uint16_t>(size_t(steps + kPadSteps) * 32);
  p.item_group = c.take<int32_t>(size_t(items) + 1);
  p.item_s0 = c.take<int32_t>(size_t(items) + 1);
  p.item_s1 = c.take<int32_t>(size_t(items) + 1);
  p.item_out = c.take<int32_t>(size_t(items) + 1);
  p.long_group = c.take<int32_t>(size_t(n_long) + 1);
  p.long_slot = c.take<int32_t>(size_t(n_long) + 1);
  p.long_nseg = c.take<int32_t>(size_t(n_long) + 1);
  p.total = c.off;
  return p;
}

struct TgBuild {
  int64_t* counts;
  uint32_t *dkey_a, *dkey_b, *drow_a, *drow_b;
  int32_t* rank;
  uint64_t *keys_a, *keys_b;
  uint32_t *head, *pos;
  int32_t* mcol;            // [nnz] merged columns, all groups back to back
  int32_t* gptr;            // [G + 1] merged-list offsets
  uint32_t *gns, *gstage0;  // [G], [G + 1] stages per group and their running sum
  uint32_t *runcount, *runclass, *runfirst, *nsteps, *step0;   // [cap_runs (+1)]
  uint32_t *nseg, *first, *lseg, *slot0, *lflag, *lidx;   // [G]
  uint64_t *ikey_a, *ikey_b;   // [cap_items]
  int32_t* itmp;               // [4 * cap_items]
  void* prim;
  size_t prim_bytes;
  int64_t G, cap_stages, cap_runs, cap_items;
  size_t total;
};

TgBuild carve_tg_build(void* base, int64_t n_rows, int64_t nnz) {
  TgBuild w{};
  Carver c(base);
  w.G = (n_rows + kRows - 1) / kRows;
  w.cap_stages = nnz / kStageCols + w.G + 1;
  w.cap_runs = w.cap_stages * kSubs;
  w.cap_items = w.G + w.cap_stages / 8 + 1;   // a group is cut only into pieces of at least 8 stages
  w.counts = c.take<int64_t>(16);
  w.dkey_a = c.take<uint32_t>(n_rows);
  w.dkey_b = c.take<uint32_t>(n_rows);
  w.drow_a = c.take<uint32_t>(n_rows);
  w.drow_b = c.take<uint32_t>(n_rows);
  w.rank = c.take<int32_t>(n_rows);
  w.keys_a = c.take<uint64_t>(nnz);
  w.keys_b = c.take<uint64_t>(nnz);
  w.head = c.take<uint32_t>(nnz);
  w.pos = c.take<uint32_t>(nnz);
  w.mcol = c.take<int32_t>(nnz);
  w.gptr = c.take<int32_t>(w.G + 1);
  w.gns = c.take<uint32_t>(w.G + 1);
  w.gstage0 = c.take<uint32_t>(w.G + 1);
  w.runcount = c.take<uint32_t>(w.cap_runs + 1);
  w.runclass = c.take<uint32_t>(w.cap_runs + 1);
  w.runfirst = c.take<uint32_t>(w.cap_runs + 1);
  w.nsteps = c.take<uint32_t>(w.cap_runs + 1);
  w.step0 = c.take<uint32_t>(w.cap_runs + 1);
  w.nseg = c.take<uint32_t>(w.G);
  w.first = c.take<uint32_t>(w.G);
  w.lseg = c.take<uint32_t>(w.G);
  w.slot0 = c.take<uint32_t>(w.G);
  w.lflag = c.take<uint32_t>(w.G);
  w.lidx = c.take<uint32_t>(w.G);
  w.ikey_a = c.take<uint64_t>(w.cap_items);
  w.ikey_b = c.take<uint64_t>(w.cap_items);
  w.itmp = c.take<int32_t>(4 * w.cap_items);
  size_t s1 = 0, s2 = 0, s3 = 0;
  uint64_t* k64 = nullptr;
  uint32_t* k32 = nullptr;
  (void)rocprim::radix_sort_keys(nullptr, s1, k64, k64, size_t(nnz > w.cap_items ? nnz : w.cap_items), 0u, 64u,
                                 hipStream_t(0));
  (void)rocprim::radix_sort_pairs(nullptr, s2, k32, k32, k32, k32, size_t(n_rows), 0u, 32u, hipStream_t(0));
  const size_t scan_n = size_t(nnz > w.cap_runs + 1 ? nnz : w.cap_runs + 1);
  (void)rocprim::exclusive_scan(nullptr, s3, k32, k32, 0u, scan_n, rocprim::plus<uint32_t>(), hipStream_t(0));
  w.prim_bytes = align_up(s1 > s2 ? (s1 > s3 ? s1 : s3) : (s2 > s3 ? s2 : s3));
  w.prim = c.take<char>(w.prim_bytes);
  w.total = c.off;
  return w;
}

// ---- plan construction, phase 0: sizes --------------------------------------------
__global__ void k_tg_deg_keys(int64_t n, const int32_t* __restrict__ rowptr, uint32_t* __restrict__ key,
                              uint32_t* __restrict__ row) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  key[i] = 0xFFFFFFFFu - uint32_t(rowptr[i + 1] - rowptr[i]);   // ascending key = descending degree
  row[i] = uint32_t(i);
}

__global__ void k_tg_rank(int64_t n, const uint32_t* __restrict__ order, int32_t* __restrict__ rank) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i < n) rank[order[i]] = int32_t(i);
}

// key of an entry = (group of its row, column, slot of its row in the group: 9 bits)
__global__ void k_tg_entry_keys(int64_t n, int64_t nnz, const int32_t* __restrict__ rowptr,
                                const int32_t* __restrict__ colidx, const int32_t* __restrict__ rank, int cbits,
                                uint64_t* __restrict__ keys) {
  int64_t e = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (e >= nnz) return;
  int64_t lo = 0, hi = n;   // last row whose rowptr <= e
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (int64_t(rowptr[mid]) <= e) lo = mid; else hi = mid;
  }
  const uint32_t rk = uint32_t(rank[lo]);
  keys[e] = (((uint64_t(rk / kRows) << cbits) | uint64_t(colidx[e])) << 9) | uint64_t(rk % kRows);
}

__global__ void k_tg_heads(int64_t nnz, const uint64_t* __restrict__ keys, int shift, uint32_t* __restrict__ head) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= nnz) return;
  head[i] = (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) ? 1u : 0u;
}

// merged columns back to back; group offsets (groups without entries get empty lists)
__global__ void k_tg_merged(int64_t G, int64_t nnz, const uint64_t* __restrict__ keys,
                            const uint32_t* __restrict__ head, const uint32_t* __restrict__ pos, int cbits,
                            int32_t* __restrict__ mcol, int32_t* __restrict__ gptr, int64_t* __restrict__ counts) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= nnz) return;
  const int64_t merged = int64_t(pos[nnz - 1]) + head[nnz - 1];
  if (head[i]) {
    const uint64_t gc = keys[i] >> 9;
    mcol[pos[i]] = int32_t(gc & ((uint64_t(1) << cbits) - 1));
    const int64_t g = int64_t(gc >> cbits);
    const int64_t gprev = i == 0 ? -1 : int64_t(keys[i - 1] >> (9 + cbits));
    for (int64_t q = gprev + 1; q <= g; ++q) gptr[q] = int32_t(pos[i]);
  }
  if (i == nnz - 1) {
    const int64_t glast = int64_t(keys[i] >> (9 + cbits));
    for (int64_t q = glast + 1; q <= G; ++q) gptr[q] = int32_t(merged);
    counts[0] = merged;
  }
}

__global__ void k_tg_group_stages(int64_t G, const int32_t* __restrict__ gptr, uint32_t* __restrict__ gns) {
  int64_t g = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (g > G) return;
  gns[g] = g < G ? uint32_t((gptr[g + 1] - gptr[g] + kStageCols - 1) / kStageCols) : 0u;
}

// second key of an entry = (stage, sub-tile of its row, slot of its column in the stage, row in the sub-tile)
__global__ void k_tg_step_keys(int64_t nnz, const uint64_t* __restrict__ keys1, const uint32_t* __restrict__ head,
                               const uint32_t* __restrict__ pos, const int32_t* __restrict__ gptr, const uint32_t* __restrict__ gstage0, int cbits,
                               uint64_t* __restrict__ keys2) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= nnz) return;
  const uint64_t k = keys1[i];
  const int64_t g = int64_t(k >> (9 + cbits));
  const uint32_t r9 = uint32_t(k & 511);
  // pos[] is the exclusive scan of the heads: an entry that is not a head sits in merged column pos[i] - 1
  const uint32_t m = (head[i] ? pos[i] : pos[i] - 1u) - uint32_t(gptr[g]);
  const uint64_t stage = uint64_t(gstage0[g]) + m / kStageCols;
  // inside a run the columns are ordered by bank class (slot & 3) first: k_tg_fill_steps deals
  // them out round-robin over the classes from that order
  const uint32_t slot = m % kStageCols;
  const uint32_t cls_major = ((slot & 3u) << (kSlotBits - 2)) | (slot >> 2);
  keys2[i] = ((stage * kSubs + (r9 >> 5)) << (5 + kSlotBits)) | (uint64_t(cls_major) << 5) | uint64_t(r9 & 31);
}

// columns of every (stage, sub-tile) run, and where the run's first (sub-tile, column) pair sits
__global__ void k_tg_run_counts(int64_t nnz, const uint64_t* __restrict__ keys2, const uint32_t* __restrict__ head,
                                const uint32_t* __restrict__ pair, uint32_t* __restrict__ runcount,
                                uint32_t* __restrict__ runclass, uint32_t* __restrict__ runfirst) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= nnz || !head[i]) return;
  const uint64_t run = keys2[i] >> (5 + kSlotBits);
  atomicAdd(&runcount[run], 1u);
  atomicAdd(&runclass[run], 1u << (8 * ((uint32_t(keys2[i]) >> (5 + kSlotBits - 2)) & 3u)));   // four counts of at most 32
  if (i == 0 || (keys2[i - 1] >> (5 + kSlotBits)) != run) runfirst[run] = pair[i];
}

__global__ void k_tg_run_steps(int64_t runs_cap, const int64_t* __restrict__ counts, const uint32_t* __restrict__ gstage0,
                               int64_t G, const uint32_t* __restrict__ runcount, uint32_t* __restrict__ nsteps) {
  int64_t r = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (r > runs_cap) return;
  const int64_t runs = int64_t(gstage0[G]) * kSubs;
  nsteps[r] = r < runs ? (runcount[r] + kStepCols - 1) / kStepCols : 0u;
  (void)counts;
}

// work items: a group's stages in pieces of about `seg` stages
// Stages per work item.  Long items mean few partial rows to write and combine, short ones a
// launch that 256 CUs finish together: 224 stages where the plan is large enough to make some
// 450 items of that length (Reddit shape: 5.45-5.6 ms per hop from 192 to 288, 5.66 at 128,
// 5.86 at 384), proportionally fewer for small plans (a rank's block of a partitioned graph),
// never under 32.
constexpr uint32_t kSegStages = 224, kSegStagesMin = 32, kItemsWanted = 448;

__device__ __forceinline__ uint32_t tg_seg_for(uint32_t total_stages, int override_seg) {
  if (override_seg > 0) return uint32_t(override_seg);
  const uint32_t s = total_stages / kItemsWanted;
  return s < kSegStagesMin ? kSegStagesMin : (s > kSegStages ? kSegStages : s);
}

__global__ void k_tg_nseg(int64_t G, int override_seg, const uint32_t* __restrict__ gns,
                          const uint32_t* __restrict__ gstage0, uint32_t* __restrict__ nseg,
                          uint32_t* __restrict__ lseg, uint32_t* __restrict__ lflag) {
  int64_t g = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (g >= G) return;
  const uint32_t seg = tg_seg_for(gstage0[G], override_seg);
  const uint32_t ns = gns[g];
  const uint32_t pieces = ns > seg + seg / 4 ? (ns + seg - 1) / seg : 1u;   // an empty group still writes its zero rows
  nseg[g] = pieces;
  lseg[g] = pieces > 1 ? pieces : 0u;
  lflag[g] = pieces > 1 ? 1u : 0u;
}

__global__ void k_tg_totals(int64_t G, const uint32_t* __restrict__ gstage0, const uint32_t* __restrict__ step0,
                            const uint32_t* __restrict__ nseg, const uint32_t* __restrict__ first,
                            const uint32_t* __restrict__ lseg, const uint32_t* __restrict__ slot0,
                            const uint32_t* __restrict__ lflag, const uint32_t* __restrict__ lidx, int seg,
                            int64_t n_rows, int64_t* __restrict__ counts) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int64_t stages = gstage0[G];
  counts[1] = stages;
  counts[2] = step0[stages * kSubs];
  counts[3] = G > 0 ? int64_t(first[G - 1]) + nseg[G - 1] : 0;
  counts[4] = G > 0 ? int64_t(slot0[G - 1]) + lseg[G - 1] : 0;
  counts[5] = G > 0 ? int64_t(lidx[G - 1]) + lflag[G - 1] : 0;
  counts[6] = tg_seg_for(uint32_t(stages), seg);
  counts[7] = n_rows;
}

// ---- plan construction, phase 1: contents -----------------------------------------
__global__ void k_tg_grows(int64_t n, int64_t slots, const uint32_t* __restrict__ order, int32_t* __restrict__ grows) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i < slots) grows[i] = i < n ? int32_t(order[i]) : -1;
}

__global__ void k_tg_copy_u32(int64_t count, const uint32_t* __restrict__ src, int32_t* __restrict__ dst,
                              int64_t fill_to) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i < count) dst[i] = int32_t(src[i]);
  else if (i < fill_to) dst[i] = int32_t(src[count - 1]);   // slack behind the last entry repeats it
}

__global__ void k_tg_stage_cols(int64_t G, int64_t stages, const int32_t* __restrict__ gptr,
                                const uint32_t* __restrict__ gstage0, const int32_t* __restrict__ mcol,
                                int32_t* __restrict__ stage_col) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= stages * kStageCols) return;
  const int64_t stage = i / kStageCols;
  const int slot = int(i % kStageCols);
  int64_t lo = 0, hi = G;   // last group whose first stage <= stage
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) >> 1;
    if (int64_t(gstage0[mid]) <= stage) lo = mid; else hi = mid;
  }
  // groups without stages share their first stage with the next group: skip to the one that owns it
  while (lo + 1 < G && gstage0[lo + 1] == gstage0[lo] && int64_t(gstage0[lo]) == stage) ++lo;
  const int64_t m0 = (stage - gstage0[lo]) * kStageCols;
  const int64_t len = gptr[lo + 1] - gptr[lo];
  const int64_t m = m0 + slot < len ? m0 + slot : m0;
  stage_col[i] = mcol[gptr[lo] + m];
}

__global__ void k_tg_fill_steps(int64_t nnz, const uint64_t* __restrict__ keys2, const uint32_t* __restrict__ head,
                                const uint32_t* __restrict__ pair, const uint32_t* __restrict__ runfirst,
                                const uint32_t* __restrict__ runclass,
                                const uint32_t* __restrict__ step0, uint16_t* __restrict__ step_slots,
                                uint32_t* __restrict__ step_bits32) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= nnz) return;
  const uint64_t k = keys2[i];
  const uint64_t run = k >> (5 + kSlotBits);
  const uint32_t cm = uint32_t(k >> 5) & uint32_t(kStageCols - 1), r5 = uint32_t(k) & 31u;
  const uint32_t cls = cm >> (kSlotBits - 2), slot = ((cm & ((1u << (kSlotBits - 2)) - 1u)) << 2) | cls;
  // pair[] is the exclusive scan of the heads: an entry that is not a head belongs to pair[i] - 1
  const uint32_t pi = head[i] ? pair[i] : pair[i] - 1u;
  // Position of the column in its run.  The run is sorted by class, then slot; dealing the
  // classes out round-robin (rank 0 of every class, rank 1 of every class, ...) makes the four
  // rows a transposed read takes from LDS in one pass differ in slot & 3, i.e. sit in
  // different bank quarters, wherever the classes last.
  const uint32_t qpos = pi - runfirst[run], packed = runclass[run];
  uint32_t before = 0;
  for (uint32_t c = 0; c < cls; ++c) before += (packed >> (8 * c)) & 255u;
  const uint32_t rank = qpos - before;
  uint32_t j = 0;
  for (uint32_t c = 0; c < 4; ++c) {
    const uint32_t nc = (packed >> (8 * c)) & 255u;
    j += nc < rank ? nc : rank;
    if (c < cls && nc > rank) ++j;
  }
  const int64_t step = int64_t(step0[run]) + j / kStepCols;
  const uint32_t kpos = j % kStepCols;
  // stored so that the two offsets one lane of the kernel needs (columns 8 h + q and 8 h + 4 + q) share a dword
  const uint32_t spos = 2u * (4u * (kpos >> 3) + (kpos & 3u)) + ((kpos >> 2) & 1u);
  if (head[i]) step_slots[step * 16 + spos] = uint16_t((slot >> 2) * (4 * kRowBytes) + ((slot & 3u) << 6));
  atomicOr(&step_bits32[step * 16 + (r5 >> 1)], 1u << (kpos + 16u * (r5 & 1u)));
}

__global__ void k_tg_items(int64_t G, const uint32_t* __restrict__ gns, const uint32_t* __restrict__ gstage0,
                           const uint32_t* __restrict__ nseg, const uint32_t* __restrict__ first,
                           const uint32_t* __restrict__ slot0, const uint32_t* __restrict__ lidx,
                           int32_t* __restrict__ tmp, int64_t cap, int32_t* __restrict__ long_group,
                           int32_t* __restrict__ long_slot, int32_t* __restrict__ long_nseg) {
  int64_t g = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (g >= G) return;
  const uint32_t pieces = nseg[g], ns = gns[g];
  for (uint32_t k = 0; k < pieces; ++k) {
    const uint32_t it = first[g] + k;
    tmp[it] = int32_t(g);
    tmp[cap + it] = int32_t(gstage0[g] + uint64_t(ns) * k / pieces);
    tmp[2 * cap + it] = int32_t(gstage0[g] + uint64_t(ns) * (k + 1) / pieces);
    tmp[3 * cap + it] = pieces > 1 ? int32_t(slot0[g] + k) : -1;
  }
  if (pieces > 1) {
    long_group[lidx[g]] = int32_t(g);
    long_slot[lidx[g]] = int32_t(slot0[g]);
    long_nseg[lidx[g]] = int32_t(pieces);
  }
}

// Launch order.  Two orders, chosen by how much the graph's rows share (tools/xcd_probe.py,
// profiles/r03_tilegroup_item_order_and_xcd_queues.txt):
// * longest item first, ties by item number (the pieces of one group's list follow each other, so
//   CUs that start together walk different stretches of the column space), one work counter for
//   the launch.  The tail of the launch is made of short items (against ordering by the column
//   block an item starts in, which is what the row-group kernel's L2 reuse wanted: 5.5-5.65 ms
//   per hop against 5.9 on the Reddit shape).
// * neighbours together: lengths compared in buckets of 32 stages and, inside a bucket, items by
//   the 4096-column block they start in, then by group -- neighbours in this order are the same
//   stretch of the lists of adjacent, equally dense groups, which fetch nearly the same rows of X
//   at the same pace -- and the kernel hands runs of 32 neighbours to the CUs of one XCD (one work
//   counter per XCD), whose L2 then serves most of a row's fetches after the first.
// On the Reddit shape the second cuts the bytes that cross the L2's fabric side by 40 % (fetch
// counter 10.87 -> 6.56 M KiB per launch, L2 hits 98.7 -> 167.7 M) and is 4 % SLOWER (main kernel
// 4.88 against 4.69 ms; the order alone, with one counter: 4.77 ms, 26 % fewer bytes): that
// kernel is not bound by what crosses the fabric, and CUs that ask for the same rows in the same
// microsecond queue at one L2 channel.  On the sparse shapes, whose kernel time IS the fabric's
// gather rate, it wins: ogbn-products 3.76 against 4.02 ms, R-MAT 10 M 10.7 against 11.7 ms.
// Hence the rule: neighbours together where a fetched row serves fewer than 4 stored entries on
// average (merged * 4 > nnz: products 2.35, R-MAT 10 M 1.6; the Reddit shape has 9.5).
__global__ void k_tg_item_keys(int64_t items, const int32_t* __restrict__ tmp, int64_t cap,
                               const int32_t* __restrict__ stage_col, int order_cols, int64_t nnz,
                               int64_t* __restrict__ plan_counts, uint64_t* __restrict__ keys) {
  int64_t it = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (it >= items) return;
  const int64_t s0 = tmp[cap + it];
  const int64_t len = tmp[2 * cap + it] - int32_t(s0);
  // order_cols 0: by the graph -- neighbours together where a fetched row serves fewer than 4 stored
  // entries on average (see above); 1: longest first; < 0: neighbours together; > 1: column blocks
  const bool neighbours = order_cols < 0 || (order_cols == 0 && plan_counts[0] * 4 > nnz);
  if (it == 0) plan_counts[8] = neighbours ? 1 : 0;
  if (order_cols > 1) {
    keys[it] = (uint64_t(stage_col[s0 * kStageCols] / order_cols) << 32) | uint64_t(it);
    return;
  }
  if (!neighbours) {
    keys[it] = (uint64_t(0x7fffffff - len) << 32) | uint64_t(it);
    return;
  }
  const uint64_t lb = uint64_t(0xFFFF) - uint64_t(len / 32 < 0xFFFF ? len / 32 : 0xFFFF);
  const uint64_t cb = len > 0 ? uint64_t(stage_col[s0 * kStageCols]) >> 12 : 0;
  keys[it] = (lb << 48) | ((cb < 0xFFFF ? cb : 0xFFFF) << 32) | uint64_t(it);
}

__global__ void k_tg_item_place(int64_t items, const uint64_t* __restrict__ keys, const int32_t* __restrict__ tmp,
                                int64_t cap, int32_t* __restrict__ a, int32_t* __restrict__ b,
                                int32_t* __restrict__ c, int32_t* __restrict__ d) {
  int64_t it = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (it >= items) return;
  const int64_t src = int64_t(keys[it] & 0xffffffffu);
  a[it] = tmp[src];
  b[it] = tmp[cap + src];
  c[it] = tmp[2 * cap + src];
  d[it] = tmp[3 * cap + src];
}

// ---- fp16 planes --------------------------------------------------------------------
// One power-of-two scale per FEATURE: columns never mix in A . X (a feature's accumulators only
// ever see that feature), so each feature column is brought into the planes' range on its own
// and the accumulator is scaled back per feature in the epilogue.  Feature matrices whose
// columns differ by many orders of magnitude stay accurate column by column.

// Per-feature max |scale_row[i] * X[i, f]| (bit patterns; NaN and inf compare above every finite
// value) into state->amax.  A workgroup's threads lie along the features, V = 1 or 2 consecutive
// ones per thread; rows are walked four at a time so that four loads are in flight per lane; a
// narrow row leaves room for several row quadruples side by side (`subs`), folded through LDS.
__global__ __launch_bounds__(1024) void k_tg_amax(int64_t n, int64_t F, int Fpad, const float* __restrict__ scale_row,
                                                 const float* __restrict__ X, int64_t ldx, TgState* __restrict__ state,
                                                 int64_t rstep, int only_redo) {
  // n rows, every rstep-th of the operand (a sample gives the split pass its provisional scales);
  // only_redo: the exact pass behind a split that met an element beyond them -- nothing to do otherwise
  if (only_redo && state->redo == 0) return;
  // the first pass over an operand also clears the state's header (flag, counters, work counters:
  // no other workgroup of this kernel touches it; the kernels behind it in the stream fill it in)
  if (!only_redo && blockIdx.x == 0 && threadIdx.x < kStateHead / 4) reinterpret_cast<uint32_t*>(state)[threadIdx.x] = 0u;
  __shared__ uint32_t smax[2048];
  const bool pairs = (ldx % 2 == 0) && (reinterpret_cast<uintptr_t>(X) % 8 == 0);
  const int V = pairs ? 2 : 1;
  const int per_row = int((F + V - 1) / V);
  const int lanes_row = per_row >= int(blockDim.x) ? int(blockDim.x) : (per_row + kWave - 1) / kWave * kWave;
  const int subs = int(blockDim.x) / lanes_row, sub = int(threadIdx.x) / lanes_row, tin = int(threadIdx.x) % lanes_row;
  for (int64_t fbase = 0; fbase < F; fbase += int64_t(V) * lanes_row) {
    const int64_t f = fbase + int64_t(V) * tin;
    uint32_t m0 = 0, m1 = 0;
    if (sub < subs && f < F) {
      for (int64_t r0 = 4 * (int64_t(blockIdx.x) * subs + sub); r0 < n; r0 += 4 * int64_t(gridDim.x) * subs) {
        float sr[4];
        float2 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          sr[i] = (r0 + i < n) ? (scale_row ? scale_row[(r0 + i) * rstep] : 1.0f) : 0.0f;
          v[i] = float2{0.0f, 0.0f};
          if (r0 + i < n) {
            const float* row = X + (r0 + i) * rstep * ldx;
            if (pairs && f + 1 < F) v[i] = *reinterpret_cast<const float2*>(row + f);
            else v[i].x = row[f];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t b0 = __builtin_bit_cast(uint32_t, v[i].x * sr[i]) & 0x7FFFFFFFu;
          const uint32_t b1 = __builtin_bit_cast(uint32_t, v[i].y * sr[i]) & 0x7FFFFFFFu;
          m0 = b0 > m0 ? b0 : m0;
          m1 = b1 > m1 ? b1 : m1;
        }
      }
    }
    if (subs > 1) {          // workgroup-uniform
      for (int i = threadIdx.x; i < 2 * lanes_row; i += blockDim.x) smax[i] = 0;
      __syncthreads();
      if (sub < subs && f < F) {
        if (m0) atomicMax(&smax[2 * tin], m0);
        if (m1) atomicMax(&smax[2 * tin + 1], m1);
      }
      __syncthreads();
      m0 = smax[2 * tin];
      m1 = smax[2 * tin + 1];
      __syncthreads();
    }
    if (sub == 0 && f < F) {
      uint32_t* mine = scratch_of(state, Fpad) + int64_t(blockIdx.x) * Fpad;
      mine[f] = m0;
      if (V == 2 && f + 1 < F) mine[f + 1] = m1;
    }
  }
}

// the workgroups' maxima folded per feature: a workgroup takes 64 features, its 16 wavefronts
// every 16th workgroup's row of the scratch (lanes along the features, several loads in flight)
__global__ __launch_bounds__(1024) void k_tg_amax_fold(int F, int Fpad, int blocks, TgState* __restrict__ state, int only_redo) {
  if (only_redo && state->redo == 0) return;
  __shared__ uint32_t part[16][64];
  const int fx = threadIdx.x & 63, by = threadIdx.x >> 6;
  const int f = blockIdx.x * 64 + fx;
  uint32_t m = 0;
  if (f < F) {
    const uint32_t* p = scratch_of(state, Fpad) + f;
    uint32_t m1 = 0, m2 = 0, m3 = 0;
    int b = by;
    for (; b + 48 < blocks; b += 64) {
      const uint32_t v0 = p[int64_t(b) * Fpad], v1 = p[int64_t(b + 16) * Fpad];
      const uint32_t v2 = p[int64_t(b + 32) * Fpad], v3 = p[int64_t(b + 48) * Fpad];
      m = v0 > m ? v0 : m;
      m1 = v1 > m1 ? v1 : m1;
      m2 = v2 > m2 ? v2 : m2;
      m3 = v3 > m3 ? v3 : m3;
    }
    for (; b < blocks; b += 16) {
      const uint32_t v = p[int64_t(b) * Fpad];
      m = v > m ? v : m;
    }
    m = m1 > m ? m1 : m;
    m2 = m3 > m2 ? m3 : m2;
    m = m2 > m ? m2 : m;
  }
  part[by][fx] = m;
  __syncthreads();
  if (by == 0 && f < Fpad) {   // (the padding features' maxima are zero: the kernels behind read them like any other)
#pragma unroll
    for (int i = 1; i < 16; ++i) m = part[i][fx] > m ? part[i][fx] : m;
    state->amax[f] = f < F ? m : 0u;
  }
}

// the power of two that brings amax into [2^14, 2^15)
__device__ __forceinline__ float tg_scale(uint32_t amax_bits) {
  if (amax_bits == 0 || amax_bits >= 0x7F800000u) return 1.0f;
  int e = int(amax_bits >> 23) - 127;
  if (e < -126) e = -126;
  int s = 14 - e;
  if (s > 126) s = 126;
  return __builtin_bit_cast(float, uint32_t(s + 127) << 23);
}

// One element into its two planes: xs = hi + lo / 2048.  For |xs| in [2^-14, 2^16) the pair
// reproduces xs to 2^-22 relative (fp16 carries 11 bits, the exact remainder times 2048 another
// 11); below, to the spacing of the lo plane's subnormals, 2^-35.  What can go wrong is therefore
// decided by the element's magnitude alone and is checked on the maxima the writers keep anyway:
// a non-finite element or one beyond twice its feature's bound (hi would overflow) raises the
// flag through tg_beyond(); `tiny` counts the nonzero elements so far below the bound
// (|xs| < 2^-19) that fewer than 17 of their bits survive.
__device__ __forceinline__ void tg_split_one(float xv, float scale, _Float16& h, _Float16& l, uint32_t& tiny) {
  const float xs = xv * scale;
  h = _Float16(xs);
  l = _Float16((xs - float(h)) * 2048.0f);
  tiny += (xv != 0.0f && __builtin_fabsf(xs) < 1.9e-6f) ? 1u : 0u;
}
// a largest magnitude (bit pattern) the planes cannot hold at the scale derived from `bound`:
// non-finite, or at least twice the bound (the scale leaves one binade of headroom below fp16's largest)
__device__ __forceinline__ bool tg_beyond(uint32_t max_bits, uint32_t bound_bits) {
  return max_bits >= 0x7F800000u || max_bits >= bound_bits + (1u << 23);
}

// planes[slice][node] = 128 halfs hi | 128 halfs lo of (scale_row[node] * X[node, slice]) * scale[feature].
// One workgroup walks whole nodes; a thread takes a pair of neighbouring features (one 8-byte
// read where the row allows it, one 4-byte store into each plane: both coalesced).
// slice_stride = bytes between a node's consecutive slices (the planes may belong to a larger
// operand of which these n nodes are a part: a rank's rows of a partitioned matrix).
__global__ __launch_bounds__(1024) void k_tg_split(int64_t n, int64_t F, int64_t n_slices,
                                                  const float* __restrict__ scale_row, const float* __restrict__ X,
                                                  int64_t ldx, char* __restrict__ planes, int64_t slice_stride,
                                                  TgState* __restrict__ state, int only_redo) {
  if (only_redo && state->redo == 0) return;
  const bool pairs = (ldx % 2 == 0) && (reinterpret_cast<uintptr_t>(X) % 8 == 0);
  const int64_t Fpad = n_slices * kSlice;
  uint32_t tiny = 0;
  // An element the planes cannot hold at its feature's scale (twice the bound or more, or not
  // finite; a nonzero one where the bound is zero).  With exact maxima only a non-finite element
  // can be one; with a sample's (state->sampled) an outlier the sample missed is one too, and the
  // operand then goes through the exact passes that stand behind this one.
  bool beyond = false;
  const uint32_t* bounds = bound_of(state, int(Fpad));
  // (a zero bound travels as a negative scale: the element test needs to know, the product does not)
  auto scale_of = [&](int64_t f) { return f < F ? (bounds[f] == 0 ? -1.0f : tg_scale(bounds[f])) : 1.0f; };
  // (one node per pass and set of threads: the 4-byte stores bound this kernel, more loads in
  // flight do not help; a node of one or two slices needs 64 or 128 threads, so the workgroup
  // takes blockDim / that many nodes side by side)
  const int per_node = int(Fpad / 2);
  const int lanes_node = per_node >= int(blockDim.x) ? int(blockDim.x) : per_node;   // a multiple of 64
  const int subs = int(blockDim.x) / lanes_node, sub = int(threadIdx.x) / lanes_node, tin = int(threadIdx.x) % lanes_node;
  // a thread's features are the same for every node: their scales, for the first few chunks, are kept in registers
  constexpr int kKeep = 4;
  float sc[kKeep][2];
#pragma unroll
  for (int c = 0; c < kKeep; ++c) {
    const int64_t f = 2 * int64_t(tin) + 2 * int64_t(lanes_node) * c;
    sc[c][0] = scale_of(f);
    sc[c][1] = scale_of(f + 1);
  }
  for (int64_t node = int64_t(blockIdx.x) * subs + sub; sub < subs && node < n; node += int64_t(gridDim.x) * subs) {
    const float sr = scale_row ? scale_row[node] : 1.0f;
    const float* row = X + node * ldx;
    auto body = [&](int64_t f, float s0, float s1) {
      float x[2] = {0.0f, 0.0f};
      if (f + 1 < F && pairs) {
        const float2 v = *reinterpret_cast<const float2*>(row + f);
        x[0] = v.x;
        x[1] = v.y;
      } else {
        if (f < F) x[0] = row[f];
        if (f + 1 < F) x[1] = row[f + 1];
      }
      _Float16 hi[2], lo[2];
      const float x0 = x[0] * sr, x1 = x[1] * sr;        // x * sr: the reference's scaling (utils.py:93), fp32
      const float a0 = s0 < 0.0f ? 1.0f : s0, a1 = s1 < 0.0f ? 1.0f : s1;
      // (the scale puts the bound into [2^14, 2^15): below 2^15 an element is at most twice its bound and fp16 holds it)
      beyond |= !(__builtin_fabsf(x0 * a0) < 32768.0f) || !(__builtin_fabsf(x1 * a1) < 32768.0f) ||
                (s0 < 0.0f && x0 != 0.0f) || (s1 < 0.0f && x1 != 0.0f);
      tg_split_one(x0, a0, hi[0], lo[0], tiny);
      tg_split_one(x1, a1, hi[1], lo[1], tiny);
      const int64_t slice = f / kSlice;
      const int within = int(f % kSlice);
      char* dst = planes + slice * slice_stride + node * kRowBytes + within * 2;
      *reinterpret_cast<uint32_t*>(dst) =
          uint32_t(__builtin_bit_cast(uint16_t, hi[0])) | (uint32_t(__builtin_bit_cast(uint16_t, hi[1])) << 16);
      *reinterpret_cast<uint32_t*>(dst + kRowBytes / 2) =
          uint32_t(__builtin_bit_cast(uint16_t, lo[0])) | (uint32_t(__builtin_bit_cast(uint16_t, lo[1])) << 16);
    };
    int64_t f = 2 * int64_t(tin);
#pragma unroll
    for (int c = 0; c < kKeep; ++c) {
      if (f < Fpad) body(f, sc[c][0], sc[c][1]);
      f += 2 * int64_t(lanes_node);
    }
    for (; f < Fpad; f += 2 * int64_t(lanes_node))
      body(f, scale_of(f), scale_of(f + 1));
  }
  // one atomic per wavefront
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tiny += uint32_t(__shfl_xor(int(tiny), off));
  const bool any_beyond = __any(beyond);
  if ((threadIdx.x & (kWave - 1)) == 0) {
    if (tiny) atomicAdd(&state->tiny, tiny);
    if (any_beyond) {
      if (state->sampled) atomicOr(&state->redo, 1u);
      else atomicOr(&state->flag, 1u);
    }
  }
}

// A continuous distribution leaves tiny elements where values cross zero: about one in 10^7 of
// a Gaussian operand, one in 10^6 once two hops of averaging have pulled the bulk of a column
// 2^10 below its largest entry (Reddit shape, 1.4 x 10^8 elements: 15 after one hop, 150 after
// two).  Each changes a sum by less than 2^-50 of its feature's largest magnitude.  Data with a
// genuinely wide dynamic range inside a column has them by the percent and goes to the
// fall-back kernel: the limit is one element in 2^14.
__host__ __device__ inline uint32_t tg_tiny_limit(int64_t elements) {
  const int64_t lim = elements >> 14;
  return uint32_t(lim > 16 ? (lim < 0x7fffffff ? lim : 0x7fffffff) : 16);
}

// an operand split from fp32: its scales derive from its own measured maxima
// (margin > 0: the maxima are a sample's; the bound is 2^margin times the sampled maximum, and an
// element beyond twice that sends the operand through the exact passes -- see sgnn_tilegroup_split_f32)
__global__ void k_tg_state_init(TgState* __restrict__ state, uint32_t tiny_limit, int Fpad, int margin, int only_redo) {
  if (only_redo && state->redo == 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    state->tiny_limit = tiny_limit;
    state->sampled = margin > 0 ? 1u : 0u;
    if (only_redo) state->tiny = 0;   // counted against the provisional scales
  }
  if (i < Fpad) {
    const uint32_t m = state->amax[i];
    uint32_t b = m;
    if (margin > 0 && m != 0 && m < 0x7F800000u) {
      b = m + (uint32_t(margin) << 23);
      if (b >= 0x7F800000u) b = 0x7F7FFFFFu;
    }
    bound_of(state, Fpad)[i] = b;
    if (m >= 0x7F800000u) atomicOr(&state->flag, 1u);   // a non-finite element
  }
}

// cycle counts of a diagnostic run (mode bit 16): kernel, waits for a stage, issue of the next
// stage, step loops; wavefronts that took part; steps
__device__ unsigned long long g_tg_dbg[8];
constexpr int kDbgSpans = 16384;
__device__ unsigned long long g_tg_span[8 * kDbgSpans];   // diagnostics, eight words per unit of work (see sgnn_tilegroup_debug_spans)

// ---- the kernel -----------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 hv4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct TgArgs {
  const int64_t* plan_counts;   // the plan's own counts, checked against the launch's
  const int32_t *grows, *stage_col, *step0;
  const uint16_t* step_slots;
  const uint16_t* step_bits;
  const int32_t *item_group, *item_s0, *item_s1, *item_out;
  const int32_t *long_group, *long_slot, *long_nseg;
  const char* planes;
  TgState* state;    // of the operand in `planes`
  uint32_t* queue;   // work counter of this launch (in the state words)
  // per-feature words of the two states, from the launch's first feature on (a launch may cover a
  // range of the operand's 128-feature slices: every pointer below is then moved to the range's
  // first slice on the host and F, Fpad, slices describe the range)
  const uint32_t *amax_in, *bound_in;
  uint32_t *amax_out, *bound_out;
  const float* dinv_row;
  // diagonal of its own (A + A^T has weight 2 off the diagonal, A + I puts 1 on it): the
  // weighted row sum is w_off * (pattern sum) + corr[i] * x'_i with x' the operand as the
  // planes hold it; read from X in fp32 (times dinv_col per row when that is given) where the
  // caller has it, else rebuilt from the planes.  Row i's own operand row is x_row_offset + i.
  const float *corr, *X, *dinv_col;
  float *Y, *partial;
  // raw sums of another column block of the same rows (a partitioned matrix multiplies its
  // own-columns block while the remote rows travel), added to the pattern sum; NULL = none
  const float* addend;
  // != NULL: the rows are written as the NEXT hop's operand planes (fp16 hi / lo, scale per
  // feature from this operand's bound) instead of fp32 Y: row i at node out_row_offset + i
  char* planes_out;
  TgState* state_out;
  int64_t ldy, ldx, ld_addend, slice_bytes, slice_bytes_out;   // slice_bytes = n_cols * 512
  int64_t x_row_offset, out_row_offset, n_rows;
  int32_t items, n_long, F, Fpad, epilogue, stages, steps;
  float w_off;
  int32_t mode;
  int32_t slices;
  uint32_t tiny_limit_out;
  int32_t adapt;   // 1: the next operand's bound is this operand's measured maximum, 0: this operand's bound again
};

// LDS-DMA: 64 lanes x 16 (4) bytes from per-lane global addresses to LDS [lds, lds + 1024 (256)).
// Issued through inline assembly so that the compiler does not order later LDS reads behind
// it (it would wait for every outstanding DMA before the first read of any stage); the
// kernel waits for the stage it needs with counted s_waitcnt vmcnt itself.
__device__ __forceinline__ void tg_dma16(const void* g, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
// the same with the global address as a wave-uniform base plus a 32-bit per-lane byte offset
__device__ __forceinline__ void tg_dma16_off(const void* base, uint32_t off, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void tg_dma4(const void* g, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}

// LDS by byte offset.  The kernel has no static LDS, so its dynamic segment starts at LDS
// address 0 and an offset into `smem` is the address itself (formed from the generic pointer,
// every access pays an add of that zero base).
template <typename T>
__device__ __forceinline__ __attribute__((address_space(3))) T* lds(uint32_t off) {
  return (__attribute__((address_space(3))) T*)(uintptr_t(off));
}

// One workgroup per (work item, feature slice); 8 wavefronts, wavefront w owns sub-tiles 2 w and 2 w + 1.
// NT = the slice's N-tiles of 32 features that hold features below F (the last slice of a
// width that is not a multiple of 128, every slice of a narrow one): the tiles beyond are
// fetched with the row but neither read from LDS nor multiplied.
// the launch's arguments as they lie in the kernel-argument segment (scalar loads at the point of use)
typedef const __attribute__((address_space(4))) TgArgs TgKernArgs;

template <int NT, bool DIAG>
__device__ __forceinline__ void tg_run(TgKernArgs& a, const int item, const int slice, const int work) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned long long wall_entry = (a.mode & 16) ? wall_clock64() : 0;
  // (the thread id passes through an empty asm statement: everything derived from it is then
  // computed per unit of work instead of being hoisted out of the kernel's work loop, where the
  // lane constants of all eight instances of this function would stay live side by side)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = a.item_group[item], s0 = a.item_s0[item], s1 = a.item_s1[item], out = a.item_out[item];
  constexpr uint32_t lds0 = 0;   // LDS address of smem: the dynamic segment follows the static one, and there is none
  const char* xs = a.planes + int64_t(slice) * a.slice_bytes;

  // Pattern table at LDS offset 0: entry v (a nibble of a row's column bits) holds two dwords of
  // fp16 pairs with 1.0 where the bit is set: half of the row's hi-plane A operand.  The
  // lo-plane operand (2^-11: the lo plane holds the remainder times 2048) is the same with one
  // bit of each half kept.  Lanes with equal nibbles read one address, lanes with different
  // nibbles different banks: no conflicts (a 256-entry table read in one go had them, and the
  // LDS array, busy throughout the step phase, is what bounds the loop).
  if (tid < 16) {
    const uint32_t d0 = ((tid & 1) ? 0x3C00u : 0u) | ((tid & 2) ? 0x3C000000u : 0u);
    const uint32_t d1 = ((tid & 4) ? 0x3C00u : 0u) | ((tid & 8) ? 0x3C000000u : 0u);
    *reinterpret_cast<u32x2*>(smem + 8 * tid) = u32x2{d0, d1};
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[tt][nt][v] = 0.0f;

  // lane constants of the transposed reads: lane = 32 h + 16 g1 + 4 q + p supplies the address
  // of columns 16 g1 + 4 p .. + 3 of the row whose k index is 8 h + 4 r + q (read r = 0, 1)
  const int h = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
  const uint32_t lane_off = uint32_t(32 * g1 + 8 * p);
  // Narrow slices (one or two N-tiles) fetch fuller instructions: a DMA instruction's 64 lanes
  // carry the hi (or lo) segments of 16 / NT rows instead of the NT live segments of 4 rows
  // -- 2 NT row instructions per wavefront and stage instead of 8 -- and the rows lie in LDS
  // in blocks of 16 / NT rows, 2 KB each: hi segment s of row r at s * (1024 / NT) + r * 64, the
  // lo segments 1 KB behind.  Rows whose slots differ in their low two bits still sit in
  // different bank quarters.  (What a narrow stage costs is mostly its instructions: see the
  // ablation in the kernel's design notes.)
  constexpr bool kPacked = NT <= 2 && kIssuers == 8;
  constexpr int kPackRows = kPacked ? 16 / NT : 4;            // rows per row instruction
  constexpr int kPackInstr = kPacked ? 2 * NT : kRowInstr;    // row instructions per wavefront and stage
  constexpr int kPackCols = kPacked ? NT : kColRegs;          // column ids a lane holds for them
  constexpr uint32_t kSegStride = kPacked ? 1024u / NT : 256u;
  constexpr uint32_t kLoBase = kPacked ? 1024u : 4u * 256u;
  // the plan's row offsets (slot / 4 * 2 KB + slot % 4 * 64) in the packed layout: slot / R * 2 KB + slot % R * 64, R = 16 / NT
  auto packed = [&](uint32_t d) -> uint32_t {
    if constexpr (!kPacked) return d;
    else if constexpr (NT == 2) return ((d >> 1) & 0x78007800u) | ((d >> 3) & 0x01000100u) | (d & 0x00FF00FFu);
    else return ((d >> 2) & 0x38003800u) | ((d >> 3) & 0x03000300u) | (d & 0x00FF00FFu);
  };
  const uint32_t offs_off = uint32_t(kStageRowsBytes + 4 * (4 * h + q));   // this lane's dword of a step's offsets (r = 0 | r = 1 << 16)
  const uint32_t bits_off = uint32_t(kStageRowsBytes + kSlotsBytes + 2 * (lane & 31) + h);
  // A row of a stage is 512 bytes: eight 64-byte segments, hi N-tiles 0..3 then lo N-tiles 0..3.
  // The rows of slots 4 b .. 4 b + 3 form a 2 KB block of LDS, stored segment-major: segment s
  // of row r at s * 256 + r * 64.  A transposed read takes one segment of four rows per pass;
  // rows whose slots differ in their low two bits sit in different bank quarters, and the
  // segments of a row are a constant 256 bytes apart (immediate offsets).  One DMA
  // instruction brings the hi half of a block, the next its lo half; lane l writes bytes
  // 16 l .. 16 l + 15 of that kilobyte: segment l >> 4, row (l >> 2) & 3.
  const uint32_t dma_row = uint32_t(lane >> 2) & 3u;
  const uint32_t pack_row = uint32_t(lane >> 2) & uint32_t(kPackRows - 1);   // packed instructions: the lane's row of the block
  const uint32_t src_off = (uint32_t(lane >> 4) << 6) | (uint32_t(lane & 3) << 4);   // inside the node's hi (or lo) 256 bytes
  const bool small = a.slice_bytes < (int64_t(1) << 32);   // row offsets fit 32 bits

  // Stage s into ring buffer (s - s0) % 2.  Wavefront w brings rows 16 w .. 16 w + 15 (half a
  // block per instruction; `col[b]` is the column of row 16 w + 4 b + dma_row) and a share
  // of the stage's records: the steps' row offsets and row bits (only as many kilobytes as
  // the stage has steps), the first-step table from this stage on, and the column ids of the
  // following stages, so that the next stage is issued from LDS and not from scalar loads
  // whose latency every wavefront would sit out once per stage.
  auto issue_rows = [&](int s, const int (&col)[kColRegs], auto lo_tag, auto hi_tag) {   // row instructions lo .. hi - 1 of the wavefront's kRowInstr
    constexpr int lo = decltype(lo_tag)::value, hi = decltype(hi_tag)::value;
    const uint32_t buf = lds0 + kTableBytes + uint32_t((s - s0) & (kRing - 1)) * kStageBytes;
    if constexpr (kPacked) {
      // instruction i: block i >> 1 of the wavefront's 16 / kPackRows, hi (even) or lo (odd) half
      if (!(a.mode & 2)) {
        const uint32_t seg_src = NT == 2 ? (uint32_t(lane >> 5) << 6) : 0u;   // which N-tile of the row this lane moves
#pragma unroll
        for (int i = 0; i < kPackInstr; ++i) {
          if (i < lo || i >= hi) continue;
          const uint32_t dst = __builtin_amdgcn_readfirstlane(buf + uint32_t(w * (16 / kPackRows) + (i >> 1)) * 2048u + uint32_t(i & 1) * 1024u);
          const uint32_t off = seg_src + (uint32_t(lane & 3) << 4) + uint32_t(i & 1) * 256u;
          const int c = (a.mode & 64) ? (col[i >> 1] & ((a.mode & 512) ? 511 : 4095)) : col[i >> 1];
          if (small) tg_dma16_off(xs, (uint32_t(c) << 9) | off, dst);
          else tg_dma16(xs + int64_t(c) * kRowBytes + off, dst);
        }
      }
      return;
    }
    // (lanes whose N-tile holds no feature below F fetch nothing: lane l moves segment l >> 4)
    if (!(a.mode & 2) && (lane >> 4) < NT && w < kIssuers) {
#pragma unroll
      for (int i = 0; i < kRowInstr; ++i) {
        if (i < lo || i >= hi) continue;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(buf + uint32_t(kRowInstr * w + i) * 1024u);
        const uint32_t off = src_off + uint32_t(i & 1) * 256u;
        const int c = (a.mode & 64) ? (col[i >> 1] & ((a.mode & 512) ? 511 : 4095)) : col[i >> 1];   // diagnostics: every fetch from the first 4096 nodes' rows (10 MB at 602 features: the memory-side cache) or, with 512, the first 512 (1.3 MB: every XCD's L2)
        if (small) tg_dma16_off(xs, (uint32_t(c) << 9) | off, dst);
        else tg_dma16(xs + int64_t(c) * kRowBytes + off, dst);
      }
    }
  };
  auto issue_records = [&](int s, int sf32, int steps) {
    const uint32_t buf = lds0 + kTableBytes + uint32_t((s - s0) & (kRing - 1)) * kStageBytes;
    const int64_t sf = sf32;
    const uint32_t rec = buf + kStageRowsBytes;
    auto slots_kb = [&](int i) {
      if (i * 32 < steps)
        tg_dma16(reinterpret_cast<const char*>(a.step_slots) + sf * 32 + i * 1024 + lane * 16,
                 __builtin_amdgcn_readfirstlane(rec + uint32_t(i) * 1024u));
    };
    auto bits_kb = [&](int i) {
      if (i * 16 < steps)
        tg_dma16(reinterpret_cast<const char*>(a.step_bits) + sf * 64 + i * 1024 + lane * 16,
                 __builtin_amdgcn_readfirstlane(rec + kSlotsBytes + uint32_t(i) * 1024u));
    };
    auto first_steps = [&]() {
      tg_dma4(a.step0 + int64_t(s) * kSubs + lane, __builtin_amdgcn_readfirstlane(rec + kSlotsBytes + kBitsBytes));
    };
    auto next_cols = [&]() {
      tg_dma16(a.stage_col + (int64_t(s) + 1) * kStageCols + lane * 4,
               __builtin_amdgcn_readfirstlane(rec + kSlotsBytes + kBitsBytes + kHdrBytes));
    };
    if (kIssuers == 8) {
      if (w >= 6) {
        slots_kb((w - 6) * 2);
        slots_kb((w - 6) * 2 + 1);
      } else if (w >= 1 && w <= 4) {
        bits_kb((w - 1) * 2);
        bits_kb((w - 1) * 2 + 1);
      } else if (w == 5) {
        first_steps();
      } else {
        next_cols();
      }
    } else if (w < kIssuers) {
      if (w == 0) next_cols();
      if (w == 1) first_steps();
      slots_kb(w);
      bits_kb(w);
      bits_kb(w + 4);
    }
  };
  auto issue = [&](int s, const int (&col)[kColRegs], int sf32, int steps) {
    issue_rows(s, col, std::integral_constant<int, 0>{}, std::integral_constant<int, kRowInstr>{});
    issue_records(s, sf32, steps);
  };

  // The wavefront's 64 rows and their scales for the write-out, one per lane (a coalesced load
  // and a gather): asked for here, two dependent trips to memory ahead of the stage loop
  // instead of behind it.
  const int my_row = a.grows[int64_t(g) * kRows + 2 * kSub * w + lane];
  float my_dr = 1.0f, my_cr = 0.0f;
  if (out < 0 && my_row >= 0) {
    if (a.dinv_row != nullptr) my_dr = a.dinv_row[my_row];
    if (DIAG) my_cr = a.corr[my_row] * ((a.X != nullptr && a.dinv_col != nullptr) ? a.dinv_col[a.x_row_offset + my_row] : 1.0f);
  }
  const bool dbg = (a.mode & 16) != 0;
  unsigned long long t_wait = 0, t_issue = 0, t_steps = 0, n_steps = 0;
  const unsigned long long t_begin = dbg ? __builtin_amdgcn_s_memtime() : 0;
  const unsigned long long wall_begin = dbg ? wall_clock64() : 0;   // 100 MHz, the same on every CU
  const int ns = s1 - s0;
  // First steps of this wavefront's two sub-tiles in the stage about to run (the stage's own
  // first step, the run of sub-tile 2 w, the run of sub-tile 2 w + 1, the end): read one stage
  // ahead out of the step table that every stage's image carries, so that the first record of a
  // stage can be asked for as soon as its barrier has fallen.
  int cur_sf = 0, cur_h0 = 0, cur_h1 = 0, cur_h2 = 0;
  if (ns > 0) {   // the item's first stage: ids and step counts from global memory
    int col[kColRegs] = {};
    if constexpr (kPacked) {
#pragma unroll
      for (int b = 0; b < kPackCols; ++b) col[b] = a.stage_col[int64_t(s0) * kStageCols + 16 * w + kPackRows * b + pack_row];
    } else {
#pragma unroll
      for (int b = 0; b < kColRegs; ++b) col[b] = a.stage_col[int64_t(s0) * kStageCols + kIssuerRows * (w & (kIssuers - 1)) + 4 * b + dma_row];
    }
    const int32_t* st0 = a.step0 + int64_t(s0) * kSubs;
    cur_sf = __builtin_amdgcn_readfirstlane(st0[0]);
    cur_h0 = __builtin_amdgcn_readfirstlane(st0[2 * w]);
    cur_h1 = __builtin_amdgcn_readfirstlane(st0[2 * w + 1]);
    cur_h2 = __builtin_amdgcn_readfirstlane(st0[2 * w + 2]);
    const int sf_second = __builtin_amdgcn_readfirstlane(st0[kSubs]);
    issue(s0, col, cur_sf, sf_second - cur_sf);
  }

  for (int k = 0; k < ns; ++k) {
    const unsigned long long t0 = dbg ? __builtin_amdgcn_s_memtime() : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's share of stage k has landed
    __builtin_amdgcn_s_barrier();                      // everyone's has; stage k - 1 is no longer read
    asm volatile("" ::: "memory");
    const unsigned long long t1 = dbg ? __builtin_amdgcn_s_memtime() : 0;

    const uint32_t boff = kTableBytes + uint32_t(k & (kRing - 1)) * kStageBytes;
    const int* hdr = reinterpret_cast<const int*>(smem + boff + kStageRowsBytes + kSlotsBytes + kBitsBytes);
    // this wavefront's steps of the stage: those of sub-tile 2 w, then those of sub-tile 2 w + 1,
    // one contiguous run of records
    const int b0 = cur_h0 - cur_sf;
    const int n0 = cur_h1 - cur_sf - b0;
    const int total = cur_h2 - cur_sf - b0;
    const bool more = k + 1 < ns;
    const bool steps_on = total > 0 && !(a.mode & 1);
    const uint32_t stage_lane = boff + lane_off;
    // A step's record as this lane needs it: the LDS offsets of the two rows it addresses in
    // the transposed reads, and its 8 row bits.  Reading past the wavefront's last record is
    // harmless (the bytes are inside the stage buffer, what is read is not used).
    uint32_t rec_o = boff + offs_off + uint32_t(b0) * 32u, rec_b = boff + bits_off + uint32_t(b0) * 64u;
    auto record = [&](uint32_t& offs, uint32_t& bt) {   // (loads only: the values are taken apart a step later)
      offs = *lds<const uint32_t>(rec_o);
      bt = *lds<const uint8_t>(rec_b);
      rec_o += 32;
      rec_b += 64;
    };
    // pattern tile of the lane's row from its 8 column bits: one table entry per nibble
    auto pattern_hi = [&](uint32_t ta, uint32_t tb) {
      const u32x2 x = *lds<const u32x2>(ta), y = *lds<const u32x2>(tb);
      return u32x4{x[0], x[1], y[0], y[1]};
    };
    auto pattern_lo = [&](const u32x4& hi) { return hi & 0x10001000u; };
    auto read_tile = [&](uint32_t base_a, uint32_t base_b, uint32_t seg_off) {   // seg_off: where the N-tile's hi or lo segment lies in a block of rows
      const hv4 f0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(lds<hv4>(base_a + seg_off));
      const hv4 f1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(lds<hv4>(base_b + seg_off));
      h8 r;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        r[jj] = _Float16(f0[jj]);
        r[4 + jj] = _Float16(f1[jj]);
      }
      return r;
    };
    // Everything the stage needs out of LDS before its first MFMA is asked for in two rounds
    // in front of the next stage's DMA instructions, whose issue (the CU's memory pipeline
    // takes them at about one per 20 cycles, 72 of them per stage) then hides the second
    // round: first the stage's first record together with the next stage's column ids and
    // step counts, then the first step's pattern tile and tiles of X and the second record.
    // Issued behind the DMA instructions, the three dependent round trips record -> pattern ->
    // tiles stood between every barrier and the first MFMA of every wavefront.
    u32x4 ahi, alo;                             // pattern tiles of the current step
    uint32_t base_a = 0, base_b = 0;            // row addresses of the step whose tiles are being read
    uint32_t offs = 0, bt = 0;                  // record of the step after that: both row offsets, row bits
    uint32_t ta, tb;
    h8 bh[NT], bl[NT];
    if (steps_on) record(offs, bt);
    int col[kColRegs] = {};
    int nsf = 0, nsf2 = 0, nh0 = 0, nh1 = 0, nh2 = 0;
    if (more) {
      if constexpr (kPacked) {
        const int* nc = hdr + kHdrBytes / 4 + 16 * w + pack_row;
#pragma unroll
        for (int b = 0; b < kPackCols; ++b) col[b] = nc[kPackRows * b];
      } else {
        const int* nc = hdr + kHdrBytes / 4 + kIssuerRows * (w & (kIssuers - 1)) + dma_row;
#pragma unroll
        for (int b = 0; b < kColRegs; ++b) col[b] = nc[4 * b];
      }
      nsf = hdr[kSubs];
      nsf2 = hdr[2 * kSubs];
      nh0 = hdr[kSubs + 2 * w];
      nh1 = hdr[kSubs + 2 * w + 1];
      nh2 = hdr[kSubs + 2 * w + 2];
      // (taken to scalar registers here, with the first round: behind the second round the
      // compiler would wait for all of it to land before the first DMA instruction)
      nsf = __builtin_amdgcn_readfirstlane(nsf);
      nsf2 = __builtin_amdgcn_readfirstlane(nsf2);
      nh0 = __builtin_amdgcn_readfirstlane(nh0);
      nh1 = __builtin_amdgcn_readfirstlane(nh1);
      nh2 = __builtin_amdgcn_readfirstlane(nh2);
    }
    // (the first row instructions go out ahead of the second round: the barrier at the end of
    // the stage waits for the last row to land, and the CU's memory pipeline, once started,
    // works through the stage's instructions back to back)
    if (more) issue_rows(s0 + k + 1, col, std::integral_constant<int, 0>{}, std::integral_constant<int, kEarlyRows>{});
    if (steps_on) {
      ahi = pattern_hi((bt & 15u) << 3, (bt >> 1) & 0x78u);
      const uint32_t op = packed(offs);
      base_a = stage_lane + (op & 0xFFFFu);
      base_b = stage_lane + (op >> 16);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bh[nt] = read_tile(base_a, base_b, uint32_t(nt) * kSegStride);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bl[nt] = read_tile(base_a, base_b, kLoBase + uint32_t(nt) * kSegStride);
      record(offs, bt);
    }
    // The next stage is issued from the ids this stage's image carries, by every wavefront
    // before its steps (taking turns between the two wavefronts of a SIMD, or spreading the
    // issue between the steps, measured slower: the CU's memory pipeline takes ~1400 cycles
    // to accept a stage either way and the earlier it starts the earlier the rows arrive).
    if (more) {
      issue_rows(s0 + k + 1, col, std::integral_constant<int, kEarlyRows>{}, std::integral_constant<int, kRowInstr>{});
      issue_records(s0 + k + 1, nsf, nsf2 - nsf);
      cur_sf = nsf;
      cur_h0 = nh0;
      cur_h1 = nh1;
      cur_h2 = nh2;
    }
    const unsigned long long t2 = dbg ? __builtin_amdgcn_s_memtime() : 0;
    // Software pipeline over the steps.  Behind every MFMA of step j the reads of the same tile
    // of X for step j + 1 go out into the registers that MFMA has just consumed, and behind the
    // last MFMA of a plane the pattern tile of step j + 1 for that plane; the record of step
    // j + 2 is fetched at the top.  Nothing is copied between registers and the segments of a
    // row are reached by immediate offsets: what bounds this loop is the number of instructions
    // a wavefront issues beside its MFMAs (an MFMA leaves room for about six).  One loop per
    // sub-tile, so that the accumulators are addressed statically; the prefetch runs across the
    // seam between the two lists and, harmlessly, one step past the end.
    if (steps_on) {
      alo = pattern_lo(ahi);
      auto body = [&](auto tt_tag) {
        constexpr int TT = decltype(tt_tag)::value;
        const uint32_t op = packed(offs);
        base_a = stage_lane + (op & 0xFFFFu);
        base_b = stage_lane + (op >> 16);
        ta = (bt & 15u) << 3;
        tb = (bt >> 1) & 0x78u;
        record(offs, bt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[TT][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ahi), bh[nt], acc[TT][nt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          bh[nt] = read_tile(base_a, base_b, uint32_t(nt) * kSegStride);
          if (nt == NT - 1) ahi = pattern_hi(ta, tb);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[TT][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, alo), bl[nt], acc[TT][nt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          bl[nt] = read_tile(base_a, base_b, kLoBase + uint32_t(nt) * kSegStride);
          if (nt == NT - 1) alo = pattern_lo(ahi);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      int j = 0;
      // The second wavefront of a SIMD (4 .. 7) runs its steps at raised priority.  Left alone
      // the arbitration (oldest first) lets the first wavefront through its steps in 0.86 of the
      // mean time and the second in 1.14; raised, the figures swap, and the hop is 1.2 % shorter
      // (the wavefront that is slow in its steps is then the one that was first through the
      // issue of the next stage).  Priority on for the whole stage loop: half of that.  The two
      // swap at the seam between the sub-tile loops (the first wavefront raised in its second
      // loop): another 0.4 %.
      const bool second = w >= kWaves / 2;
      if (second) __builtin_amdgcn_s_setprio(1);
      for (; j < n0; ++j) body(std::integral_constant<int, 0>{});
      if (second) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
      for (; j < total; ++j) body(std::integral_constant<int, 1>{});
      if (!second) __builtin_amdgcn_s_setprio(0);
    }
    if (dbg) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      t_wait += t1 - t0;
      t_issue += t2 - t1;
      t_steps += t3 - t2;
      n_steps += uint32_t(total);
    }
  }
  if (dbg && lane == 0) {
    atomicAdd(&g_tg_dbg[0], __builtin_amdgcn_s_memtime() - t_begin);
    if (w == 0) atomicAdd(&g_tg_dbg[7], wall_clock64() - wall_begin);   // workgroup residence, summed
    if (w == 0 && work < kDbgSpans) {
      g_tg_span[8 * work] = wall_begin;
      g_tg_span[8 * work + 1] = wall_clock64();
      // XCC_ID (hardware register 20) and HW_ID (4: CU in bits 8-11, SH 12, SE 13-15)
      g_tg_span[8 * work + 2] = ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) << 16) |
                                      ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xFFu);
      g_tg_span[8 * work + 3] = (unsigned long long)ns;
      g_tg_span[8 * work + 5] = wall_entry;
    }
    atomicAdd(&g_tg_dbg[1], t_wait);
    atomicAdd(&g_tg_dbg[2], t_issue);
    atomicAdd(&g_tg_dbg[3], t_steps);
    atomicAdd(&g_tg_dbg[4], 1ull);
    atomicAdd(&g_tg_dbg[5], n_steps);
    atomicAdd(&g_tg_dbg[6], (unsigned long long)ns);
  }

  // ---- write-out -------------------------------------------------------------------------
  // Accumulator register v of a lane: row (v & 3) + 8 (v >> 2) + 4 h of the sub-tile, feature
  // lane & 31 of the N-tile.  Stored from there a wave instruction would carry two 128-byte row
  // pieces (or, for fp16 planes, two of 64 bytes); the rows go through LDS instead -- the ring
  // is free once every wavefront has left the stage loop -- and leave as whole 512-byte rows,
  // 16 bytes per lane.  Each wavefront stages its own 32 rows of a sub-tile, so nothing but the
  // one barrier in front is needed (a wavefront's LDS accesses execute in order).
  // The wavefront's 64 rows and their scales, looked up once per lane before the stage loop, go
  // to a small table in LDS behind the staging areas (written after the barrier below); a lane of
  // the row-major pass reads the entry of the row it carries.  (Handed out by v_readlane they
  // took 200 scalar registers, which the compiler spilled.)
  if (a.mode & 32) return;   // diagnostics: nothing written
  const bool fin = out < 0;                                  // uniform: finished rows (else raw partial sums)
  const bool to_planes = fin && a.planes_out != nullptr;     // uniform
  const bool diag = DIAG && fin;                             // uniform (DIAG: the launch has a diagonal correction)
  const bool add = fin && a.addend != nullptr;               // uniform
  // Two passes per sub-tile.  The accumulators go to LDS as they are; the rows are then taken
  // back 16 bytes per lane -- lane l holds features 4 (l & 31) .. + 3 of row 2 j + (l >> 5) -- and
  // everything a finished row needs is done in that layout: the scale back, the other column
  // block's raw sums and the row's own operand elements for the diagonal correction (read as
  // whole 512-byte row pieces, all eight row pairs of a batch asked for before the first is
  // used, the next batch before this one is worked on: done in the accumulators' layout these
  // loads came to eight round trips to memory per wavefront and unit of work, a twentieth of
  // a unit's time), the row scaling, the split into the next operand's planes.
  const int c = lane & 31;                     // which 16 bytes of a staged row
  const int f0 = slice * kSlice + 4 * c;       // the lane's first feature
  const bool live = (c >> 3) < NT;             // N-tiles this unit did not compute hold nothing
  // The bounds of this lane's features: the power of two the accumulators are scaled back by,
  // and the scale of the operand written for the next hop.  A row of D^-1 (A + I) sums to one,
  // so no element of that operand exceeds the largest of the operand read here: that measured
  // maximum (or, where ranks would have to agree on it first, the older bound) is the next
  // operand's bound.
  float inv4[4], sc4[4];
  uint32_t nxt4[4];
  u32x4 mx4 = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t bnd = a.bound_in[f0 + i];
    const uint32_t nxt = (to_planes && a.adapt && !a.state->sampled) ? a.amax_in[f0 + i] : bnd;
    const uint32_t sb = __builtin_bit_cast(uint32_t, tg_scale(bnd));
    inv4[i] = __builtin_bit_cast(float, (254u - (sb >> 23)) << 23);   // 1 / scale, exact: a power of two
    sc4[i] = tg_scale(nxt);
    nxt4[i] = nxt;
  }
  uint32_t tiny = 0;
  __builtin_amdgcn_s_barrier();   // the ring is no longer read
  const unsigned long long wall_bar = dbg ? wall_clock64() : 0;
  const uint32_t stg = uint32_t(kTableBytes) + uint32_t(w) * uint32_t(kStageOutBytes);
  const uint32_t rows_tab = uint32_t(kTableBytes + kWaves * kStageOutBytes) + uint32_t(w) * uint32_t(kRowTabBytes);
  *lds<int>(rows_tab + uint32_t(lane) * 4u) = my_row;
  *lds<float>(rows_tab + 256u + uint32_t(lane) * 4u) = my_dr;
  *lds<float>(rows_tab + 512u + uint32_t(lane) * 4u) = my_cr;
  const bool whole = int64_t(slice + 1) * kSlice <= a.F;   // every feature of the slice exists
  const bool y4 = (a.ldy % 4 == 0) && (reinterpret_cast<uintptr_t>(a.Y) % 16 == 0);
  const bool y2 = (a.ldy % 2 == 0) && (reinterpret_cast<uintptr_t>(a.Y) % 8 == 0);
  const bool x4 = whole && (a.ldx % 4 == 0) && (reinterpret_cast<uintptr_t>(a.X) % 16 == 0);
  const bool x2 = whole && (a.ldx % 2 == 0) && (reinterpret_cast<uintptr_t>(a.X) % 8 == 0);
  const bool d4 = whole && (a.ld_addend % 4 == 0) && (reinterpret_cast<uintptr_t>(a.addend) % 16 == 0);
  const bool d2 = whole && (a.ld_addend % 2 == 0) && (reinterpret_cast<uintptr_t>(a.addend) % 8 == 0);
  // four floats of a row from feature f0 on, in the widest pieces the row's alignment allows; in
  // a slice that F cuts short, feature by feature (one past F reads the last: the value is not used)
  auto load4 = [&](const float* rowp, bool v4, bool v2) -> f32x4 {
    if (v4) return *reinterpret_cast<const f32x4*>(rowp + f0);
    if (v2) {
      const f32x2 p0 = *reinterpret_cast<const f32x2*>(rowp + f0), p1 = *reinterpret_cast<const f32x2*>(rowp + f0 + 2);
      return f32x4{p0[0], p0[1], p1[0], p1[1]};
    }
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = rowp[min(f0 + i, int(a.F) - 1)];
    return r;
  };
  struct HalfRows { hv4 hi, lo; };
  const uint32_t my_tab = rows_tab + uint32_t(h) * 4u;   // entry 2 j + h of a sub-tile: the row this lane carries in row pair j
  auto lane_row = [&](int tt, int j) { return *lds<const int>(my_tab + uint32_t(kSub * tt + 2 * j) * 4u); };
  constexpr int kBatch = 4;   // row pairs whose loads are in flight together
  auto fetch = [&](int tt, int b, f32x4 (&xq)[kBatch], f32x4 (&aq)[kBatch]) {
#pragma unroll
    for (int jj = 0; jj < kBatch; ++jj) {
      const int row = max(lane_row(tt, b * kBatch + jj), 0);   // a row that does not exist reads row 0
      if (a.mode & 2048) {   // diagnostics: nothing read for the write-out (results wrong)
        xq[jj] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        aq[jj] = xq[jj];
        continue;
      }
      if constexpr (DIAG) if (diag) {
        const int64_t node = a.x_row_offset + row;
        if (a.X != nullptr) {
          xq[jj] = load4(a.X + node * a.ldx, x4, x2);
        } else {   // the operand exists only as planes
          const char* pr = xs + node * kRowBytes + c * 8;
          xq[jj] = __builtin_bit_cast(f32x4, HalfRows{*reinterpret_cast<const hv4*>(pr), *reinterpret_cast<const hv4*>(pr + kRowBytes / 2)});
        }
      }
      if (add) aq[jj] = load4(a.addend + int64_t(row) * a.ld_addend, d4, d2);
    }
  };
  auto stage_out = [&](auto tt_tag) {   // accumulator register v of a lane: row (v & 3) + 8 (v >> 2) + 4 h, feature lane & 31 of the N-tile
    constexpr int TT = decltype(tt_tag)::value;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const uint32_t rl = uint32_t((v & 3) + 8 * (v >> 2) + 4 * h);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *lds<float>(stg + rl * uint32_t(kRowBytes) + uint32_t(nt * 32 + (lane & 31)) * 4u) = acc[TT][nt][v];
    }
  };
  auto finish = [&](int tt, int b, const f32x4 (&xq)[kBatch], const f32x4 (&aq)[kBatch]) {
#pragma unroll
    for (int jj = 0; jj < kBatch; ++jj) {
      const int j = b * kBatch + jj;
      const int row = lane_row(tt, j);
      const float dr = *lds<const float>(my_tab + 256u + uint32_t(kSub * tt + 2 * j) * 4u);
      const float cr = *lds<const float>(my_tab + 512u + uint32_t(kSub * tt + 2 * j) * 4u);
      const f32x4 q4 = *lds<const f32x4>(stg + uint32_t(j) * 1024u + uint32_t(lane) * 16u);
      f32x4 y4v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float y = q4[i] * inv4[i];   // exact: a power of two
        if (fin) {
          if (add) y += aq[jj][i];
          if constexpr (DIAG) {
            float xv;
            if (a.X != nullptr) {
              xv = xq[jj][i];
            } else {   // hi + lo / 2048, scaled back
              const HalfRows hr = __builtin_bit_cast(HalfRows, xq[jj]);
              xv = (float(hr.hi[i]) + float(hr.lo[i]) * (1.0f / 2048.0f)) * inv4[i];
            }
            y = fmaf(a.w_off, y, cr * xv);
          } else if (a.w_off != 1.0f) {
            y *= a.w_off;
          }
          if (a.dinv_row != nullptr) {
            y = y * dr;
            if (a.epilogue == SGNN_EPI_SCALE_ROW2) y = y * dr;
          }
          // rows that do not exist, features past F and N-tiles not computed: zero, as the planes' padding must be
          if (row < 0 || f0 + i >= a.F || !live) y = 0.0f;
        }
        y4v[i] = y;
      }
      if (to_planes) {
        hv4 hq, lq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          _Float16 hv, lv;
          const float yv = y4v[i];   // (a copy: __builtin_bit_cast of a vector element reads element 0)
          tg_split_one(yv, sc4[i], hv, lv, tiny);
          const uint32_t yb = __builtin_bit_cast(uint32_t, yv) & 0x7FFFFFFFu;
          mx4[i] = yb > mx4[i] ? yb : mx4[i];
          hq[i] = hv;
          lq[i] = lv;
        }
        if (row < 0) continue;
        char* dst = a.planes_out + int64_t(slice) * a.slice_bytes_out + (a.out_row_offset + row) * kRowBytes + c * 8;
        *reinterpret_cast<hv4*>(dst) = hq;
        *reinterpret_cast<hv4*>(dst + kRowBytes / 2) = lq;
      } else if (fin) {
        float* dst = a.Y + int64_t(max(row, 0)) * a.ldy + f0;
        if (y4) {
          if (row < 0) continue;
          if (f0 + 4 <= a.F) {
            *reinterpret_cast<f32x4*>(dst) = y4v;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (f0 + i < a.F) dst[i] = y4v[i];
          }
        } else if (y2) {
          // Rows aligned to 8 bytes only (F = 602): stored as they lie, 8 bytes per lane at a
          // stride of 16, every 32-byte sector of Y would be written in two halves by two
          // instructions.  The finished row goes back to its staging area and is taken up again
          // as two runs of 256 contiguous bytes: lane c carries features 2 c, 2 c + 1 and
          // 64 + 2 c, 65 + 2 c.
          const uint32_t back = stg + uint32_t(j) * 1024u + uint32_t(h) * 512u;
          *lds<f32x4>(back + uint32_t(c) * 16u) = y4v;
          const f32x2 pa = *lds<const f32x2>(back + uint32_t(c) * 8u), pb = *lds<const f32x2>(back + 256u + uint32_t(c) * 8u);
          if (row < 0) continue;
          float* d2 = a.Y + int64_t(row) * a.ldy + slice * kSlice + 2 * c;
          const int fa = slice * kSlice + 2 * c, fb = fa + 64;
          if (fa + 2 <= a.F) *reinterpret_cast<f32x2*>(d2) = pa;
          else if (fa < a.F) d2[0] = pa[0];
          if (fb + 2 <= a.F) *reinterpret_cast<f32x2*>(d2 + 64) = pb;
          else if (fb < a.F) d2[64] = pb[0];
        } else {
          if (row < 0) continue;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (f0 + i < a.F) dst[i] = y4v[i];
        }
      } else {
        const int rr = 2 * kSub * w + kSub * tt + 2 * j + h;
        *reinterpret_cast<f32x4*>(a.partial + (int64_t(out) * kRows + rr) * a.Fpad + slice * kSlice + 4 * c) = y4v;
      }
    }
  };
  {
    [[maybe_unused]] f32x4 xa[kBatch], xb[kBatch], aa[kBatch], ab[kBatch];
    constexpr int kBatches = 16 / kBatch;   // per sub-tile
    fetch(0, 0, xa, aa);
    stage_out(std::integral_constant<int, 0>{});
#pragma unroll
    for (int b = 0; b < 2 * kBatches; ++b) {   // batch b of the two sub-tiles' 2 * kBatches; the next batch's loads go out first
      const int tt = b / kBatches, bb = b % kBatches;
      if (b + 1 < 2 * kBatches) {
        if (b % 2 == 0) fetch((b + 1) / kBatches, (b + 1) % kBatches, xb, ab);
        else fetch((b + 1) / kBatches, (b + 1) % kBatches, xa, aa);
      }
      if (b == kBatches) stage_out(std::integral_constant<int, 1>{});
      if (b % 2 == 0) finish(tt, bb, xa, aa);
      else finish(tt, bb, xb, ab);
      if (b == 0 && dbg && lane == 0 && w == 0 && work < kDbgSpans) {   // since the end of the stage loop: the barrier, the first batch stored (100 MHz ticks)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g_tg_span[8 * work + 7] = ((wall_clock64() - wall_begin) << 32) | ((wall_bar - wall_begin) & 0xFFFFFFFFull);
      }
    }
  }
  if (to_planes) {
    // what the next hop needs to know about the operand just written: whether the planes hold
    // it, and each feature's largest magnitude (the bound of the hop after next)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tiny += uint32_t(__shfl_xor(int(tiny), off));
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) bad |= tg_beyond(mx4[i], nxt4[i]);
    const bool any_bad = __any(bad);
    if (lane == 0) {
      if (tiny) atomicAdd(&a.state_out->tiny, tiny);
      if (any_bad) atomicOr(&a.state_out->flag, 1u);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t o = uint32_t(__shfl_xor(int(mx4[i]), 32));
      const uint32_t m = o > mx4[i] ? o : mx4[i];
      if (h == 0 && m) atomicMax(reinterpret_cast<uint32_t*>(smem + kAmaxOff) + (4 * c + i), m);
    }
    __syncthreads();
    if (tid < kSlice) {
      const uint32_t m = *lds<uint32_t>(uint32_t(kAmaxOff) + uint32_t(tid) * 4u);
      if (m) {
        atomicMax(&a.amax_out[slice * kSlice + tid], m);
        *lds<uint32_t>(uint32_t(kAmaxOff) + uint32_t(tid) * 4u) = 0u;
      }
    }
  }
  if (dbg && lane == 0 && w == 0 && work < kDbgSpans) g_tg_span[8 * work + 4] = wall_clock64();   // wavefront 0 has written its rows
}

// Work order: slice-major, inside a slice the items longest first (the plan's order).  Slices
// stay together in time, which keeps a slice's hot rows in L2 (slice index fastest was tried:
// 9 % slower), and a slice ends on its short items; launching the long items of all slices
// ahead of all the short ones was tried and made no difference once the counter was in.
// the operand cannot go through the planes (or the launch does not match the plan): nothing is
// computed here and the fall-back behind the launch, guarded by state->flag, takes over
__device__ __forceinline__ bool tg_refused(const TgArgs& a, bool first_thread) {
  const TgState* st = a.state;
  const bool many_tiny = st->tiny > st->tiny_limit;
  const int64_t* pc = a.plan_counts;
  const bool mismatch = pc[1] != a.stages || pc[2] != a.steps || pc[3] != a.items || pc[5] != a.n_long || pc[7] != a.n_rows;
  const bool refused = st->flag != 0 || many_tiny || mismatch;
  if (refused && first_thread) {
    const uint32_t bits = (many_tiny ? 2u : 0u) | (mismatch ? 4u : 0u);
    if (bits) atomicOr(&a.state->flag, bits);
    if (a.state_out != nullptr) atomicOr(&a.state_out->flag, 8u);   // the hops behind this one have no operand
  }
  return refused;
}

__global__ __launch_bounds__(kWaves* kWave, 2) void k_tg_spmm(TgArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  if (tg_refused(a, blockIdx.x == 0 && threadIdx.x == 0)) return;
  if (a.state_out != nullptr) {
    // the next operand's bounds (see tg_run) and its limit of tiny elements
    for (int sl = blockIdx.x; sl < a.slices; sl += gridDim.x)
      if (threadIdx.x < kSlice) {
        const int f = sl * kSlice + threadIdx.x;
        a.bound_out[f] = (a.adapt && !a.state->sampled) ? a.amax_in[f] : a.bound_in[f];
      }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.state_out->tiny_limit = a.tiny_limit_out;
  }
  if (threadIdx.x < kSlice) *lds<uint32_t>(uint32_t(kAmaxOff) + threadIdx.x * 4u) = 0u;
  const int total = a.items * a.slices;
  // One workgroup per CU for the whole launch, taking (item, slice) pairs off a counter: a
  // workgroup holds all of a CU's LDS, so a fresh one can only be placed when the last has gone,
  // and between two workgroups of a one-per-pair grid a CU sat idle for tens of microseconds
  // (Reddit shape: 5.27 ms per hop with the counter against 5.48 ms with one workgroup per
  // pair).  Every wavefront leaves through the same test of b.
  int* next = reinterpret_cast<int*>(smem + kTicketOff);   // behind the pattern table
  int prev = -1;
  // Which units a workgroup takes: the next of the launch order (one counter), or, for a plan
  // ordered neighbours-together (see k_tg_item_keys), the order dealt out in runs of 32 units (an
  // XCD has 32 CUs), a workgroup taking the next unit of its own XCD's runs (one counter per XCD)
  // and, when those are finished, of the other XCDs'.  (Mode bits 256 / 1024 force it on / off.)
  const int xcd = int(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u);   // XCC_ID
  const bool by_xcd = ((a.plan_counts[8] != 0) || (a.mode & 256)) && !(a.mode & 1024);
  int victim = 0;   // how many other XCDs' queues this workgroup has found empty
  for (;;) {
    int b = blockIdx.x;
    if (!(a.mode & 128)) {
      if (threadIdx.x == 0) {
        int t = total;
        if (by_xcd) {
          // ticket k of queue q is unit k % 32 of the queue's run k / 32, i.e. of run (k / 32) * 8 + q
          // of the launch; tickets only grow, so a queue whose ticket lies past the end is finished
          while (victim < 8) {
            const int q = (xcd + victim) & 7;
            const int k = int(atomicAdd(&a.state->xcd_queue[q], 1u));
            // (rounds of eight runs go to the XCDs forwards and backwards in turn: the order is by
            // falling length, and dealt one way only XCD 0 would get the longest run of every round)
            const int round = k >> 5;
            t = (round * 8 + ((round & 1) ? 7 - q : q)) * 32 + (k & 31);
            if (t < total) break;
            t = total;
            ++victim;
          }
        } else {
          t = int(atomicAdd(a.queue, 1u));
        }
        *next = t;
      }
      __syncthreads();
      b = __builtin_amdgcn_readfirstlane(*next);
      __syncthreads();   // everyone has read it before the next round writes it; the last item's LDS reads are over
      if ((a.mode & 16) && threadIdx.x == 0 && prev >= 0 && prev < kDbgSpans) g_tg_span[8 * prev + 6] = wall_clock64();
      prev = b;
    }
    if (b >= total) break;
    const int slice = b / a.items, item = b - slice * a.items;
    const int tiles = (min(kSlice, a.F - slice * kSlice) + 31) / 32;   // workgroup-uniform
    // A unit of work reads the launch's arguments where it needs them, through a pointer the
    // compiler cannot see through: held in registers across the work loop (eight instances of
    // tg_run, each with its own derived values) they cost more registers than the wavefront has.
    TgKernArgs* ka = (TgKernArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    if (a.corr != nullptr) {   // the same for every unit of the launch
      if (tiles >= 4) tg_run<4, true>(*ka, item, slice, b);
      else if (tiles == 3) tg_run<3, true>(*ka, item, slice, b);
      else if (tiles == 2) tg_run<2, true>(*ka, item, slice, b);
      else tg_run<1, true>(*ka, item, slice, b);
    } else {
      if (tiles >= 4) tg_run<4, false>(*ka, item, slice, b);
      else if (tiles == 3) tg_run<3, false>(*ka, item, slice, b);
      else if (tiles == 2) tg_run<2, false>(*ka, item, slice, b);
      else tg_run<1, false>(*ka, item, slice, b);
    }
    if (a.mode & 128) break;   // diagnostics: one workgroup per (item, slice) pair
  }
  // The last workgroup to get here zeroes the work counters for the next launch that uses this
  // state (every workgroup is past its last ticket by then), which saves a memset per launch.
  if (!(a.mode & 128) && threadIdx.x == 0) {
    const uint32_t gone = atomicAdd(&a.state->done, 1u);
    if (gone == gridDim.x - 1) {
      a.state->queue = 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) a.state->xcd_queue[q] = 0u;
      a.state->done = 0u;
    }
  }
}

// rows of a group cut into several items: add the items' partial rows in order, then the row epilogue
// The launch's workgroup size is a multiple of the (row, four features) pairs of one row
// (tg_combine_threads), so a thread keeps the same four features for every row it visits.
__host__ __device__ inline int tg_combine_quads(int F, int Fpad, bool to_planes) { return ((to_planes ? Fpad : F) + 3) / 4; }
inline unsigned tg_combine_threads(int quads) {
  if (quads >= 1024) return 1024;
  const int fit = 512 / quads;                // rows side by side
  const int rows = fit > 8 ? 8 : (fit < 1 ? 1 : fit);
  const int t = quads * rows;
  return unsigned(t < 64 ? 64 : t);
}

__global__ __launch_bounds__(1024) void k_tg_combine(TgArgs a) {
  if (tg_refused(a, false)) return;
  const bool to_planes = a.planes_out != nullptr;
  __shared__ uint32_t smax[4096];   // rows_par > 1 means at most 512 features
  if (to_planes) {
    for (int i = threadIdx.x; i < 4096 && i < a.F; i += blockDim.x) smax[i] = 0;
    __syncthreads();
  }
  const int li = blockIdx.x;
  const int g = a.long_group[li], slot = a.long_slot[li], nseg = a.long_nseg[li];
  const int64_t stride = int64_t(kRows) * a.Fpad;
  // A workgroup takes the rows blockIdx.y, blockIdx.y + gridDim.y, ... of the group, several of
  // them side by side; a thread owns four features (the partial rows are padded to whole slices
  // and 16-byte aligned).  The partial sums are loaded four at a time and added in item order.
  // As planes the rows are written whole, padding features included (they are zero).
  const int quads = tg_combine_quads(a.F, a.Fpad, to_planes);
  const int rows_par = int(blockDim.x) / quads > 0 ? int(blockDim.x) / quads : 1;
  const int my_rows = (kRows - int(blockIdx.y) + int(gridDim.y) - 1) / int(gridDim.y);
  const bool scale = a.dinv_row != nullptr;
  bool bad = false;
  uint32_t tiny = 0;
  for (int q = threadIdx.x % quads; q < quads && int(threadIdx.x) < quads * rows_par; q += int(blockDim.x) >= quads ? quads : int(blockDim.x)) {
    const int f4 = q * 4;
    uint32_t mx[4] = {0, 0, 0, 0};
    uint32_t nb[4] = {0, 0, 0, 0};
    float inv_in[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (to_planes || (a.corr != nullptr && a.X == nullptr)) {   // (uniform) the features' scales are needed only then
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = f4 + j;
        const uint32_t bnd = f < a.F ? a.bound_in[f] : 0u;
        nb[j] = (to_planes && a.adapt && !a.state->sampled && f < a.F) ? a.amax_in[f] : bnd;
        const uint32_t sb = __builtin_bit_cast(uint32_t, tg_scale(bnd));
        inv_in[j] = __builtin_bit_cast(float, (254u - (sb >> 23)) << 23);
      }
    }
    for (int r = int(threadIdx.x) / quads; r < my_rows; r += rows_par) {
      const int rr = int(blockIdx.y) + r * int(gridDim.y);
      const int row = a.grows[int64_t(g) * kRows + rr];
      if (row < 0) continue;
      const float dr = scale ? a.dinv_row[row] : 1.0f;
      const float cr = a.corr != nullptr
                           ? a.corr[row] * ((a.X != nullptr && a.dinv_col != nullptr) ? a.dinv_col[a.x_row_offset + row] : 1.0f)
                           : 0.0f;
      const float* pp = a.partial + (int64_t(slot) * kRows + rr) * a.Fpad + f4;
      f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
      // four loads in flight whatever the number of pieces (most groups are cut in two to four): a
      // piece that does not exist reads the last one again and adds zero, which changes no bit
      // (the sum starts from +0, so its zeros are +0 already)
      for (int k = 0; k < nseg; k += 4) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(pp + (k + i < nseg ? k + i : nseg - 1) * stride);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (k + i >= nseg) v[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
          s += v[i];
        }
      }
      _Float16 hv[4], lv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = f4 + j;
        float y = 0.0f;
        if (f < a.F) {
          y = s[j];
          if (a.addend != nullptr) y += a.addend[int64_t(row) * a.ld_addend + f];
          if (a.corr != nullptr) {
            float xv;
            if (a.X != nullptr) {
              xv = a.X[(a.x_row_offset + row) * a.ldx + f];
            } else {
              const char* pr = a.planes + int64_t(f / kSlice) * a.slice_bytes + (a.x_row_offset + row) * kRowBytes + (f % kSlice) * 2;
              xv = (float(*reinterpret_cast<const _Float16*>(pr)) +
                    float(*reinterpret_cast<const _Float16*>(pr + kRowBytes / 2)) * (1.0f / 2048.0f)) * inv_in[j];
            }
            y = fmaf(a.w_off, y, cr * xv);
          } else if (a.w_off != 1.0f) {
            y *= a.w_off;
          }
          if (scale) {
            y = y * dr;
            if (a.epilogue == SGNN_EPI_SCALE_ROW2) y = y * dr;
          }
        }
        if (to_planes) {
          tg_split_one(y, tg_scale(nb[j]), hv[j], lv[j], tiny);
          const uint32_t yb = __builtin_bit_cast(uint32_t, y) & 0x7FFFFFFFu;
          mx[j] = yb > mx[j] ? yb : mx[j];
        } else if (f < a.F) {
          a.Y[int64_t(row) * a.ldy + f] = y;
        }
      }
      if (to_planes) {
        char* dst = a.planes_out + int64_t(f4 / kSlice) * a.slice_bytes_out + (a.out_row_offset + row) * kRowBytes + (f4 % kSlice) * 2;
        auto pack = [](_Float16 x, _Float16 y2) {
          return uint32_t(__builtin_bit_cast(uint16_t, x)) | (uint32_t(__builtin_bit_cast(uint16_t, y2)) << 16);
        };
        *reinterpret_cast<u32x2*>(dst) = u32x2{pack(hv[0], hv[1]), pack(hv[2], hv[3])};
        *reinterpret_cast<u32x2*>(dst + kRowBytes / 2) = u32x2{pack(lv[0], lv[1]), pack(lv[2], lv[3])};
      }
    }
    if (to_planes) {
      // the threads that share these features fold their maxima in LDS first: one atomic per
      // feature and workgroup reaches memory (every workgroup of the launch adds to the same words)
#pragma unroll
      for (int j = 0; j < 4; ++j) bad |= tg_beyond(mx[j], nb[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (mx[j] && f4 + j < a.F) {
          if (f4 + j < 4096 && rows_par > 1) atomicMax(&smax[f4 + j], mx[j]);
          else atomicMax(&a.amax_out[f4 + j], mx[j]);
        }
    }
    if (int(blockDim.x) >= quads) break;   // one set of features per thread
  }
  if (to_planes) {
    if (rows_par > 1) {
      __syncthreads();
      for (int f = threadIdx.x; f < a.F && f < 4096; f += blockDim.x)
        if (smax[f]) atomicMax(&a.amax_out[f], smax[f]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tiny += uint32_t(__shfl_xor(int(tiny), off));
    const bool any_bad = __any(bad);
    if ((threadIdx.x & (kWave - 1)) == 0) {
      if (tiny) atomicAdd(&a.state_out->tiny, tiny);
      if (any_bad) atomicOr(&a.state_out->flag, 1u);
    }
  }
}

dim3 blocks_for(int64_t items, unsigned T = 256) { return dim3(unsigned((items + T - 1) / T)); }

}  // namespace

namespace sgnn {
void set_tilegroup_mode(int m) { g_tg_mode = m; }
void set_tilegroup_segment(int stages) { g_seg_stages = stages >= 8 ? stages : 0; }
void set_tilegroup_order(int cols) { g_order_cols = cols; }
void set_tilegroup_combine_rows(int gy) { g_combine_gy = gy; }
void set_tilegroup_no_sample(int v) { g_no_sample = v; }
void set_tilegroup_split_shape(int blocks, int threads) {
  if (blocks >= 0) g_split_blocks = blocks;
  if (threads >= 0) g_split_threads = threads;
}
}  // namespace sgnn

extern "C" {

// diagnostics: the cycle counters of runs made with tuning key tg_mode bit 16 set; reading clears them
int sgnn_tilegroup_debug_counters(unsigned long long* out8) {
  unsigned long long zero[8] = {};
  HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tg_dbg), sizeof(zero)));
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_tg_dbg), zero, sizeof(zero)));
  return SGNN_OK;
}

// diagnostics: per unit of work (item, slice) of the last k_tg_spmm launch made with tg_mode bit
// 16 set, eight words: wall clock (100 MHz) at the start and the end of the stage loop, CU
// (XCC << 16 | SE, SH, CU bits), stages walked, wall clock when wavefront 0 had written its
// rows, at entry, and when the whole workgroup had fetched its next unit; one spare.  At most
// 16384 units are kept
int sgnn_tilegroup_debug_spans(unsigned long long* out, int64_t count) {
  if (!out || count < 0 || count > kDbgSpans) return fail(SGNN_ERR_INVALID, "at most 16384 spans are kept");
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tg_span), size_t(count) * 64));
  return SGNN_OK;
}

size_t sgnn_tilegroup_build_workspace_bytes(int64_t n_rows, int64_t nnz) {
  if (n_rows <= 0 || nnz <= 0) return 0;
  return carve_tg_build(nullptr, n_rows, nnz).total;
}

size_t sgnn_tilegroup_plan_bytes(int64_t n_rows, int64_t stages, int64_t steps, int64_t items, int64_t long_groups) {
  if (n_rows <= 0 || stages < 0 || steps < 0 || items < 0 || long_groups < 0) return 0;
  return carve_tg_plan(nullptr, n_rows, stages, steps, items, long_groups).total;
}

int sgnn_tilegroup_analyze(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                           void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_rows <= 0 || n_cols <= 0 || nnz <= 0) return fail(SGNN_ERR_INVALID, "empty matrix");
  if (n_rows >= (int64_t(1) << 31) || n_cols >= (int64_t(1) << 31) || nnz >= (int64_t(1) << 31))
    return fail(SGNN_ERR_INVALID, "size exceeds int32 index range");
  if (!rowptr || !colidx || !workspace) return fail(SGNN_ERR_INVALID, "null pointer");
  TgBuild w = carve_tg_build(workspace, n_rows, nnz);
  if (workspace_bytes < w.total) return fail(SGNN_ERR_WORKSPACE, "tile-group build workspace too small");
  const int cbits = ceil_log2(n_cols);
  const int gbits = ceil_log2(w.G + 1);
  if (9 + cbits + gbits > 64) return fail(SGNN_ERR_INVALID, "key does not fit 64 bits");
  const int rbits = ceil_log2(w.cap_runs + 1);
  if (5 + kSlotBits + rbits > 64) return fail(SGNN_ERR_INVALID, "key does not fit 64 bits");
  const unsigned T = 256;
  HIP_TRY(hipMemsetAsync(w.counts, 0, 128, st));
  // rank the rows by degree (stable: ties keep row order)
  hipLaunchKernelGGL(k_tg_deg_keys, blocks_for(n_rows), dim3(T), 0, st, n_rows, rowptr, w.dkey_a, w.drow_a);
  HIP_TRY(hipGetLastError());
  size_t pb = w.prim_bytes;
  HIP_TRY(rocprim::radix_sort_pairs(w.prim, pb, w.dkey_a, w.dkey_b, w.drow_a, w.drow_b, size_t(n_rows), 0u, 32u, st));
  hipLaunchKernelGGL(k_tg_rank, blocks_for(n_rows), dim3(T), 0, st, n_rows, w.drow_b, w.rank);
  HIP_TRY(hipGetLastError());
  // merged column list of every group
  hipLaunchKernelGGL(k_tg_entry_keys, blocks_for(nnz), dim3(T), 0, st, n_rows, nnz, rowptr, colidx, w.rank, cbits,
                     w.keys_a);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::radix_sort_keys(w.prim, pb, w.keys_a, w.keys_b, size_t(nnz), 0u, unsigned(9 + cbits + gbits), st));
  hipLaunchKernelGGL(k_tg_heads, blocks_for(nnz), dim3(T), 0, st, nnz, w.keys_b, 9, w.head);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.head, w.pos, 0u, size_t(nnz), rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_tg_merged, blocks_for(nnz), dim3(T), 0, st, w.G, nnz, w.keys_b, w.head, w.pos, cbits, w.mcol,
                     w.gptr, w.counts);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_tg_group_stages, blocks_for(w.G + 1), dim3(T), 0, st, w.G, w.gptr, w.gns);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.gns, w.gstage0, 0u, size_t(w.G + 1), rocprim::plus<uint32_t>(), st));
  // per (stage, sub-tile): its columns, packed 16 to a step
  hipLaunchKernelGGL(k_tg_step_keys, blocks_for(nnz), dim3(T), 0, st, nnz, w.keys_b, w.head, w.pos, w.gptr, w.gstage0, cbits,
                     w.keys_a);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::radix_sort_keys(w.prim, pb, w.keys_a, w.keys_b, size_t(nnz), 0u, unsigned(5 + kSlotBits + rbits), st));
  hipLaunchKernelGGL(k_tg_heads, blocks_for(nnz), dim3(T), 0, st, nnz, w.keys_b, 5, w.head);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.head, w.pos, 0u, size_t(nnz), rocprim::plus<uint32_t>(), st));
  HIP_TRY(hipMemsetAsync(w.runcount, 0, size_t(w.cap_runs + 1) * 4, st));
  HIP_TRY(hipMemsetAsync(w.runclass, 0, size_t(w.cap_runs + 1) * 4, st));
  hipLaunchKernelGGL(k_tg_run_counts, blocks_for(nnz), dim3(T), 0, st, nnz, w.keys_b, w.head, w.pos, w.runcount,
                     w.runclass, w.runfirst);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_tg_run_steps, blocks_for(w.cap_runs + 1), dim3(T), 0, st, w.cap_runs, w.counts, w.gstage0, w.G,
                     w.runcount, w.nsteps);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.nsteps, w.step0, 0u, size_t(w.cap_runs + 1), rocprim::plus<uint32_t>(),
                                  st));
  // work items
  const int seg = g_seg_stages;
  hipLaunchKernelGGL(k_tg_nseg, blocks_for(w.G), dim3(T), 0, st, w.G, seg, w.gns, w.gstage0, w.nseg, w.lseg, w.lflag);
  HIP_TRY(hipGetLastError());
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.nseg, w.first, 0u, size_t(w.G), rocprim::plus<uint32_t>(), st));
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.lseg, w.slot0, 0u, size_t(w.G), rocprim::plus<uint32_t>(), st));
  pb = w.prim_bytes;
  HIP_TRY(rocprim::exclusive_scan(w.prim, pb, w.lflag, w.lidx, 0u, size_t(w.G), rocprim::plus<uint32_t>(), st));
  hipLaunchKernelGGL(k_tg_totals, dim3(1), dim3(64), 0, st, w.G, w.gstage0, w.step0, w.nseg, w.first, w.lseg, w.slot0,
                     w.lflag, w.lidx, seg, n_rows, w.counts);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

int sgnn_tilegroup_fill(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t stages, int64_t steps, int64_t items,
                        int64_t long_groups, void* plan, size_t plan_bytes, void* workspace, size_t workspace_bytes,
                        void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_rows <= 0 || n_cols <= 0 || nnz <= 0) return fail(SGNN_ERR_INVALID, "empty matrix");
  if (!plan || !workspace) return fail(SGNN_ERR_INVALID, "null pointer");
  TgBuild w = carve_tg_build(workspace, n_rows, nnz);
  if (workspace_bytes < w.total) return fail(SGNN_ERR_WORKSPACE, "tile-group build workspace too small");
  if (stages <= 0 || stages > w.cap_stages || steps <= 0 || items <= 0 || items > w.cap_items || long_groups < 0 ||
      long_groups > w.G)
    return fail(SGNN_ERR_INVALID, "counts do not match the analysed matrix");
  TgPlan p = carve_tg_plan(plan, n_rows, stages, steps, items, long_groups);
  if (plan_bytes < p.total) return fail(SGNN_ERR_WORKSPACE, "tile-group plan buffer too small");
  const unsigned T = 256;
  HIP_TRY(hipMemcpyAsync(p.counts, w.counts, 128, hipMemcpyDeviceToDevice, st));   // counts[7] = n_rows: the SpMM kernel checks its launch against them
  hipLaunchKernelGGL(k_tg_grows, blocks_for(p.G * kRows), dim3(T), 0, st, n_rows, p.G * kRows, w.drow_b, p.grows);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_tg_copy_u32, blocks_for(p.G + 1), dim3(T), 0, st, p.G + 1, w.gstage0, p.gstage0, p.G + 1);
  HIP_TRY(hipGetLastError());
  const int64_t runs = stages * kSubs;
  hipLaunchKernelGGL(k_tg_copy_u32, blocks_for(runs + 1 + 128), dim3(T), 0, st, runs + 1, w.step0, p.step0,
                     runs + 1 + 128);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemsetAsync(p.stage_col + stages * kStageCols, 0, 3 * kStageCols * 4, st));
  hipLaunchKernelGGL(k_tg_stage_cols, blocks_for(stages * kStageCols), dim3(T), 0, st, p.G, stages, w.gptr, w.gstage0,
                     w.mcol, p.stage_col);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemsetAsync(p.step_slots, 0, size_t(steps + kPadSteps) * 32, st));
  HIP_TRY(hipMemsetAsync(p.step_bits, 0, size_t(steps + kPadSteps) * 64, st));
  hipLaunchKernelGGL(k_tg_fill_steps, blocks_for(nnz), dim3(T), 0, st, nnz, w.keys_b, w.head, w.pos, w.runfirst,
                     w.runclass, w.step0, p.step_slots, reinterpret_cast<uint32_t*>(p.step_bits));
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_tg_items, blocks_for(p.G), dim3(T), 0, st, p.G, w.gns, w.gstage0, w.nseg, w.first, w.slot0,
                     w.lidx, w.itmp, w.cap_items, p.long_group, p.long_slot, p.long_nseg);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_tg_item_keys, blocks_for(items), dim3(T), 0, st, items, w.itmp, w.cap_items, p.stage_col,
                     int(g_order_cols.load()), nnz, p.counts, w.ikey_a);
  HIP_TRY(hipGetLastError());
  size_t pb = w.prim_bytes;
  HIP_TRY(rocprim::radix_sort_keys(w.prim, pb, w.ikey_a, w.ikey_b, size_t(items), 0u, 64u, st));
  hipLaunchKernelGGL(k_tg_item_place, blocks_for(items), dim3(T), 0, st, items, w.ikey_b, w.itmp, w.cap_items,
                     p.item_group, p.item_s0, p.item_s1, p.item_out);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

size_t sgnn_tilegroup_planes_bytes(int64_t n_cols, int64_t F) {
  if (n_cols <= 0 || F <= 0) return 0;
  const int64_t slices = (F + kSlice - 1) / kSlice;
  return align_up(size_t(slices) * size_t(n_cols) * kRowBytes + 256);
}

size_t sgnn_tilegroup_state_bytes(int64_t F) {
  if (F <= 0) return 0;
  const int64_t Fpad = (F + kSlice - 1) / kSlice * kSlice;
  return align_up(tg_state_live_bytes(Fpad) + size_t(kAmaxBlocks) * Fpad * sizeof(uint32_t));
}

namespace {
// per-feature maxima of every rstep-th row (rstep = 1: all of them) into state->amax
int tg_amax_launch(int64_t n, int64_t F, const float* scale_row, const float* X, int64_t ldx, TgState* s, int64_t rstep,
                   int only_redo, hipStream_t st) {
  const int Fpad = int((F + kSlice - 1) / kSlice * kSlice);
  const int64_t rows = (n + rstep - 1) / rstep;
  const int64_t per_row = (F + 1) / 2;
  const unsigned T = unsigned(per_row >= 1024 ? 1024 : (per_row <= 256 ? 256 : (per_row + kWave - 1) / kWave * kWave));
  // two workgroups per CU, each with many rows; their maxima meet in the fold kernel
  const int64_t quads = (rows + 3) / 4;
  const int blocks = int(quads < kAmaxBlocks ? quads : kAmaxBlocks);
  hipLaunchKernelGGL(k_tg_amax, dim3(unsigned(blocks)), dim3(T), 0, st, rows, F, Fpad, scale_row, X, ldx, s, rstep, only_redo);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_tg_amax_fold, dim3(unsigned(Fpad / 64)), dim3(1024), 0, st, int(F), Fpad, blocks, s, only_redo);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

// bounds from the maxima (times 2^margin), then the planes
int tg_planes_launch(int64_t n, int64_t F, const float* scale_row, const float* X, int64_t ldx, void* planes,
                     int64_t plane_rows, int64_t operand_rows, TgState* s, int margin, int only_redo, hipStream_t st) {
  const int64_t slices = (F + kSlice - 1) / kSlice;
  const int Fpad = int(slices * kSlice);
  hipLaunchKernelGGL(k_tg_state_init, dim3(unsigned((Fpad + 255) / 256)), dim3(256), 0, st, s,
                     tg_tiny_limit(operand_rows * F), Fpad, margin, only_redo);
  HIP_TRY(hipGetLastError());
  // Narrow operands have little work per row: fewer, longer-lived workgroups measure faster there
  // (Reddit rows, 128 features: 0.073 -> 0.059 ms for max + split); wide operands and those with
  // millions of rows prefer more of them (10 M rows of 128: 2.09 ms with 16384, 2.30 ms with 4096).
  const int64_t max_blocks = g_split_blocks.load() > 0 ? g_split_blocks.load() : (F <= 128 ? std::min<int64_t>(16384, std::max<int64_t>(4096, n / 64)) : 16384);
  const unsigned row_blocks = unsigned(n < max_blocks ? n : max_blocks);
  const unsigned split_threads = unsigned(g_split_threads.load() > 0 ? g_split_threads.load() : 256);
  hipLaunchKernelGGL(k_tg_split, dim3(row_blocks), dim3(split_threads), 0, st, n, F, slices, scale_row, X, ldx,
                     static_cast<char*>(planes), plane_rows * int64_t(kRowBytes), s, only_redo);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}
}  // namespace

int sgnn_tilegroup_amax_f32(int64_t n, int64_t F, const float* scale_row, const float* X, int64_t ldx, void* state,
                            void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0 || F <= 0) return fail(SGNN_ERR_INVALID, "empty operand");
  if (!X || !state) return fail(SGNN_ERR_INVALID, "null pointer");
  if (ldx < F) return fail(SGNN_ERR_INVALID, "leading dimension smaller than F");
  TgState* s = static_cast<TgState*>(state);
  const int Fpad = int((F + kSlice - 1) / kSlice * kSlice);
  return tg_amax_launch(n, F, scale_row, X, ldx, s, 1, 0, st);
}

int sgnn_tilegroup_planes_f32(int64_t n, int64_t F, const float* scale_row, const float* X, int64_t ldx, void* planes,
                              int64_t plane_rows, int64_t operand_rows, void* state, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n <= 0 || F <= 0) return fail(SGNN_ERR_INVALID, "empty operand");
  if (!X || !planes || !state) return fail(SGNN_ERR_INVALID, "null pointer");
  if (ldx < F) return fail(SGNN_ERR_INVALID, "leading dimension smaller than F");
  if (plane_rows < n || operand_rows < n) return fail(SGNN_ERR_INVALID, "the planes hold fewer rows than are written");
  if (reinterpret_cast<uintptr_t>(planes) % 16) return fail(SGNN_ERR_INVALID, "planes must be 16-byte aligned");
  return tg_planes_launch(n, F, scale_row, X, ldx, planes, plane_rows, operand_rows, static_cast<TgState*>(state), 0, 0, st);
}

// Operands of kSampleMinRows rows and more are split at PROVISIONAL scales: the per-feature
// maxima of every kSampleStep-th row, times 2^kSampleMargin, stand in for the maxima of all rows,
// which saves the pass that reads the whole operand once just to find them (a third of the
// max + split time; 8 % of a hop on R-MAT 10 M, 6 % on ogbn-products, 3 % on the Reddit
// shape).  The split pass tests every element against its feature's bound anyway; one beyond it
// (an outlier the sample missed: more than 2^kSampleMargin times the sampled maximum) raises
// `redo` in the state, and the exact passes -- maxima of all rows, bounds without margin, the
// split again -- which are launched behind the provisional ones in any case, then run instead of
// returning at once.  No host synchronisation either way.  What the margin costs: up to
// kSampleMargin of the 29 bits of headroom below the bound in which an element keeps 22 bits.
constexpr int64_t kSampleMinRows = 65536, kSampleStep = 32;
constexpr int kSampleMargin = 3;

int sgnn_tilegroup_split_f32(int64_t n_cols, int64_t F, const float* scale_row, const float* X, int64_t ldx,
                             void* planes, size_t planes_bytes, void* state, void* stream) {
  if (n_cols <= 0 || F <= 0) return fail(SGNN_ERR_INVALID, "empty operand");
  if (planes_bytes < sgnn_tilegroup_planes_bytes(n_cols, F)) return fail(SGNN_ERR_WORKSPACE, "planes buffer too small");
  if (n_cols < kSampleMinRows || g_no_sample.load()) {
    int rc = sgnn_tilegroup_amax_f32(n_cols, F, scale_row, X, ldx, state, stream);
    if (rc != SGNN_OK) return rc;
    return sgnn_tilegroup_planes_f32(n_cols, F, scale_row, X, ldx, planes, n_cols, n_cols, state, stream);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!X || !planes || !state) return fail(SGNN_ERR_INVALID, "null pointer");
  if (ldx < F) return fail(SGNN_ERR_INVALID, "leading dimension smaller than F");
  if (reinterpret_cast<uintptr_t>(planes) % 16) return fail(SGNN_ERR_INVALID, "planes must be 16-byte aligned");
  TgState* s = static_cast<TgState*>(state);
  const int Fpad = int((F + kSlice - 1) / kSlice * kSlice);
  int rc = tg_amax_launch(n_cols, F, scale_row, X, ldx, s, kSampleStep, 0, st);
  if (rc != SGNN_OK) return rc;
  rc = tg_planes_launch(n_cols, F, scale_row, X, ldx, planes, n_cols, n_cols, s, kSampleMargin, 0, st);
  if (rc != SGNN_OK) return rc;
  // the exact passes, which do nothing unless the provisional split raised `redo`
  rc = tg_amax_launch(n_cols, F, scale_row, X, ldx, s, 1, 1, st);
  if (rc != SGNN_OK) return rc;
  return tg_planes_launch(n_cols, F, scale_row, X, ldx, planes, n_cols, n_cols, s, 0, 1, st);
}

size_t sgnn_spmm_tilegroup_workspace_bytes(int64_t partial_slots, int64_t F) {
  if (partial_slots < 0 || F <= 0) return 0;
  const int64_t Fpad = (F + kSlice - 1) / kSlice * kSlice;
  return align_up(size_t(partial_slots) * kRows * size_t(Fpad) * 4 + 256);
}

int sgnn_spmm_tilegroup_slices_f32(int64_t n_rows, int64_t n_cols, int64_t F, const void* plan, size_t plan_bytes,
                                   int64_t stages, int64_t steps, int64_t items, int64_t partial_slots, int64_t long_groups,
                                   const float* dinv_row, float w_off, const float* corr, const void* planes,
                                   size_t planes_bytes,
                                   void* state, float* Y, int64_t ldy, int epilogue, void* workspace,
                                   size_t workspace_bytes, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                                   const float* vals, const float* dinv_col, const float* X, int64_t ldx,
                                   void* csr_workspace, size_t csr_workspace_bytes, int64_t x_row_offset, const float* addend,
                                   int64_t ld_addend, void* planes_out, int64_t planes_out_rows, int64_t out_row_offset,
                                   void* state_out, int adapt, int64_t slice_begin, int64_t slice_count, int reserve_cus,
                                   void* stream) {
  if (n_rows <= 0 || n_cols <= 0 || F <= 0) return fail(SGNN_ERR_INVALID, "empty problem");
  if (!plan || !planes || !state) return fail(SGNN_ERR_INVALID, "null pointer");
  if (!planes_out && !Y) return fail(SGNN_ERR_INVALID, "neither Y nor planes_out given");
  if (!planes_out && ldy < F) return fail(SGNN_ERR_INVALID, "leading dimension smaller than F");
  if (epilogue != SGNN_EPI_SCALE_ROW && epilogue != SGNN_EPI_SCALE_ROW2)
    return fail(SGNN_ERR_INVALID, "unknown epilogue mode");
  if (corr != nullptr && (x_row_offset < 0 || x_row_offset + n_rows > n_cols || (X != nullptr && ldx < F)))
    return fail(SGNN_ERR_INVALID, "a diagonal correction needs the rows' own operand rows inside the operand");
  if (addend != nullptr && ld_addend < F) return fail(SGNN_ERR_INVALID, "leading dimension of the addend smaller than F");
  if (planes_out != nullptr) {
    if (!state_out) return fail(SGNN_ERR_INVALID, "planes_out needs state_out");
    if (out_row_offset < 0 || out_row_offset + n_rows > planes_out_rows)
      return fail(SGNN_ERR_INVALID, "the written rows do not fit planes_out");
    if (reinterpret_cast<uintptr_t>(planes_out) % 16) return fail(SGNN_ERR_INVALID, "planes_out must be 16-byte aligned");
    if (rowptr != nullptr) return fail(SGNN_ERR_INVALID, "a hop that writes planes has no fall-back of its own: the chain's caller launches it");
    if (planes_out == planes) return fail(SGNN_ERR_INVALID, "planes_out must not be the operand");
  }
  if (stages <= 0 || steps <= 0 || items <= 0 || partial_slots < 0 || long_groups < 0)
    return fail(SGNN_ERR_INVALID, "counts do not match the plan");
  TgPlan p = carve_tg_plan(const_cast<void*>(plan), n_rows, stages, steps, items, long_groups);
  if (plan_bytes < p.total) return fail(SGNN_ERR_WORKSPACE, "tile-group plan buffer too small");
  if (planes_bytes < sgnn_tilegroup_planes_bytes(n_cols, F)) return fail(SGNN_ERR_WORKSPACE, "planes buffer too small");
  if (partial_slots > 0 && (!workspace || workspace_bytes < sgnn_spmm_tilegroup_workspace_bytes(partial_slots, F)))
    return fail(SGNN_ERR_WORKSPACE, "tile-group spmm workspace too small");
  if (reinterpret_cast<uintptr_t>(workspace) % 16) return fail(SGNN_ERR_INVALID, "workspace must be 16-byte aligned");
  const int64_t all_slices = (F + kSlice - 1) / kSlice;
  if (slice_count <= 0) slice_count = all_slices - slice_begin;
  if (slice_begin < 0 || slice_begin >= all_slices || slice_begin + slice_count > all_slices)
    return fail(SGNN_ERR_INVALID, "slice range outside the operand");
  if (reserve_cus < 0) return fail(SGNN_ERR_INVALID, "negative number of reserved CUs");
  const bool part = slice_begin != 0 || slice_count != all_slices;
  if (part && rowptr != nullptr) return fail(SGNN_ERR_INVALID, "a launch over a range of slices has no fall-back");
  const int64_t f_base = slice_begin * kSlice;                                  // the range's first feature
  const int64_t F_all = F, Fpad_all = all_slices * kSlice;
  F = (F_all - f_base < slice_count * kSlice) ? F_all - f_base : slice_count * kSlice;   // features of the range
  const int64_t slices = slice_count;
  if (slices * items >= (int64_t(1) << 31)) return fail(SGNN_ERR_INVALID, "too many workgroups for one launch");
  TgArgs a{};
  a.plan_counts = p.counts;
  a.grows = p.grows; a.stage_col = p.stage_col; a.step0 = p.step0;
  a.step_slots = p.step_slots; a.step_bits = p.step_bits;
  a.item_group = p.item_group; a.item_s0 = p.item_s0; a.item_s1 = p.item_s1; a.item_out = p.item_out;
  a.long_group = p.long_group; a.long_slot = p.long_slot; a.long_nseg = p.long_nseg;
  a.slice_bytes = n_cols * kRowBytes;
  a.slice_bytes_out = planes_out_rows * int64_t(kRowBytes);
  a.planes = static_cast<const char*>(planes) + slice_begin * a.slice_bytes;
  a.state = static_cast<TgState*>(state);
  a.queue = &static_cast<TgState*>(state)->queue;
  a.amax_in = a.state->amax + f_base;
  a.bound_in = a.state->amax + Fpad_all + f_base;   // (addresses only: the bounds follow the maxima)
  a.dinv_row = dinv_row; a.Y = Y ? Y + f_base : nullptr; a.partial = static_cast<float*>(workspace);
  a.corr = corr; a.X = X ? X + f_base : nullptr; a.dinv_col = dinv_col; a.ldx = ldx;
  a.addend = addend ? addend + f_base : nullptr; a.ld_addend = ld_addend;
  a.planes_out = planes_out ? static_cast<char*>(planes_out) + slice_begin * a.slice_bytes_out : nullptr;
  a.state_out = static_cast<TgState*>(state_out);
  if (a.state_out != nullptr) {
    a.amax_out = a.state_out->amax + f_base;
    a.bound_out = a.state_out->amax + Fpad_all + f_base;
  }
  a.x_row_offset = x_row_offset; a.out_row_offset = out_row_offset; a.n_rows = n_rows;
  a.ldy = ldy;
  a.items = int32_t(items); a.n_long = int32_t(long_groups);
  a.stages = int32_t(stages); a.steps = int32_t(steps);
  a.F = int32_t(F); a.Fpad = int32_t(slices * kSlice); a.epilogue = epilogue; a.w_off = w_off;
  a.mode = g_tg_mode;
  a.slices = int32_t(slices);
  a.tiny_limit_out = tg_tiny_limit(planes_out_rows * F_all);
  a.adapt = adapt ? 1 : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // per device: the attribute belongs to the device's copy of the kernel, and the number of
  // workgroups to the device's CUs (a process may drive more than one GPU, from more than one thread)
  static std::atomic<int> cus_of[64];
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(SGNN_ERR_INVALID, "device ordinal out of range");
  int n_cus = cus_of[dev].load(std::memory_order_acquire);
  if (n_cus == 0) {   // (two threads may both get here: they set the same attribute and store the same count)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tg_spmm),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    HIP_TRY(hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (n_cus <= 0) n_cus = 256;
    cus_of[dev].store(n_cus, std::memory_order_release);
  }
  // (the next operand's state is cleared by the launch that covers the operand's first slice:
  // launches over ranges of slices go in ascending order on one stream)
  if (planes_out != nullptr && slice_begin == 0) HIP_TRY(hipMemsetAsync(state_out, 0, tg_state_live_bytes(Fpad_all), st));
  // (the work counters -- one for the launch, one per XCD -- are zero between launches: the state
  // starts cleared and the last workgroup to leave a launch's work loop zeroes them again)
  // one workgroup per CU (a workgroup takes the whole LDS of one), fewer when there is less work
  sgnn::kernel_timer_begin(st);
  // (reserve_cus: CUs left to kernels of other streams -- a collective that runs beside this
  // launch finds no CU once the workgroups of this one hold them all until the work is done)
  const int64_t mine = n_cus - reserve_cus > 8 ? n_cus - reserve_cus : 8;
  const int64_t wgs = (items * slices < mine || (a.mode & 128)) ? items * slices : mine;
  hipLaunchKernelGGL(k_tg_spmm, dim3(unsigned(wgs)), dim3(kWaves * kWave), kLdsBytes, st, a);
  HIP_TRY(hipGetLastError());
  sgnn::kernel_timer_end(st);
  if (long_groups > 0) {
    const int quads = tg_combine_quads(a.F, a.Fpad, planes_out != nullptr);
    // 64 workgroups per split group where there are few groups, fewer where there are thousands
    // (every workgroup ends in one atomic per feature when it writes planes)
    unsigned gy = unsigned(long_groups >= 16384 ? 1 : (16384 / long_groups > 64 ? 64 : 16384 / long_groups));
    if (g_combine_gy.load() > 0) gy = unsigned(g_combine_gy.load());
    hipLaunchKernelGGL(k_tg_combine, dim3(unsigned(long_groups), gy), dim3(tg_combine_threads(quads)), 0, st, a);
    HIP_TRY(hipGetLastError());
  }
  if (rowptr != nullptr) {
    // fall-back for an X the planes cannot represent: the plain CSR kernel on the original
    // operand, behind the same flag (its kernels return at once while the flag is 0)
    if (!colidx || !X || !Y) return fail(SGNN_ERR_INVALID, "null pointer");
    if (addend != nullptr) return fail(SGNN_ERR_INVALID, "a launch with an addend has no fall-back");
    return spmm_csr_guarded(n_rows, n_cols, nnz, F_all, rowptr, colidx, vals, dinv_row, dinv_col, X, ldx, Y, ldy, epilogue,
                            csr_workspace, csr_workspace_bytes, &static_cast<const TgState*>(state)->flag, st);
  }
  return SGNN_OK;
}

int sgnn_spmm_tilegroup_f32(int64_t n_rows, int64_t n_cols, int64_t F, const void* plan, size_t plan_bytes,
                            int64_t stages, int64_t steps, int64_t items, int64_t partial_slots, int64_t long_groups,
                            const float* dinv_row, float w_off, const float* corr, const void* planes,
                            size_t planes_bytes,
                            void* state, float* Y, int64_t ldy, int epilogue, void* workspace,
                            size_t workspace_bytes, int64_t nnz, const int32_t* rowptr, const int32_t* colidx,
                            const float* vals, const float* dinv_col, const float* X, int64_t ldx,
                            void* csr_workspace, size_t csr_workspace_bytes, int64_t x_row_offset, const float* addend,
                            int64_t ld_addend, void* planes_out, int64_t planes_out_rows, int64_t out_row_offset,
                            void* state_out, int adapt, void* stream) {
  return sgnn_spmm_tilegroup_slices_f32(n_rows, n_cols, F, plan, plan_bytes, stages, steps, items, partial_slots, long_groups,
                                        dinv_row, w_off, corr, planes, planes_bytes, state, Y, ldy, epilogue, workspace,
                                        workspace_bytes, nnz, rowptr, colidx, vals, dinv_col, X, ldx, csr_workspace,
                                        csr_workspace_bytes, x_row_offset, addend, ld_addend, planes_out, planes_out_rows,
                                        out_row_offset, state_out, adapt, 0, 0, 0, stream);
}

int sgnn_tilegroup_chain_f32(int K, int64_t n, int64_t F, const void* plan, size_t plan_bytes, int64_t stages,
                             int64_t steps, int64_t items, int64_t partial_slots, int64_t long_groups, const float* dinv,
                             float w_off, const float* corr, const float* X, int64_t ldx, float* Y, int64_t ldy,
                             void* planes_a, void* planes_b, size_t planes_bytes, void* states, size_t states_bytes,
                             void* workspace, size_t workspace_bytes, int64_t nnz, const int32_t* rowptr,
                             const int32_t* colidx, const float* vals, void* csr_workspace, size_t csr_workspace_bytes,
                             void* stream) {
  if (K < 1) return fail(SGNN_ERR_INVALID, "K must be at least 1");
  if (n <= 0 || F <= 0) return fail(SGNN_ERR_INVALID, "empty problem");
  if (!dinv || !X || !Y || !planes_a || !states || !rowptr || !colidx) return fail(SGNN_ERR_INVALID, "null pointer");
  if (K > 1 && !planes_b) return fail(SGNN_ERR_INVALID, "a second planes buffer is required when K > 1");
  const size_t sb = sgnn_tilegroup_state_bytes(F);
  if (states_bytes < sb * size_t(K)) return fail(SGNN_ERR_WORKSPACE, "one state block per hop is required");
  if (planes_bytes < sgnn_tilegroup_planes_bytes(n, F)) return fail(SGNN_ERR_WORKSPACE, "planes buffer too small");
  char* sp = static_cast<char*>(states);
  // operand 0: utils.py:93 applied by the split pass
  int rc = sgnn_tilegroup_split_f32(n, F, dinv, X, ldx, planes_a, planes_bytes, sp, stream);
  if (rc != SGNN_OK) return rc;
  for (int h = 0; h < K; ++h) {
    const bool last = h == K - 1;
    void* src = (h % 2 == 0) ? planes_a : planes_b;
    void* dst = (h % 2 == 0) ? planes_b : planes_a;
    rc = sgnn_spmm_tilegroup_f32(n, n, F, plan, plan_bytes, stages, steps, items, partial_slots, long_groups, dinv,
                                 w_off, corr, src, planes_bytes, sp + sb * h, last ? Y : nullptr, ldy,
                                 last ? SGNN_EPI_SCALE_ROW : SGNN_EPI_SCALE_ROW2, workspace, workspace_bytes, nnz, nullptr,
                                 nullptr, nullptr, h == 0 ? dinv : nullptr, h == 0 ? X : nullptr, ldx, nullptr, 0, 0,
                                 nullptr, 0, last ? nullptr : dst, n, 0, last ? nullptr : sp + sb * (h + 1), 1, stream);
    if (rc != SGNN_OK) return rc;
  }
  // Fall-back, behind the last operand's flag (a refusal anywhere in the chain is handed down to
  // it): the whole chain again through the plain CSR kernel on the fp32 operand.  Its
  // intermediates live in the planes' memory, which is dead once the flag is up; while the flag
  // is 0 the guarded kernels return at once and touch nothing.
  const uint32_t* guard = &reinterpret_cast<const TgState*>(sp + sb * (K - 1))->flag;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float* src = X;
  int64_t lds = ldx;
  for (int h = 0; h < K; ++h) {
    const bool last = h == K - 1;
    float* dst = last ? Y : static_cast<float*>((h % 2 == 0) ? planes_a : planes_b);
    const int64_t ldd = last ? ldy : F;
    rc = spmm_csr_guarded(n, n, nnz, F, rowptr, colidx, vals, dinv, h == 0 ? dinv : nullptr, src, lds, dst, ldd,
                          last ? SGNN_EPI_SCALE_ROW : SGNN_EPI_SCALE_ROW2, csr_workspace, csr_workspace_bytes, guard, st);
    if (rc != SGNN_OK) return rc;
    src = dst;
    lds = ldd;
  }
  return SGNN_OK;
}

}  // extern "C"
