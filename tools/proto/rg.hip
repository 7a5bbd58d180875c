// Prototype: row-group SpMM.  R output rows share every fetched row of X.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ f32x2 raw_buffer_load_f32x2(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");

__device__ __forceinline__ i32x4 make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  i32x4 r = {int(uint32_t(b)), int(uint32_t(b >> 32) & 0xffffu), int(bytes), 0x00020000};
  return r;
}

struct RgArgs {
  const int32_t* item_group;  // [items]
  const int32_t* item_m0;     // [items]
  const int32_t* item_m1;     // [items]
  const int32_t* item_out;    // [items] -1: write Y, else partial slot
  const int32_t* grows;       // [groups * R]
  const int32_t* gcol;        // [merged]
  const uint32_t* gmask;      // [merged]
  const float* dinv;
  const float* X;             // pre-scaled
  float* Y;
  float* partial;             // [slots * R * F]
  int32_t items, F, row_bytes;
  uint32_t x_bytes;
};

template <int R, int C, int U, int ABL>
__global__ __launch_bounds__(256) void k_rg(RgArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (item >= a.items) return;
  const int g = a.item_group[item];
  const int m0 = a.item_m0[item], m1 = a.item_m1[item];
  if (a.item_out[item] == -2) return;
  const i32x4 rsrc = make_rsrc(a.X, a.x_bytes);
  int boff[C];
#pragma unroll
  for (int c = 0; c < C; ++c) boff[c] = (blockIdx.y * C * 128 + (c * 64 + lane) * 2) * 4;
  f32x2 acc[R][C];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int c = 0; c < C; ++c) acc[r][c] = f32x2{0.f, 0.f};

  int32_t ncols = 0; uint32_t nmask = 0;
  if (m0 + lane < m1) { ncols = a.gcol[m0 + lane]; nmask = a.gmask[m0 + lane]; }
  for (int m = m0; m < m1; m += 64) {
    const int32_t cols = ncols; const uint32_t masks = nmask;
    const int cnt = min(64, m1 - m);
    if (m + 64 < m1) {
      const bool mine = m + 64 + lane < m1;
      ncols = mine ? a.gcol[m + 64 + lane] : 0;
      nmask = mine ? a.gmask[m + 64 + lane] : 0;
    }
    int k = 0;
    for (; k + U <= cnt; k += U) {
      f32x2 xv[U][C]; uint32_t mk[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int col = __builtin_amdgcn_readlane(cols, k + u);
        mk[u] = __builtin_amdgcn_readlane(masks, k + u);
        const int soff = col * a.row_bytes;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          if constexpr (ABL == 2) xv[u][c] = f32x2{float(soff), float(boff[c])};
          else xv[u][c] = raw_buffer_load_f32x2(rsrc, boff[c], soff, 0);
        }
      }
      if constexpr (ABL == 1) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int c = 0; c < C; ++c) asm volatile("" ::"v"(xv[u][c]));
      } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (mk[u] & (1u << r)) {
#pragma unroll
            for (int c = 0; c < C; ++c) acc[r][c] += xv[u][c];
          }
      }
    }
    for (; k < cnt; ++k) {
      f32x2 xv[C];
      const int col = __builtin_amdgcn_readlane(cols, k);
      const uint32_t mk = __builtin_amdgcn_readlane(masks, k);
      const int soff = col * a.row_bytes;
#pragma unroll
      for (int c = 0; c < C; ++c) xv[c] = raw_buffer_load_f32x2(rsrc, boff[c], soff, 0);
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (mk & (1u << r)) {
#pragma unroll
          for (int c = 0; c < C; ++c) acc[r][c] += xv[c];
        }
    }
  }
  const int out = a.item_out[item];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = a.grows[g * R + r];
    if (row < 0) continue;
    float* dst; float dr = 1.0f;
    if (out < 0) { dst = a.Y + int64_t(row) * a.F; dr = a.dinv[row]; }
    else dst = a.partial + (int64_t(out) * R + r) * a.F;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int f = blockIdx.y * C * 128 + (c * 64 + lane) * 2;
      if (f < a.F) { f32x2 o = acc[r][c] * dr; *reinterpret_cast<f32x2*>(dst + f) = o; }
    }
  }
}

extern "C" int rg_spmm(int R, int U, RgArgs* a, void* stream) {
  dim3 grid((a->items + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int C = (a->F + 127) / 128;
  const int abl = U / 100; U %= 100;
#define L(RR, CC, UU, AA) hipLaunchKernelGGL((k_rg<RR, CC, UU, AA>), grid, block, 0, st, *a)
  if (C == 5 && R == 16 && U == 2) { if (abl == 0) L(16, 5, 2, 0); else if (abl == 1) L(16, 5, 2, 1); else L(16, 5, 2, 2); }
  else if (C == 1 && R == 8 && U == 8) { if (abl == 0) L(8, 1, 8, 0); else if (abl == 1) L(8, 1, 8, 1); else L(8, 1, 8, 2); }
  else return -1;
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
