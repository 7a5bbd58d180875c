// Shared by the translation units of libsgnn_hip.so: error reporting, small host
// helpers and the buffer-load primitives the SpMM kernels are built on.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "sgnn_hip.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LLVM's raw buffer loads, bound by name (the clang builtins for the 8- and
// 16-byte forms return a splat of the first dword on this toolchain).
__device__ float raw_buffer_load_f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.f32");
__device__ f32x2 raw_buffer_load_f32x2(i32x4 rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.v2f32");
__device__ f32x4 raw_buffer_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.v4f32");

namespace sgnn {

// records the calling thread's error message and returns `code`
int fail(int code, const std::string& msg);

// tuning hook: merged columns per row-group work item for plans built afterwards (0 = policy)
void set_rowgroup_segment(int cols);
void set_rowgroup_order(int cols);   // column-block width the work items are ordered by (0 = default)
void set_rowgroup_R(int R);   // 8 or 16: group size for plans built afterwards (0 = policy)
void set_tilegroup_mode(int mode);        // diagnostic switches of the tile-group kernel (0 = normal)
void set_tilegroup_segment(int stages);   // stages per tile-group work item for plans built afterwards
void set_tilegroup_combine_rows(int gy);  // workgroups per split group of the tile-group combine pass (0 = built-in)
void set_tilegroup_no_sample(int v);      // 1: the tile-group split pass measures the maxima of all rows (no provisional scales)
void set_tilegroup_split_shape(int blocks, int threads);   // launch shape of the tile-group split pass (-1: leave, 0: built-in)
void set_tilegroup_order(int cols);       // tile-group items by column blocks of this width (0 = longest first)

// sgnn_spmm_csr_f32 whose kernels do nothing while the device word *guard is 0 (guard may
// be NULL: always run).  Used as the fall-back behind the tile-group kernel.
int spmm_csr_guarded(int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t F, const int32_t* rowptr,
                     const int32_t* colidx, const float* vals, const float* dinv_row, const float* dinv_col,
                     const float* X, int64_t ldx, float* Y, int64_t ldy, int epilogue, void* workspace,
                     size_t workspace_bytes, const uint32_t* guard, hipStream_t stream);

// Kernel timer (sgnn_tune_set("time_kernel", 1)): while it is on, every SpMM entry point
// brackets its main kernel (k_tg_spmm / k_rg_spmm / k_spmm; not the scale, split or combine
// kernels around it) with a pair of HIP events on the launch stream; sgnn_kernel_time_ms
// adds them up.  Off, both calls return at once and create nothing.
void kernel_timer_begin(hipStream_t stream);
void kernel_timer_end(hipStream_t stream);

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return ::sgnn::fail(SGNN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
  } while (0)

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

inline int ceil_log2(int64_t n) {
  int b = 1;
  while ((int64_t(1) << b) < n) ++b;
  return b;
}

// carve consecutive 256-byte aligned regions out of one scratch buffer; with a
// null base it only measures
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base(static_cast<char*>(b)) {}
  template <typename T>
  T* take(size_t count) {
    char* q = base ? base + off : nullptr;
    off += align_up(count * sizeof(T));
    return reinterpret_cast<T*>(q);
  }
};

constexpr int kWave = 64;

// A buffer load's range check looks at voffset (+ the instruction offset) only.  Lanes
// that must not touch memory get this voffset: it lies beyond every descriptor the
// kernels build (those span less than kBigX bytes), so the load returns 0.
constexpr int kOobVoffset = int(0xFFFFF000u);
// Operands of at least this many bytes are addressed by re-basing the descriptor on
// the fetched row instead of a 32-bit byte offset.
constexpr int64_t kBigX = (int64_t(1) << 32) - 8192;

// untyped raw buffer over [base, base + bytes): stride 0, range-checked
__device__ __forceinline__ i32x4 make_rsrc(const void* base, uint32_t bytes) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  i32x4 r = {int(uint32_t(b)), int(uint32_t(b >> 32) & 0xffffu), int(bytes), 0x00020000};
  return r;
}

template <int V>
__device__ __forceinline__ void buffer_load_vec(i32x4 rsrc, int voff, int soff, float (&r)[V]) {
  if constexpr (V == 1) {
    r[0] = raw_buffer_load_f32(rsrc, voff, soff, 0);
  } else if constexpr (V == 2) {
    f32x2 x = raw_buffer_load_f32x2(rsrc, voff, soff, 0);
    r[0] = x[0];
    r[1] = x[1];
  } else {
    f32x4 x = raw_buffer_load_f32x4(rsrc, voff, soff, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = x[i];
  }
}

template <int V>
__device__ __forceinline__ void store_vec(float* p, const float (&r)[V], int aligned_for) {
  if constexpr (V == 1) {
    *p = r[0];
  } else {
    if (aligned_for >= V) {
      typedef float vt __attribute__((ext_vector_type(V)));
      vt v;
#pragma unroll
      for (int i = 0; i < V; ++i) v[i] = r[i];
      *reinterpret_cast<vt*>(p) = v;
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) p[i] = r[i];
    }
  }
}

}  // namespace sgnn
