// The small dense pieces of one optimiser iteration of a decoupled layer
// (model/SGNN.py:417-432): the Adam update of the layer's few parameter
// tensors, the cross-entropy of the label head (model/SGNN.py:397-399) and the squared
// error against the embedding target (model/SGNN.py:80-86).  A batch is 512 rows, so every one of these is a handful of
// microseconds of arithmetic; issued as stock framework kernels they are ~25 launches of
// 3-17 us each (softmax, nll, reductions, a multi-tensor Adam pass per parameter group
// plus its step bookkeeping).  Here each is one launch.
//
// Sums are formed in a fixed order (per row by one wavefront whose lanes fold in halves,
// then over the rows by the last workgroup to finish), so a loss value does not depend on
// scheduling.
#include "common.h"

namespace {

constexpr int kMaxAdamTensors = 16;

struct AdamArgs {
  float* p[kMaxAdamTensors];
  const float* g[kMaxAdamTensors];
  float* m[kMaxAdamTensors];
  float* v[kMaxAdamTensors];
  int64_t end[kMaxAdamTensors];   // running element count after tensor k
  float wd[kMaxAdamTensors];
  float l1[kMaxAdamTensors];
  int count;
};

// state[0] = steps taken so far, state[1] = workgroups of the current launch that have
// finished.  Every workgroup reads state[0] before it counts itself in, and the last one
// to count itself in writes the new step number, so a replayed HIP graph advances by
// itself and no workgroup of a launch can see the number change under it.
__global__ __launch_bounds__(256) void k_adam(AdamArgs a, float lr, float beta1, float beta2, float eps,
                                              int64_t total, unsigned long long* state) {
  __shared__ float corr[2];
  __shared__ float* s_p[kMaxAdamTensors];
  __shared__ const float* s_g[kMaxAdamTensors];
  __shared__ float* s_m[kMaxAdamTensors];
  __shared__ float* s_v[kMaxAdamTensors];
  __shared__ int64_t s_end[kMaxAdamTensors];
  __shared__ float s_wd[kMaxAdamTensors];
  __shared__ float s_l1[kMaxAdamTensors];
  const unsigned long long step = state[0] + 1ull;
  if (threadIdx.x == 0) {
    corr[0] = float(1.0 - pow(double(beta1), double(step)));
    corr[1] = float(1.0 - pow(double(beta2), double(step)));
  }
#pragma unroll
  for (int k = 0; k < kMaxAdamTensors; ++k)      // constant indices: the table never leaves registers
    if (threadIdx.x == 64 + k) {
      const bool live = k < a.count;
      s_p[k] = a.p[k]; s_g[k] = a.g[k]; s_m[k] = a.m[k]; s_v[k] = a.v[k]; s_wd[k] = a.wd[k]; s_l1[k] = a.l1[k];
      s_end[k] = live ? a.end[k] : INT64_MAX;
    }
  __syncthreads();
  const float step_size = lr / corr[0];
  const float inv_sqrt_c2 = 1.f / sqrtf(corr[1]);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t i = int64_t(blockIdx.x) * 1024 + e * 256 + threadIdx.x;
    if (i >= total) break;
    int k = 0;
    while (i >= s_end[k]) ++k;
    const int64_t j = i - (k ? s_end[k - 1] : 0);
    float p = s_p[k][j];
    const float sign = p > 0.f ? 1.f : (p < 0.f ? -1.f : 0.f);
    const float g = s_g[k][j] + s_wd[k] * p + s_l1[k] * sign;   // penalties enter through their gradients
    float m = s_m[k][j], v = s_v[k][j];
    m = m + (1.f - beta1) * (g - m);
    v = beta2 * v + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) * inv_sqrt_c2 + eps;
    p -= step_size * (m / denom);
    s_p[k][j] = p;
    s_m[k][j] = m;
    s_v[k][j] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned long long done = atomicAdd(&state[1], 1ull);
    if (done == gridDim.x - 1ull) {
      state[1] = 0ull;
      state[0] = step;
      __threadfence();
    }
  }
}

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
  return x;
}

constexpr int kLossWaves = 4;                    // rows per workgroup: one wavefront each
constexpr int kLossThreads = kLossWaves * 64;

// Adds row_loss[0..B) in a fixed order (lane l takes l, l + 64, ..., then the lanes fold in
// halves) once every workgroup of the launch has written its rows.  `ticket` counts the
// finished workgroups; the last one sums, stores and puts the ticket back to zero.
__device__ __forceinline__ void finish_loss(int64_t B, const float* row_loss, unsigned int* ticket, float factor,
                                            float* loss) {
  __shared__ bool last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1u;
    __threadfence();
  }
  __syncthreads();
  if (!last || threadIdx.x >= 64) return;
  float t = 0.f;
  for (int64_t r = threadIdx.x; r < B; r += 64) t += __builtin_nontemporal_load(row_loss + r);
  t = wave_sum(t);
  if (threadIdx.x == 0) {
    loss[0] = t * factor;
    *ticket = 0u;
  }
}

// One wavefront per row; its lanes stride over the classes.
__global__ __launch_bounds__(kLossThreads) void k_softmax_xent(int64_t B, int64_t C, const float* __restrict__ z,
                                                               int64_t ldz, const int64_t* __restrict__ labels,
                                                               const int64_t* __restrict__ idx, int64_t n_labels,
                                                               float* loss, float* __restrict__ dz, int64_t ldd,
                                                               float* row_loss, unsigned int* ticket) {
  const int lane = threadIdx.x & 63;
  const int64_t r = int64_t(blockIdx.x) * kLossWaves + (threadIdx.x >> 6);
  const float inv_b = 1.f / float(B);
  if (r < B) {
    const float* zr = z + r * ldz;
    float mx = -INFINITY;
    for (int64_t c = lane; c < C; c += 64) mx = fmaxf(mx, zr[c]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int64_t c = lane; c < C; c += 64) s += expf(zr[c] - mx);
    s = wave_sum(s);
    const int64_t at = idx ? idx[r] : r;
    const int64_t y = (at >= 0 && at < n_labels) ? labels[at] : -1;
    const float inv_s = 1.f / s;
    for (int64_t c = lane; c < C; c += 64)
      dz[r * ldd + c] = (expf(zr[c] - mx) * inv_s - (c == y ? 1.f : 0.f)) * inv_b;
    if (lane == 0) row_loss[r] = (y >= 0 && y < C) ? mx + logf(s) - zr[y] : 0.f;
  }
  finish_loss(B, row_loss, ticket, inv_b, loss);
}

// loss = scale * sum (a - t)^2, grad = 2 * scale * (a - t); t's rows optionally picked by idx.
__global__ __launch_bounds__(kLossThreads) void k_sq_err(int64_t B, int64_t D, const float* __restrict__ a,
                                                         int64_t lda, const float* __restrict__ t, int64_t ldt,
                                                         const int64_t* __restrict__ idx, int64_t t_rows, float scale,
                                                         float* loss, float* __restrict__ grad, int64_t ldg,
                                                         float* row_loss, unsigned int* ticket) {
  const int lane = threadIdx.x & 63;
  const int64_t r = int64_t(blockIdx.x) * kLossWaves + (threadIdx.x >> 6);
  if (r < B) {
    const int64_t at = idx ? idx[r] : r;
    const bool ok = at >= 0 && at < t_rows;
    const float* tr = t + (ok ? at : 0) * ldt;
    float acc = 0.f;
    for (int64_t c = lane; c < D; c += 64) {
      const float d = a[r * lda + c] - (ok ? tr[c] : 0.f);
      acc += d * d;
      grad[r * ldg + c] = 2.f * scale * d;
    }
    acc = wave_sum(acc);
    if (lane == 0) row_loss[r] = acc;
  }
  finish_loss(B, row_loss, ticket, scale, loss);
}

}  // namespace

extern "C" int sgnn_adam_step_f32(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const int64_t* sizes, const float* weight_decay,
                                  const float* l1_decay, float lr, float beta1, float beta2, float eps, uint64_t* state, void* stream) {
  using sgnn::fail;
  if (count < 0 || count > kMaxAdamTensors) return fail(SGNN_ERR_INVALID, "at most 16 tensors per Adam step");
  if (!state) return fail(SGNN_ERR_INVALID, "null pointer");
  if (!(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f))
    return fail(SGNN_ERR_INVALID, "need 0 <= beta < 1 and eps >= 0");
  if (count > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !sizes))
    return fail(SGNN_ERR_INVALID, "null pointer");
  AdamArgs a{};
  int64_t total = 0;
  for (int k = 0; k < count; ++k) {
    if (sizes[k] < 0) return fail(SGNN_ERR_INVALID, "negative size");
    if (sizes[k] > 0 && (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k]))
      return fail(SGNN_ERR_INVALID, "null pointer");
    total += sizes[k];
    a.p[k] = params[k]; a.g[k] = grads[k]; a.m[k] = exp_avg[k]; a.v[k] = exp_avg_sq[k];
    a.end[k] = total;
    a.wd[k] = weight_decay ? weight_decay[k] : 0.f;
    a.l1[k] = l1_decay ? l1_decay[k] : 0.f;
  }
  a.count = count;
  if (total == 0) return SGNN_OK;               // nothing to update: the step number stays
  const int64_t blocks = (total + 1023) / 1024;
  if (blocks > 0x7fffffff) return fail(SGNN_ERR_INVALID, "too many elements for one Adam step");
  hipLaunchKernelGGL(k_adam, dim3(unsigned(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), a, lr, beta1,
                     beta2, eps, total, reinterpret_cast<unsigned long long*>(state));
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

extern "C" size_t sgnn_loss_workspace_bytes(int64_t B) {
  return B > 0 ? size_t(B) * sizeof(float) : 0;
}

namespace {
int check_loss_ws(int64_t B, const void* ws, size_t ws_bytes, const void* ticket) {
  using sgnn::fail;
  if (B > 0x7fffffff) return fail(SGNN_ERR_INVALID, "more than 2^31-1 rows");
  if (!ticket || (B > 0 && !ws)) return fail(SGNN_ERR_INVALID, "null pointer");
  if (ws_bytes < sgnn_loss_workspace_bytes(B)) return fail(SGNN_ERR_WORKSPACE, "loss workspace too small");
  return SGNN_OK;
}
}  // namespace

extern "C" int sgnn_softmax_xent_f32(int64_t B, int64_t C, const float* logits, int64_t ldz, const int64_t* labels,
                                     int64_t n_labels, const int64_t* idx, float* loss, float* dlogits, int64_t ldd,
                                     void* workspace, size_t workspace_bytes, uint32_t* ticket, void* stream) {
  using sgnn::fail;
  if (B <= 0 || C <= 0) return fail(SGNN_ERR_INVALID, "need B > 0 and C > 0");
  if (ldz < C || ldd < C) return fail(SGNN_ERR_INVALID, "row stride smaller than the class count");
  if (n_labels < 0 || (!idx && n_labels < B)) return fail(SGNN_ERR_INVALID, "fewer labels than rows");
  if (!logits || !labels || !loss || !dlogits) return fail(SGNN_ERR_INVALID, "null pointer");
  if (int rc = check_loss_ws(B, workspace, workspace_bytes, ticket)) return rc;
  const unsigned blocks = unsigned((B + kLossWaves - 1) / kLossWaves);
  hipLaunchKernelGGL(k_softmax_xent, dim3(blocks), dim3(kLossThreads), 0, static_cast<hipStream_t>(stream), B, C,
                     logits, ldz, labels, idx, n_labels, loss, dlogits, ldd, static_cast<float*>(workspace), ticket);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

extern "C" int sgnn_sq_err_f32(int64_t B, int64_t D, const float* a, int64_t lda, const float* target, int64_t ldt,
                               int64_t target_rows, const int64_t* idx, float scale, float* loss, float* grad,
                               int64_t ldg, void* workspace, size_t workspace_bytes, uint32_t* ticket, void* stream) {
  using sgnn::fail;
  if (B <= 0 || D <= 0 || target_rows < 0) return fail(SGNN_ERR_INVALID, "need B > 0, D > 0");
  if (lda < D || ldt < D || ldg < D) return fail(SGNN_ERR_INVALID, "row stride smaller than the width");
  if (!idx && target_rows < B) return fail(SGNN_ERR_INVALID, "fewer target rows than rows");
  if (!loss || !a || !target || !grad) return fail(SGNN_ERR_INVALID, "null pointer");
  if (int rc = check_loss_ws(B, workspace, workspace_bytes, ticket)) return rc;
  const unsigned blocks = unsigned((B + kLossWaves - 1) / kLossWaves);
  hipLaunchKernelGGL(k_sq_err, dim3(blocks), dim3(kLossThreads), 0, static_cast<hipStream_t>(stream), B, D, a, lda,
                     target, ldt, idx, target_rows, scale, loss, grad, ldg, static_cast<float*>(workspace), ticket);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}
