// The small dense pieces of one optimiser iteration of a decoupled layer
// (model/SGNN.py:417-432): the Adam update of the layer's few parameter
// tensors, the cross-entropy of the label head (model/SGNN.py:397-399) and the squared
// error against the embedding target (model/SGNN.py:80-86).  A batch is 512 rows, so every one of these is a handful of
// microseconds of arithmetic; issued as stock framework kernels they are ~25 launches of
// 3-17 us each (softmax, nll, reductions, a multi-tensor Adam pass per parameter group
// plus its step bookkeeping).  Here each is one launch.
//
// Sums are formed in a fixed order (per row by one wavefront whose lanes fold in halves,
// then over the rows by one wavefront in a launch of its own), so a loss value does not
// depend on scheduling.
#include "common.h"

namespace {

constexpr int kMaxAdamTensors = 16;
constexpr int kAdamPerThread = 4;
// The workgroups of a launch count themselves in so that the last one can advance the step
// number.  One atomic per workgroup on a single word costs ~12 ns apiece once there are more
// than ~100 of them (measured: 512 workgroups add 5.5 us to a 2.4 us kernel), so they count on
// kAdamGroups words, one cache line each, and only the last of each group counts on the root.
constexpr int kAdamGroups = 64, kAdamLineWords = 16;

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

// state[0] = steps taken so far, state[1] = groups of the current launch that have finished,
// state[2 + 16 g] = workgroups of group g (blockIdx % 64 == g) that have finished.  Every
// workgroup reads state[0] before it counts itself in, and the last one of the last group
// writes the new step number, so a replayed HIP graph advances by itself and no workgroup of a
// launch can see the number change under it.
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
  if (threadIdx.x < 2) {        // beta^step by repeated squaring in double: ~40 multiplications, no libm call
    double base = threadIdx.x == 0 ? double(beta1) : double(beta2), r = 1.0;
    for (unsigned long long e = step; e != 0; e >>= 1) {
      if (e & 1ull) r *= base;
      base *= base;
    }
    corr[threadIdx.x] = float(1.0 - r);
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
  const auto update = [&](float* __restrict__ P, const float* __restrict__ G, float* __restrict__ Mm,
                          float* __restrict__ V, float wd, float l1, int64_t j) {
    float p = P[j];
    const float sign = p > 0.f ? 1.f : (p < 0.f ? -1.f : 0.f);
    const float g = G[j] + wd * p + l1 * sign;    // penalties enter through their gradients
    float m = Mm[j], v = V[j];
    m = m + (1.f - beta1) * (g - m);
    v = beta2 * v + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) * inv_sqrt_c2 + eps;
    p -= step_size * (m / denom);
    P[j] = p;
    Mm[j] = m;
    V[j] = v;
  };
  const int64_t first = int64_t(blockIdx.x) * (256 * kAdamPerThread);
  const int64_t last = first + 256 * kAdamPerThread - 1 < total ? first + 256 * kAdamPerThread - 1 : total - 1;
  int kf = 0, kl = 0;                             // tensors of the workgroup's first and last element
  while (first >= s_end[kf]) ++kf;
  while (last >= s_end[kl]) ++kl;
  if (kf == kl) {                                 // the usual case: one tensor, uniform pointers, no per-element search
    const int64_t base = kf ? s_end[kf - 1] : 0;
    float* P = s_p[kf]; const float* G = s_g[kf]; float* Mm = s_m[kf]; float* V = s_v[kf];
    const float wd = s_wd[kf], l1 = s_l1[kf];
#pragma unroll
    for (int e = 0; e < kAdamPerThread; ++e) {
      const int64_t i = first + e * 256 + threadIdx.x;
      if (i <= last) update(P, G, Mm, V, wd, l1, i - base);
    }
  } else {
#pragma unroll
    for (int e = 0; e < kAdamPerThread; ++e) {
      const int64_t i = first + e * 256 + threadIdx.x;
      if (i > last) break;
      int k = kf;
      while (i >= s_end[k]) ++k;
      update(s_p[k], s_g[k], s_m[k], s_v[k], s_wd[k], s_l1[k], i - (k ? s_end[k - 1] : 0));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {       // no data passes between workgroups, so the counts need no fence
    const unsigned g = blockIdx.x % kAdamGroups;
    const unsigned long long in_group = (gridDim.x - g + kAdamGroups - 1) / kAdamGroups;
    const unsigned long long groups = gridDim.x < unsigned(kAdamGroups) ? gridDim.x : kAdamGroups;
    unsigned long long* ticket = state + 2 + kAdamLineWords * g;
    if (atomicAdd(ticket, 1ull) == in_group - 1ull) {
      *ticket = 0ull;
      if (atomicAdd(&state[1], 1ull) == groups - 1ull) {
        state[1] = 0ull;
        state[0] = step;
      }
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

// loss[0] = factor * (row_loss[0] + ... + row_loss[B-1]) in a fixed order: lane l takes
// l, l + 64, ..., then the lanes fold in halves.  A launch of its own, so that the rows
// written by the workgroups of the previous launch are visible without a device-wide fence
// (which costs more than a launch when hundreds of workgroups issue it).
__global__ __launch_bounds__(64) void k_sum_rows(int64_t B, const float* __restrict__ row_loss, float factor,
                                                 float* __restrict__ loss) {
  float t = 0.f;
  for (int64_t r = threadIdx.x; r < B; r += 64) t += row_loss[r];
  t = wave_sum(t);
  if (threadIdx.x == 0) loss[0] = t * factor;
}

// One wavefront per row; its lanes stride over the classes.
__global__ __launch_bounds__(kLossThreads) void k_softmax_xent(int64_t B, int64_t C, const float* __restrict__ z,
                                                               int64_t ldz, const int64_t* __restrict__ labels,
                                                               const int64_t* __restrict__ idx, int64_t n_labels,
                                                               float* __restrict__ dz, int64_t ldd,
                                                               float* __restrict__ row_loss) {
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
}

// loss = scale * sum (a - t)^2, grad = 2 * scale * (a - t); t's rows optionally picked by idx.
__global__ __launch_bounds__(kLossThreads) void k_sq_err(int64_t B, int64_t D, const float* __restrict__ a,
                                                         int64_t lda, const float* __restrict__ t, int64_t ldt,
                                                         const int64_t* __restrict__ idx, int64_t t_rows, float scale,
                                                         float* __restrict__ grad, int64_t ldg,
                                                         float* __restrict__ row_loss) {
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
}


// ---------------------------------------------------------------------------------
// The label head on one batch, end to end: logits = E . Wt, their cross-entropy against the
// labels, and both gradients dE = dz . Wt^T and dWt = E^T . dz (model/SGNN.py:397-399 and its
// backward).  E is b x D (D <= 256), Wt is D x C (C <= 64): a few hundred KFLOP that stock
// kernels spread over five launches (GEMM, softmax, nll, two GEMMs).  Here a workgroup keeps
// Wt in LDS and takes kHeadRows rows; a wavefront owns a row at a time with lane c holding
// class c.  dWt needs all rows: every workgroup writes its share and k_head_finish adds the
// shares (and the rows' loss terms) in a fixed order in a second launch.
constexpr int kHeadRows = 4;                     // one per wavefront
constexpr int kHeadMaxD = 256, kHeadMaxC = 64;

__global__ __launch_bounds__(256) void k_head_xent(int64_t B, int D, int C, const float* __restrict__ E, int64_t lde,
                                                   const float* __restrict__ Wt, int64_t ldw,
                                                   const int64_t* __restrict__ labels, const int64_t* __restrict__ idx,
                                                   int64_t n_labels, float* __restrict__ dE, int64_t ldde,
                                                   float* __restrict__ row_loss, float* __restrict__ dWt_part) {
  extern __shared__ float smem[];
  float* sW = smem;                               // [D][C]
  float* sE = sW + D * C;                         // [kHeadRows][D]
  float* sZ = sE + kHeadRows * D;                 // [kHeadRows][C]   dz
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t r0 = int64_t(blockIdx.x) * kHeadRows;
  // (k, c) of element e = t, t + 256, ... of a D x C array, stepped without dividing every time
  const int step_k = 256 / C, step_c = 256 % C;
  {
    int k = t / C, c = t % C;
    for (int e = t; e < D * C; e += 256) {
      sW[e] = Wt[int64_t(k) * ldw + c];
      k += step_k; c += step_c;
      if (c >= C) { c -= C; ++k; }
    }
  }
  for (int lr = 0; lr < kHeadRows; ++lr) {
    const int64_t r = r0 + lr;
    for (int k = t; k < D; k += 256) sE[lr * D + k] = r < B ? E[r * lde + k] : 0.f;
  }
  __syncthreads();
  const float inv_b = 1.f / float(B);
  for (int lr = w; lr < kHeadRows; lr += 4) {     // this wavefront's rows
    const int64_t r = r0 + lr;
    if (r >= B) {                                 // uniform over the wavefront
      if (lane < C) sZ[lr * C + lane] = 0.f;
      continue;
    }
    const float* e = sE + lr * D;
    float z = 0.f;
    if (lane < C) {
      float z1 = 0.f, z2 = 0.f, z3 = 0.f;
      int k = 0;
      for (; k + 3 < D; k += 4) {                 // four independent chains
        z = fmaf(e[k], sW[k * C + lane], z);
        z1 = fmaf(e[k + 1], sW[(k + 1) * C + lane], z1);
        z2 = fmaf(e[k + 2], sW[(k + 2) * C + lane], z2);
        z3 = fmaf(e[k + 3], sW[(k + 3) * C + lane], z3);
      }
      for (; k < D; ++k) z = fmaf(e[k], sW[k * C + lane], z);
      z = (z + z1) + (z2 + z3);
    }
    const float zm = lane < C ? z : -INFINITY;
    const float mx = wave_max(zm);
    const float ex = lane < C ? expf(z - mx) : 0.f;
    const float sum = wave_sum(ex);
    const int64_t at = idx ? idx[r] : r;
    const int64_t y = (at >= 0 && at < n_labels) ? labels[at] : -1;
    const float dz = (ex / sum - (lane == y ? 1.f : 0.f)) * inv_b;
    if (lane < C) sZ[lr * C + lane] = dz;
    const float zy = __shfl(z, (y >= 0 && y < C) ? int(y) : 0);
    if (lane == 0) row_loss[r] = (y >= 0 && y < C) ? mx + logf(sum) - zy : 0.f;
    // dE[r][k] = sum_c dz[c] * Wt[k][c]; the wavefront's own dz row is visible to it after the store
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < D; k += 64) {
      float acc = 0.f;
      for (int c = 0; c < C; ++c) acc = fmaf(sZ[lr * C + c], sW[k * C + c], acc);
      dE[r * ldde + k] = acc;
    }
  }
  __syncthreads();
  float* mine = dWt_part + int64_t(blockIdx.x) * D * C;
  int k = t / C, c = t % C;
  for (int e = t; e < D * C; e += 256) {          // dWt share of these rows
    float acc = 0.f;
#pragma unroll
    for (int lr = 0; lr < kHeadRows; ++lr) acc = fmaf(sE[lr * D + k], sZ[lr * C + c], acc);
    mine[e] = acc;
    k += step_k; c += step_c;
    if (c >= C) { c -= C; ++k; }
  }
}

// dWt[k][c] = sum over the workgroups' shares in a fixed order; the last workgroup of this
// launch adds the rows' loss terms (one wavefront, fixed order).
__global__ __launch_bounds__(256) void k_head_finish(int64_t B, int D, int C, int shares,
                                                     const float* __restrict__ dWt_part, float* __restrict__ dWt,
                                                     int64_t lddw, const float* __restrict__ row_loss,
                                                     float* __restrict__ loss) {
  if (blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x >= 64) return;
    float s = 0.f;
    for (int64_t r = threadIdx.x; r < B; r += 64) s += row_loss[r];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s / float(B);
    return;
  }
  // four lanes per element: lane q adds shares q, q + 4, ... in order, then the four fold
  const int e = blockIdx.x * 64 + (threadIdx.x >> 2), q = threadIdx.x & 3;
  float acc = 0.f;
  if (e < D * C) {
#pragma unroll 8
    for (int z = q; z < shares; z += 4) acc += dWt_part[int64_t(z) * D * C + e];
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  if (e < D * C && q == 0) dWt[lddw == C ? int64_t(e) : int64_t(e / C) * lddw + e % C] = acc;
}

// ---------------------------------------------------------------------------------
// The layer's projection of one batch, E = act(X[idx]) . W, and its weight gradient
// dW = act(X[idx])^T . dE, with the row gather and the inner activation
// act(x) = x > 0 ? x : slope * x applied while the operand is loaded
// (model/SGNN.py:73-78 on the rows get_samples picked, :106-121, while U is the identity).
// The products are 512 x 602 x 128-sized: far too small to fill the chip, so what counts is
// latency.  A stock library call picks a large macro-tile (8 workgroups busy, ~19 us) after
// separate gather and activation launches; here one launch covers all three with 32 x 32
// output tiles, operands staged through LDS and the next K-slab prefetched into registers
// while the current one is multiplied.
constexpr int kT = 64;                            // output tile edge
constexpr int kChunk = 32;                        // reduction depth handled by one workgroup

// The reduction dimension is cut into chunks of 32 handled by different workgroups
// (blockIdx.z), so a 512 x 602 x 128 product runs as ~300 short workgroups instead of a few
// long ones whose every K-slab costs a global-memory round trip.  Each thread forms a 4 x 4
// block of its workgroup's 64 x 64 partial tile; k_project_sum then adds the partial tiles in
// chunk order (a launch of its own rather than a device-wide fence inside this one).
template <bool GRAD, bool TRANSB>
__global__ __launch_bounds__(256) void k_batch_project(int64_t M, int64_t N, int64_t K, int64_t n_rows,
                                                       const float* __restrict__ X, int64_t ldx,
                                                       const int64_t* __restrict__ idx, float slope,
                                                       const float* __restrict__ Bm, int64_t ldb,
                                                       float* __restrict__ C, int64_t ldc,
                                                       float4* __restrict__ partial) {
  // C[M x N] = A . B over K, one 64 x 64 tile and one 32-deep chunk of K per workgroup.
  //   forward    : A(m, k) = act(X[idx[m]][k])           B(k, n) = Bm[k][n]   (W,  K x N)
  //   GRAD       : A(m, k) = act(X[idx[k]][m])           B(k, n) = Bm[k][n]   (dE, K x N)
  //   TRANSB     : A(m, k) = X[m][k] (idx NULL, slope 1) B(k, n) = Bm[n][k]   (W,  N x K)
  // idx NULL = the rows of X in order.
  __shared__ __align__(16) float As[kChunk][kT + 4];   // As[k][m]
  __shared__ __align__(16) float Bs[kChunk][kT + 4];   // Bs[k][n]
  const int64_t m0 = int64_t(blockIdx.y) * kT, n0 = int64_t(blockIdx.x) * kT;
  const int64_t k0 = int64_t(blockIdx.z) * kChunk;
  const int t = threadIdx.x;
  if (!GRAD) {                                    // row t / 4 of the tile, 8 consecutive k
    const int lm = t >> 2, lk = (t & 3) * 8;
    const int64_t m = m0 + lm;
    int64_t r = m < M ? (idx ? idx[m] : m) : -1;
    if (r >= n_rows) r = -1;
    const float* src = X + r * ldx + k0 + lk;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (r >= 0 && k0 + lk + e < K) ? src[e] : 0.f;
      As[lk + e][lm] = x > 0.f ? x : slope * x;
    }
  } else {                                        // batch position t / 8 of the chunk, 8 consecutive features
    const int lk = t >> 3, lm = (t & 7) * 8;
    const int64_t k = k0 + lk;
    int64_t r = k < K ? (idx ? idx[k] : k) : -1;
    if (r >= n_rows) r = -1;
    const float* src = X + r * ldx + m0 + lm;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (r >= 0 && m0 + lm + e < M) ? src[e] : 0.f;
      As[lk][lm + e] = x > 0.f ? x : slope * x;
    }
  }
  if (!TRANSB) {
    const int lk = t >> 3, ln = (t & 7) * 8;
    const int64_t k = k0 + lk;
    const float* src = Bm + k * ldb + n0 + ln;
#pragma unroll
    for (int e = 0; e < 8; ++e) Bs[lk][ln + e] = (k < K && n0 + ln + e < N) ? src[e] : 0.f;
  } else {                                        // row n of Bm, 8 consecutive k
    const int ln = t >> 2, lk = (t & 3) * 8;
    const int64_t n = n0 + ln;
    const float* src = Bm + n * ldb + k0 + lk;
#pragma unroll
    for (int e = 0; e < 8; ++e) Bs[lk + e][ln] = (n < N && k0 + lk + e < K) ? src[e] : 0.f;
  }
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;             // 4 x 4 outputs at (4 ty, 4 tx)
  float c[4][4] = {};
#pragma unroll
  for (int k = 0; k < kChunk; ++k) {
    const float4 av = *reinterpret_cast<const float4*>(&As[k][4 * ty]);
    const float4 bv = *reinterpret_cast<const float4*>(&Bs[k][4 * tx]);
    const float a[4] = {av.x, av.y, av.z, av.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) c[i][j] = fmaf(a[i], bb[j], c[i][j]);
  }
  if (gridDim.z > 1) {
    const int64_t tile = int64_t(blockIdx.y) * gridDim.x + blockIdx.x, tiles = int64_t(gridDim.x) * gridDim.y;
    float4* mine = partial + (int64_t(blockIdx.z) * tiles + tile) * 1024 + t;
#pragma unroll
    for (int i = 0; i < 4; ++i) mine[i * 256] = make_float4(c[i][0], c[i][1], c[i][2], c[i][3]);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + 4 * ty + i;
    if (m >= M) break;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + 4 * tx + j;
      if (n < N) C[m * ldc + n] = c[i][j];
    }
  }
}

// C tile = sum over the chunks of the partial tiles, in chunk order.  Same thread-to-output
// mapping as k_batch_project: thread t holds rows 4 (t / 16) + i, columns 4 (t % 16) ..; one
// workgroup per (tile, i), so every thread adds one float4 per chunk.
__global__ __launch_bounds__(256) void k_project_sum(int64_t M, int64_t N, int splits,
                                                     const float4* __restrict__ partial, float* __restrict__ C,
                                                     int64_t ldc, const float* __restrict__ pre, int64_t ldpre,
                                                     float slope) {
  const int64_t tile = int64_t(blockIdx.y) * gridDim.x + blockIdx.x, tiles = int64_t(gridDim.x) * gridDim.y;
  const int t = threadIdx.x, ty = t >> 4, tx = t & 15, i = blockIdx.z;
  const float4* src = partial + tile * 1024 + i * 256 + t;
  const int64_t stride = tiles * 1024;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int z = 0; z < splits; ++z) {
    const float4 v = src[z * stride];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const int64_t m = int64_t(blockIdx.y) * kT + 4 * ty + i, n = int64_t(blockIdx.x) * kT + 4 * tx;
  if (m >= M) return;
  float out[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (n + j >= N) break;
    // pre given: times act'(pre) — the chain rule through act(x) = x > 0 ? x : slope * x
    if (pre != nullptr && !(pre[m * ldpre + n + j] > 0.f)) out[j] *= slope;
    C[m * ldc + n + j] = out[j];
  }
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
  const int64_t blocks = (total + 256 * kAdamPerThread - 1) / (256 * kAdamPerThread);
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
int check_loss_ws(int64_t B, const void* ws, size_t ws_bytes) {
  using sgnn::fail;
  if (B > 0x7fffffff) return fail(SGNN_ERR_INVALID, "more than 2^31-1 rows");
  if (B > 0 && !ws) return fail(SGNN_ERR_INVALID, "null pointer");
  if (ws_bytes < sgnn_loss_workspace_bytes(B)) return fail(SGNN_ERR_WORKSPACE, "loss workspace too small");
  return SGNN_OK;
}
}  // namespace

extern "C" int sgnn_softmax_xent_f32(int64_t B, int64_t C, const float* logits, int64_t ldz, const int64_t* labels,
                                     int64_t n_labels, const int64_t* idx, float* loss, float* dlogits, int64_t ldd,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  using sgnn::fail;
  if (B <= 0 || C <= 0) return fail(SGNN_ERR_INVALID, "need B > 0 and C > 0");
  if (ldz < C || ldd < C) return fail(SGNN_ERR_INVALID, "row stride smaller than the class count");
  if (n_labels < 0 || (!idx && n_labels < B)) return fail(SGNN_ERR_INVALID, "fewer labels than rows");
  if (!logits || !labels || !loss || !dlogits) return fail(SGNN_ERR_INVALID, "null pointer");
  if (int rc = check_loss_ws(B, workspace, workspace_bytes)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* rows = static_cast<float*>(workspace);
  const unsigned blocks = unsigned((B + kLossWaves - 1) / kLossWaves);
  hipLaunchKernelGGL(k_softmax_xent, dim3(blocks), dim3(kLossThreads), 0, st, B, C, logits, ldz, labels, idx, n_labels,
                     dlogits, ldd, rows);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(64), 0, st, B, rows, 1.f / float(B), loss);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

extern "C" int sgnn_sq_err_f32(int64_t B, int64_t D, const float* a, int64_t lda, const float* target, int64_t ldt,
                               int64_t target_rows, const int64_t* idx, float scale, float* loss, float* grad,
                               int64_t ldg, void* workspace, size_t workspace_bytes, void* stream) {
  using sgnn::fail;
  if (B <= 0 || D <= 0 || target_rows < 0) return fail(SGNN_ERR_INVALID, "need B > 0, D > 0");
  if (lda < D || ldt < D || ldg < D) return fail(SGNN_ERR_INVALID, "row stride smaller than the width");
  if (!idx && target_rows < B) return fail(SGNN_ERR_INVALID, "fewer target rows than rows");
  if (!loss || !a || !target || !grad) return fail(SGNN_ERR_INVALID, "null pointer");
  if (int rc = check_loss_ws(B, workspace, workspace_bytes)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* rows = static_cast<float*>(workspace);
  const unsigned blocks = unsigned((B + kLossWaves - 1) / kLossWaves);
  hipLaunchKernelGGL(k_sq_err, dim3(blocks), dim3(kLossThreads), 0, st, B, D, a, lda, target, ldt, idx, target_rows,
                     scale, grad, ldg, rows);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(64), 0, st, B, rows, scale, loss);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

namespace {
struct ProjectGrid { int64_t tiles_n, tiles_m, splits; };
ProjectGrid project_grid(int64_t M, int64_t N, int64_t K) {
  return {(N + kT - 1) / kT, (M + kT - 1) / kT, (K + kChunk - 1) / kChunk};
}
int check_project(int64_t n_rows, int64_t F, const float* X, int64_t ldx, int64_t b, const int64_t* idx, int64_t N) {
  using sgnn::fail;
  if (n_rows < 0 || F <= 0 || b <= 0 || N <= 0) return fail(SGNN_ERR_INVALID, "need F > 0, b > 0, N > 0");
  if (ldx < F) return fail(SGNN_ERR_INVALID, "ldx smaller than the feature width");
  if (!X) return fail(SGNN_ERR_INVALID, "null pointer");
  if (!idx && n_rows < b) return fail(SGNN_ERR_INVALID, "fewer rows than the batch and no index list");
  if (b > 65535 * int64_t(kT) || F > 65535 * int64_t(kT) || N > (int64_t(1) << 20) * kT)
    return fail(SGNN_ERR_INVALID, "dimension too large for one launch");
  return SGNN_OK;
}
template <bool GRAD, bool TRANSB>
int launch_project(int64_t M, int64_t N, int64_t K, int64_t n_rows, const float* X, int64_t ldx, const int64_t* idx,
                   float slope, const float* Bm, int64_t ldb, float* C, int64_t ldc, const float* pre, int64_t ldpre,
                   float pre_slope, void* ws, size_t ws_bytes, void* stream) {
  using sgnn::fail;
  const ProjectGrid g = project_grid(M, N, K);
  const bool staged = g.splits > 1 || pre != nullptr;        // the derivative mask is applied by the summing launch
  // with a single chunk and a mask the partial path is forced by a grid of depth 2 whose second chunk is empty
  const unsigned depth = unsigned(staged && g.splits == 1 ? 2 : g.splits);
  if (staged) {
    if (ws_bytes < size_t(depth) * size_t(g.tiles_m * g.tiles_n) * 1024 * sizeof(float4))
      return fail(SGNN_ERR_WORKSPACE, "projection workspace too small");
    if (!ws) return fail(SGNN_ERR_INVALID, "null pointer");
    if (depth > 65535) return fail(SGNN_ERR_INVALID, "dimension too large for one launch");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(unsigned(g.tiles_n), unsigned(g.tiles_m), depth);
  hipLaunchKernelGGL((k_batch_project<GRAD, TRANSB>), grid, dim3(256), 0, st, M, N, K, n_rows, X, ldx, idx, slope, Bm,
                     ldb, C, ldc, static_cast<float4*>(ws));
  HIP_TRY(hipGetLastError());
  if (staged) {
    hipLaunchKernelGGL(k_project_sum, dim3(unsigned(g.tiles_n), unsigned(g.tiles_m), 4), dim3(256), 0, st, M, N,
                       int(depth), static_cast<const float4*>(ws), C, ldc, pre, ldpre, pre_slope);
    HIP_TRY(hipGetLastError());
  }
  return SGNN_OK;
}
}  // namespace

extern "C" size_t sgnn_batch_project_workspace_bytes(int64_t b, int64_t F, int64_t N, int which) {
  if (b <= 0 || F <= 0 || N <= 0) return 0;
  // which: 0 = sgnn_batch_project_f32, 1 = sgnn_batch_project_grad_f32, 2 = sgnn_batch_project_input_grad_f32
  const ProjectGrid g = which == 1 ? project_grid(F, N, b) : which == 2 ? project_grid(b, F, N) : project_grid(b, N, F);
  const int64_t depth = which == 2 && g.splits == 1 ? 2 : g.splits;
  return (depth > 1) ? size_t(depth * g.tiles_m * g.tiles_n) * 1024 * sizeof(float4) : 0;
}

extern "C" int sgnn_batch_project_f32(int64_t n_rows, int64_t F, const float* X, int64_t ldx, int64_t b,
                                      const int64_t* idx, float slope, const float* W, int64_t ldw, int64_t N, float* E,
                                      int64_t lde, void* workspace, size_t workspace_bytes, void* stream) {
  using sgnn::fail;
  if (int rc = check_project(n_rows, F, X, ldx, b, idx, N)) return rc;
  if (ldw < N || lde < N) return fail(SGNN_ERR_INVALID, "row stride smaller than the output width");
  if (!W || !E) return fail(SGNN_ERR_INVALID, "null pointer");
  return launch_project<false, false>(b, N, F, n_rows, X, ldx, idx, slope, W, ldw, E, lde, nullptr, 0, 1.f, workspace,
                                      workspace_bytes, stream);
}

extern "C" int sgnn_batch_project_grad_f32(int64_t n_rows, int64_t F, const float* X, int64_t ldx, int64_t b,
                                           const int64_t* idx, float slope, const float* dE, int64_t ldde, int64_t N,
                                           float* dW, int64_t lddw, void* workspace, size_t workspace_bytes,
                                           void* stream) {
  using sgnn::fail;
  if (int rc = check_project(n_rows, F, X, ldx, b, idx, N)) return rc;
  if (ldde < N || lddw < N) return fail(SGNN_ERR_INVALID, "row stride smaller than the output width");
  if (!dE || !dW) return fail(SGNN_ERR_INVALID, "null pointer");
  return launch_project<true, false>(F, N, b, n_rows, X, ldx, idx, slope, dE, ldde, dW, lddw, nullptr, 0, 1.f,
                                     workspace, workspace_bytes, stream);
}

extern "C" int sgnn_batch_project_input_grad_f32(int64_t b, int64_t F, int64_t N, const float* dE, int64_t ldde,
                                                 const float* W, int64_t ldw, const float* pre, int64_t ldpre,
                                                 float slope, float* dA, int64_t ldda, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
  using sgnn::fail;
  if (b <= 0 || F <= 0 || N <= 0) return fail(SGNN_ERR_INVALID, "need b > 0, F > 0, N > 0");
  if (ldde < N || ldw < N || ldda < F || (pre && ldpre < F))
    return fail(SGNN_ERR_INVALID, "row stride smaller than the width");
  if (!dE || !W || !dA) return fail(SGNN_ERR_INVALID, "null pointer");
  if (b > 65535 * int64_t(kT) || F > (int64_t(1) << 20) * kT) return fail(SGNN_ERR_INVALID, "dimension too large for one launch");
  return launch_project<false, true>(b, F, N, b, dE, ldde, nullptr, 1.f, W, ldw, dA, ldda, pre, ldpre, slope, workspace,
                                     workspace_bytes, stream);
}

extern "C" int sgnn_sum_f32(int64_t n, const float* x, float factor, float* out, void* stream) {
  using sgnn::fail;
  if (n < 0) return fail(SGNN_ERR_INVALID, "negative size");
  if (!out || (n > 0 && !x)) return fail(SGNN_ERR_INVALID, "null pointer");
  hipLaunchKernelGGL(k_sum_rows, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), n, x, factor, out);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}

extern "C" size_t sgnn_head_xent_workspace_bytes(int64_t B, int64_t D, int64_t C) {
  if (B <= 0 || D <= 0 || C <= 0) return 0;
  const int64_t shares = (B + kHeadRows - 1) / kHeadRows;
  return size_t(B + shares * D * C) * sizeof(float);
}

extern "C" int sgnn_head_xent_f32(int64_t B, int64_t D, int64_t C, const float* E, int64_t lde, const float* Wt,
                                  int64_t ldw, const int64_t* labels, int64_t n_labels, const int64_t* idx, float* loss,
                                  float* dE, int64_t ldde, float* dWt, int64_t lddw, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  using sgnn::fail;
  if (B <= 0 || D <= 0 || C <= 0) return fail(SGNN_ERR_INVALID, "need B > 0, D > 0, C > 0");
  if (D > kHeadMaxD || C > kHeadMaxC) return fail(SGNN_ERR_UNSUPPORTED, "label head larger than 256 x 64");
  if (B > (int64_t(1) << 24)) return fail(SGNN_ERR_INVALID, "batch too large");
  if (lde < D || ldde < D || ldw < C || lddw < C) return fail(SGNN_ERR_INVALID, "row stride smaller than the width");
  if (n_labels < 0 || (!idx && n_labels < B)) return fail(SGNN_ERR_INVALID, "fewer labels than rows");
  if (!E || !Wt || !labels || !loss || !dE || !dWt || !workspace) return fail(SGNN_ERR_INVALID, "null pointer");
  if (workspace_bytes < sgnn_head_xent_workspace_bytes(B, D, C))
    return fail(SGNN_ERR_WORKSPACE, "label-head workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t shares = (B + kHeadRows - 1) / kHeadRows;
  float* rows = static_cast<float*>(workspace);
  float* part = rows + B;
  const size_t lds = size_t(D * C + kHeadRows * D + kHeadRows * C) * sizeof(float);
  if (lds > 64 * 1024) return fail(SGNN_ERR_UNSUPPORTED, "label head does not fit 64 KB of LDS");
  hipLaunchKernelGGL(k_head_xent, dim3(unsigned(shares)), dim3(256), lds, st, B, int(D), int(C), E, lde, Wt, ldw, labels,
                     idx, n_labels, dE, ldde, rows, part);
  HIP_TRY(hipGetLastError());
  const unsigned blocks = unsigned((D * C + 63) / 64) + 1;
  hipLaunchKernelGGL(k_head_finish, dim3(blocks), dim3(256), 0, st, B, int(D), int(C), int(shares), part, dWt, lddw,
                     rows, loss);
  HIP_TRY(hipGetLastError());
  return SGNN_OK;
}
