// The elementwise half of a TCNN block (reference mrgcn/models/temporal_cnn.py:6-156: Conv1d -> BatchNorm1d -> ReLU
// [-> MaxPool1d(k, stride k) | AdaptiveMaxPool1d(n)]): batch statistics, normalisation, ReLU and the pooling
// window in ONE pass over the convolution's output, and the matching backward (pooled gradient read back through the
// saved argmax + ReLU mask + batch-norm backward: two passes over x, no intermediate buffer unless the windows of an
// adaptive pool overlap).  The convolutions themselves are implicit-im2col products on the matrix cores (encoders.hip).
// x / dx: [B][C][T] float32; y / dy / argmax: [B][C][Tout].  All of it is HBM-bound elementwise / reduction work.
#include <algorithm>

#include "common.hpp"

namespace mrgcn {
namespace {

enum { POOL_NONE = 0, POOL_MAX = 1, POOL_ADAPTIVE = 2 };

__host__ __device__ inline int pool_out_len(int kind, int arg, int T) {
  if (kind == POOL_NONE) return T;
  if (kind == POOL_MAX) return T >= arg ? (T - arg) / arg + 1 : 0;  // MaxPool1d(k, stride k), ceil_mode off
  return arg;                                                       // AdaptiveMaxPool1d(n)
}
__device__ inline void pool_window(int kind, int arg, int T, int to, int &t0, int &t1) {
  if (kind == POOL_NONE) {
    t0 = to;
    t1 = to + 1;
  } else if (kind == POOL_MAX) {
    t0 = to * arg;
    t1 = t0 + arg;
  } else {  // ATen's adaptive windows: [floor(to T / n), ceil((to + 1) T / n))
    t0 = (int)(((int64_t)to * T) / arg);
    t1 = (int)((((int64_t)to + 1) * T + arg - 1) / arg);
  }
}

__device__ inline double block_sum(double v, double *s) {  // 256 threads
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s[wv] = v;
  __syncthreads();
  return s[0] + s[1] + s[2] + s[3];
}

// batch statistics: block (c, y) sums channel c over the batch slab y (a wave per batch row, lanes along t:
// coalesced), fp64 partials meet in acc[c][2] (zeroed by the launcher); a second tiny kernel turns them into mean
// and biased variance.  (One block per channel left 64-channel layers on a quarter of the chip.)
__global__ __launch_bounds__(256) void k_bn_stats_part(const float *__restrict__ x, int B, int C, int T, int tp_log2,
                                                       double *__restrict__ acc) {
  __shared__ double s[4];
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // a wave takes 64 / tp batch rows at a time (tp = the power of two >= min(T, 64)): short sequences fill the lanes
  const int rpw = 64 >> tp_log2, sub = lane >> tp_log2, t0 = lane & ((1 << tp_log2) - 1);
  double a = 0.0, q = 0.0;
  for (int b = (blockIdx.y * 4 + wv) * rpw + sub; b < B; b += gridDim.y * 4 * rpw) {
    const float *row = x + ((int64_t)b * C + c) * T;
    for (int t = t0; t < T; t += 1 << tp_log2) {
      const double v = row[t];
      a += v;
      q += v * v;
    }
  }
  a = block_sum(a, s);
  q = block_sum(q, s);
  if (threadIdx.x == 0) {
    atomicAdd(&acc[2 * c], a);
    atomicAdd(&acc[2 * c + 1], q);
  }
}
__global__ void k_bn_stats_fin(const double *__restrict__ acc, int C, int64_t n, float *__restrict__ mean,
                               float *__restrict__ var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = acc[2 * c] / (double)n;
  const double vv = acc[2 * c + 1] / (double)n - m * m;
  mean[c] = (float)m;
  var[c] = (float)(vv > 0.0 ? vv : 0.0);
}

// block (c, y): channel c over the y-th slab of the batch; a wave takes 64 / tp rows of OUTPUT positions at a time
// (tp = the power of two >= min(Tout, 64)), a lane walks its pooling window (no per-element index divisions)
__global__ __launch_bounds__(256) void k_bn_relu_pool_fwd(const float *__restrict__ x, int B, int C, int T, int Tout,
                                                          int tp_log2, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta,
                                                          const float *__restrict__ mean, const float *__restrict__ var,
                                                          float eps, int kind, int arg, float *__restrict__ y,
                                                          int32_t *__restrict__ argmax) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rpw = 64 >> tp_log2, sub = lane >> tp_log2, o0 = lane & ((1 << tp_log2) - 1);
  // (x - mean) first: folding the mean into a bias (x * sc + sh) cancels badly when |mean| >> std
  const float mu = mean[c], sc = (gamma ? gamma[c] : 1.f) / sqrtf(var[c] + eps), sh = beta ? beta[c] : 0.f;
  for (int b = (blockIdx.y * 4 + wv) * rpw + sub; b < B; b += gridDim.y * 4 * rpw) {
    const int64_t bc = (int64_t)b * C + c;
    const float *row = x + bc * T;
    for (int to = o0; to < Tout; to += 1 << tp_log2) {
      int t0, t1;
      pool_window(kind, arg, T, to, t0, t1);
      float best = -1.f;
      int bi = t0;
      for (int t = t0; t < t1; ++t) {
        const float z = fmaxf((row[t] - mu) * sc + sh, 0.f);
        if (z > best) {  // first maximum wins (ATen's order)
          best = z;
          bi = t;
        }
      }
      const int64_t o = bc * Tout + to;
      y[o] = best;
      if (argmax) argmax[o] = bi;
    }
  }
}

// dz[b][c][argmax] += dy where y > 0 (dz zeroed by the caller); windows of an adaptive pool may overlap
__global__ void k_pool_relu_bwd(const float *__restrict__ y, const float *__restrict__ dy,
                                const int32_t *__restrict__ argmax, int64_t n_out, int T, int Tout, int kind,
                                float *__restrict__ dz) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  if (!(y[o] > 0.f)) return;
  const int64_t bc = o / Tout;
  const int t = argmax ? argmax[o] : (int)(o - bc * Tout);
  if (kind == POOL_ADAPTIVE) atomicAdd(dz + bc * T + t, dy[o]);
  else dz[bc * T + t] = dy[o];
}

// gradient at the block's pre-pool activation z = relu(bn(x)) for position (bc, t), read straight from the pooled
// gradient: a non-overlapping window (POOL_NONE, POOL_MAX) hands its dy to its argmax where y > 0 — no dz buffer,
// no zero fill, no scatter.  (Adaptive windows may overlap: they keep the scatter into dz.)
template <int KIND>
__device__ __forceinline__ float dz_at(const float *__restrict__ y, const float *__restrict__ dy,
                                       const int32_t *__restrict__ argmax, const float *__restrict__ dz, int64_t bc,
                                       int t, int T, int Tout, uint32_t arg_magic) {
  if constexpr (KIND == POOL_ADAPTIVE) {
    return dz[bc * T + t];
  } else if constexpr (KIND == POOL_NONE) {
    const int64_t o = bc * T + t;
    const float yy = y[o], dd = dy[o];   // both loads issue together (a branch on y would chain them)
    return yy > 0.f ? dd : 0.f;
  } else {
    // window of t: t / arg as a multiply-high (exact while t * arg < 2^32; 0 stands for arg == 1)
    int to = arg_magic ? (int)__umulhi((uint32_t)t, arg_magic) : t;
    const bool in = to < Tout;            // positions behind the last whole window have none
    to = in ? to : 0;
    const int64_t o = bc * Tout + to;
    const float yy = y[o], dd = dy[o];
    const int am = argmax[o];
    return (in && yy > 0.f && am == t) ? dd : 0.f;
  }
}

// dbeta = sum dz, dgamma = sum dz * xhat: the same two-stage reduction as the statistics
template <int KIND>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce_part(const float *__restrict__ x, const float *__restrict__ dz,
                                                            const float *__restrict__ y, const float *__restrict__ dy,
                                                            const int32_t *__restrict__ argmax, int B, int C, int T,
                                                            int Tout, uint32_t arg_magic, int tp_log2,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ var, float eps,
                                                            double *__restrict__ acc) {
  __shared__ double s[4];
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double m = mean[c], istd = 1.0 / sqrt((double)var[c] + (double)eps);
  double a = 0.0, q = 0.0;
  // a wave takes 64 / tp batch rows at a time (tp = the power of two >= min(T, 64)): short sequences fill the lanes
  const int rpw = 64 >> tp_log2, sub = lane >> tp_log2, t0 = lane & ((1 << tp_log2) - 1);
  for (int b = (blockIdx.y * 4 + wv) * rpw + sub; b < B; b += gridDim.y * 4 * rpw) {
    const int64_t bc = (int64_t)b * C + c, base = bc * T;
    for (int t = t0; t < T; t += 1 << tp_log2) {
      const double gg = dz_at<KIND>(y, dy, argmax, dz, bc, t, T, Tout, arg_magic);
      a += gg;
      q += gg * ((double)x[base + t] - m) * istd;
    }
  }
  a = block_sum(a, s);
  q = block_sum(q, s);
  if (threadIdx.x == 0) {
    atomicAdd(&acc[2 * c], a);
    atomicAdd(&acc[2 * c + 1], q);
  }
}
__global__ void k_bn_bwd_reduce_fin(const double *__restrict__ acc, int C, float *__restrict__ dgamma,
                                    float *__restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = (float)acc[2 * c];
  dgamma[c] = (float)acc[2 * c + 1];
}

// training: dx = gamma istd (dz - mean(dz) - xhat mean(dz xhat));  eval: dx = gamma istd dz.  A wave per (b, c) row,
// lanes along t (no per-element index divisions)
template <int KIND>
__global__ __launch_bounds__(256) void k_bn_bwd_dx(const float *__restrict__ x, const float *__restrict__ dz,
                                                   const float *__restrict__ y, const float *__restrict__ dy,
                                                   const int32_t *__restrict__ argmax, int B, int C, int T, int Tout,
                                                   uint32_t arg_magic, int tp_log2, const float *__restrict__ gamma,
                                                   const float *__restrict__ mean, const float *__restrict__ var,
                                                   float eps, const float *__restrict__ dgamma,
                                                   const float *__restrict__ dbeta, int training,
                                                   float *__restrict__ dx, float *__restrict__ chan_sum) {
  // block (c, y): channel c over the y-th slab of the batch (as the reductions above), a wave takes 64 / tp batch rows
  // at a time (tp = the power of two >= min(T, 64)); the block's sum of dx goes to chan_sum[c] with ONE atomic
  // (an atomic per row put 131 k - 1 M of them on two cache lines: +3.8 ms on the TCNN-M step)
  __shared__ double s_sum[4];
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rpw = 64 >> tp_log2, sub = lane >> tp_log2, t0 = lane & ((1 << tp_log2) - 1);
  const float istd = 1.f / sqrtf(var[c] + eps);
  const float g = (gamma ? gamma[c] : 1.f) * istd;
  const float inv_n = 1.f / (float)((int64_t)B * T);
  const float mu = mean[c], k_b = training ? dbeta[c] * inv_n : 0.f, k_g = training ? dgamma[c] * inv_n : 0.f;
  double rsum = 0.0;
  for (int b = (blockIdx.y * 4 + wv) * rpw + sub; b < B; b += gridDim.y * 4 * rpw) {
    const int64_t bc = (int64_t)b * C + c, base = bc * T;
    for (int t = t0; t < T; t += 1 << tp_log2) {
      float v = dz_at<KIND>(y, dy, argmax, dz, bc, t, T, Tout, arg_magic);
      if (training) v = v - k_b - (x[base + t] - mu) * istd * k_g;
      v *= g;
      dx[base + t] = v;
      rsum += (double)v;
    }
  }
  if (chan_sum) {
    rsum = block_sum(rsum, s_sum);
    if (threadIdx.x == 0 && rsum != 0.0) atomicAdd(&chan_sum[c], (float)rsum);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// SHORT sequences (T <= 32: rows of under 128 bytes — the S-size TCNN runs at T = 20, 10, 5, 1).  The kernels above
// give a block ONE channel: a wave then reads T floats per batch row, C T floats apart — 80-byte, 40-byte, 20-byte,
// 4-byte pieces of different cache lines (the seven BatchNorm blocks of TCNN-S on 20 000 literals: 5.2 ms for
// passes over 100 MB arrays, 0.3 TB/s).  Here a block takes a slab of batch rows over ALL channels: thread t owns the
// positions p = t, t + 256, ... of a batch row's C T floats (a fixed channel each) and walks the slab's rows — every
// load instruction reads 1 KB of consecutive floats.  Per-channel sums meet in LDS (fp64), one global atomic per
// channel and block.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFlatUnroll = 4;

__global__ __launch_bounds__(256) void k_bn_stats_flat(const float *__restrict__ x, int B, int C, int T, int rows_per,
                                                       double *__restrict__ acc) {
  extern __shared__ double s_acc[];  // [C][2]
  for (int i = threadIdx.x; i < 2 * C; i += 256) s_acc[i] = 0.0;
  __syncthreads();
  const int CT = C * T;
  const int b0 = blockIdx.x * rows_per, b1 = min(B, b0 + rows_per);
  for (int p = threadIdx.x; p < CT; p += 256) {
    const int c = p / T;
    const float *col = x + (int64_t)b0 * CT + p;
    double a = 0.0, q = 0.0;
    int b = b0;
    for (; b + kFlatUnroll <= b1; b += kFlatUnroll) {
      float v[kFlatUnroll];
#pragma unroll
      for (int u = 0; u < kFlatUnroll; ++u) v[u] = col[(int64_t)(b - b0 + u) * CT];
#pragma unroll
      for (int u = 0; u < kFlatUnroll; ++u) { a += (double)v[u]; q += (double)v[u] * (double)v[u]; }
    }
    for (; b < b1; ++b) { const double v = col[(int64_t)(b - b0) * CT]; a += v; q += v * v; }
    atomicAdd(&s_acc[2 * c], a);
    atomicAdd(&s_acc[2 * c + 1], q);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256)
    if (s_acc[i] != 0.0) atomicAdd(&acc[i], s_acc[i]);
}

__global__ __launch_bounds__(256) void k_bn_relu_pool_fwd_flat(const float *__restrict__ x, int B, int C, int T, int Tout,
                                                               int rows_per, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ var, float eps, int kind, int arg,
                                                               float *__restrict__ y, int32_t *__restrict__ argmax) {
  const int CT = C * T, CTo = C * Tout;
  const int b0 = blockIdx.x * rows_per, b1 = min(B, b0 + rows_per);
  for (int po = threadIdx.x; po < CTo; po += 256) {  // output position (c, to) of a batch row
    const int c = po / Tout, to = po - c * Tout;
    const float mu = mean[c], sc = (gamma ? gamma[c] : 1.f) / sqrtf(var[c] + eps), sh = beta ? beta[c] : 0.f;
    int t0, t1;
    pool_window(kind, arg, T, to, t0, t1);
    int b = b0;
    for (; b + kFlatUnroll <= b1; b += kFlatUnroll) {  // four rows side by side: their loads of a window step in flight
      float best[kFlatUnroll];
      int bi[kFlatUnroll];
#pragma unroll
      for (int u = 0; u < kFlatUnroll; ++u) { best[u] = -1.f; bi[u] = t0; }
      for (int t = t0; t < t1; ++t) {
        float v[kFlatUnroll];
#pragma unroll
        for (int u = 0; u < kFlatUnroll; ++u) v[u] = x[(int64_t)(b + u) * CT + c * T + t];
#pragma unroll
        for (int u = 0; u < kFlatUnroll; ++u) {
          const float z = fmaxf((v[u] - mu) * sc + sh, 0.f);
          if (z > best[u]) {  // first maximum wins (ATen's order)
            best[u] = z;
            bi[u] = t;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kFlatUnroll; ++u) {
        const int64_t o = (int64_t)(b + u) * CTo + po;
        y[o] = best[u];
        if (argmax) argmax[o] = bi[u];
      }
    }
    for (; b < b1; ++b) {
      const float *row = x + (int64_t)b * CT + c * T;
      float best = -1.f;
      int bi = t0;
      for (int t = t0; t < t1; ++t) {
        const float z = fmaxf((row[t] - mu) * sc + sh, 0.f);
        if (z > best) {  // first maximum wins (ATen's order)
          best = z;
          bi = t;
        }
      }
      const int64_t o = (int64_t)b * CTo + po;
      y[o] = best;
      if (argmax) argmax[o] = bi;
    }
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce_flat(const float *__restrict__ x, const float *__restrict__ dz,
                                                            const float *__restrict__ y, const float *__restrict__ dy,
                                                            const int32_t *__restrict__ argmax, int B, int C, int T,
                                                            int Tout, uint32_t arg_magic, int rows_per,
                                                            const float *__restrict__ mean,
                                                            const float *__restrict__ var, float eps,
                                                            double *__restrict__ acc) {
  extern __shared__ double s_acc[];  // [C][2]
  for (int i = threadIdx.x; i < 2 * C; i += 256) s_acc[i] = 0.0;
  __syncthreads();
  const int CT = C * T;
  const int b0 = blockIdx.x * rows_per, b1 = min(B, b0 + rows_per);
  for (int p = threadIdx.x; p < CT; p += 256) {
    const int c = p / T, t = p - c * T;
    const double m = mean[c], istd = 1.0 / sqrt((double)var[c] + (double)eps);
    double a = 0.0, q = 0.0;
    int b = b0;
    for (; b + kFlatUnroll <= b1; b += kFlatUnroll) {  // the rows' loads first (a plain loop waits for each row in turn)
      float gv[kFlatUnroll], xv[kFlatUnroll];
#pragma unroll
      for (int u = 0; u < kFlatUnroll; ++u) {
        const int64_t bc = (int64_t)(b + u) * C + c;
        gv[u] = dz_at<KIND>(y, dy, argmax, dz, bc, t, T, Tout, arg_magic);
        xv[u] = x[bc * T + t];
      }
#pragma unroll
      for (int u = 0; u < kFlatUnroll; ++u) {
        const double gg = gv[u];
        a += gg;
        q += gg * ((double)xv[u] - m) * istd;
      }
    }
    for (; b < b1; ++b) {
      const int64_t bc = (int64_t)b * C + c;
      const double gg = dz_at<KIND>(y, dy, argmax, dz, bc, t, T, Tout, arg_magic);
      a += gg;
      q += gg * ((double)x[bc * T + t] - m) * istd;
    }
    atomicAdd(&s_acc[2 * c], a);
    atomicAdd(&s_acc[2 * c + 1], q);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256)
    if (s_acc[i] != 0.0) atomicAdd(&acc[i], s_acc[i]);
}

template <int KIND>
__global__ __launch_bounds__(256) void k_bn_bwd_dx_flat(const float *__restrict__ x, const float *__restrict__ dz,
                                                        const float *__restrict__ y, const float *__restrict__ dy,
                                                        const int32_t *__restrict__ argmax, int B, int C, int T, int Tout,
                                                        uint32_t arg_magic, int rows_per, const float *__restrict__ gamma,
                                                        const float *__restrict__ mean, const float *__restrict__ var,
                                                        float eps, const float *__restrict__ dgamma,
                                                        const float *__restrict__ dbeta, int training,
                                                        float *__restrict__ dx, float *__restrict__ chan_sum) {
  extern __shared__ double s_acc[];  // [C]
  if (chan_sum) {
    for (int i = threadIdx.x; i < C; i += 256) s_acc[i] = 0.0;
    __syncthreads();
  }
  const int CT = C * T;
  const int b0 = blockIdx.x * rows_per, b1 = min(B, b0 + rows_per);
  const float inv_n = 1.f / (float)((int64_t)B * T);
  for (int p = threadIdx.x; p < CT; p += 256) {
    const int c = p / T, t = p - c * T;
    const float istd = 1.f / sqrtf(var[c] + eps);
    const float g = (gamma ? gamma[c] : 1.f) * istd;
    const float mu = mean[c], k_b = training ? dbeta[c] * inv_n : 0.f, k_g = training ? dgamma[c] * inv_n : 0.f;
    double rsum = 0.0;
    int b = b0;
    for (; b + kFlatUnroll <= b1; b += kFlatUnroll) {  // (loads of four rows in flight)
      float gv[kFlatUnroll], xv[kFlatUnroll];
#pragma unroll
      for (int u = 0; u < kFlatUnroll; ++u) {
        const int64_t bc = (int64_t)(b + u) * C + c;
        gv[u] = dz_at<KIND>(y, dy, argmax, dz, bc, t, T, Tout, arg_magic);
        xv[u] = training ? x[bc * T + t] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kFlatUnroll; ++u) {
        const int64_t e = ((int64_t)(b + u) * C + c) * T + t;
        float v = gv[u];
        if (training) v = v - k_b - (xv[u] - mu) * istd * k_g;
        v *= g;
        dx[e] = v;
        rsum += (double)v;
      }
    }
    for (; b < b1; ++b) {
      const int64_t bc = (int64_t)b * C + c, e = bc * T + t;
      float v = dz_at<KIND>(y, dy, argmax, dz, bc, t, T, Tout, arg_magic);
      if (training) v = v - k_b - (x[e] - mu) * istd * k_g;
      v *= g;
      dx[e] = v;
      rsum += (double)v;
    }
    if (chan_sum) atomicAdd(&s_acc[c], rsum);
  }
  if (chan_sum) {
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256)
      if (s_acc[i] != 0.0) atomicAdd(&chan_sum[i], (float)s_acc[i]);
  }
}

__global__ __launch_bounds__(256) void k_chan_sum_flat(const float *__restrict__ x, int B, int C, int T, int rows_per,
                                                       float *__restrict__ out) {
  extern __shared__ double s_acc[];  // [C]
  for (int i = threadIdx.x; i < C; i += 256) s_acc[i] = 0.0;
  __syncthreads();
  const int CT = C * T;
  const int b0 = blockIdx.x * rows_per, b1 = min(B, b0 + rows_per);
  for (int p = threadIdx.x; p < CT; p += 256) {
    const float *col = x + (int64_t)b0 * CT + p;
    double a = 0.0;
    for (int b = b0; b < b1; ++b) a += (double)col[(int64_t)(b - b0) * CT];
    atomicAdd(&s_acc[p / T], a);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256)
    if (s_acc[i] != 0.0) atomicAdd(&out[i], (float)s_acc[i]);
}

// out[c] += sum over (b, t) of x[b][c][t] (out zeroed by the launcher): the bias gradient of a Conv1d.  Block (c, y)
// sums its batch slab in fp64, one float atomic per block
__global__ __launch_bounds__(256) void k_chan_sum_part(const float *__restrict__ x, int B, int C, int T, int tp_log2,
                                                       float *__restrict__ out) {
  __shared__ double s[4];
  const int c = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rpw = 64 >> tp_log2, sub = lane >> tp_log2, t0 = lane & ((1 << tp_log2) - 1);
  double a = 0.0;
  for (int b = (blockIdx.y * 4 + wv) * rpw + sub; b < B; b += gridDim.y * 4 * rpw) {
    const float *row = x + ((int64_t)b * C + c) * T;
    for (int t = t0; t < T; t += 1 << tp_log2) a += (double)row[t];
  }
  a = block_sum(a, s);
  if (threadIdx.x == 0) atomicAdd(&out[c], (float)a);
}

// running statistics of nn.BatchNorm1d in training mode: r = (1 - momentum) r + momentum stat, the variance unbiased
// (one launch instead of the six elementwise kernels of the module's own update)
__global__ void k_bn_running(const float *__restrict__ mean, const float *__restrict__ var, int C, float momentum,
                             float unbias, float *__restrict__ rmean, float *__restrict__ rvar) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[c];
  rvar[c] = (1.f - momentum) * rvar[c] + momentum * (var[c] * unbias);
}

inline unsigned nb(int64_t n) { return (unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1); }

}  // namespace
}  // namespace mrgcn

using namespace mrgcn;

extern "C" {

int32_t mrgcn_pool_out_len(int32_t pool_kind, int32_t pool_arg, int32_t T) {
  if (pool_kind < 0 || pool_kind > 2 || (pool_kind != POOL_NONE && pool_arg <= 0) || T < 0) return -1;
  return pool_out_len(pool_kind, pool_arg, T);
}

size_t mrgcn_bn_workspace_bytes(int32_t C) { return C > 0 ? (size_t)C * 2 * sizeof(double) : 0; }

static inline int tp_log2_of(int T) {  // log2 of the power of two >= min(T, 64)
  int l = 0;
  while (l < 6 && (1 << l) < T) ++l;
  return l;
}

// the short-sequence kernels (k_*_flat): rows of under 128 bytes, per-channel sums in LDS
static inline bool bn_flat(int C, int T) { return T <= 32 && C <= 2048; }
static inline int bn_flat_rows(int B) { return std::max(1, (B + 1023) / 1024); }  // batch rows per block: <= 1 024 blocks

static inline unsigned bn_slabs(int B, int C) {
  int s = 2048 / (C > 0 ? C : 1);
  if (s < 1) s = 1;
  if (s > (B + 3) / 4) s = (B + 3) / 4;
  return (unsigned)(s < 1 ? 1 : s);
}

int mrgcn_bn_relu_pool_fwd_f32(const float *x, int32_t B, int32_t C, int32_t T, const float *gamma,
                               const float *beta, float eps, int32_t training, float *mean, float *var,
                               int32_t pool_kind, int32_t pool_arg, float *y, int32_t *argmax, void *workspace,
                               void *stream) {
  MRGCN_REQUIRE(x && y && mean && var, "NULL");
  MRGCN_REQUIRE(!training || workspace, "training mode needs the workspace (mrgcn_bn_workspace_bytes)");
  MRGCN_REQUIRE(B > 0 && C > 0 && T > 0, "B / C / T");
  MRGCN_REQUIRE(pool_kind >= 0 && pool_kind <= 2 && (pool_kind == POOL_NONE || pool_arg > 0), "pool");
  MRGCN_REQUIRE(pool_kind == POOL_NONE || argmax, "a pooled block needs the argmax buffer");
  const int Tout = pool_out_len(pool_kind, pool_arg, T);
  MRGCN_REQUIRE(Tout > 0, "the pooling window is longer than the sequence");
  hipStream_t s = (hipStream_t)stream;
  if (training) {
    double *acc = (double *)workspace;
    MRGCN_HIP_TRY(mrgcn::fill_async(acc, 0, mrgcn_bn_workspace_bytes(C), s));
    if (bn_flat(C, T)) {
      const int rp = bn_flat_rows(B);
      k_bn_stats_flat<<<dim3((B + rp - 1) / rp), dim3(256), (size_t)C * 2 * sizeof(double), s>>>(x, B, C, T, rp, acc);
    } else
      k_bn_stats_part<<<dim3(C, bn_slabs(B, C)), dim3(256), 0, s>>>(x, B, C, T, tp_log2_of(T), acc);
    k_bn_stats_fin<<<dim3((C + 255) / 256), dim3(256), 0, s>>>(acc, C, (int64_t)B * T, mean, var);
  }
  if (bn_flat(C, T)) {
    const int rp = bn_flat_rows(B);
    k_bn_relu_pool_fwd_flat<<<dim3((B + rp - 1) / rp), dim3(256), 0, s>>>(x, B, C, T, Tout, rp, gamma, beta, mean, var, eps,
                                                                        pool_kind, pool_arg, y, argmax);
  } else {
    const int tpo = tp_log2_of(Tout), rpw = 64 >> tpo;
    const unsigned slabs = (unsigned)std::max<int64_t>(1, std::min<int64_t>(((int64_t)B + 4 * rpw - 1) / (4 * rpw),
                                                                             std::max<int64_t>(1, 8192 / C)));
    k_bn_relu_pool_fwd<<<dim3(C, slabs), dim3(256), 0, s>>>(x, B, C, T, Tout, tpo, gamma, beta, mean, var, eps, pool_kind,
                                                           pool_arg, y, argmax);
  }
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_bn_running_stats_f32(const float *mean, const float *var, int32_t C, int64_t n, float momentum,
                               float *running_mean, float *running_var, void *stream) {
  MRGCN_REQUIRE(mean && var && running_mean && running_var && C > 0 && n > 0, "operands");
  const float unbias = n > 1 ? (float)((double)n / (double)(n - 1)) : 1.f;
  k_bn_running<<<dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(mean, var, C, momentum, unbias, running_mean,
                                                                            running_var);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_channel_sum_f32(const float *x, int32_t B, int32_t C, int32_t T, float *out, void *stream) {
  MRGCN_REQUIRE(x && out && B > 0 && C > 0 && T > 0, "operands");
  hipStream_t s = (hipStream_t)stream;
  MRGCN_HIP_TRY(mrgcn::fill_async(out, 0, (size_t)C * sizeof(float), s));
  if (bn_flat(C, T)) {
    const int rp = bn_flat_rows(B);
    k_chan_sum_flat<<<dim3((B + rp - 1) / rp), dim3(256), (size_t)C * sizeof(double), s>>>(x, B, C, T, rp, out);
  } else
    k_chan_sum_part<<<dim3(C, bn_slabs(B, C)), dim3(256), 0, s>>>(x, B, C, T, tp_log2_of(T), out);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_bn_relu_pool_bwd_f32(const float *x, const float *y, const float *dy, const int32_t *argmax, int32_t B,
                               int32_t C, int32_t T, const float *gamma, const float *mean, const float *var,
                               float eps, int32_t training, int32_t pool_kind, int32_t pool_arg, float *dz,
                               float *dx, float *dgamma, float *dbeta, void *workspace, void *stream) {
  return mrgcn_bn_relu_pool_bwd_sum_f32(x, y, dy, argmax, B, C, T, gamma, mean, var, eps, training, pool_kind, pool_arg,
                                        dz, dx, dgamma, dbeta, nullptr, workspace, stream);
}

int mrgcn_bn_relu_pool_bwd_sum_f32(const float *x, const float *y, const float *dy, const int32_t *argmax, int32_t B,
                                   int32_t C, int32_t T, const float *gamma, const float *mean, const float *var,
                                   float eps, int32_t training, int32_t pool_kind, int32_t pool_arg, float *dz,
                                   float *dx, float *dgamma, float *dbeta, float *dx_chan_sum, void *workspace,
                                   void *stream) {
  MRGCN_REQUIRE(x && y && dy && mean && var && dx && dgamma && dbeta && workspace, "NULL");
  MRGCN_REQUIRE(dz || pool_kind != POOL_ADAPTIVE, "adaptive pooling needs the dz workspace");
  MRGCN_REQUIRE(B > 0 && C > 0 && T > 0, "B / C / T");
  MRGCN_REQUIRE(pool_kind >= 0 && pool_kind <= 2 && (pool_kind == POOL_NONE || (pool_arg > 0 && argmax)), "pool");
  const int Tout = pool_out_len(pool_kind, pool_arg, T);
  MRGCN_REQUIRE(Tout > 0, "the pooling window is longer than the sequence");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_in = (int64_t)B * C * T, n_out = (int64_t)B * C * Tout;
  double *acc = (double *)workspace;
  MRGCN_HIP_TRY(mrgcn::fill_async(acc, 0, mrgcn_bn_workspace_bytes(C), s));
  if (dx_chan_sum) MRGCN_HIP_TRY(mrgcn::fill_async(dx_chan_sum, 0, (size_t)C * sizeof(float), s));
  const dim3 rgrid(C, bn_slabs(B, C));
  const uint32_t arg_magic = (pool_kind == POOL_MAX && pool_arg > 1) ? (uint32_t)((((uint64_t)1) << 32) / (uint64_t)pool_arg + 1) : 0u;
  MRGCN_REQUIRE(pool_kind != POOL_MAX || (int64_t)T * pool_arg < ((int64_t)1 << 32), "sequence too long");
  const int tp_log2 = tp_log2_of(T);
  const int rpw = 64 >> tp_log2;
  unsigned xslabs = (unsigned)std::max<int64_t>(1, std::min<int64_t>(((int64_t)B + 4 * rpw - 1) / (4 * rpw), std::max<int64_t>(1, 8192 / C)));
  const dim3 xgrid(C, xslabs);
  const bool flat = bn_flat(C, T);
  const int frp = bn_flat_rows(B);
  const dim3 fgrid((B + frp - 1) / frp);
#define BN_BWD_GO(KIND_)                                                                                             \
  do {                                                                                                               \
    if (flat) {                                                                                                      \
      k_bn_bwd_reduce_flat<KIND_><<<fgrid, dim3(256), (size_t)C * 2 * sizeof(double), s>>>(                          \
          x, dz, y, dy, argmax, B, C, T, Tout, arg_magic, frp, mean, var, eps, acc);                                 \
      k_bn_bwd_reduce_fin<<<dim3((C + 255) / 256), dim3(256), 0, s>>>(acc, C, dgamma, dbeta);                        \
      k_bn_bwd_dx_flat<KIND_><<<fgrid, dim3(256), (size_t)C * sizeof(double), s>>>(                                  \
          x, dz, y, dy, argmax, B, C, T, Tout, arg_magic, frp, gamma, mean, var, eps, dgamma, dbeta, training, dx,   \
          dx_chan_sum);                                                                                              \
      break;                                                                                                         \
    }                                                                                                                \
    k_bn_bwd_reduce_part<KIND_><<<rgrid, dim3(256), 0, s>>>(x, dz, y, dy, argmax, B, C, T, Tout, arg_magic,         \
                                                            tp_log2, mean, var, eps, acc);                                          \
    k_bn_bwd_reduce_fin<<<dim3((C + 255) / 256), dim3(256), 0, s>>>(acc, C, dgamma, dbeta);                          \
    k_bn_bwd_dx<KIND_><<<xgrid, dim3(256), 0, s>>>(x, dz, y, dy, argmax, B, C, T, Tout, arg_magic, tp_log2, gamma,   \
                                                   mean, var, eps, dgamma, dbeta, training, dx, dx_chan_sum);         \
  } while (0)
  if (pool_kind == POOL_ADAPTIVE) {  // windows may overlap: scatter into dz first
    MRGCN_HIP_TRY(mrgcn::fill_async(dz, 0, (size_t)n_in * sizeof(float), s));
    k_pool_relu_bwd<<<dim3(nb(n_out)), dim3(256), 0, s>>>(y, dy, argmax, n_out, T, Tout, pool_kind, dz);
    BN_BWD_GO(POOL_ADAPTIVE);
  } else if (pool_kind == POOL_MAX) {
    BN_BWD_GO(POOL_MAX);
  } else {
    BN_BWD_GO(POOL_NONE);
  }
#undef BN_BWD_GO
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // extern "C"
