// Experiment harness (NOT product): what a float4 copy / triad reaches on this box, by launch shape.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/lab/copy_lab.hip -o /tmp/copy_lab && /tmp/copy_lab
// MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; round 5's probes (2 048 persistent blocks, one 16-byte piece per
// thread and trip) read 4.7-4.8 TB/s on the bench boxes.  Which of grid shape, pieces in flight per thread, or cache
// policy is the difference?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using f4 = __attribute__((ext_vector_type(4))) float;

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_gs(const f4 *__restrict__ a, f4 *__restrict__ b, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * stride < n4 ? i + u * stride : i;
      v[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < n4) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
  }
}
// a block owns one contiguous chunk; U pieces in flight per thread
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_chunk(const f4 *__restrict__ a, f4 *__restrict__ b, int64_t n4) {
  const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
  for (int64_t i = lo + threadIdx.x; i < hi; i += U * 256) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 256 < hi ? i + u * 256 : i;
      v[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * 256 < hi) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * 256); else b[i + u * 256] = v[u]; }
  }
}
// one-shot grid: thread = U pieces, block = 256 U consecutive pieces
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_once(const f4 *__restrict__ a, f4 *__restrict__ b, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t j = i + u * 256 < n4 ? i + u * 256 : 0;
    v[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (i + u * 256 < n4) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * 256); else b[i + u * 256] = v[u]; }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_triad_gs(f4 *__restrict__ p, f4 *__restrict__ m, f4 *__restrict__ v, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += U * stride) {
    f4 P[U], M[U], V[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * stride < n4 ? i + u * stride : i;
      P[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
      M[u] = NT ? __builtin_nontemporal_load(m + j) : m[j];
      V[u] = NT ? __builtin_nontemporal_load(v + j) : v[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g = 1e-3f;
        M[u][k] = fmaf(0.9f, M[u][k], 0.1f * g);
        V[u][k] = fmaf(0.999f, V[u][k], 0.001f * g * g);
        P[u][k] -= 0.01f * (M[u][k] / (sqrtf(V[u][k]) + 1e-8f));
      }
      const int64_t j = i + u * stride;
      if (j < n4) {
        if (NT) { __builtin_nontemporal_store(P[u], p + j); __builtin_nontemporal_store(M[u], m + j); __builtin_nontemporal_store(V[u], v + j); }
        else { p[j] = P[u]; m[j] = M[u]; v[j] = V[u]; }
      }
    }
  }
}
// read-only sweep (sum into a sink) and write-only fill: what each direction reaches alone
template <int U>
__global__ __launch_bounds__(256) void k_read(const f4 *__restrict__ a, float *sink, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  f4 s = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) { const int64_t j = i + u * stride < n4 ? i + u * stride : i; s += __builtin_nontemporal_load(a + j); }
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) *sink = s.x;
}
__global__ __launch_bounds__(256) void k_write(f4 *__restrict__ b, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const f4 z = {1.f, 2.f, 3.f, 4.f};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(z, b + i);
}

// persistent blocks that take their tiles IN ORDER from a ticket counter (one atomic per wave and tile of 64 U pieces):
// the hardware's own block dispatch order, without giving up a resident block's state
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_ticket(const f4 *__restrict__ a, f4 *__restrict__ b, int64_t n4,
                                                     unsigned long long *__restrict__ counter) {
  const int lane = threadIdx.x & 63;
  const int64_t tiles = (n4 + 64 * U - 1) / (64 * U);
  for (;;) {
    unsigned long long t = 0;
    if (lane == 0) t = atomicAdd(counter, 1ull);
    t = __shfl(t, 0, 64);
    if ((int64_t)t >= tiles) break;
    const int64_t i = (int64_t)t * 64 * U + lane;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 64 < n4 ? i + u * 64 : i;
      v[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * 64 < n4) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * 64); else b[i + u * 64] = v[u]; }
  }
}
// the ticket fetched one tile AHEAD (the atomic's round trip flies under the current tile's loads)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy_ticket2(const f4 *__restrict__ a, f4 *__restrict__ b, int64_t n4,
                                                      unsigned long long *__restrict__ counter) {
  const int lane = threadIdx.x & 63;
  const int64_t tiles = (n4 + 64 * U - 1) / (64 * U);
  unsigned long long t = 0, tn = 0;
  if (lane == 0) t = atomicAdd(counter, 1ull);
  t = __shfl(t, 0, 64);
  while ((int64_t)t < tiles) {
    if (lane == 0) tn = atomicAdd(counter, 1ull);
    const int64_t i = (int64_t)t * 64 * U + lane;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 64 < n4 ? i + u * 64 : i;
      v[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * 64 < n4) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * 64); else b[i + u * 64] = v[u]; }
    t = __shfl(tn, 0, 64);
  }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_triad_once(f4 *__restrict__ p, f4 *__restrict__ m, f4 *__restrict__ v, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
  f4 P[U], M[U], V[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t j = i + u * 256 < n4 ? i + u * 256 : 0;
    P[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
    M[u] = NT ? __builtin_nontemporal_load(m + j) : m[j];
    V[u] = NT ? __builtin_nontemporal_load(v + j) : v[j];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = 1e-3f;
      M[u][k] = fmaf(0.9f, M[u][k], 0.1f * g);
      V[u][k] = fmaf(0.999f, V[u][k], 0.001f * g * g);
      P[u][k] -= 0.01f * (M[u][k] / (sqrtf(V[u][k]) + 1e-8f));
    }
    const int64_t j = i + u * 256;
    if (j < n4) {
      if (NT) { __builtin_nontemporal_store(P[u], p + j); __builtin_nontemporal_store(M[u], m + j); __builtin_nontemporal_store(V[u], v + j); }
      else { p[j] = P[u]; m[j] = M[u]; v[j] = V[u]; }
    }
  }
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_triad_ticket(f4 *__restrict__ p, f4 *__restrict__ m, f4 *__restrict__ v, int64_t n4,
                                                      unsigned long long *__restrict__ counter) {
  const int lane = threadIdx.x & 63;
  const int64_t tiles = (n4 + 64 * U - 1) / (64 * U);
  unsigned long long t = 0, tn = 0;
  if (lane == 0) t = atomicAdd(counter, 1ull);
  t = __shfl(t, 0, 64);
  while ((int64_t)t < tiles) {
    if (lane == 0) tn = atomicAdd(counter, 1ull);
    const int64_t i = (int64_t)t * 64 * U + lane;
    f4 P[U], M[U], V[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 64 < n4 ? i + u * 64 : i;
      P[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
      M[u] = NT ? __builtin_nontemporal_load(m + j) : m[j];
      V[u] = NT ? __builtin_nontemporal_load(v + j) : v[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g = 1e-3f;
        M[u][k] = fmaf(0.9f, M[u][k], 0.1f * g);
        V[u][k] = fmaf(0.999f, V[u][k], 0.001f * g * g);
        P[u][k] -= 0.01f * (M[u][k] / (sqrtf(V[u][k]) + 1e-8f));
      }
      const int64_t j = i + u * 64;
      if (j < n4) {
        if (NT) { __builtin_nontemporal_store(P[u], p + j); __builtin_nontemporal_store(M[u], m + j); __builtin_nontemporal_store(V[u], v + j); }
        else { p[j] = P[u]; m[j] = M[u]; v[j] = V[u]; }
      }
    }
    t = __shfl(tn, 0, 64);
  }
}
// C ticket counters (4 KB apart: different channels), counter c serves the c-th contiguous slice of the array; a wave
// draws from counter (its global wave id mod C), the next ticket one tile ahead
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_triad_mticket(f4 *__restrict__ p, f4 *__restrict__ m, f4 *__restrict__ v, int64_t n4,
                                                       unsigned long long *__restrict__ counters, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int c = (int)(wid % C);
  unsigned long long *counter = counters + (int64_t)c * 512;
  const int64_t tiles_all = (n4 + 64 * U - 1) / (64 * U);
  const int64_t per = (tiles_all + C - 1) / C;
  const int64_t t_lo = c * per, t_hi = t_lo + per < tiles_all ? t_lo + per : tiles_all;
  unsigned long long t = 0, tn = 0;
  if (lane == 0) t = atomicAdd(counter, 1ull);
  t = __shfl(t, 0, 64);
  while (t_lo + (int64_t)t < t_hi) {
    if (lane == 0) tn = atomicAdd(counter, 1ull);
    const int64_t i = (t_lo + (int64_t)t) * 64 * U + lane;
    f4 P[U], M[U], V[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 64 < n4 ? i + u * 64 : i;
      P[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
      M[u] = NT ? __builtin_nontemporal_load(m + j) : m[j];
      V[u] = NT ? __builtin_nontemporal_load(v + j) : v[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g = 1e-3f;
        M[u][k] = fmaf(0.9f, M[u][k], 0.1f * g);
        V[u][k] = fmaf(0.999f, V[u][k], 0.001f * g * g);
        P[u][k] -= 0.01f * (M[u][k] / (sqrtf(V[u][k]) + 1e-8f));
      }
      const int64_t j = i + u * 64;
      if (j < n4) {
        if (NT) { __builtin_nontemporal_store(P[u], p + j); __builtin_nontemporal_store(M[u], m + j); __builtin_nontemporal_store(V[u], v + j); }
        else { p[j] = P[u]; m[j] = M[u]; v[j] = V[u]; }
      }
    }
    t = __shfl(tn, 0, 64);
  }
}
// persistent, but a wave's trips are short runs: wave w of W takes tiles w, w + W, ... only within a WINDOW of the
// array, windows one after the other behind a grid-wide... (no grid barrier exists: instead every window is its own
// LAUNCH of a one-shot grid — `k_triad_once` over a slice — to see what many short launches cost)
// one-shot read-only / write-only
__global__ __launch_bounds__(256) void k_read_once(const f4 *__restrict__ a, float *sink, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const f4 s = a[i < n4 ? i : 0];
  if (s.x + s.y + s.z + s.w == 12345.678f) *sink = s.x;
}
__global__ __launch_bounds__(256) void k_write_once(f4 *__restrict__ b, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const f4 z = {1.f, 2.f, 3.f, 4.f};
  if (i < n4) b[i] = z;
}

template <typename Fn> static float time_ms(Fn fn, int iters = 10) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) fn();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}
int main() {
  const int64_t n = (int64_t)1 << 28, n4 = n / 4;  // 1 GiB per array
  float *a, *b, *c, *sink;
  unsigned long long *ctr;
  CK(hipMalloc(&ctr, 8 * 512 * 1024));
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4));
  const double cp = 2.0 * n * 4 / 1e9, tr = 6.0 * n * 4 / 1e9;
#define RUN(label, bytes, ...) do { float ms = time_ms([&] { __VA_ARGS__; }); printf("%-46s %8.1f us  %7.0f GB/s\n", label, ms * 1e3, bytes / (ms * 1e-3)); } while (0)
  RUN("copy grid-stride 2048 blk U=1 (round 5's probe)", cp, (k_copy_gs<1, false><<<2048, 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy grid-stride 2048 blk U=4", cp, (k_copy_gs<4, false><<<2048, 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy grid-stride 2048 blk U=4 nt", cp, (k_copy_gs<4, true><<<2048, 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy grid-stride 4096 blk U=4 nt", cp, (k_copy_gs<4, true><<<4096, 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy grid-stride 1024 blk U=8 nt", cp, (k_copy_gs<8, true><<<1024, 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy grid-stride 8192 blk U=2 nt", cp, (k_copy_gs<2, true><<<8192, 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy chunk/block 2048 blk U=4 nt", cp, (k_copy_chunk<4, true><<<2048, 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy chunk/block 2048 blk U=4", cp, (k_copy_chunk<4, false><<<2048, 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy one-shot U=1", cp, (k_copy_once<1, false><<<(unsigned)(n4 / 256), 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy one-shot U=4", cp, (k_copy_once<4, false><<<(unsigned)(n4 / 1024), 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy one-shot U=4 nt", cp, (k_copy_once<4, true><<<(unsigned)(n4 / 1024), 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy one-shot U=8 nt", cp, (k_copy_once<8, true><<<(unsigned)(n4 / 2048), 256>>>((f4 *)a, (f4 *)b, n4)));
  RUN("copy ticket (wave tiles) 2048 blk U=4", cp, CK(hipMemsetAsync(ctr, 0, 8, 0)); (k_copy_ticket<4, false><<<2048, 256>>>((f4 *)a, (f4 *)b, n4, ctr)));
  RUN("copy ticket-ahead 2048 blk U=4", cp, CK(hipMemsetAsync(ctr, 0, 8, 0)); (k_copy_ticket2<4, false><<<2048, 256>>>((f4 *)a, (f4 *)b, n4, ctr)));
  RUN("hipMemcpyDtoD", cp, CK(hipMemcpyAsync(b, a, n * 4, hipMemcpyDeviceToDevice, 0)));
  RUN("read only U=4 nt (1 GiB)", (n * 4 / 1e9), (k_read<4><<<2048, 256>>>((f4 *)a, sink, n4)));
  RUN("write only nt (1 GiB)", (n * 4 / 1e9), (k_write<<<2048, 256>>>((f4 *)b, n4)));
  RUN("read only one-shot (1 GiB)", (n * 4 / 1e9), (k_read_once<<<(unsigned)(n4 / 256), 256>>>((f4 *)a, sink, n4)));
  RUN("write only one-shot (1 GiB)", (n * 4 / 1e9), (k_write_once<<<(unsigned)(n4 / 256), 256>>>((f4 *)b, n4)));
  RUN("triad one-shot U=1", tr, (k_triad_once<1, false><<<(unsigned)(n4 / 256), 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4)));
  RUN("triad one-shot U=1 nt", tr, (k_triad_once<1, true><<<(unsigned)(n4 / 256), 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4)));
  RUN("triad one-shot U=2", tr, (k_triad_once<2, false><<<(unsigned)(n4 / 512), 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4)));
  RUN("triad ticket-ahead 2048 blk U=2 nt", tr, CK(hipMemsetAsync(ctr, 0, 8, 0)); (k_triad_ticket<2, true><<<2048, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4, ctr)));
  RUN("triad 16 counters 2048 blk U=1 nt", tr, CK(hipMemsetAsync(ctr, 0, 8 * 512 * 16, 0)); (k_triad_mticket<1, true><<<2048, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4, ctr, 16)));
  RUN("triad 64 counters 2048 blk U=1 nt", tr, CK(hipMemsetAsync(ctr, 0, 8 * 512 * 64, 0)); (k_triad_mticket<1, true><<<2048, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4, ctr, 64)));
  RUN("triad 256 counters 2048 blk U=1 nt", tr, CK(hipMemsetAsync(ctr, 0, 8 * 512 * 256, 0)); (k_triad_mticket<1, true><<<2048, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4, ctr, 256)));
  RUN("triad 1024 counters 2048 blk U=1 nt", tr, CK(hipMemsetAsync(ctr, 0, 8 * 512 * 1024, 0)); (k_triad_mticket<1, true><<<2048, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4, ctr, 1024)));
  RUN("triad 256 counters 2048 blk U=2 nt", tr, CK(hipMemsetAsync(ctr, 0, 8 * 512 * 256, 0)); (k_triad_mticket<2, true><<<2048, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4, ctr, 256)));
  RUN("triad 256 counters 1024 blk U=2 nt", tr, CK(hipMemsetAsync(ctr, 0, 8 * 512 * 256, 0)); (k_triad_mticket<2, true><<<1024, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4, ctr, 256)));
  RUN("triad grid-stride 2048 blk U=1 (round 5's probe)", tr, (k_triad_gs<1, false><<<2048, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4)));
  RUN("triad grid-stride 2048 blk U=2 nt", tr, (k_triad_gs<2, true><<<2048, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4)));
  RUN("triad grid-stride 4096 blk U=2 nt", tr, (k_triad_gs<2, true><<<4096, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4)));
  RUN("triad grid-stride 1024 blk U=4 nt", tr, (k_triad_gs<4, true><<<1024, 256>>>((f4 *)a, (f4 *)b, (f4 *)c, n4)));
  return 0;
}
