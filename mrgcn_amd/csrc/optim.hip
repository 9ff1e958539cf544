// Streaming kernels of the epoch that surround the layer: masked-ReLU backward, the
// cross-entropy over labelled rows (mrgcn/tasks/node_classification.py:439-444), the global
// gradient norm + clip coefficient (clip_grad_norm_(…, 1.0), :192) and Adam (:35-37, :193).
// All are HBM-bound: float4 accesses, grid-stride over <= 2048 blocks.  At AM scale the
// Adam pass over weight_I (2.67 GB x 7 streams) dominates the epoch.
#include <cstdlib>

#include <algorithm>

#include "common.hpp"
#include "config.hpp"

namespace mrgcn {
namespace {

constexpr int kTB = 256;

inline int stream_grid(int64_t n_vec) {
  int64_t b = (n_vec + kTB - 1) / kTB;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
  return x;
}

__device__ __forceinline__ float block_sum(float x) {  // result valid in thread 0
  __shared__ float s_part[kTB / kWave];
  x = wave_sum(x);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_part[w] = x;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < kTB / kWave; ++i) t += s_part[i];
  return t;
}

// the same on row-strided operands (a layer output kept in a buffer with padded rows): 16 lanes per row
__global__ void k_relu_bwd_rows(const float *__restrict__ dY, int64_t ldd, const float *__restrict__ Y, int64_t ldy,
                                int64_t rows, int F, float *__restrict__ out, int64_t ldo) {
  const int f0 = threadIdx.x & 15;
  const int64_t step = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; r < rows; r += step)
    for (int f = f0; f < F; f += 16) out[r * ldo + f] = Y[r * ldy + f] > 0.f ? dY[r * ldd + f] : 0.f;
}

// out = dY * (Y > 0)
__global__ void k_relu_bwd(const float *__restrict__ dY, const float *__restrict__ Y, int64_t n,
                           float *__restrict__ out) {
  const int64_t nv = n >> 2;
  const float4 *d4 = reinterpret_cast<const float4 *>(dY);
  const float4 *y4 = reinterpret_cast<const float4 *>(Y);
  float4 *o4 = reinterpret_cast<float4 *>(out);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 d = d4[i], y = y4[i], o;
    o.x = y.x > 0.f ? d.x : 0.f;
    o.y = y.y > 0.f ? d.y : 0.f;
    o.z = y.z > 0.f ? d.z : 0.f;
    o.w = y.w > 0.f ? d.w : 0.f;
    o4[i] = o;
  }
  if (blockIdx.x == 0)
    for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += blockDim.x)
      out[i] = Y[i] > 0.f ? dY[i] : 0.f;
}

// flags[i] = 1 when row i of X holds anything but zeros (NaN counts as something)
__global__ void k_rows_nonzero(const float *__restrict__ X, int64_t ld, int F, int64_t nrows,
                               uint8_t *__restrict__ flags, int vec_ok) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrows;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float *row = X + i * ld;
    bool nz = false;
    int q = 0;
    if (vec_ok)
      for (; q + 4 <= F; q += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(row + q);
        nz |= (t.x != 0.f) | (t.y != 0.f) | (t.z != 0.f) | (t.w != 0.f);
      }
    for (; q < F; ++q) nz |= row[q] != 0.f;
    flags[i] = nz ? 1 : 0;
  }
}

// the same for wide rows (F > 32): a wave per row, lanes along the row (a thread per row reads 64 rows of F floats
// with a stride of F: 104 us for 14 541 x 200 at the FB15k-237 shape against a few us of bytes)
__global__ __launch_bounds__(256) void k_rows_nonzero_wide(const float *__restrict__ X, int64_t ld, int F, int64_t nrows,
                                                           uint8_t *__restrict__ flags) {
  const int lane = threadIdx.x & 63;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < nrows;
       i += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const float *row = X + i * ld;
    bool nz = false;
    for (int q = lane; q < F; q += 64) nz |= row[q] != 0.f;
    const uint64_t any = __ballot(nz);
    if (lane == 0) flags[i] = any ? 1 : 0;
  }
}

// sum of squares -> double accumulator (one atomic per block).  At most 512 blocks, four 16-byte pieces in flight per
// thread: a double atomic on ONE address costs ~12 ns — with 2 048 blocks the adds alone took 25 of the kernel's 31 us
// (AIFB's 48 MB gradient, FB15k-237's 23 MB).
__global__ void k_sumsq(const float *__restrict__ x, int64_t n, double *__restrict__ accum) {
  const int64_t nv = n >> 2;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x4[i + u * stride < nv ? i + u * stride : i];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i + u * stride < nv) {
        s = fmaf(v[u].x, v[u].x, s);
        s = fmaf(v[u].y, v[u].y, s);
        s = fmaf(v[u].z, v[u].z, s);
        s = fmaf(v[u].w, v[u].w, s);
      }
    }
  }
  if (blockIdx.x == 0)
    for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += blockDim.x) s = fmaf(x[i], x[i], s);
  float t = block_sum(s);
  if (threadIdx.x == 0) atomicAdd(accum, (double)t);
}

// clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6))
// Adam on a parameter made of `nrows` rows of `rowlen` floats (weight_I in its node-major layout: one row
// = the B*F floats of a node) whose gradient exists for some rows only — a semi-supervised epoch gives
// gradient to the nodes within reach of a label, with few labelled nodes half of the AM node table gets
// none, ever.  `ever[r]` = 0: row r never received gradient — g = m = v = 0, the update is the identity
// and nothing is read or written (weight_decay must be 0); `cur[r]` = 0 but ever: the gradient is zero
// this step (g is not read: its row may be unwritten).  Rows with cur = 1 are marked ever.
// One thread per float4 of a row (rowlen % 4 == 0; k_adam_rows_scalar: per float), rows walked by the grid.
__global__ __launch_bounds__(256) void k_adam_rows_scalar(float *__restrict__ p, const float *__restrict__ g,
                                                          float *__restrict__ m, float *__restrict__ v,
                                                          int64_t nrows, int rowlen, const uint8_t *__restrict__ cur,
                                                          uint8_t *__restrict__ ever, float lr, float b1, float b2,
                                                          float eps, float bc1, float bc2_sqrt,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ bc_dev) {
  if (bc_dev) {
    bc1 = bc_dev[0];
    bc2_sqrt = bc_dev[1];
  }
  const float sc = scale ? *scale : 1.f;
  const float step = lr / bc1;
  const int64_t n = nrows * rowlen;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / rowlen;
    const bool c = cur[r] != 0;
    if (!c && !ever[r]) continue;
    float gg = (c ? g[i] : 0.f) * sc;
    float mm = fmaf(b1, m[i], (1.f - b1) * gg);
    float vv = fmaf(b2, v[i], (1.f - b2) * gg * gg);
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] -= step * (mm / denom);
    m[i] = mm;
    v[i] = vv;
  }
}
// rows that took gradient count as `ever` from now on (after the update pass: no thread may see the flag change
// under its feet)
__global__ void k_rows_mark_ever(const uint8_t *__restrict__ cur, uint8_t *__restrict__ ever, int64_t nrows) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < nrows && cur[r]) ever[r] = 1;
}

template <int RB>  // rows per block step
__global__ __launch_bounds__(256) void k_adam_rows(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v,
                                                   int64_t nrows, int rowlen4, const uint8_t *__restrict__ cur,
                                                   uint8_t *__restrict__ ever, float lr, float b1, float b2,
                                                   float eps, float bc1, float bc2_sqrt,
                                                   const float *__restrict__ scale,
                                                   const float *__restrict__ bc_dev) {
  if (bc_dev) {
    bc1 = bc_dev[0];
    bc2_sqrt = bc_dev[1];
  }
  const float sc = scale ? *scale : 1.f;
  const float step = lr / bc1;
  float4 *p4 = reinterpret_cast<float4 *>(p);
  const float4 *g4 = reinterpret_cast<const float4 *>(g);
  float4 *m4 = reinterpret_cast<float4 *>(m);
  float4 *v4 = reinterpret_cast<float4 *>(v);
  auto upd = [&](float &pp, float gg, float &mm, float &vv) {  // == k_adam with wd = 0
    gg *= sc;
    mm = fmaf(b1, mm, (1.f - b1) * gg);
    vv = fmaf(b2, vv, (1.f - b2) * gg * gg);
    float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step * (mm / denom);
  };
  // a block takes RB consecutive rows per step: RB * rowlen4 float4s, thread t the t-th, t + 256-th, ...
  const int per = RB * rowlen4;
  for (int64_t r0 = (int64_t)blockIdx.x * RB; r0 < nrows; r0 += (int64_t)gridDim.x * RB) {
    for (int t = threadIdx.x; t < per; t += blockDim.x) {
      const int rr = t / rowlen4;
      const int64_t r = r0 + rr;
      if (r >= nrows) break;
      const bool c = cur[r] != 0;
      if (!c && !ever[r]) continue;
      const int64_t i = r * rowlen4 + (t - rr * rowlen4);
      float4 P = p4[i], M = m4[i], V = v4[i];
      float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c) G = g4[i];
      upd(P.x, G.x, M.x, V.x);
      upd(P.y, G.y, M.y, V.y);
      upd(P.z, G.z, M.z, V.z);
      upd(P.w, G.w, M.w, V.w);
      p4[i] = P;
      m4[i] = M;
      v4[i] = V;
      if (c && t == rr * rowlen4) ever[r] = 1;  // (readers of the same row see cur = 1 either way)
    }
  }
}

// Adam on the rows `index[c]` of a parameter (rows of rowlen4 16-byte pieces) with the gradient of row index[c] in
// row c of a COMPACT gradient — the literal operand of a featureless layer without bases (functional._SpmmLiteral): the
// only rows of the (R*N) x F table that ever see gradient are the graph's compact columns, whatever the label set; the
// others keep zero moments and never move (weight_decay = 0).  One thread per piece, a one-shot grid; the arithmetic is
// k_adam's with wd = 0.  `index` holds distinct rows.
__global__ __launch_bounds__(256) void k_adam_index_rows(float *__restrict__ p, const float *__restrict__ g,
                                                         int64_t ldg4, float *__restrict__ m, float *__restrict__ v,
                                                         const int32_t *__restrict__ index, int64_t n, int rowlen4,
                                                         float lr, float b1, float b2, float eps, float bc1,
                                                         float bc2_sqrt, const float *__restrict__ scale,
                                                         const float *__restrict__ bc_dev) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * rowlen4) return;
  if (bc_dev) {
    bc1 = bc_dev[0];
    bc2_sqrt = bc_dev[1];
  }
  const float sc = scale ? *scale : 1.f;
  const float step = lr / bc1;
  const int64_t c = i / rowlen4;
  const int q = (int)(i - c * rowlen4);
  const int64_t at = (int64_t)index[c] * rowlen4 + q;
  float4 *p4 = reinterpret_cast<float4 *>(p) + at;
  float4 *m4 = reinterpret_cast<float4 *>(m) + at;
  float4 *v4 = reinterpret_cast<float4 *>(v) + at;
  const float4 G = reinterpret_cast<const float4 *>(g)[c * ldg4 + q];
  float4 P = *p4, M = *m4, V = *v4;
  auto upd = [&](float &pp, float gg, float &mm, float &vv) {  // == k_adam with wd = 0
    gg *= sc;
    mm = fmaf(b1, mm, (1.f - b1) * gg);
    vv = fmaf(b2, vv, (1.f - b2) * gg * gg);
    float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step * (mm / denom);
  };
  upd(P.x, G.x, M.x, V.x);
  upd(P.y, G.y, M.y, V.y);
  upd(P.z, G.z, M.z, V.z);
  upd(P.w, G.w, M.w, V.w);
  *p4 = P;
  *m4 = M;
  *v4 = V;
}

__global__ void k_clip_coef(const double *__restrict__ sumsq, float max_norm, float *__restrict__ coef,
                            float *__restrict__ norm) {
  float nrm = (float)sqrt(*sumsq);
  float c = max_norm / (nrm + 1e-6f);
  *coef = c < 1.f ? c : 1.f;
  if (norm) *norm = nrm;
}

// torch.optim.Adam (amsgrad = False, maximize = False), gradient pre-scaled by *scale
template <bool NT>
__global__ void k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                       float *__restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                       float wd, float bc1, float bc2_sqrt, const float *__restrict__ scale,
                       const float *__restrict__ bc_dev) {
  if (bc_dev) {  // bias corrections kept on the device (graph-capturable steps)
    bc1 = bc_dev[0];
    bc2_sqrt = bc_dev[1];
  }
  const float sc = scale ? *scale : 1.f;
  const float step = lr / bc1;
  const int64_t nv = n >> 2;
  float4 *p4 = reinterpret_cast<float4 *>(p);
  const float4 *g4 = reinterpret_cast<const float4 *>(g);
  float4 *m4 = reinterpret_cast<float4 *>(m);
  float4 *v4 = reinterpret_cast<float4 *>(v);
  auto upd = [&](float &pp, float gg, float &mm, float &vv) {
    gg *= sc;
    if (wd != 0.f) gg = fmaf(wd, pp, gg);
    mm = fmaf(b1, mm, (1.f - b1) * gg);
    vv = fmaf(b2, vv, (1.f - b2) * gg * gg);
    float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step * (mm / denom);
  };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 P, G, M, V;
    if (NT) {  // streamed once per step: keep it out of the way of resident data
      using v4f = __attribute__((ext_vector_type(4))) float;
      auto ld = [](const float4 *q) {
        v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(q));
        return make_float4(t.x, t.y, t.z, t.w);
      };
      P = ld(&p4[i]); G = ld(&g4[i]); M = ld(&m4[i]); V = ld(&v4[i]);
    } else {
      P = p4[i]; G = g4[i]; M = m4[i]; V = v4[i];
    }
    upd(P.x, G.x, M.x, V.x);
    upd(P.y, G.y, M.y, V.y);
    upd(P.z, G.z, M.z, V.z);
    upd(P.w, G.w, M.w, V.w);
    if (NT) {
      using v4f = __attribute__((ext_vector_type(4))) float;
      auto st = [](float4 x, float4 *q) {
        v4f t = {x.x, x.y, x.z, x.w};
        __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(q));
      };
      st(P, &p4[i]); st(M, &m4[i]); st(V, &v4[i]);
    } else {
      p4[i] = P; m4[i] = M; v4[i] = V;
    }
  }
  if (blockIdx.x == 0)
    for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += blockDim.x) {
      float P = p[i], M = m[i], V = v[i];
      upd(P, g[i], M, V);
      p[i] = P;
      m[i] = M;
      v[i] = V;
    }
}

// The same with the gradient kept COMPACT: drows[i, :] = d loss / d logits[idx[i], :] (n x C) — the dense N x C
// gradient (73 MB at the AM shape, all zeros but n rows) is only formed by the backward (k_xent_scatter), scaled by
// the upstream gradient on the way: no dense multiply, and the rows that hold anything are known (row flags).
using xf4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(1024) void k_xent_rows(const float *__restrict__ logits, int64_t ld, int C,
                                                    const int64_t *__restrict__ idx, const int64_t *__restrict__ target,
                                                    int64_t n, float *__restrict__ loss, float *__restrict__ drows,
                                                    int single_block) {
  float my = 0.f;
  const float inv_n = 1.f / (float)n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // (the next row's index and target are fetched under this row's work: one round trip less per row and thread)
  int64_t row_n = idx[first < n ? first : n - 1], t_n = target[first < n ? first : n - 1];
  for (int64_t i = first; i < n; i += stride) {
    const float *z = logits + row_n * ld;
    const int64_t t = t_n;
    {
      const int64_t nx = i + stride < n ? i + stride : n - 1;
      row_n = idx[nx];
      t_n = target[nx];
    }
    if (C <= 16) {  // the row in registers: all its loads in flight at once (a loop over z[c] waits for each)
      float zz[16];
      if (C >= 4) {
        // four 16-byte pieces per row (dword-aligned addresses: rows may be 44 bytes), each starting inside the row
        // (min(4p, C - 4)) and shifted into place — a quarter of the load instructions of sixteen 4-byte loads: the one
        // block's CU is bound by its address unit when the rows are scattered (10 000 labelled rows: 155 us)
        xf4 v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const xf4 *>(z + (4 * p < C - 4 ? 4 * p : C - 4));
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int shift = 4 * p - (4 * p < C - 4 ? 4 * p : C - 4);  // (wave uniform; > 3: the whole piece lies past C)
          const xf4 w = v[p];
          zz[4 * p] = shift == 0 ? w.x : shift == 1 ? w.y : shift == 2 ? w.z : w.w;
          zz[4 * p + 1] = shift == 0 ? w.y : shift == 1 ? w.z : w.w;
          zz[4 * p + 2] = shift == 0 ? w.z : w.w;
          zz[4 * p + 3] = w.w;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) zz[c] = z[c < C ? c : C - 1];  // (unconditional loads at clamped addresses:
      }                                                             //  a load behind `c < C` waits for itself)
#pragma unroll
      for (int c = 0; c < 16; ++c) zz[c] = c < C ? zz[c] : -INFINITY;
      float mx = zz[0];
#pragma unroll
      for (int c = 1; c < 16; ++c) mx = fmaxf(mx, zz[c]);
      float se = 0.f, zt = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (c < C) se += expf(zz[c] - mx);
        if (c == t) zt = zz[c];
      }
      const float lse = logf(se) + mx;
      my += (lse - zt) * inv_n;
      if (drows) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c < C) drows[i * C + c] = (expf(zz[c] - lse) - (c == t ? 1.f : 0.f)) * inv_n;
      }
      continue;
    }
    float mx = z[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
    const float lse = logf(se) + mx;
    my += (lse - z[t]) * inv_n;
    if (drows)
      for (int c = 0; c < C; ++c) drows[i * C + c] = (expf(z[c] - lse) - (c == t ? 1.f : 0.f)) * inv_n;
  }
  // block sum over up to 16 waves
  __shared__ float s_part[16];
  my = wave_sum(my);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = my;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_part[i];
    if (single_block) *loss = t; else atomicAdd(loss, t);
  }
}

// dlogits[idx[i], :] += g * drows[i, :] (a node may be listed twice: atomics), flags[idx[i]] = 1
__global__ void k_xent_scatter(const float *__restrict__ drows, const int64_t *__restrict__ idx, int64_t n, int C,
                               const float *__restrict__ g, float *__restrict__ dlogits, int64_t ldd,
                               uint8_t *__restrict__ flags) {
  const float gg = g ? *g : 1.f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * C; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C;
    const int c = (int)(t - i * C);
    const float v = gg * drows[t];
    atomicAdd(&dlogits[idx[i] * ldd + c], v);
    if (flags && c == 0) flags[idx[i]] = 1;
  }
}

// ---- the optimizer tail of the SMALL parameters in two launches --------------------------------------------------
// An epoch's dense parameters besides the node table are a handful of KB-sized tensors (comp tables, W_F, biases,
// the decoder's relations).  One kernel per tensor and phase (sum of squares, Adam) is ~6 us of launch each inside a
// replayed hipGraph — 15 nodes for 5 tensors.  Here: ONE launch squares every tensor, adds the sums that arrived
// from elsewhere (the row-sparse node table's ||g||^2: device doubles), and its last block turns the total into the
// clip coefficient and advances the device step counter; ONE launch applies Adam to all of them.
constexpr int kMultiMax = 16;
struct MultiSumsq {
  const float *g[kMultiMax];
  int64_t n[kMultiMax];
  int32_t blk0[kMultiMax + 1];  // first block of each tensor
  int32_t n_tensors;
  const double *extra[kMultiMax];  // further squared norms to add (device doubles)
  int32_t n_extra;
};
// *accum += the squared norms of up to kMultiMax tensors: the launches in front of k_sumsq_multi when a model has more
// small tensors than one launch takes (an MRGCN with encoders: ~40)
__global__ __launch_bounds__(kTB) void k_sumsq_multi_accum(MultiSumsq a, double *__restrict__ accum) {
  int t = 0;
  while (t + 1 < a.n_tensors && (int)blockIdx.x >= a.blk0[t + 1]) ++t;
  const int nb = a.blk0[t + 1] - a.blk0[t], b = blockIdx.x - a.blk0[t];
  const float *x = a.g[t];
  const int64_t n = a.n[t], nv = n >> 2;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  float s = 0.f;
  for (int64_t i = (int64_t)b * kTB + threadIdx.x; i < nv; i += (int64_t)nb * kTB) {
    float4 v = x4[i];
    s = fmaf(v.x, v.x, s);
    s = fmaf(v.y, v.y, s);
    s = fmaf(v.z, v.z, s);
    s = fmaf(v.w, v.w, s);
  }
  if (b == 0)
    for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += kTB) s = fmaf(x[i], x[i], s);
  const float tot = block_sum(s);
  if (threadIdx.x == 0 && tot != 0.f) atomicAdd(accum, (double)tot);
}

// `accum` / `ticket` must be zero on entry; the last block leaves them zero again (self-cleaning)
__global__ __launch_bounds__(kTB) void k_sumsq_multi(MultiSumsq a, double *__restrict__ accum,
                                                     unsigned int *__restrict__ ticket, float max_norm,
                                                     double *__restrict__ sumsq_out, float *__restrict__ coef,
                                                     float *__restrict__ norm, int64_t *__restrict__ step, float b1,
                                                     float b2, float *__restrict__ bc) {
  int t = 0;
  while (t + 1 < a.n_tensors && (int)blockIdx.x >= a.blk0[t + 1]) ++t;
  const int nb = a.blk0[t + 1] - a.blk0[t], b = blockIdx.x - a.blk0[t];
  const float *x = a.g[t];
  const int64_t n = a.n[t], nv = n >> 2;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  float s = 0.f;
  const int64_t stride = (int64_t)nb * kTB;
  for (int64_t i = (int64_t)b * kTB + threadIdx.x; i < nv; i += 4 * stride) {  // four pieces in flight (clamped loads)
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x4[i + u * stride < nv ? i + u * stride : i];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i + u * stride < nv) {
        s = fmaf(v[u].x, v[u].x, s);
        s = fmaf(v[u].y, v[u].y, s);
        s = fmaf(v[u].z, v[u].z, s);
        s = fmaf(v[u].w, v[u].w, s);
      }
    }
  }
  if (b == 0)
    for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += kTB) s = fmaf(x[i], x[i], s);
  const float tot = block_sum(s);
  if (threadIdx.x == 0) {
    atomicAdd(accum, (double)tot);
    __threadfence();
    const unsigned int seen = atomicAdd(ticket, 1u);
    if (seen == gridDim.x - 1) {  // every block's sum is in
      double total = atomicAdd(accum, 0.0);
      for (int e = 0; e < a.n_extra; ++e) total += *a.extra[e];
      if (sumsq_out) *sumsq_out = total;
      const float nrm = (float)sqrt(total);
      if (coef) {
        const float c = max_norm / (nrm + 1e-6f);
        *coef = (max_norm > 0.f && c < 1.f) ? c : 1.f;
      }
      if (norm) *norm = nrm;
      if (step) {  // ++*step; bc = {1 - b1^step, sqrt(1 - b2^step)} (k_adam_bias)
        const int64_t tt = *step + 1;
        *step = tt;
        bc[0] = (float)(1.0 - pow((double)b1, (double)tt));
        bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)tt));
      }
      *accum = 0.0;
      *ticket = 0u;
    }
  }
}

struct MultiAdam {
  float *p[kMultiMax];
  const float *g[kMultiMax];
  float *m[kMultiMax];
  float *v[kMultiMax];
  int64_t n[kMultiMax];
  float lr[kMultiMax], wd[kMultiMax];
  int32_t blk0[kMultiMax + 1];
  int32_t n_tensors;
};
__global__ __launch_bounds__(kTB) void k_adam_multi(MultiAdam a, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                    const float *__restrict__ scale, const float *__restrict__ bc_dev) {
  if (bc_dev) {
    bc1 = bc_dev[0];
    bc2_sqrt = bc_dev[1];
  }
  int t = 0;
  while (t + 1 < a.n_tensors && (int)blockIdx.x >= a.blk0[t + 1]) ++t;
  const int nb = a.blk0[t + 1] - a.blk0[t], b = blockIdx.x - a.blk0[t];
  float *p = a.p[t];
  const float *g = a.g[t];
  float *m = a.m[t], *v = a.v[t];
  const int64_t n = a.n[t];
  const float sc = scale ? *scale : 1.f, wd = a.wd[t];
  const float step = a.lr[t] / bc1;
  for (int64_t i = (int64_t)b * kTB + threadIdx.x; i < n; i += (int64_t)nb * kTB) {  // == k_adam's update
    float gg = g[i] * sc, pp = p[i], mm = m[i], vv = v[i];
    if (wd != 0.f) gg = fmaf(wd, pp, gg);
    mm = fmaf(b1, mm, (1.f - b1) * gg);
    vv = fmaf(b2, vv, (1.f - b2) * gg * gg);
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step * (mm / denom);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
  }
}

// mean cross-entropy over labelled rows + its gradient w.r.t. the logits.
// one thread per labelled row (C is small: the number of classes)
__global__ void k_xent(const float *__restrict__ logits, int64_t ld, int C,
                       const int64_t *__restrict__ idx, const int64_t *__restrict__ target, int64_t n,
                       float *__restrict__ loss, float *__restrict__ dlogits, int64_t ldd) {
  float my = 0.f;
  const float inv_n = 1.f / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float *z = logits + idx[i] * ld;
    float mx = z[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
    const float lse = logf(se) + mx;
    const int64_t t = target[i];
    my += (lse - z[t]) * inv_n;
    if (dlogits) {
      float *d = dlogits + idx[i] * ldd;
      for (int c = 0; c < C; ++c) {
        float pr = expf(z[c] - lse);
        atomicAdd(&d[c], (pr - (c == t ? 1.f : 0.f)) * inv_n);  // a node may be listed twice
      }
    }
  }
  float t = block_sum(my);
  if (threadIdx.x == 0) atomicAdd(loss, t);
}

}  // namespace
}  // namespace mrgcn

using namespace mrgcn;

namespace mrgcn {
// Streaming yardsticks (measurement only: bench.py's extra.device_copy_gbps_hip / extra.triad_gbps): what a plain
// float4 copy and a 3-read / 3-write elementwise pass (the memory shape of a dense Adam step) reach on the box.
using yf4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(256) void k_probe_copy4(const yf4 *__restrict__ a, yf4 *__restrict__ b, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}
template <bool NT>
__global__ __launch_bounds__(256) void k_probe_triad4(yf4 *__restrict__ p, yf4 *__restrict__ m, yf4 *__restrict__ v,
                                                      int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    yf4 P, M, V;
    if (NT) { P = __builtin_nontemporal_load(p + i); M = __builtin_nontemporal_load(m + i); V = __builtin_nontemporal_load(v + i); }
    else { P = p[i]; M = m[i]; V = v[i]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // Adam's arithmetic on a constant gradient
      const float g = 1e-3f;
      M[k] = fmaf(0.9f, M[k], 0.1f * g);
      V[k] = fmaf(0.999f, V[k], 0.001f * g * g);
      P[k] -= 0.01f * (M[k] / (sqrtf(V[k]) + 1e-8f));
    }
    if (NT) { __builtin_nontemporal_store(P, p + i); __builtin_nontemporal_store(M, m + i); __builtin_nontemporal_store(V, v + i); }
    else { p[i] = P; m[i] = M; v[i] = V; }
  }
}
}  // namespace mrgcn

namespace mrgcn {
// dst[row, 0:F] = src[perm[k], 0:F] where row == sorted_rows[k], zeros for every other row: the dense gradient of a
// table of which only some rows received any (the literal (R*N) x out weight_I of a layer without bases: autograd of
// graph.py:75 — torch produced it as zeros_like + index_copy_, two passes over the table).  A lane group per row, a
// binary search in the sorted list of touched rows (cache resident), one pass over dst.
__global__ __launch_bounds__(256) void k_scatter_rows_zero_fill(const int32_t *__restrict__ sorted_rows,
                                                                const int32_t *__restrict__ perm, int64_t n_touched,
                                                                const float *__restrict__ src, int64_t ldS, int F,
                                                                float *__restrict__ dst, int64_t n_rows) {
  const int P = (F + 3) >> 2;  // 16-byte pieces of a row
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t / P;
  const int piece = (int)(t - row * P);
  if (row >= n_rows) return;
  int64_t lo = 0, hi = n_touched;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_rows[mid] < row) lo = mid + 1; else hi = mid;
  }
  const bool hit = lo < n_touched && sorted_rows[lo] == row;
  const float *s = hit ? src + (int64_t)perm[lo] * ldS : nullptr;
  float *d = dst + row * F;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int f = 4 * piece + u;
    if (f < F) d[f] = hit ? s[f] : 0.f;
  }
}
// out[f] = sum over the rows m of X[m][f] (rows with row_flags[m] == 0 are neither read nor added): the bias gradient of a
// narrow layer (autograd of `AFW + self.b`, graph.py:98-101) — torch's reduction of a 1.67 M x 10 matrix along its
// long side runs at 0.12 TB/s (0.55 ms); this is one coalesced pass and a fixed-order sum of the blocks' partials.
constexpr int kColsumBlocks = 1024;
template <int FT>
__global__ __launch_bounds__(256) void k_colsum_rows(const float *__restrict__ X, int64_t ld, int64_t M, int F,
                                                     const uint8_t *__restrict__ row_flags,
                                                     float *__restrict__ partials /* [gridDim.x][FT] */) {
  float acc[FT];
#pragma unroll
  for (int f = 0; f < FT; ++f) acc[f] = 0.f;
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    if (row_flags && !row_flags[m]) continue;
    const float *row = X + m * ld;
#pragma unroll
    for (int f = 0; f < FT; ++f)
      if (f < F) acc[f] += row[f];
  }
  __shared__ float s_part[4][FT];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    float v = acc[f];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) s_part[wv][f] = v;
  }
  __syncthreads();
  if (threadIdx.x < FT)
    partials[(int64_t)blockIdx.x * FT + threadIdx.x] =
        (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}

template <int FT>
__global__ __launch_bounds__(64) void k_colsum_final(const float *__restrict__ partials, int n_blocks, int F,
                                                     float *__restrict__ out) {
  // lane (f, q): column f, blocks q, q + 64 / FT, ... in order; the q sums meet through a fixed butterfly
  constexpr int Q = 64 / FT;
  const int f = threadIdx.x % FT, q = threadIdx.x / FT;
  float v = 0.f;
  for (int b = q; b < n_blocks; b += 8 * Q) {  // eight loads in flight (one at a time: 60 us for 1 024 partials), same order
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = partials[(int64_t)(b + u * Q < n_blocks ? b + u * Q : b) * FT + f];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (b + u * Q < n_blocks) v += x[u];
  }
#pragma unroll
  for (int off = FT; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
  if (q == 0 && f < F) out[f] = v;
}
}  // namespace mrgcn

extern "C" {

int mrgcn_scatter_rows_zero_fill_f32(const int32_t *sorted_rows, const int32_t *perm, int64_t n_touched,
                                     const float *src, int64_t ldS, int32_t F, float *dst, int64_t n_rows, void *stream) {
  MRGCN_REQUIRE(dst && n_rows >= 0 && F > 0 && n_touched >= 0, "NULL / sizes");
  MRGCN_REQUIRE(n_touched == 0 || (sorted_rows && perm && src && ldS >= F), "NULL / ldS");
  if (n_rows == 0) return MRGCN_OK;
  const int64_t threads = n_rows * ((F + 3) / 4);
  mrgcn::k_scatter_rows_zero_fill<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      sorted_rows, perm, n_touched, src, ldS, F, dst, n_rows);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int64_t mrgcn_colsum_rows_workspace(int32_t F) { return F > 0 && F <= 16 ? (int64_t)mrgcn::kColsumBlocks * 16 : -1; }

int mrgcn_colsum_rows_f32(const float *X, int64_t ld, int64_t M, int32_t F, const uint8_t *row_flags, float *out,
                          float *workspace, int64_t workspace_floats, void *stream) {
  MRGCN_REQUIRE(X && out && workspace, "NULL");
  MRGCN_REQUIRE(F > 0 && F <= 16 && ld >= F && M >= 0, "F (1..16) / ld / M");
  MRGCN_REQUIRE(workspace_floats >= mrgcn_colsum_rows_workspace(F), "workspace (mrgcn_colsum_rows_workspace floats)");
  hipStream_t s = (hipStream_t)stream;
  int blocks = (int)std::min<int64_t>(mrgcn::kColsumBlocks, (M + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (F <= 4) {
    mrgcn::k_colsum_rows<4><<<dim3(blocks), dim3(256), 0, s>>>(X, ld, M, F, row_flags, workspace);
    mrgcn::k_colsum_final<4><<<dim3(1), dim3(64), 0, s>>>(workspace, blocks, F, out);
  } else if (F <= 8) {
    mrgcn::k_colsum_rows<8><<<dim3(blocks), dim3(256), 0, s>>>(X, ld, M, F, row_flags, workspace);
    mrgcn::k_colsum_final<8><<<dim3(1), dim3(64), 0, s>>>(workspace, blocks, F, out);
  } else {
    mrgcn::k_colsum_rows<16><<<dim3(blocks), dim3(256), 0, s>>>(X, ld, M, F, row_flags, workspace);
    mrgcn::k_colsum_final<16><<<dim3(1), dim3(64), 0, s>>>(workspace, blocks, F, out);
  }
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

// The yardsticks as ONE-SHOT grids (a thread per 16-byte piece): what the memory system gives a stream whose work is
// handed out in address order by the block dispatcher — 6.25 TB/s (copy) / 6.0 TB/s (triad, nontemporal) on the bench
// boxes, MI355X_MICROARCH.md's 6.29.  The `_persistent` forms are round 5's (2 048 resident blocks striding through
// the arrays: 4.8 / 4.9 TB/s): the shape a kernel with per-block state is stuck with (tools/lab/copy_lab.hip).
int mrgcn_probe_copy_f32(const float *src, float *dst, int64_t n, void *stream) {
  MRGCN_REQUIRE(src && dst && n >= 0 && n % 4 == 0, "NULL / n % 4");
  MRGCN_REQUIRE(((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, "16-byte alignment");
  if (n == 0) return MRGCN_OK;
  const int64_t n4 = n / 4;
  mrgcn::k_probe_copy4<<<dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      (const mrgcn::yf4 *)src, (mrgcn::yf4 *)dst, n4);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_probe_copy_persistent_f32(const float *src, float *dst, int64_t n, void *stream) {
  MRGCN_REQUIRE(src && dst && n >= 0 && n % 4 == 0, "NULL / n % 4");
  MRGCN_REQUIRE(((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, "16-byte alignment");
  if (n == 0) return MRGCN_OK;
  mrgcn::k_probe_copy4<<<dim3(256 * 8), dim3(256), 0, (hipStream_t)stream>>>((const mrgcn::yf4 *)src, (mrgcn::yf4 *)dst, n / 4);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_probe_triad_persistent_f32(float *p, float *m, float *v, int64_t n, void *stream) {
  MRGCN_REQUIRE(p && m && v && n >= 0 && n % 4 == 0, "NULL / n % 4");
  MRGCN_REQUIRE(((((uintptr_t)p) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0, "16-byte alignment");
  if (n == 0) return MRGCN_OK;
  mrgcn::k_probe_triad4<false><<<dim3(256 * 8), dim3(256), 0, (hipStream_t)stream>>>((mrgcn::yf4 *)p, (mrgcn::yf4 *)m, (mrgcn::yf4 *)v, n / 4);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_probe_triad_f32(float *p, float *m, float *v, int64_t n, void *stream) {
  MRGCN_REQUIRE(p && m && v && n >= 0 && n % 4 == 0, "NULL / n % 4");
  MRGCN_REQUIRE(((((uintptr_t)p) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0, "16-byte alignment");
  if (n == 0) return MRGCN_OK;
  const int64_t n4 = n / 4;
  mrgcn::k_probe_triad4<true><<<dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>((mrgcn::yf4 *)p, (mrgcn::yf4 *)m, (mrgcn::yf4 *)v, n4);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_relu_bwd_f32(const float *dY, const float *Y, int64_t n, float *out, void *stream) {
  MRGCN_REQUIRE(dY && Y && out, "NULL");
  MRGCN_REQUIRE((((uintptr_t)dY | (uintptr_t)Y | (uintptr_t)out) & 15) == 0, "16-byte alignment");
  if (n == 0) return MRGCN_OK;
  k_relu_bwd<<<dim3(stream_grid(n >> 2)), dim3(kTB), 0, (hipStream_t)stream>>>(dY, Y, n, out);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_relu_bwd_rows_f32(const float *dY, int64_t ld_dY, const float *Y, int64_t ldY, int64_t rows, int32_t F,
                            float *out, int64_t ld_out, void *stream) {
  MRGCN_REQUIRE(dY && Y && out, "NULL");
  MRGCN_REQUIRE(rows >= 0 && F > 0 && ld_dY >= F && ldY >= F && ld_out >= F, "F / leading dimensions");
  if (rows == 0) return MRGCN_OK;
  k_relu_bwd_rows<<<dim3(stream_grid(rows * 16)), dim3(kTB), 0, (hipStream_t)stream>>>(dY, ld_dY, Y, ldY, rows, F, out,
                                                                                      ld_out);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_rows_nonzero_f32(const float *X, int64_t ld, int32_t F, int64_t nrows, uint8_t *flags,
                           void *stream) {
  MRGCN_REQUIRE(X && flags, "NULL");
  MRGCN_REQUIRE(F > 0 && ld >= F, "F / ld");
  if (nrows == 0) return MRGCN_OK;
  const int vec_ok = (ld % 4 == 0) && (((uintptr_t)X & 15) == 0);
  if (F > 32) k_rows_nonzero_wide<<<dim3(stream_grid(nrows * 64)), dim3(256), 0, (hipStream_t)stream>>>(X, ld, F, nrows, flags);
  else k_rows_nonzero<<<dim3(stream_grid(nrows)), dim3(kTB), 0, (hipStream_t)stream>>>(X, ld, F, nrows, flags, vec_ok);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_sumsq_accum_f32(const float *x, int64_t n, double *accum, void *stream) {
  MRGCN_REQUIRE(x && accum, "NULL");
  MRGCN_REQUIRE(((uintptr_t)x & 15) == 0, "16-byte alignment");
  if (n == 0) return MRGCN_OK;
  k_sumsq<<<dim3(std::min(stream_grid(((n >> 2) + 3) / 4), 512)), dim3(kTB), 0, (hipStream_t)stream>>>(x, n, accum);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_clip_coef_f32(const double *sumsq, float max_norm, float *coef, float *norm, void *stream) {
  MRGCN_REQUIRE(sumsq && coef, "NULL");
  k_clip_coef<<<dim3(1), dim3(1), 0, (hipStream_t)stream>>>(sumsq, max_norm, coef, norm);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

static int adam_impl(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int64_t step,
                     const float *grad_scale, const float *bc_dev, void *stream);

int mrgcn_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                        float lr, float beta1, float beta2, float eps, float weight_decay,
                        int64_t step, const float *grad_scale, void *stream) {
  return adam_impl(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                   nullptr, stream);
}

static int adam_impl(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int64_t step,
                     const float *grad_scale, const float *bc_dev, void *stream) {
  MRGCN_REQUIRE(param && grad && exp_avg && exp_avg_sq, "NULL");
  MRGCN_REQUIRE(step >= 1, "step counts from 1");
  MRGCN_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "16-byte alignment");
  if (n == 0) return MRGCN_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int nt_mode = (int)cfg(CFG_ADAM_NT);
  const int grid_cap = (int)cfg(CFG_ADAM_GRID);
  int64_t blocks = ((n >> 2) + kTB - 1) / kTB;
  if (blocks < 1) blocks = 1;
  if (blocks > grid_cap) blocks = grid_cap;
  if (nt_mode)
    k_adam<true><<<dim3((unsigned)blocks), dim3(kTB), 0, (hipStream_t)stream>>>(
        param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
        (float)sqrt(bc2), grad_scale, bc_dev);
  else
    k_adam<false><<<dim3((unsigned)blocks), dim3(kTB), 0, (hipStream_t)stream>>>(
        param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
        (float)sqrt(bc2), grad_scale, bc_dev);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // extern "C"

namespace mrgcn {
namespace {
// step counter and Adam bias corrections on the device: ++*step; bc = {1 - b1^step, sqrt(1 - b2^step)}
__global__ void k_adam_bias(int64_t *__restrict__ step, float b1, float b2, float *__restrict__ bc) {
  const int64_t t = *step + 1;
  *step = t;
  bc[0] = (float)(1.0 - pow((double)b1, (double)t));
  bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
}
}  // namespace
}  // namespace mrgcn

extern "C" {

int mrgcn_adam_step_rows_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t nrows,
                             int32_t rowlen, const uint8_t *row_cur, uint8_t *row_ever, float lr, float beta1,
                             float beta2, float eps, int64_t step, const float *bc_dev, const float *grad_scale,
                             void *stream) {
  MRGCN_REQUIRE(param && grad && exp_avg && exp_avg_sq && row_cur && row_ever, "NULL");
  MRGCN_REQUIRE(step >= 1 || bc_dev, "step counts from 1");
  MRGCN_REQUIRE(nrows >= 0 && rowlen > 0, "nrows / rowlen");
  if (nrows == 0) return MRGCN_OK;
  const double bc1 = bc_dev ? 1.0 : 1.0 - pow((double)beta1, (double)step);
  const double bc2 = bc_dev ? 1.0 : 1.0 - pow((double)beta2, (double)step);
  const int rowlen4 = rowlen >> 2;
  hipStream_t s = (hipStream_t)stream;
  if (rowlen % 4 != 0 ||
      ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)) {
    int64_t blocks = (nrows * rowlen + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    mrgcn::k_adam_rows_scalar<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(
        param, grad, exp_avg, exp_avg_sq, nrows, rowlen, row_cur, row_ever, lr, beta1, beta2, eps, (float)bc1,
        (float)sqrt(bc2), grad_scale, bc_dev);
    mrgcn::k_rows_mark_ever<<<dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s>>>(row_cur, row_ever, nrows);
    MRGCN_HIP_TRY(hipGetLastError());
    return MRGCN_OK;
  }
  // rows per block step: about 2 float4s per thread
#define ROWS_GO(RB_)                                                                                        \
  do {                                                                                                      \
    int64_t blocks = (nrows + RB_ - 1) / RB_;                                                               \
    if (blocks > 8192) blocks = 8192;                                                                       \
    mrgcn::k_adam_rows<RB_><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(                                   \
        param, grad, exp_avg, exp_avg_sq, nrows, rowlen4, row_cur, row_ever, lr, beta1, beta2, eps,         \
        (float)bc1, (float)sqrt(bc2), grad_scale, bc_dev);                                                  \
  } while (0)
  if (rowlen4 >= 256) ROWS_GO(1);
  else if (rowlen4 >= 64) ROWS_GO(4);
  else if (rowlen4 >= 16) ROWS_GO(16);
  else ROWS_GO(64);
#undef ROWS_GO
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_adam_step_index_rows_f32(float *param, const float *grad, int64_t ld_grad, float *exp_avg,
                                   float *exp_avg_sq, const int32_t *index, int64_t n_index, int32_t rowlen, float lr,
                                   float beta1, float beta2, float eps, int64_t step, const float *bc_dev,
                                   const float *grad_scale, void *stream) {
  MRGCN_REQUIRE(param && grad && exp_avg && exp_avg_sq && index, "NULL");
  MRGCN_REQUIRE(step >= 1 || bc_dev, "step counts from 1");
  MRGCN_REQUIRE(n_index >= 0 && rowlen > 0 && rowlen % 4 == 0 && ld_grad >= rowlen && ld_grad % 4 == 0,
                "rows of whole 16-byte pieces (rowlen, ld_grad multiples of 4)");
  MRGCN_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "param / grad / moments must be 16-byte aligned");
  if (n_index == 0) return MRGCN_OK;
  const double bc1 = bc_dev ? 1.0 : 1.0 - pow((double)beta1, (double)step);
  const double bc2 = bc_dev ? 1.0 : 1.0 - pow((double)beta2, (double)step);
  const int rowlen4 = rowlen >> 2;
  const int64_t blocks = (n_index * rowlen4 + 255) / 256;
  MRGCN_REQUIRE(blocks <= 0x7fffffff, "too many rows for one launch");
  mrgcn::k_adam_index_rows<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
      param, grad, ld_grad >> 2, exp_avg, exp_avg_sq, index, n_index, rowlen4, lr, beta1, beta2, eps, (float)bc1,
      (float)sqrt(bc2), grad_scale, bc_dev);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_adam_bias_f32(int64_t *step_dev, float beta1, float beta2, float *bc_dev, void *stream) {
  MRGCN_REQUIRE(step_dev && bc_dev, "NULL");
  mrgcn::k_adam_bias<<<dim3(1), dim3(1), 0, (hipStream_t)stream>>>(step_dev, beta1, beta2, bc_dev);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_adam_step_dev_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                            float lr, float beta1, float beta2, float eps, float weight_decay,
                            const float *bc_dev, const float *grad_scale, void *stream) {
  MRGCN_REQUIRE(bc_dev, "bc_dev is NULL");
  return adam_impl(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, 1, grad_scale,
                   bc_dev, stream);
}

int mrgcn_softmax_xent_f32(const float *logits, int64_t ld, int32_t C, const int64_t *idx,
                           const int64_t *target, int64_t n, float *loss, float *dlogits,
                           int64_t ldd, int64_t num_rows, void *stream) {
  MRGCN_REQUIRE(logits && idx && target && loss, "NULL");
  MRGCN_REQUIRE(C > 0 && ld >= C && n > 0, "C / ld / n");
  hipStream_t s = (hipStream_t)stream;
  MRGCN_HIP_TRY(mrgcn::fill_async(loss, 0, sizeof(float), s));
  if (dlogits) {
    MRGCN_REQUIRE(ldd >= C && num_rows > 0, "ldd / num_rows");
    MRGCN_HIP_TRY(mrgcn::fill_async(dlogits, 0, (size_t)num_rows * ldd * sizeof(float), s));
  }
  int grid = (int)((n + kTB - 1) / kTB);
  if (grid > 1024) grid = 1024;
  k_xent<<<dim3(grid), dim3(kTB), 0, s>>>(logits, ld, C, idx, target, n, loss, dlogits, ldd);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}


int mrgcn_softmax_xent_rows_f32(const float *logits, int64_t ld, int32_t C, const int64_t *idx, const int64_t *target,
                                int64_t n, float *loss, float *drows, void *stream) {
  MRGCN_REQUIRE(logits && idx && target && loss, "NULL");
  MRGCN_REQUIRE(C > 0 && ld >= C && n > 0, "C / ld / n");
  hipStream_t s = (hipStream_t)stream;
  const int single = n <= 16384;
  if (!single) MRGCN_HIP_TRY(mrgcn::fill_async(loss, 0, sizeof(float), s));
  int grid = single ? 1 : (int)((n + 1023) / 1024);
  if (grid > 256) grid = 256;
  mrgcn::k_xent_rows<<<dim3(grid), dim3(1024), 0, s>>>(logits, ld, C, idx, target, n, loss, drows, single);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_softmax_xent_bwd_f32(const float *drows, const int64_t *idx, int64_t n, int32_t C, const float *g,
                               float *dlogits, int64_t ldd, int64_t num_rows, uint8_t *row_flags, void *stream) {
  MRGCN_REQUIRE(drows && idx && dlogits, "NULL");
  MRGCN_REQUIRE(C > 0 && ldd >= C && n > 0 && num_rows > 0, "C / ldd / n / num_rows");
  hipStream_t s = (hipStream_t)stream;
  MRGCN_HIP_TRY(mrgcn::fill_async(dlogits, 0, (size_t)num_rows * ldd * sizeof(float), s));
  if (row_flags) MRGCN_HIP_TRY(mrgcn::fill_async(row_flags, 0, (size_t)num_rows, s));
  int grid = (int)((n * C + kTB - 1) / kTB);
  if (grid > 1024) grid = 1024;
  mrgcn::k_xent_scatter<<<dim3(grid), dim3(kTB), 0, s>>>(drows, idx, n, C, g, dlogits, ldd, row_flags);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_sumsq_clip_multi_f32(int32_t n_tensors, const float *const *grads, const int64_t *numel, int32_t n_extra,
                               const double *const *extra, double *accum, uint32_t *ticket, float max_norm,
                               double *sumsq_out, float *coef, float *norm, int64_t *step_dev, float beta1,
                               float beta2, float *bc_dev, void *stream) {
  MRGCN_REQUIRE(n_tensors >= 1 && n_tensors <= mrgcn::kMultiMax && n_extra >= 0 && n_extra <= mrgcn::kMultiMax,
                "at most 16 tensors / extra sums per call, at least one tensor");
  MRGCN_REQUIRE(grads && numel && accum && ticket && (n_extra == 0 || extra), "NULL");
  MRGCN_REQUIRE(!step_dev || bc_dev, "bc_dev is NULL");
  mrgcn::MultiSumsq a{};
  a.n_tensors = n_tensors;
  a.n_extra = n_extra;
  int blk = 0;
  for (int t = 0; t < n_tensors; ++t) {
    MRGCN_REQUIRE(grads[t] && numel[t] >= 0 && ((uintptr_t)grads[t] & 15) == 0, "gradient: NULL / 16-byte alignment");
    a.g[t] = grads[t];
    a.n[t] = numel[t];
    a.blk0[t] = blk;
    int64_t nb = ((numel[t] >> 2) + kTB - 1) / kTB;
    blk += (int)(nb < 1 ? 1 : (nb > 64 ? 64 : nb));
  }
  a.blk0[n_tensors] = blk;
  for (int e = 0; e < n_extra; ++e) {
    MRGCN_REQUIRE(extra[e], "extra sum is NULL");
    a.extra[e] = extra[e];
  }
  mrgcn::k_sumsq_multi<<<dim3(blk), dim3(kTB), 0, (hipStream_t)stream>>>(a, accum, ticket, max_norm, sumsq_out, coef,
                                                                        norm, step_dev, beta1, beta2, bc_dev);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_sumsq_accum_multi_f32(int32_t n_tensors, const float *const *grads, const int64_t *numel, double *accum,
                                void *stream) {
  MRGCN_REQUIRE(n_tensors >= 1 && n_tensors <= mrgcn::kMultiMax && grads && numel && accum, "1..16 tensors per call");
  mrgcn::MultiSumsq a{};
  a.n_tensors = n_tensors;
  int blk = 0;
  for (int t = 0; t < n_tensors; ++t) {
    MRGCN_REQUIRE(grads[t] && numel[t] >= 0 && ((uintptr_t)grads[t] & 15) == 0, "gradient: NULL / 16-byte alignment");
    a.g[t] = grads[t];
    a.n[t] = numel[t];
    a.blk0[t] = blk;
    int64_t nb = (numel[t] / 4 + kTB - 1) / kTB;
    blk += (int)(nb < 1 ? 1 : (nb > 64 ? 64 : nb));
  }
  a.blk0[n_tensors] = blk;
  a.n_extra = 0;
  mrgcn::k_sumsq_multi_accum<<<dim3(blk), dim3(kTB), 0, (hipStream_t)stream>>>(a, accum);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_adam_step_multi_f32(int32_t n_tensors, float *const *params, const float *const *grads,
                              float *const *exp_avg, float *const *exp_avg_sq, const int64_t *numel, const float *lr,
                              const float *weight_decay, float beta1, float beta2, float eps, int64_t step,
                              const float *bc_dev, const float *grad_scale, void *stream) {
  MRGCN_REQUIRE(n_tensors >= 1 && n_tensors <= mrgcn::kMultiMax, "1..16 tensors per call");
  MRGCN_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && lr && weight_decay, "NULL");
  MRGCN_REQUIRE(step >= 1 || bc_dev, "step counts from 1");
  mrgcn::MultiAdam a{};
  a.n_tensors = n_tensors;
  int blk = 0;
  for (int t = 0; t < n_tensors; ++t) {
    MRGCN_REQUIRE(params[t] && grads[t] && exp_avg[t] && exp_avg_sq[t] && numel[t] >= 0, "NULL tensor");
    a.p[t] = params[t]; a.g[t] = grads[t]; a.m[t] = exp_avg[t]; a.v[t] = exp_avg_sq[t];
    a.n[t] = numel[t]; a.lr[t] = lr[t]; a.wd[t] = weight_decay[t];
    a.blk0[t] = blk;
    int64_t nb = (numel[t] + kTB - 1) / kTB;
    blk += (int)(nb < 1 ? 1 : (nb > 256 ? 256 : nb));
  }
  a.blk0[n_tensors] = blk;
  const double bc1 = bc_dev ? 1.0 : 1.0 - pow((double)beta1, (double)step);
  const double bc2 = bc_dev ? 1.0 : 1.0 - pow((double)beta2, (double)step);
  mrgcn::k_adam_multi<<<dim3(blk), dim3(kTB), 0, (hipStream_t)stream>>>(a, beta1, beta2, eps, (float)bc1,
                                                                       (float)sqrt(bc2), grad_scale, bc_dev);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // extern "C"
