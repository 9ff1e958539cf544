// Streaming kernels of the epoch that surround the layer: masked-ReLU backward, the
// cross-entropy over labelled rows (mrgcn/tasks/node_classification.py:439-444), the global
// gradient norm + clip coefficient (clip_grad_norm_(…, 1.0), :192) and Adam (:35-37, :193).
// All are HBM-bound: float4 accesses, grid-stride over <= 2048 blocks.  At AM scale the
// Adam pass over weight_I (2.67 GB x 7 streams) dominates the epoch.
#include <cstdlib>

#include "common.hpp"

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

// sum of squares -> double accumulator (one atomic per block)
__global__ void k_sumsq(const float *__restrict__ x, int64_t n, double *__restrict__ accum) {
  const int64_t nv = n >> 2;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = x4[i];
    s = fmaf(v.x, v.x, s);
    s = fmaf(v.y, v.y, s);
    s = fmaf(v.z, v.z, s);
    s = fmaf(v.w, v.w, s);
  }
  if (blockIdx.x == 0)
    for (int64_t i = (nv << 2) + threadIdx.x; i < n; i += blockDim.x) s = fmaf(x[i], x[i], s);
  float t = block_sum(s);
  if (threadIdx.x == 0) atomicAdd(accum, (double)t);
}

// clip_grad_norm_: coef = min(1, max_norm / (norm + 1e-6))
// Adam on a [B][slab] parameter whose gradient lives only in some 1024-float chunks of each slab
// (the same chunks in every slab: weight_I with few labelled nodes).  `ever[c]` = 0: chunk c never
// received gradient — g = m = v = 0, the update is a no-op and nothing is touched (weight_decay must
// be 0); `cur[c]` = 0 but ever: the gradient is zero this step (g is not read: its chunk may be unwritten).
__global__ __launch_bounds__(256) void k_adam_chunked(float *__restrict__ p, const float *__restrict__ g,
                                                      float *__restrict__ m, float *__restrict__ v,
                                                      int64_t slab4, int B, int64_t nch,
                                                      const uint8_t *__restrict__ cur,
                                                      const uint8_t *__restrict__ ever, float lr, float b1,
                                                      float b2, float eps, float bc1, float bc2_sqrt,
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
  // the loop of k_adam (one linear stream over the whole parameter) with the chunk flags looked up
  // per float4: q = position inside its slab, chunk = q / 256
  const int64_t nv = slab4 * B;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = i % slab4;
    const int64_t c = q >> 8;
    if (!ever[c]) continue;
    float4 P = p4[i], M = m4[i], V = v4[i];
    float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cur[c]) G = g4[i];
    upd(P.x, G.x, M.x, V.x);
    upd(P.y, G.y, M.y, V.y);
    upd(P.z, G.z, M.z, V.z);
    upd(P.w, G.w, M.w, V.w);
    p4[i] = P;
    m4[i] = M;
    v4[i] = V;
  }
}

// Adam on a [B][N][F] parameter whose gradient and moments are kept NODE-MAJOR ([N][B][F]: the B rows
// of a node are one contiguous block).  Nodes that never had gradient (`ever` = 0) are skipped in any
// node numbering — with few labelled nodes that is half of the AM node table — and nodes without
// gradient this step (`cur` = 0) are updated with g = 0 without reading their (unwritten) block.
// A block transposes a tile of T consecutive nodes through LDS: the parameter tile enters and leaves
// as B runs of T*F floats (whole cache lines), gradient and moments as one contiguous run per node.
// tools/micro/adam_nodemajor.hip: 5.1 TB/s — 2.36 ms for the AM table at 50 % live nodes, 3.25 ms with
// every node live, against 3.36 ms for the 7-stream k_adam.
constexpr int kNmTB = 512;  // tools/micro/adam_nodemajor.hip: 512 threads 3 % ahead of 256 at T = 32
template <int T>
__global__ __launch_bounds__(kNmTB) void k_adam_nodemajor(float *__restrict__ p, const float *__restrict__ g,
                                                        float *__restrict__ m, float *__restrict__ v, int64_t N,
                                                        int B, int F, const uint8_t *__restrict__ cur,
                                                        const uint8_t *__restrict__ ever, float lr, float b1,
                                                        float b2, float eps, float bc1, float bc2_sqrt,
                                                        const float *__restrict__ scale,
                                                        const float *__restrict__ bc_dev) {
  extern __shared__ __align__(16) float s_p[];  // [B][RS]
  __shared__ int s_any;
  if (bc_dev) {
    bc1 = bc_dev[0];
    bc2_sqrt = bc_dev[1];
  }
  const float sc = scale ? *scale : 1.f;
  const float step = lr / bc1;
  auto upd = [&](float &pp, float gg, float &mm, float &vv) {  // == k_adam with wd = 0
    gg *= sc;
    mm = fmaf(b1, mm, (1.f - b1) * gg);
    vv = fmaf(b2, vv, (1.f - b2) * gg * gg);
    float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step * (mm / denom);
  };
  const int RS = T * F + 4;
  const int nf4 = (B * F) >> 2;  // float4s of one node's block
  const int64_t slab = N * F;
  const int64_t ntiles = (N + T - 1) / T;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t j0 = tile * T;
    const int nt = (int)((N - j0 < T) ? N - j0 : T);
    __syncthreads();  // s_any / s_p of the previous tile are done with
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    if ((int)threadIdx.x < nt && ever[j0 + threadIdx.x]) s_any = 1;
    __syncthreads();
    if (!s_any) continue;  // block uniform: nothing in this tile ever had gradient
    const int run = nt * F;
    if ((run & 3) == 0) {
      const int run4 = run >> 2;
      for (int q = threadIdx.x; q < B * run4; q += kNmTB) {
        const int b = q / run4, x = q - b * run4;
        *reinterpret_cast<float4 *>(&s_p[b * RS + 4 * x]) =
            *reinterpret_cast<const float4 *>(p + (int64_t)b * slab + j0 * F + 4 * x);
      }
    } else {  // the last, partial tile
      for (int q = threadIdx.x; q < B * run; q += kNmTB) {
        const int b = q / run, x = q - b * run;
        s_p[b * RS + x] = p[(int64_t)b * slab + j0 * F + x];
      }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nt * nf4; q += kNmTB) {
      const int t = q / nf4, w = q - t * nf4;
      if (!ever[j0 + t]) continue;
      const int64_t i4 = (j0 + t) * (int64_t)nf4 + w;
      float4 G = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cur[j0 + t]) G = reinterpret_cast<const float4 *>(g)[i4];
      float4 M = reinterpret_cast<const float4 *>(m)[i4], V = reinterpret_cast<const float4 *>(v)[i4];
      const int e0 = 4 * w;
      int b = e0 / F, o = e0 - b * F;
      float *pp = &s_p[b * RS + t * F + o];
      upd(*pp, G.x, M.x, V.x);
      if (++o == F) { o = 0; ++b; } pp = &s_p[b * RS + t * F + o];
      upd(*pp, G.y, M.y, V.y);
      if (++o == F) { o = 0; ++b; } pp = &s_p[b * RS + t * F + o];
      upd(*pp, G.z, M.z, V.z);
      if (++o == F) { o = 0; ++b; } pp = &s_p[b * RS + t * F + o];
      upd(*pp, G.w, M.w, V.w);
      reinterpret_cast<float4 *>(m)[i4] = M;
      reinterpret_cast<float4 *>(v)[i4] = V;
    }
    __syncthreads();
    if ((run & 3) == 0) {
      const int run4 = run >> 2;
      for (int q = threadIdx.x; q < B * run4; q += kNmTB) {
        const int b = q / run4, x = q - b * run4;
        *reinterpret_cast<float4 *>(p + (int64_t)b * slab + j0 * F + 4 * x) =
            *reinterpret_cast<const float4 *>(&s_p[b * RS + 4 * x]);
      }
    } else {
      for (int q = threadIdx.x; q < B * run; q += kNmTB) {
        const int b = q / run, x = q - b * run;
        p[(int64_t)b * slab + j0 * F + x] = s_p[b * RS + x];
      }
    }
  }
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

extern "C" {

int mrgcn_relu_bwd_f32(const float *dY, const float *Y, int64_t n, float *out, void *stream) {
  MRGCN_REQUIRE(dY && Y && out, "NULL");
  MRGCN_REQUIRE((((uintptr_t)dY | (uintptr_t)Y | (uintptr_t)out) & 15) == 0, "16-byte alignment");
  if (n == 0) return MRGCN_OK;
  k_relu_bwd<<<dim3(stream_grid(n >> 2)), dim3(kTB), 0, (hipStream_t)stream>>>(dY, Y, n, out);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_rows_nonzero_f32(const float *X, int64_t ld, int32_t F, int64_t nrows, uint8_t *flags,
                           void *stream) {
  MRGCN_REQUIRE(X && flags, "NULL");
  MRGCN_REQUIRE(F > 0 && ld >= F, "F / ld");
  if (nrows == 0) return MRGCN_OK;
  const int vec_ok = (ld % 4 == 0) && (((uintptr_t)X & 15) == 0);
  k_rows_nonzero<<<dim3(stream_grid(nrows)), dim3(kTB), 0, (hipStream_t)stream>>>(X, ld, F, nrows, flags, vec_ok);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_sumsq_accum_f32(const float *x, int64_t n, double *accum, void *stream) {
  MRGCN_REQUIRE(x && accum, "NULL");
  MRGCN_REQUIRE(((uintptr_t)x & 15) == 0, "16-byte alignment");
  if (n == 0) return MRGCN_OK;
  k_sumsq<<<dim3(stream_grid(n >> 2)), dim3(kTB), 0, (hipStream_t)stream>>>(x, n, accum);
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
  static const int nt_mode = getenv("MRGCN_ADAM_NT") ? atoi(getenv("MRGCN_ADAM_NT")) : 0;
  static const int grid_cap = getenv("MRGCN_ADAM_GRID") ? atoi(getenv("MRGCN_ADAM_GRID")) : 8192;
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

int mrgcn_adam_step_chunked_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                int64_t slab_elems, int32_t B, const uint8_t *cur, const uint8_t *ever,
                                float lr, float beta1, float beta2, float eps, int64_t step,
                                const float *bc_dev, const float *grad_scale, void *stream) {
  MRGCN_REQUIRE(param && grad && exp_avg && exp_avg_sq && cur && ever, "NULL");
  MRGCN_REQUIRE(step >= 1 || bc_dev, "step counts from 1");
  MRGCN_REQUIRE(slab_elems > 0 && slab_elems % 4 == 0 && B > 0, "slab_elems must be a multiple of 4");
  MRGCN_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "16-byte alignment");
  const double bc1 = bc_dev ? 1.0 : 1.0 - pow((double)beta1, (double)step);
  const double bc2 = bc_dev ? 1.0 : 1.0 - pow((double)beta2, (double)step);
  const int64_t nch = (slab_elems + 1023) >> 10;
  int64_t blocks = ((slab_elems >> 2) * B + 255) / 256;
  if (blocks > 8192) blocks = 8192;  // as mrgcn_adam_step_f32
  mrgcn::k_adam_chunked<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
      param, grad, exp_avg, exp_avg_sq, slab_elems >> 2, B, nch, cur, ever, lr, beta1, beta2, eps, (float)bc1,
      (float)sqrt(bc2), grad_scale, bc_dev);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_adam_step_nodemajor_f32(float *param, const float *grad_nm, float *exp_avg_nm, float *exp_avg_sq_nm,
                                  int64_t N, int32_t B, int32_t F, const uint8_t *node_cur,
                                  const uint8_t *node_ever, float lr, float beta1, float beta2, float eps,
                                  int64_t step, const float *bc_dev, const float *grad_scale, void *stream) {
  MRGCN_REQUIRE(param && grad_nm && exp_avg_nm && exp_avg_sq_nm && node_cur && node_ever, "NULL");
  MRGCN_REQUIRE(step >= 1 || bc_dev, "step counts from 1");
  MRGCN_REQUIRE(N > 0 && B > 0 && F > 0 && (N * F) % 4 == 0 && (B * F) % 4 == 0, "N*F and B*F must be multiples of 4");
  MRGCN_REQUIRE((((uintptr_t)param | (uintptr_t)grad_nm | (uintptr_t)exp_avg_nm | (uintptr_t)exp_avg_sq_nm) & 15) == 0,
                "16-byte alignment");
  const double bc1 = bc_dev ? 1.0 : 1.0 - pow((double)beta1, (double)step);
  const double bc2 = bc_dev ? 1.0 : 1.0 - pow((double)beta2, (double)step);
  auto lds_for = [&](int T) { return (size_t)B * (T * F + 4) * sizeof(float); };
  const int T = lds_for(32) <= 60 * 1024 ? 32 : lds_for(16) <= 60 * 1024 ? 16 : 8;
  MRGCN_REQUIRE(lds_for(T) <= 60 * 1024, "B * F too large for the node-major Adam tile");
  int64_t blocks = (N + T - 1) / T;
  if (blocks > 3072) blocks = 3072;
  hipStream_t s = (hipStream_t)stream;
#define NM_GO(T_)                                                                                          \
  mrgcn::k_adam_nodemajor<T_><<<dim3((unsigned)blocks), dim3(mrgcn::kNmTB), lds_for(T_), s>>>(                      \
      param, grad_nm, exp_avg_nm, exp_avg_sq_nm, N, B, F, node_cur, node_ever, lr, beta1, beta2, eps,      \
      (float)bc1, (float)sqrt(bc2), grad_scale, bc_dev)
  if (T == 32) NM_GO(32);
  else if (T == 16) NM_GO(16);
  else NM_GO(8);
#undef NM_GO
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
  MRGCN_HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), s));
  if (dlogits) {
    MRGCN_REQUIRE(ldd >= C && num_rows > 0, "ldd / num_rows");
    MRGCN_HIP_TRY(hipMemsetAsync(dlogits, 0, (size_t)num_rows * ldd * sizeof(float), s));
  }
  int grid = (int)((n + kTB - 1) / kTB);
  if (grid > 1024) grid = 1024;
  k_xent<<<dim3(grid), dim3(kTB), 0, s>>>(logits, ld, C, idx, target, n, loss, dlogits, ldd);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // extern "C"
