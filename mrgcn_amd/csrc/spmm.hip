// Stacked-CSR sparse x dense product  Y[i,:] = sum_e val[e] * D[idx[e],:]  for gfx950.
//
// Replaces torch.mm(A.float(), dense) of mrgcn/layers/graph.py:75,:95 and its autograd
// (dDense = A^T dY) when run on the transposed view.
//
// HBM-bound gather: a dense row is only F*4 bytes (40-64 B at the R-GCN hidden sizes), so
// the design goal is many independent row gathers in flight per wave, vector loads
// (VEC floats per lane) and coalesced index streaming — not FLOPs.
//
//   short rows (<= kLongThreshold entries): a subgroup of G lanes owns one row; a 64-lane
//       wave therefore walks 64/G adjacent rows at once and every load instruction
//       carries 64/G independent row gathers.  Deterministic (fixed summation order).
//   long rows: cut at plan time into chunks of <= kChunk entries; one wave per chunk
//       strides over the entries with all 64/G slots, butterfly-reduces across slots
//       and stores one partial row; a finalise pass adds a long row's partials in chunk
//       order (no atomics -> bitwise reproducible; no pre-zeroing of Y needed).
#include "common.hpp"

namespace mrgcn {
namespace {

template <int VEC> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };

template <int VEC> __device__ __forceinline__ void load_vec(const float *p, float (&x)[VEC]) {
  using T = typename Vec<VEC>::T;
  T t = *reinterpret_cast<const T *>(p);
  const float *f = reinterpret_cast<const float *>(&t);
#pragma unroll
  for (int i = 0; i < VEC; ++i) x[i] = f[i];
}

template <int VEC>
__device__ __forceinline__ void store_row(float *y, const float (&acc)[VEC], int f0, int F,
                                          const float *bias, int relu, bool vec_ok) {
  float o[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    float v = acc[i];
    if (bias && f0 + i < F) v += bias[f0 + i];
    if (relu) v = fmaxf(v, 0.f);
    o[i] = v;
  }
  if (vec_ok && f0 + VEC <= F) {
    using T = typename Vec<VEC>::T;
    *reinterpret_cast<T *>(y + f0) = *reinterpret_cast<const T *>(o);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if (f0 + i < F) y[f0 + i] = o[i];
  }
}

// ---- short rows ----------------------------------------------------------------------
template <int G, int VEC>
__global__ __launch_bounds__(256) void k_spmm_short(SparseView v, const float *__restrict__ D,
                                                    int64_t ldD, int F, float *__restrict__ Y,
                                                    int64_t ldY, const float *__restrict__ bias,
                                                    int relu, const int32_t *__restrict__ out_index,
                                                    int store_vec_ok) {
  constexpr int SLOTS = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int slot = lane / G, q = lane % G;
  const int64_t row = wave * SLOTS + slot;
  if (row >= v.rows) return;
  const int f0 = q * VEC;
  const bool active = f0 < F;
  int32_t b = v.ptr[row], e = v.ptr[row + 1];
  if (e - b > kLongThreshold) return;  // split-row path owns this row

  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

  const float *Dq = D + f0;
  int32_t k = b;
  // 4 gathers in flight per lane
  for (; k + 4 <= e; k += 4) {
    int32_t c0 = v.idx[k], c1 = v.idx[k + 1], c2 = v.idx[k + 2], c3 = v.idx[k + 3];
    float a0 = v.val[k], a1 = v.val[k + 1], a2 = v.val[k + 2], a3 = v.val[k + 3];
    if (active) {
      float x0[VEC], x1[VEC], x2[VEC], x3[VEC];
      load_vec<VEC>(Dq + (int64_t)c0 * ldD, x0);
      load_vec<VEC>(Dq + (int64_t)c1 * ldD, x1);
      load_vec<VEC>(Dq + (int64_t)c2 * ldD, x2);
      load_vec<VEC>(Dq + (int64_t)c3 * ldD, x3);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc[i] = fmaf(a0, x0[i], acc[i]);
        acc[i] = fmaf(a1, x1[i], acc[i]);
        acc[i] = fmaf(a2, x2[i], acc[i]);
        acc[i] = fmaf(a3, x3[i], acc[i]);
      }
    }
  }
  for (; k < e; ++k) {
    int32_t c = v.idx[k];
    float a = v.val[k];
    if (active) {
      float x[VEC];
      load_vec<VEC>(Dq + (int64_t)c * ldD, x);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(a, x[i], acc[i]);
    }
  }
  if (active) {
    const int64_t orow = out_index ? (int64_t)out_index[row] : row;
    store_row<VEC>(Y + orow * ldY, acc, f0, F, bias, relu, store_vec_ok != 0);
  }
}

// ---- long rows: one wave per chunk -----------------------------------------------------
template <int G, int VEC>
__global__ __launch_bounds__(256) void k_spmm_chunks(SparseView v, const float *__restrict__ D,
                                                     int64_t ldD, int F, float *__restrict__ partials,
                                                     int ldP) {
  constexpr int SLOTS = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t chunk = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  if (chunk >= v.n_chunks) return;
  const int slot = lane / G, q = lane % G;
  const int f0 = q * VEC;
  const bool active = f0 < F;
  const int32_t b = v.chunk_beg[chunk], e = v.chunk_end[chunk];

  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  const float *Dq = D + f0;
  int32_t k = b + slot;
  for (; k + SLOTS < e; k += 2 * SLOTS) {  // two gathers in flight per lane
    int32_t c0 = v.idx[k], c1 = v.idx[k + SLOTS];
    float a0 = v.val[k], a1 = v.val[k + SLOTS];
    if (active) {
      float x0[VEC], x1[VEC];
      load_vec<VEC>(Dq + (int64_t)c0 * ldD, x0);
      load_vec<VEC>(Dq + (int64_t)c1 * ldD, x1);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc[i] = fmaf(a0, x0[i], acc[i]);
        acc[i] = fmaf(a1, x1[i], acc[i]);
      }
    }
  }
  if (k < e) {
    int32_t c = v.idx[k];
    float a = v.val[k];
    if (active) {
      float x[VEC];
      load_vec<VEC>(Dq + (int64_t)c * ldD, x);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = fmaf(a, x[i], acc[i]);
    }
  }
  // butterfly over the slots (lanes with equal q); every lane takes part
#pragma unroll
  for (int off = G; off < kWave; off <<= 1) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
  }
  if (slot == 0 && active) {
    float *p = partials + chunk * (int64_t)ldP + f0;
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if (f0 + i < F) p[i] = acc[i];
  }
}

// ---- long rows: ordered sum of partials ------------------------------------------------
__global__ void k_spmm_finalize(SparseView v, const float *__restrict__ partials, int ldP, int F,
                                float *__restrict__ Y, int64_t ldY, const float *__restrict__ bias,
                                int relu, const int32_t *__restrict__ out_index) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t li = t / F;
  int f = (int)(t - li * F);
  if (li >= v.n_long) return;
  int32_t c0 = v.long_cptr[li], c1 = v.long_cptr[li + 1];
  float s = 0.f;
  for (int32_t c = c0; c < c1; ++c) s += partials[(int64_t)c * ldP + f];
  if (bias) s += bias[f];
  if (relu) s = fmaxf(s, 0.f);
  int64_t row = v.long_row[li];
  if (out_index) row = out_index[row];
  Y[row * ldY + f] = s;
}

template <int G, int VEC>
int launch(const SparseView &v, const float *D, int64_t ldD, int F, float *Y, int64_t ldY,
           const float *bias, int relu, const int32_t *out_index, float *partials, hipStream_t s) {
  constexpr int SLOTS = kWave / G;
  const bool store_vec_ok = (ldY % VEC == 0) && (((uintptr_t)Y) % (VEC * 4) == 0);
  if (v.rows > 0) {
    int64_t waves = (v.rows + SLOTS - 1) / SLOTS;
    int64_t blocks = (waves + 3) / 4;
    k_spmm_short<G, VEC><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(v, D, ldD, F, Y, ldY, bias, relu,
                                                                       out_index, store_vec_ok ? 1 : 0);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  if (v.n_chunks > 0) {
    int64_t blocks = ((int64_t)v.n_chunks + 3) / 4;
    k_spmm_chunks<G, VEC><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(v, D, ldD, F, partials, kWsFeatures);
    MRGCN_HIP_TRY(hipGetLastError());
    int64_t threads = (int64_t)v.n_long * F;
    k_spmm_finalize<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s>>>(
        v, partials, kWsFeatures, F, Y, ldY, bias, relu, out_index);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  return MRGCN_OK;
}

// picks lanes-per-row G and vector width VEC for one feature tile of width F <= 256
int dispatch(const SparseView &v, const float *D, int64_t ldD, int64_t avail, int F, float *Y,
             int64_t ldY, const float *bias, int relu, const int32_t *out_index, float *partials,
             hipStream_t s) {
  // widest vector the operand layout allows; loads past F must stay inside the row
  // (`avail` = floats left in a row of D from this tile's first column)
  int vec = 1;
  auto ok = [&](int w) {
    int64_t padded = ((int64_t)F + w - 1) / w * w;
    return ldD % w == 0 && ((uintptr_t)D) % (w * 4) == 0 && avail >= padded;
  };
  if (ok(4)) vec = 4; else if (ok(2)) vec = 2;
  const int lanes = (F + vec - 1) / vec;  // lanes needed per row
#define MRGCN_GO(G, V) return launch<G, V>(v, D, ldD, F, Y, ldY, bias, relu, out_index, partials, s)
  if (vec == 4) {
    if (lanes <= 1) MRGCN_GO(1, 4);
    if (lanes <= 2) MRGCN_GO(2, 4);
    if (lanes <= 4) MRGCN_GO(4, 4);
    if (lanes <= 8) MRGCN_GO(8, 4);
    if (lanes <= 16) MRGCN_GO(16, 4);
    if (lanes <= 32) MRGCN_GO(32, 4);
    MRGCN_GO(64, 4);
  } else if (vec == 2) {
    if (lanes <= 1) MRGCN_GO(1, 2);
    if (lanes <= 2) MRGCN_GO(2, 2);
    if (lanes <= 4) MRGCN_GO(4, 2);
    if (lanes <= 8) MRGCN_GO(8, 2);
    if (lanes <= 16) MRGCN_GO(16, 2);
    if (lanes <= 32) MRGCN_GO(32, 2);
    if (lanes <= 64) MRGCN_GO(64, 2);
  } else {
    if (lanes <= 1) MRGCN_GO(1, 1);
    if (lanes <= 2) MRGCN_GO(2, 1);
    if (lanes <= 4) MRGCN_GO(4, 1);
    if (lanes <= 8) MRGCN_GO(8, 1);
    if (lanes <= 16) MRGCN_GO(16, 1);
    if (lanes <= 32) MRGCN_GO(32, 1);
    if (lanes <= 64) MRGCN_GO(64, 1);
  }
#undef MRGCN_GO
  set_error("internal: no SpMM instantiation for this feature tile");
  return MRGCN_ERR_UNSUPPORTED;
}

}  // namespace
}  // namespace mrgcn

extern "C" int mrgcn_spmm_f32(const mrgcn_plan_t *plan, int32_t view, const float *D, int64_t ldD,
                              int32_t F, float *Y, int64_t ldY, const float *bias, int32_t relu,
                              const int32_t *out_index, void *stream) {
  using namespace mrgcn;
  MRGCN_REQUIRE(plan, "plan is NULL");
  MRGCN_REQUIRE(view >= MRGCN_VIEW_LITERAL && view <= MRGCN_VIEW_TRANSPOSED, "view");
  MRGCN_REQUIRE(F > 0 && ldD >= F && ldY >= F, "F / leading dimensions");
  MRGCN_REQUIRE(D && Y, "NULL operand");
  SparseView v = plan->view(view);
  hipStream_t s = (hipStream_t)stream;
  // feature tiles: one pass covers up to 64 lanes x VEC floats; the split-row workspace
  // holds kWsFeatures floats per chunk
  int tile = 64;  // scalar-load worst case
  if (ldD % 4 == 0 && ((uintptr_t)D) % 16 == 0) tile = 256;
  else if (ldD % 2 == 0 && ((uintptr_t)D) % 8 == 0) tile = 128;
  for (int f = 0; f < F; f += tile) {
    int w = (F - f < tile) ? (F - f) : tile;
    int rc = dispatch(v, D + f, ldD, ldD - f, w, Y + f, ldY, bias ? bias + f : nullptr, relu,
                      out_index, plan->partials, s);
    if (rc != MRGCN_OK) return rc;
  }
  return MRGCN_OK;
}
