// Fused construction of the compact dense operand M (one row per *touched* column of A)
// and its backward — the MI355X replacement for the reference's materialisation of the
// (R*N) x out operands W_I and FW_F (mrgcn/layers/graph.py:69-72, :83-85, :93-94) and of
// their dense gradients (autograd of the same lines).
//
//   M[c, :] = sum_b comp_I[r_c, b] * V_I[b, j_c, :]      basis mix   (input term)
//           + X[j_c, :] . W_F[r_c]                        relation transform (feature term)
//
// Compact columns are numbered in (source node j, relation r) order, so everything that
// belongs to one node is contiguous: the basis tables V[b, j, :] are streamed exactly once
// (HBM-bound, coalesced over j for every b) and dV needs no atomics.  The per-relation
// dense transforms run relation-major over `rperm` with the relation's weight tile and a
// gathered X tile staged in LDS.
#include "common.hpp"

namespace mrgcn {
namespace {

constexpr int kTB = 256;

// =====================================================================================
// basis mix, forward:  thread = (node j, feature o); V[., j, o] lives in registers
// =====================================================================================
template <int BT>
__global__ __launch_bounds__(kTB) void k_mix_fwd(const int32_t *__restrict__ nptr,
                                                 const int32_t *__restrict__ urel,
                                                 const int32_t *__restrict__ mpos,
                                                 const float *__restrict__ V,
                                                 const float *__restrict__ comp, int64_t N, int R,
                                                 int B, int b0, int F, float *__restrict__ M,
                                                 int64_t ldM, int accumulate, int comp_in_lds) {
  extern __shared__ float s_comp[];  // [R][BT] slice b0..b0+BT of comp when it fits
  const int nb = min(BT, B - b0);
  if (comp_in_lds) {
    for (int t = threadIdx.x; t < R * BT; t += blockDim.x) {
      int r = t / BT, b = t - r * BT;
      s_comp[t] = (b < nb) ? comp[(int64_t)r * B + b0 + b] : 0.f;
    }
    __syncthreads();
  }
  const int64_t total = N * F;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / F;
    const int o = (int)(t - j * F);
    const int32_t c0 = nptr[j], c1 = nptr[j + 1];
    if (c0 == c1) continue;
    float v[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b)
      v[b] = (b < nb) ? V[((int64_t)(b0 + b) * N + j) * F + o] : 0.f;
    for (int32_t c = c0; c < c1; ++c) {
      const int r = urel[c];
      float s = 0.f;
      if (comp_in_lds) {
        const float *cr = s_comp + r * BT;
#pragma unroll
        for (int b = 0; b < BT; ++b) s = fmaf(cr[b], v[b], s);
      } else {
        const float *cr = comp + (int64_t)r * B + b0;
#pragma unroll
        for (int b = 0; b < BT; ++b)
          if (b < nb) s = fmaf(cr[b], v[b], s);
      }
      float *m = M + (int64_t)mpos[c] * ldM + o;
      *m = accumulate ? (*m + s) : s;
    }
  }
}

// =====================================================================================
// basis mix, backward — two passes, both free of per-node dependent load chains:
//
//   k_mix_bwd_dv    thread = (node j, feature o), accumulators over the bases in registers:
//                     dV[b, j, o] = sum_{c in node j} comp[r_c, b] * dM[c, o]
//                   (mirror of the forward: for a fixed b a wave writes 256 contiguous bytes)
//   k_mix_bwd_dcomp thread = compact column c (lanes = consecutive columns: coalesced index
//                   and dM reads, no divergence however skewed the node degrees are):
//                     dcomp[r_c, b] += <dM[c, :], V[b, j_c, :]>      for every b
//                   accumulated with LDS float atomics per block, one global flush per block.
// =====================================================================================
template <int BT>
__global__ __launch_bounds__(kTB) void k_mix_bwd_dv(const int32_t *__restrict__ nptr,
                                                    const int32_t *__restrict__ urel,
                                                    const float *__restrict__ dM, int64_t ldM,
                                                    const float *__restrict__ comp, int64_t N, int R,
                                                    int B, int b0, int F, float *__restrict__ dV,
                                                    int comp_in_lds) {
  extern __shared__ float s_comp[];  // [R][BT]
  const int nb = min(BT, B - b0);
  if (comp_in_lds) {
    for (int t = threadIdx.x; t < R * BT; t += blockDim.x) {
      int r = t / BT, b = t - r * BT;
      s_comp[t] = (b < nb) ? comp[(int64_t)r * B + b0 + b] : 0.f;
    }
    __syncthreads();
  }
  const int64_t total = N * F;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / F;
    const int o = (int)(t - j * F);
    const int32_t c0 = nptr[j], c1 = nptr[j + 1];
    float acc[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[b] = 0.f;
    for (int32_t c = c0; c < c1; ++c) {
      const int r = urel[c];
      const float d = dM[(int64_t)c * ldM + o];
      if (comp_in_lds) {
        const float *cr = s_comp + r * BT;
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[b] = fmaf(cr[b], d, acc[b]);
      } else {
        const float *cr = comp + (int64_t)r * B + b0;
#pragma unroll
        for (int b = 0; b < BT; ++b)
          if (b < nb) acc[b] = fmaf(cr[b], d, acc[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < BT; ++b)
      if (b < nb) dV[((int64_t)(b0 + b) * N + j) * F + o] = acc[b];
  }
}

template <int FT>
__global__ __launch_bounds__(kTB) void k_mix_bwd_dcomp(const int32_t *__restrict__ urel,
                                                       const int32_t *__restrict__ unode,
                                                       const float *__restrict__ dM, int64_t ldM,
                                                       const float *__restrict__ V, int64_t N, int R,
                                                       int B, int F, int64_t ncols,
                                                       float *__restrict__ dcomp, int dcomp_in_lds) {
  extern __shared__ float s_dcomp[];  // [R][B]
  if (dcomp_in_lds) {
    for (int t = threadIdx.x; t < R * B; t += blockDim.x) s_dcomp[t] = 0.f;
    __syncthreads();
  }
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols;
       c += (int64_t)gridDim.x * blockDim.x) {
    const int r = urel[c];
    const int64_t j = unode[c];
    float dm[FT];
    const float *dp = dM + c * ldM;
#pragma unroll
    for (int o = 0; o < FT; ++o) dm[o] = (o < F) ? dp[o] : 0.f;
    for (int b = 0; b < B; ++b) {
      const float *vp = V + ((int64_t)b * N + j) * F;
      float dot = 0.f;
#pragma unroll
      for (int o = 0; o < FT; ++o)
        if (o < F) dot = fmaf(dm[o], vp[o], dot);
      if (dcomp_in_lds) atomicAdd(&s_dcomp[r * B + b], dot);
      else atomicAdd(&dcomp[(int64_t)r * B + b], dot);
    }
  }
  if (dcomp_in_lds) {
    __syncthreads();
    for (int t = threadIdx.x; t < R * B; t += blockDim.x) {
      const float x = s_dcomp[t];
      if (x != 0.f) atomicAdd(&dcomp[t], x);
    }
  }
}

// =====================================================================================
// no-bases input term: M[c, :] = W[ulcol[c], :]   (row gather of weight_I)
// =====================================================================================
__global__ void k_gather_rows(const int32_t *__restrict__ ulcol, const int32_t *__restrict__ mpos,
                              int64_t ncols,
                              const float *__restrict__ W, int F, float *__restrict__ M, int64_t ldM,
                              int accumulate) {
  const int64_t total = ncols * F;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = t / F;
    const int o = (int)(t - c * F);
    float x = W[(int64_t)ulcol[c] * F + o];
    float *m = M + (int64_t)mpos[c] * ldM + o;
    *m = accumulate ? (*m + x) : x;
  }
}

// =====================================================================================
// relation transform, forward: one block per relation chunk (<= kRelChunk compact columns of
// one relation r, relation-major order `rperm`).  W[r] (K x F) and a gathered X tile
// (TK columns x K) are staged in LDS; thread = (column in tile, feature o).
//   M[c, o] (+)= sum_i X[j_c, i] * W[r, i, o]
// K is processed in slabs of KS so that LDS use is bounded for any K.
// =====================================================================================
constexpr int kTK = 32;   // columns per LDS tile
constexpr int kKS = 160;  // K slab held in LDS at once

__global__ __launch_bounds__(kTB) void k_xform_fwd(const int32_t *__restrict__ relchunk_rel,
                                                   const int32_t *__restrict__ relchunk_beg,
                                                   const int32_t *__restrict__ relchunk_end,
                                                   const int32_t *__restrict__ rperm,
                                                   const int32_t *__restrict__ unode,
                                                   const int32_t *__restrict__ mpos,
                                                   const float *__restrict__ X, int64_t ldX, int K,
                                                   const float *__restrict__ W, int F,
                                                   float *__restrict__ M, int64_t ldM,
                                                   int accumulate) {
  extern __shared__ float smem[];
  float *Ws = smem;                 // [KS][F]
  float *Xs = smem + kKS * F;       // [TK][KS+1]
  __shared__ int32_t s_c[kTK], s_j[kTK];
  const int chunk = blockIdx.x;
  const int r = relchunk_rel[chunk];
  const int32_t beg = relchunk_beg[chunk], end = relchunk_end[chunk];
  const float *Wr = W + (int64_t)r * K * F;
  const int XS = kKS + 1;

  for (int32_t t0 = beg; t0 < end; t0 += kTK) {
    const int nk = min(kTK, end - t0);
    __syncthreads();
    if (threadIdx.x < nk) {
      int32_t c = rperm[t0 + threadIdx.x];
      s_c[threadIdx.x] = mpos[c];
      s_j[threadIdx.x] = unode[c];
    }
    // per-thread outputs: pairs (kk, o), strided over the block (static register indices)
    constexpr int NQ = (kTK * 64 + kTB - 1) / kTB;  // F <= 64
    float acc[NQ];
    const int npairs = nk * F;
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
    for (int k0 = 0; k0 < K; k0 += kKS) {
      const int ks = min(kKS, K - k0);
      __syncthreads();
      for (int t = threadIdx.x; t < ks * F; t += kTB) Ws[t] = Wr[(int64_t)k0 * F + t];
      for (int t = threadIdx.x; t < nk * ks; t += kTB) {
        int kk = t / ks, i = t - kk * ks;
        Xs[kk * XS + i] = X[(int64_t)s_j[kk] * ldX + k0 + i];
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int p = q * kTB + threadIdx.x;
        if (p < npairs) {
          const int kk = p / F, o = p - kk * F;
          const float *xr = Xs + kk * XS;
          float s = acc[q];
          for (int i = 0; i < ks; ++i) s = fmaf(xr[i], Ws[i * F + o], s);
          acc[q] = s;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int p = q * kTB + threadIdx.x;
      if (p < npairs) {
        const int kk = p / F, o = p - kk * F;
        float *m = M + (int64_t)s_c[kk] * ldM + o;
        *m = accumulate ? (*m + acc[q]) : acc[q];
      }
    }
  }
}

// =====================================================================================
// relation transform, backward w.r.t. W:  dW[r, i, o] = sum_{c in r} X[j_c, i] * dM[c, o]
// one block per relation chunk; X tile and dM tile in LDS; thread owns pairs (i, o);
// one atomicAdd per (block, i, o) into the zero-initialised dW.
// =====================================================================================
constexpr int kPP = 8;  // (i, o) pairs per thread per pass

__global__ __launch_bounds__(kTB) void k_xform_bwd_dw(const int32_t *__restrict__ relchunk_rel,
                                                      const int32_t *__restrict__ relchunk_beg,
                                                      const int32_t *__restrict__ relchunk_end,
                                                      const int32_t *__restrict__ rperm,
                                                      const int32_t *__restrict__ unode,
                                                      const float *__restrict__ X, int64_t ldX, int K,
                                                      const float *__restrict__ dM, int64_t ldM, int F,
                                                      float *__restrict__ dW) {
  extern __shared__ float smem[];
  const int chunk = blockIdx.x;
  const int r = relchunk_rel[chunk];
  const int32_t beg = relchunk_beg[chunk], end = relchunk_end[chunk];
  __shared__ int32_t s_c[kTK], s_j[kTK];
  const int XS = kKS + 1;
  float *Xs = smem;               // [TK][KS+1]
  float *Ds = smem + kTK * XS;    // [TK][F]
  float *dWr = dW + (int64_t)r * K * F;

  for (int k0 = 0; k0 < K; k0 += kKS) {
    const int ks = min(kKS, K - k0);
    const int npairs = ks * F;
    for (int pbase = 0; pbase < npairs; pbase += kTB * kPP) {
      float acc[kPP];
#pragma unroll
      for (int q = 0; q < kPP; ++q) acc[q] = 0.f;
      for (int32_t t0 = beg; t0 < end; t0 += kTK) {
        const int nk = min(kTK, end - t0);
        __syncthreads();
        if (threadIdx.x < nk) {
          int32_t c = rperm[t0 + threadIdx.x];
          s_c[threadIdx.x] = c;
          s_j[threadIdx.x] = unode[c];
        }
        __syncthreads();
        for (int t = threadIdx.x; t < nk * ks; t += kTB) {
          int kk = t / ks, i = t - kk * ks;
          Xs[kk * XS + i] = X[(int64_t)s_j[kk] * ldX + k0 + i];
        }
        for (int t = threadIdx.x; t < nk * F; t += kTB) {
          int kk = t / F, o = t - kk * F;
          Ds[t] = dM[(int64_t)s_c[kk] * ldM + o];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kPP; ++q) {
          const int p = pbase + q * kTB + threadIdx.x;
          if (p < npairs) {
            const int i = p / F, o = p - i * F;
            float s = acc[q];
            for (int kk = 0; kk < nk; ++kk) s = fmaf(Xs[kk * XS + i], Ds[kk * F + o], s);
            acc[q] = s;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < kPP; ++q) {
        const int p = pbase + q * kTB + threadIdx.x;
        if (p < npairs) atomicAdd(&dWr[(int64_t)k0 * F + p], acc[q]);
      }
    }
  }
}

// =====================================================================================
// relation transform, backward w.r.t. X (node-major; no atomics):
//   dX[j, i] = sum_{c in node j} sum_o dM[c, o] * W[r_c, i, o]
// thread = (node j, input feature i)
// =====================================================================================
__global__ __launch_bounds__(kTB) void k_xform_bwd_dx(const int32_t *__restrict__ nptr,
                                                      const int32_t *__restrict__ urel,
                                                      const float *__restrict__ dM, int64_t ldM,
                                                      const float *__restrict__ W, int64_t N, int K,
                                                      int F, float *__restrict__ dX, int64_t lddX) {
  const int64_t total = N * K;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / K;
    const int i = (int)(t - j * K);
    const int32_t c0 = nptr[j], c1 = nptr[j + 1];
    float s = 0.f;
    for (int32_t c = c0; c < c1; ++c) {
      const float *w = W + ((int64_t)urel[c] * K + i) * F;
      const float *dm = dM + (int64_t)c * ldM;
      for (int o = 0; o < F; ++o) s = fmaf(dm[o], w[o], s);
    }
    dX[j * lddX + i] = s;
  }
}

int grid_for(int64_t work_items, int max_blocks = 256 * 8) {
  int64_t b = (work_items + kTB - 1) / kTB;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

constexpr size_t kLdsBudget = 64 * 1024;  // dynamic LDS these kernels may take

}  // namespace
}  // namespace mrgcn

using namespace mrgcn;

extern "C" {

int mrgcn_basis_mix_fwd_f32(const mrgcn_plan_t *p, const float *V, const float *comp, int32_t B,
                            int32_t F, float *M, int64_t ldM, int32_t accumulate, void *stream) {
  MRGCN_REQUIRE(p && V && comp && M, "NULL");
  MRGCN_REQUIRE(B > 0 && F > 0 && ldM >= F, "B / F / ldM");
  if (p->ncols == 0) return MRGCN_OK;
  hipStream_t s = (hipStream_t)stream;
  const int R = (int)p->num_relations;
  const int64_t N = p->num_nodes;
  int acc = accumulate;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int nb = (B - b0 < 64) ? (B - b0) : 64;
    int BT = nb <= 2 ? 2 : nb <= 4 ? 4 : nb <= 8 ? 8 : nb <= 16 ? 16 : nb <= 32 ? 32 : nb <= 40 ? 40 : 64;
    size_t lds = (size_t)R * BT * sizeof(float);
    int in_lds = lds <= kLdsBudget;
    if (!in_lds) lds = 0;
    int grid = grid_for(N * F);
#define MIX_GO(T)                                                                               \
  k_mix_fwd<T><<<dim3(grid), dim3(kTB), lds, s>>>(p->nptr, p->urel, p->mpos, V, comp, N, R, B, b0, F, M, \
                                                  ldM, acc, in_lds)
    switch (BT) {
      case 2: MIX_GO(2); break;
      case 4: MIX_GO(4); break;
      case 8: MIX_GO(8); break;
      case 16: MIX_GO(16); break;
      case 32: MIX_GO(32); break;
      case 40: MIX_GO(40); break;
      default: MIX_GO(64); break;
    }
#undef MIX_GO
    MRGCN_HIP_TRY(hipGetLastError());
    acc = 1;
  }
  return MRGCN_OK;
}

int mrgcn_basis_mix_bwd_f32(const mrgcn_plan_t *p, const float *dM, int64_t ldM, const float *V,
                            const float *comp, int32_t B, int32_t F, float *dV, float *dcomp,
                            void *stream) {
  MRGCN_REQUIRE(p && dM && V && comp && dV && dcomp, "NULL");
  MRGCN_REQUIRE(B > 0 && F > 0 && ldM >= F, "B / F / ldM");
  MRGCN_REQUIRE(F <= 64, "basis_mix_bwd supports F <= 64 (tile the feature dimension)");
  hipStream_t s = (hipStream_t)stream;
  const int R = (int)p->num_relations;
  const int64_t N = p->num_nodes;
  MRGCN_HIP_TRY(hipMemsetAsync(dcomp, 0, (size_t)R * B * sizeof(float), s));
  // nodes without any column never get written by the kernel: zero dV first only then
  // (full-batch graphs carry the identity block, so every node owns >= 1 column)
  // pass 1: dV
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int nb = (B - b0 < 64) ? (B - b0) : 64;
    int BT = nb <= 2 ? 2 : nb <= 4 ? 4 : nb <= 8 ? 8 : nb <= 16 ? 16 : nb <= 32 ? 32 : nb <= 40 ? 40 : 64;
    size_t lds = (size_t)R * BT * sizeof(float);
    int in_lds = lds <= kLdsBudget;
    if (!in_lds) lds = 0;
    int grid = grid_for(N * F);
#define MIXDV_GO(T)                                                                               \
  k_mix_bwd_dv<T><<<dim3(grid), dim3(kTB), lds, s>>>(p->nptr, p->urel, dM, ldM, comp, N, R, B, b0, \
                                                     F, dV, in_lds)
    switch (BT) {
      case 2: MIXDV_GO(2); break;
      case 4: MIXDV_GO(4); break;
      case 8: MIXDV_GO(8); break;
      case 16: MIXDV_GO(16); break;
      case 32: MIXDV_GO(32); break;
      case 40: MIXDV_GO(40); break;
      default: MIXDV_GO(64); break;
    }
#undef MIXDV_GO
    MRGCN_HIP_TRY(hipGetLastError());
  }
  // pass 2: dcomp
  if (p->ncols > 0) {
    size_t lds = (size_t)R * B * sizeof(float);
    int in_lds = lds <= kLdsBudget;
    if (!in_lds) lds = 0;
    int per_cu = in_lds && lds > 0 ? (int)((160 * 1024) / (lds + 512)) : 8;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int64_t want = (p->ncols + kTB - 1) / kTB;
    int grid = 256 * per_cu;
    if (grid > want) grid = (int)want;
#define MIXDC_GO(T)                                                                                 \
  k_mix_bwd_dcomp<T><<<dim3(grid), dim3(kTB), lds, s>>>(p->urel, p->unode, dM, ldM, V, N, R, B, F, \
                                                        p->ncols, dcomp, in_lds)
    if (F <= 4) MIXDC_GO(4);
    else if (F <= 8) MIXDC_GO(8);
    else if (F <= 12) MIXDC_GO(12);
    else if (F <= 16) MIXDC_GO(16);
    else if (F <= 32) MIXDC_GO(32);
    else MIXDC_GO(64);
#undef MIXDC_GO
  }
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_gather_rows_f32(const mrgcn_plan_t *p, const float *W, int32_t F, float *M, int64_t ldM,
                          int32_t accumulate, void *stream) {
  MRGCN_REQUIRE(p && W && M, "NULL");
  MRGCN_REQUIRE(F > 0 && ldM >= F, "F / ldM");
  if (p->ncols == 0) return MRGCN_OK;
  k_gather_rows<<<dim3(grid_for(p->ncols * F)), dim3(kTB), 0, (hipStream_t)stream>>>(
      p->ulcol, p->mpos, p->ncols, W, F, M, ldM, accumulate);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_rel_transform_fwd_f32(const mrgcn_plan_t *p, const float *X, int64_t ldX, int32_t K,
                                const float *W, int32_t F, float *M, int64_t ldM, int32_t accumulate,
                                void *stream) {
  MRGCN_REQUIRE(p && X && W && M, "NULL");
  MRGCN_REQUIRE(K > 0 && F > 0 && ldX >= K && ldM >= F, "K / F / leading dimensions");
  MRGCN_REQUIRE(F <= 64, "rel_transform supports F <= 64 (tile the feature dimension)");
  if (p->n_relchunks == 0) return MRGCN_OK;
  size_t lds = ((size_t)kKS * F + (size_t)kTK * (kKS + 1)) * sizeof(float);
  k_xform_fwd<<<dim3(p->n_relchunks), dim3(kTB), lds, (hipStream_t)stream>>>(
      p->relchunk_rel, p->relchunk_beg, p->relchunk_end, p->rperm, p->unode, p->mpos, X, ldX, K, W, F, M,
      ldM, accumulate);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_rel_transform_bwd_f32(const mrgcn_plan_t *p, const float *dM, int64_t ldM, const float *X,
                                int64_t ldX, int32_t K, const float *W, int32_t F, float *dX,
                                int64_t lddX, float *dW, void *stream) {
  MRGCN_REQUIRE(p && dM && X && W, "NULL");
  MRGCN_REQUIRE(K > 0 && F > 0 && ldX >= K && ldM >= F, "K / F / leading dimensions");
  MRGCN_REQUIRE(F <= 64, "rel_transform supports F <= 64 (tile the feature dimension)");
  hipStream_t s = (hipStream_t)stream;
  if (dW) {
    MRGCN_HIP_TRY(hipMemsetAsync(dW, 0, (size_t)p->num_relations * K * F * sizeof(float), s));
    if (p->n_relchunks > 0) {
      size_t lds = ((size_t)kTK * (kKS + 1) + (size_t)kTK * F) * sizeof(float);
      k_xform_bwd_dw<<<dim3(p->n_relchunks), dim3(kTB), lds, s>>>(
          p->relchunk_rel, p->relchunk_beg, p->relchunk_end, p->rperm, p->unode, X, ldX, K, dM, ldM, F, dW);
      MRGCN_HIP_TRY(hipGetLastError());
    }
  }
  if (dX) {
    MRGCN_REQUIRE(lddX >= K, "lddX");
    k_xform_bwd_dx<<<dim3(grid_for(p->num_nodes * K)), dim3(kTB), 0, s>>>(
        p->nptr, p->urel, dM, ldM, W, p->num_nodes, K, F, dX, lddX);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  return MRGCN_OK;
}

}  // extern "C"
