// The backward of a semi-supervised epoch on a GRADIENT SUPPORT (common.hpp: mrgcn_support; built in plan.hip).
//
// The reference's autograd runs the backward of graph.py:62-102 densely: dD = A^T dY over all (R*N) rows although the
// loss (node_classification.py:439-444) only touches the labelled rows.  Rounds 1-3 skipped the zeros per epoch (flag
// the live rows of dY, mark the columns they touch, compact lists in LDS, skip dead nodes).  With the label set fixed
// all of that is the same every epoch, so it is decided ONCE: the support numbers the live columns 0..L-1 in
// (node, relation) order and keeps, per live column, only the entries that sit in live rows.  Per epoch:
//
//   dM[k]      = sum over kept entries (k_spmm on the filtered view: a plain CSR product over L short rows)
//   D[k][b]    = <dM[k], V[j_k][b]>,  ||dV||^2            one pass over the live nodes' V blocks (k_mix_bwd_sup):
//                                                         lane = basis, a node's live columns are consecutive rows
//   dcomp[r]   = sum of the D rows of relation r          relation-major walk of D (k_dcomp_chunks / _final)
//   dW, dX     = the matrix-core transforms on the support's relation-major lists (xform_mfma.hip, unchanged kernels)
//   Adam       = k_adam_rows_fused with (node -> live range, relation per live column) in place of the plan's arrays
//
// No atomics, no flags, no zero fills: every output is written whole, in a fixed order.
#include <cstdlib>

#include "common.hpp"

namespace mrgcn {
namespace {

constexpr int kSupTB = 512;
constexpr int kSqParts = 2048;  // per-block partial sums of ||dV||^2 (doubles), added in block order by k_dcomp_final

// F floats of one basis row (rows 4 F bytes apart: 8-byte aligned when F is even)
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
template <int FT>
__device__ __forceinline__ void load_basis_row(const float *__restrict__ p, int F, float (&v)[FT]) {
  if ((F & 1) == 0) {
#pragma unroll
    for (int o = 0; o < FT; o += 4) {
      if (o + 4 <= F && o + 4 <= FT) {
        const f32x4_a8 t = *reinterpret_cast<const f32x4_a8 *>(p + o);
        v[o] = t.x; v[o + 1] = t.y; v[o + 2] = t.z; v[o + 3] = t.w;
      } else {
#pragma unroll
        for (int u = o; u < o + 4 && u < FT; u += 2) {
          if (u < F) {
            const float2 t = *reinterpret_cast<const float2 *>(p + u);
            v[u] = t.x;
            if (u + 1 < FT) v[u + 1] = t.y;
          } else {
            v[u] = 0.f;
            if (u + 1 < FT) v[u + 1] = 0.f;
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int o = 0; o < FT; ++o) v[o] = (o < F) ? p[o] : 0.f;
  }
}

// Norm-only backward of the basis mix over the live nodes (see the file comment).  A wave fetches the (node, live
// range) pairs of 64 live nodes with one coalesced load each and walks them; per node: lane b reads its row of the
// node's V block, the node's dM rows arrive four at a time (16-lane group q reads row kb + q) and are handed round with
// v_readlane; lane b stores D[k][b] (a live column's B products: one 4 B-byte row per column, coalesced) and keeps
// the column's share of its dV row for the squared norm.
template <int FT>
__global__ __launch_bounds__(kSupTB) void k_mix_bwd_sup(const int32_t *__restrict__ lnode,
                                                        const int32_t *__restrict__ lnptr,
                                                        const int32_t *__restrict__ lrel,
                                                        const float *__restrict__ dM, int64_t ldM,
                                                        const float *__restrict__ V, const float *__restrict__ comp,
                                                        int64_t NL, int B, int F, float *__restrict__ D,
                                                        double *__restrict__ sq_part) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = blockDim.x >> 6;
  const bool on = lane < B;
  const int b = on ? lane : 0;
  const int kq = lane >> 4, oq = lane & 15;
  const int64_t nwaves = (int64_t)gridDim.x * nw;
  float sq = 0.f;
  for (int64_t base = ((int64_t)blockIdx.x * nw + wv) * 64; base < NL; base += nwaves * 64) {
    const int64_t me = (base + lane < NL) ? base + lane : NL - 1;
    const int32_t jn = lnode[me], k0v = lnptr[me], k1v = lnptr[me + 1];
    const int cnt = (int)((NL - base < 64) ? NL - base : 64);
    for (int i = 0; i < cnt; ++i) {
      const int32_t j = __builtin_amdgcn_readlane(jn, i);
      const int32_t klo = __builtin_amdgcn_readlane(k0v, i), khi = __builtin_amdgcn_readlane(k1v, i);
      float v[FT];
      load_basis_row<FT>(V + ((int64_t)j * B + b) * F, F, v);
      float acc[FT];
#pragma unroll
      for (int o = 0; o < FT; ++o) acc[o] = 0.f;
      for (int32_t kb = klo; kb < khi; kb += 4) {
        const int32_t kk = kb + kq;
        const bool cin = kk < khi;
        const float dmine = (cin && oq < F) ? dM[(int64_t)kk * ldM + oq] : 0.f;
        const int32_t rmine = cin ? lrel[kk] : 0;
        int r[4];
        float w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) r[t] = __builtin_amdgcn_readlane(rmine, 16 * t);
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = comp[(int64_t)r[t] * B + b];  // R * B floats: cache resident
        const int nc = (khi - kb < 4) ? khi - kb : 4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if (t < nc) {  // wave uniform
            float d[FT];
#pragma unroll
            for (int o = 0; o < FT; ++o)
              d[o] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dmine), 16 * t + o));
            float dot = 0.f;
#pragma unroll
            for (int o = 0; o < FT; ++o) dot = fmaf(d[o], v[o], dot);
            if (on) D[(int64_t)(kb + t) * B + b] = dot;
#pragma unroll
            for (int o = 0; o < FT; ++o) acc[o] = fmaf(w[t], d[o], acc[o]);
          }
        }
      }
      if (on) {
#pragma unroll
        for (int o = 0; o < FT; ++o)
          if (o < F) sq = fmaf(acc[o], acc[o], sq);
      }
    }
  }
  __shared__ float s_sq[kSupTB / 64];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
  if (lane == 0) s_sq[wv] = sq;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += (double)s_sq[i];
    sq_part[blockIdx.x] = t;
  }
}

// slab[chunk][b] = sum of D[k][b] over the live columns k of one relation-major chunk (<= kRelChunk columns of one
// relation).  Lane = basis; the four waves take a quarter of the chunk each, 64 row numbers per load, eight row
// gathers in flight.
__global__ __launch_bounds__(256) void k_dcomp_chunks(const int32_t *__restrict__ chunk_beg,
                                                      const int32_t *__restrict__ chunk_end,
                                                      const int32_t *__restrict__ lperm, const float *__restrict__ D,
                                                      int B, float *__restrict__ slab) {
  __shared__ float s_part[4][64];
  const int chunk = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int32_t beg = chunk_beg[chunk], end = chunk_end[chunk];
  const int32_t per = (end - beg + 3) >> 2;
  const int32_t a0 = beg + wv * per, a1 = (a0 + per < end) ? a0 + per : end;
  const int bb = lane < B ? lane : 0;
  float acc0 = 0.f, acc1 = 0.f;
  for (int32_t e0 = a0; e0 < a1; e0 += 64) {
    const int32_t mine = (e0 + lane < a1) ? lperm[e0 + lane] : 0;
    const int cnt = (a1 - e0 < 64) ? a1 - e0 : 64;
    for (int t = 0; t < cnt; t += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int tt = (t + u < cnt) ? t + u : cnt - 1;
        const int32_t k = __builtin_amdgcn_readlane(mine, tt);
        x[u] = D[(int64_t)k * B + bb];
      }
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        if (t + u < cnt) acc0 += x[u];
        if (t + u + 1 < cnt) acc1 += x[u + 1];
      }
    }
  }
  s_part[wv][lane] = acc0 + acc1;
  __syncthreads();
  if (threadIdx.x < (unsigned)B)
    slab[(int64_t)chunk * B + threadIdx.x] =
        (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}

// dcomp[r][b] = sum of relation r's chunk slabs (in chunk order; zeros for a relation without live columns);
// the block after the last relation adds the per-block parts of ||dV||^2
__global__ __launch_bounds__(64) void k_dcomp_final(const int32_t *__restrict__ chunk_ptr,
                                                    const int32_t *__restrict__ chunk_ids,
                                                    const float *__restrict__ slab, int R, int B,
                                                    float *__restrict__ dcomp, const double *__restrict__ sq_part,
                                                    int n_parts, double *__restrict__ sumsq) {
  const int r = blockIdx.x;
  if (r == R) {
    if (!sumsq) return;
    double t = 0.0;
    for (int i = threadIdx.x; i < n_parts; i += 64) t += sq_part[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (threadIdx.x == 0) *sumsq = t;
    return;
  }
  if ((int)threadIdx.x >= B) return;
  float s0 = 0.f, s1 = 0.f;
  int c = chunk_ptr[r];
  const int c1 = chunk_ptr[r + 1];
  for (; c + 2 <= c1; c += 2) {
    s0 += slab[(int64_t)chunk_ids[c] * B + threadIdx.x];
    s1 += slab[(int64_t)chunk_ids[c + 1] * B + threadIdx.x];
  }
  if (c < c1) s0 += slab[(int64_t)chunk_ids[c] * B + threadIdx.x];
  dcomp[(int64_t)r * B + threadIdx.x] = s0 + s1;
}

__global__ void k_xent_scatter_rows(const float *__restrict__ drows, const int64_t *__restrict__ idx, int64_t n, int C,
                                    const float *__restrict__ g, float *__restrict__ dlogits, int64_t ldd) {
  const float gg = g ? *g : 1.f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * C; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / C;
    const int c = (int)(t - i * C);
    dlogits[idx[i] * ldd + c] = gg * drows[t];
  }
}

}  // namespace
}  // namespace mrgcn

extern "C" {

using namespace mrgcn;

int mrgcn_support_spmm_t_f32(const mrgcn_support_t *q, const float *dY, int64_t ldY, int32_t F, float *dM, int64_t ldM,
                             void *stream) {
  MRGCN_REQUIRE(q && dY && dM, "NULL");
  MRGCN_REQUIRE(F > 0 && ldY >= F && ldM >= F, "F / leading dimensions");
  return spmm_on_view(q->tview(), dY, ldY, F, dM, ldM, q->partials, (hipStream_t)stream);
}

int64_t mrgcn_support_mix_bwd_workspace(const mrgcn_support_t *q, int32_t B) {
  if (!q || B <= 0) return 0;
  // D [L][B] | slab [chunks][B] | parts of ||dV||^2 (doubles, 8-byte aligned)
  return ((q->L + q->wide.n_chunks) * (int64_t)B + 1) / 2 * 2 + 2 * (int64_t)kSqParts;
}

int mrgcn_support_mix_bwd_f32(const mrgcn_support_t *q, const float *dM, int64_t ldM, const float *V,
                              const float *comp, int32_t B, int32_t F, float *dV, int32_t dense, float *dcomp,
                              double *dV_sumsq, float *workspace, int64_t workspace_floats, void *stream) {
  MRGCN_REQUIRE(q && dM && V && comp && dcomp, "NULL");
  MRGCN_REQUIRE(B > 0 && F > 0 && ldM >= F, "B / F / ldM");
  hipStream_t s = (hipStream_t)stream;
  const mrgcn_plan *p = q->plan;
  const int R = (int)p->num_relations;
  const int64_t N = p->num_nodes;
  if (dV) {  // the gradient itself: the wave-over-nodes kernel of rgcn_fused.hip on the support's arrays
    MRGCN_HIP_TRY(hipMemsetAsync(dcomp, 0, (size_t)R * B * sizeof(float), s));
    int rc = mix_bwd_nm_arrays(q->nlptr, q->lrel, N, R, -1, dM, ldM, V, comp, B, F, dV, dcomp, dV_sumsq, s, nullptr,
                               dense ? nullptr : q->node_scratch);
    if (rc < 0) {
      set_error("mrgcn_support_mix_bwd_f32: shape outside the node-major kernel's limits (F <= 16, B <= 64)");
      return MRGCN_ERR_UNSUPPORTED;
    }
    return rc;
  }
  MRGCN_REQUIRE(F <= 16 && B <= 64, "the norm-only pass needs F <= 16, B <= 64");
  MRGCN_REQUIRE((F & 1) || (((uintptr_t)V) & 7) == 0, "V must be 8-byte aligned");
  MRGCN_REQUIRE(workspace && workspace_floats >= mrgcn_support_mix_bwd_workspace(q, B) && (((uintptr_t)workspace) & 7) == 0,
                "workspace (mrgcn_support_mix_bwd_workspace floats, 8-byte aligned)");
  float *D = workspace;
  float *slab = D + q->L * (int64_t)B;
  double *sq_part = reinterpret_cast<double *>(workspace + ((q->L + q->wide.n_chunks) * (int64_t)B + 1) / 2 * 2);
  const int nw = kSupTB / 64;
  int64_t grid = (q->NL + 64 * nw - 1) / (64 * nw);
  static const int per_cu = getenv("MRGCN_SUP_MIX_PER_CU") ? atoi(getenv("MRGCN_SUP_MIX_PER_CU")) : 4;
  if (grid > 256 * per_cu) grid = 256 * per_cu;
  if (grid > kSqParts) grid = kSqParts;
  if (grid < 1) grid = 1;
  const int FT = (F == 10 || F == 11) ? F : (F + 3) / 4 * 4;
#define SUP_GO(T)                                                                                                  \
  k_mix_bwd_sup<T><<<dim3((unsigned)grid), dim3(kSupTB), 0, s>>>(q->lnode, q->lnptr, q->lrel, dM, ldM, V, comp,   \
                                                                  q->NL, B, F, D, sq_part)
  if (q->NL > 0) {
    switch (FT) {
      case 4: SUP_GO(4); break;
      case 8: SUP_GO(8); break;
      case 10: SUP_GO(10); break;
      case 11: SUP_GO(11); break;
      case 12: SUP_GO(12); break;
      default: SUP_GO(16); break;
    }
  }
#undef SUP_GO
  MRGCN_HIP_TRY(hipGetLastError());
  const mrgcn_support::Order &o = q->wide;
  if (o.n_chunks > 0) {
    k_dcomp_chunks<<<dim3((unsigned)o.n_chunks), dim3(256), 0, s>>>(o.chunk_beg, o.chunk_end, o.lperm, D, B, slab);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  k_dcomp_final<<<dim3((unsigned)(R + 1)), dim3(64), 0, s>>>(o.chunk_ptr, o.chunk_ids, slab, R, B, dcomp, sq_part,
                                                            q->NL > 0 ? (int)grid : 0, dV_sumsq);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_support_adam_rows_fused_f32(const mrgcn_support_t *q, const float *dM, int64_t ldM, const float *comp,
                                      int32_t B, int32_t F, float *param, float *exp_avg, float *exp_avg_sq,
                                      uint8_t *row_ever, float lr, float beta1, float beta2, float eps, int64_t step,
                                      const float *bc_dev, const float *grad_scale, void *stream) {
  MRGCN_REQUIRE(q && dM && comp && param && exp_avg && exp_avg_sq && row_ever, "NULL");
  MRGCN_REQUIRE(mrgcn_adam_rows_fused_supported(q->plan, B, F), "shape outside mrgcn_adam_rows_fused_supported");
  MRGCN_REQUIRE(ldM >= F, "ldM");
  MRGCN_REQUIRE(((((uintptr_t)param) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15) == 0,
                "param / moments must be 16-byte aligned");
  MRGCN_REQUIRE(bc_dev || step >= 1, "step");
  const mrgcn_plan *p = q->plan;
  return adam_rows_fused_arrays(q->nlptr, q->lrel, nullptr, p->num_nodes, (int)p->num_relations, dM, ldM, comp, B, F,
                                param, exp_avg, exp_avg_sq, q->node_flags, row_ever, lr, beta1, beta2, eps, step,
                                bc_dev, grad_scale, (hipStream_t)stream);
}

int64_t mrgcn_support_rel_transform_bwd_workspace(const mrgcn_support_t *q, int32_t K, int32_t F, int32_t need_dX,
                                                  int32_t need_dW) {
  if (!q) return -1;
  if (!xform_use_mfma()) return -1;
  if (need_dW && !xform_mfma_dw_supported(K, F)) return -1;
  if (need_dX && !(xform_mfma_fwd_supported(F, K))) return -1;
  const int64_t a = need_dX ? q->L * (((int64_t)K + 3) / 4 * 4) : 0;
  const int64_t b = need_dW ? (int64_t)q->order_for(K).n_relchunks * K * F : 0;
  const int64_t m = a > b ? a : b;
  return m > 0 ? m : 1;
}

int mrgcn_support_rel_transform_bwd_f32(const mrgcn_support_t *q, const float *dM, int64_t ldM, const float *X,
                                        int64_t ldX, int32_t K, const float *W, int32_t F, float *dX, int64_t lddX,
                                        float *dW, float *workspace, int64_t workspace_floats,
                                        int32_t relu_mask_from_x, void *stream) {
  MRGCN_REQUIRE(q && dM && X && W, "NULL");
  MRGCN_REQUIRE(K > 0 && F > 0 && ldX >= K && ldM >= F, "K / F / leading dimensions");
  const int64_t need = mrgcn_support_rel_transform_bwd_workspace(q, K, F, dX != nullptr, dW != nullptr);
  if (need < 0) {
    set_error("mrgcn_support_rel_transform_bwd_f32: shape outside the matrix-core transforms' limits");
    return MRGCN_ERR_UNSUPPORTED;
  }
  MRGCN_REQUIRE(workspace && workspace_floats >= need, "workspace (mrgcn_support_rel_transform_bwd_workspace floats)");
  MRGCN_REQUIRE(!relu_mask_from_x || (dX && K <= 16), "the masked dX needs K <= 16");
  hipStream_t s = (hipStream_t)stream;
  const mrgcn_plan *p = q->plan;
  if (dW) {
    const RelOrder o = q->order_for(K);
    int rc = xform_mfma_dw(p, o, o.rnode, X, ldX, K, dM, ldM, F, dW, workspace, workspace_floats, s, nullptr);
    if (rc != MRGCN_OK) return rc;
  }
  if (dX) {
    MRGCN_REQUIRE(lddX >= K, "lddX");
    const int64_t ldZ = ((int64_t)K + 3) / 4 * 4;
    // Z[k, 0:K] = dM[k, 0:F] . W[r_k]^T on the matrix cores, then dX[j] = sum of node j's Z rows (every row written)
    int rc = xform_mfma_fwd(p, q->order_for(F), nullptr, nullptr, dM, ldM, F, W, true, K, workspace, ldZ, s, false,
                            nullptr);
    if (rc != MRGCN_OK) return rc;
    rc = segment_sum_arrays(q->nlptr, p->num_nodes, q->L, workspace, ldZ, K, dX, lddX, s,
                            relu_mask_from_x ? X : nullptr, ldX);
    if (rc != MRGCN_OK) return rc;
  }
  return MRGCN_OK;
}

int mrgcn_softmax_xent_bwd_rows_f32(const float *drows, const int64_t *idx, int64_t n, int32_t C, const float *g,
                                    float *dlogits, int64_t ldd, void *stream) {
  MRGCN_REQUIRE(drows && idx && dlogits, "NULL");
  MRGCN_REQUIRE(C > 0 && ldd >= C && n > 0, "C / ldd / n");
  int grid = (int)((n * C + 255) / 256);
  if (grid > 1024) grid = 1024;
  k_xent_scatter_rows<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(drows, idx, n, C, g, dlogits, ldd);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // extern "C"
