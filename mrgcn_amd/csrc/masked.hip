// A mini-batch layer as a MASKED PASS over the full graph's plan (mrgcn/data/batch.py:185-263: `A[sample_idx]`,
// getNeighboursSparse, getAdjacencyNodeColumnIdx + sliceSparseCOO; mrgcn/models/rgcn.py:91-128; mrgcn/layers/graph.py:
// 62-102 with A_idx) — no per-batch plan.
//
// The rows a layer computes (its sample) are a row-flag array on the plan; a gradient support created with
// MRGCN_SUPPORT_FORWARD (plan.hip: build_support) numbers what those rows touch:
//     flagged rows   0..NR-1   rising row id            = the sample (sorted, distinct)
//     live columns   0..L-1    (node, relation) order   = the columns of A[sample] that hold an entry
//     live nodes     0..NL-1   rising node id           = getNeighboursSparse(A, sample)
// and keeps A[sample] as a CSR over (flagged row, live column) plus its transpose.  Everything a layer touches is
// then compact: activations [NR] x F, the neighbours' features / embeddings [NL] x K, per-column operands [L] x F;
// nothing of size N is read or written per step except weight_I's blocks of the live nodes.
//
// graph.py's two terms on a slice:
//   input term    A[sample] (stored values, global columns) . (comp (x) weight_I)      -> fview(values) . M_I
//   feature term  sliceSparseCOO(A[sample], A_idx) (ALL-ONES values, quirk A-3) . (X W_F)  -> fview(ones) . T
// so a layer with both terms takes two products.  Backward: the transposed view over the compact output gradient
// (lrow_rank), then the support's own per-column backward kernels (support.hip) — the weight_I rows through
// mrgcn_support_mix_bwd_f32 / the fused row update, the transform through the matrix-core kernels on the order
// by live-node rank.
#include "common.hpp"

namespace mrgcn {
namespace {
// weight_I without bases is the reference's literal (R*N) x F table (graph.py:72-74): live column k reads row
// lrel[k] * N + node(k).  SCATTER: the transposed move (rows of the zeroed gradient table; a live column is one
// (relation, node) pair, so no two of them meet).
template <bool SCATTER>
__global__ void k_sup_literal_rows(const int32_t *__restrict__ lcol, const int32_t *__restrict__ lrel,
                                   const int32_t *__restrict__ unode, int64_t L, int64_t N, int F, float *table,
                                   float *M, int64_t ldM) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t k = t / F;
  const int f = (int)(t - k * F);
  if (k >= L) return;
  const int64_t row = (int64_t)lrel[k] * N + unode[lcol[k]];
  if (SCATTER) table[row * F + f] = M[k * ldM + f];
  else M[k * ldM + f] = table[row * F + f];
}
}  // namespace
}  // namespace mrgcn

extern "C" {

using namespace mrgcn;

#define REQUIRE_FORWARD(q) \
  MRGCN_REQUIRE((q) && (q)->has_forward, "the support was not created with MRGCN_SUPPORT_FORWARD")

int mrgcn_support_spmm_fwd_f32(const mrgcn_support_t *q, int32_t use_values, const float *D, int64_t ldD, int32_t F,
                               float *Y, int64_t ldY, const float *bias, int32_t relu, void *stream) {
  REQUIRE_FORWARD(q);
  MRGCN_REQUIRE(D && Y, "NULL");
  MRGCN_REQUIRE(F > 0 && ldD >= F && ldY >= F, "F / leading dimensions");
  return spmm_on_view(q->fview(use_values != 0), D, ldD, F, Y, ldY, q->f_partials, (hipStream_t)stream, bias, relu);
}

int mrgcn_support_spmm_t_compact_f32(const mrgcn_support_t *q, int32_t use_values, const float *dY, int64_t ldY,
                                     int32_t F, float *dM, int64_t ldM, void *stream) {
  REQUIRE_FORWARD(q);
  MRGCN_REQUIRE(dY && dM, "NULL");
  MRGCN_REQUIRE(F > 0 && ldY >= F && ldM >= F, "F / leading dimensions");
  return spmm_on_view(q->tview_ranked(use_values != 0), dY, ldY, F, dM, ldM, q->partials, (hipStream_t)stream);
}

int mrgcn_support_mix_fwd_f32(const mrgcn_support_t *q, const float *V, const float *comp, int32_t B, int32_t F,
                              float *M, int64_t ldM, void *stream) {
  MRGCN_REQUIRE(q && V && comp && M, "NULL");
  MRGCN_REQUIRE(B > 0 && F > 0 && ldM >= F, "B / F / ldM");
  // the live nodes' column ranges (lnptr) and the relation of every live column (lrel); row k of M = live column k
  return mix_fwd_arrays(q->lnptr, q->lrel, q->lnode, q->NL, q->L, (int)q->plan->num_relations, V, comp, B, F, M, ldM,
                        (hipStream_t)stream);
}

int mrgcn_support_literal_rows_f32(const mrgcn_support_t *q, int32_t scatter, float *table, int32_t F, float *M,
                                   int64_t ldM, void *stream) {
  MRGCN_REQUIRE(q && table && M, "NULL");
  MRGCN_REQUIRE(F > 0 && ldM >= F, "F / ldM");
  hipStream_t s = (hipStream_t)stream;
  const mrgcn_plan *p = q->plan;
  if (scatter)
    MRGCN_HIP_TRY(mrgcn::fill_async(table, 0, (size_t)p->num_relations * p->num_nodes * F * sizeof(float), s));
  if (q->L == 0) return MRGCN_OK;
  const unsigned grid = (unsigned)((q->L * F + 255) / 256);
  if (scatter)
    k_sup_literal_rows<true><<<dim3(grid), dim3(256), 0, s>>>(q->lcol, q->lrel, p->unode, q->L, p->num_nodes, F, table,
                                                             M, ldM);
  else
    k_sup_literal_rows<false><<<dim3(grid), dim3(256), 0, s>>>(q->lcol, q->lrel, p->unode, q->L, p->num_nodes, F,
                                                              table, M, ldM);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int32_t mrgcn_support_rel_transform_supported(const mrgcn_support_t *q, int32_t K, int32_t F, int32_t need_dX) {
  return (q && q->has_forward && xform_mfma_fwd_supported(K, F) && xform_mfma_dw_supported(K, F) &&
          (!need_dX || xform_mfma_dx_supported(F, K))) ? 1 : 0;
}

int mrgcn_support_rel_transform_fwd_f32(const mrgcn_support_t *q, const float *X, int64_t ldX, int32_t x_by_node,
                                        int32_t K, const float *W, int32_t F, float *T, int64_t ldT, void *stream) {
  REQUIRE_FORWARD(q);
  MRGCN_REQUIRE(X && W && T, "NULL");
  MRGCN_REQUIRE(K > 0 && F > 0 && ldX >= K && ldT >= F, "K / F / leading dimensions");
  if (!xform_mfma_fwd_supported(K, F)) {
    set_error("mrgcn_support_rel_transform_fwd_f32: shape outside the matrix-core transforms' limits");
    return MRGCN_ERR_UNSUPPORTED;
  }
  const RelOrder o = q->order_for(K, x_by_node == 0);
  // row t of the order: input = X[rank of its node among the live nodes] (x_by_node: X[its node]), output row = its
  // live number
  return xform_mfma_fwd(q->plan, o, o.rnode, nullptr, X, ldX, K, W, false, F, T, ldT, (hipStream_t)stream, false,
                        nullptr);
}

int mrgcn_support_rel_transform_bwd_compact_f32(const mrgcn_support_t *q, const float *dM, int64_t ldM, const float *X,
                                                int64_t ldX, int32_t x_by_node, int32_t K, const float *W, int32_t F, float *dX,
                                                int64_t lddX, float *dW, float *workspace, int64_t workspace_floats,
                                                int32_t relu_mask_from_x, void *stream) {
  REQUIRE_FORWARD(q);
  MRGCN_REQUIRE(dM && X && W, "NULL");
  MRGCN_REQUIRE(K > 0 && F > 0 && ldX >= K && ldM >= F, "K / F / leading dimensions");
  const int64_t need = mrgcn_support_rel_transform_bwd_workspace(q, K, F, dX != nullptr, dW != nullptr);
  if (need < 0) {
    set_error("mrgcn_support_rel_transform_bwd_compact_f32: shape outside the matrix-core transforms' limits");
    return MRGCN_ERR_UNSUPPORTED;
  }
  MRGCN_REQUIRE(workspace && workspace_floats >= need, "workspace (mrgcn_support_rel_transform_bwd_workspace floats)");
  MRGCN_REQUIRE(!relu_mask_from_x || (dX && K <= 16), "the masked dX needs K <= 16");
  MRGCN_REQUIRE(!(relu_mask_from_x && x_by_node), "the masked dX reads X by live-node rank");
  hipStream_t s = (hipStream_t)stream;
  if (dW) {
    const RelOrder o = q->order_for(K, x_by_node == 0);
    int rc = xform_mfma_dw(q->plan, o, o.rnode, X, ldX, K, dM, ldM, F, dW, workspace, workspace_floats, s, nullptr);
    if (rc != MRGCN_OK) return rc;
  }
  if (dX) {
    MRGCN_REQUIRE(lddX >= K, "lddX");
    const int64_t ldZ = ((int64_t)K + 3) / 4 * 4;
    // Z[k] = dM[k] . W[r_k]^T by live number, then row t of dX = the sum of live node t's Z rows (lnptr)
    int rc = xform_mfma_fwd(q->plan, q->order_for(F), nullptr, nullptr, dM, ldM, F, W, true, K, workspace, ldZ, s,
                            false, nullptr);
    if (rc != MRGCN_OK) return rc;
    rc = segment_sum_arrays(q->lnptr, q->NL, q->L, workspace, ldZ, K, dX, lddX, s, relu_mask_from_x ? X : nullptr,
                            ldX, 8);
    if (rc != MRGCN_OK) return rc;
  }
  return MRGCN_OK;
}

}  // extern "C"
