// Mini-batch frontier on the device (reference mrgcn/data/batch.py:185-263: `A[sample_idx]`,
// getNeighboursSparse, getAdjacencyNodeColumnIdx + sliceSparseCOO).  The stacked adjacency's CSR stays resident
// in HBM; one layer of a batch is two calls with ONE two-word readback between them (the sizes of the outputs,
// which the caller allocates):
//   mrgcn_frontier_count : row offsets of the slice A[sample] and, for every node, its position among the
//                          source nodes those rows touch (exclusive scan of the touched flags)
//   mrgcn_frontier_emit  : the slice as COO in the order `A[sample].nonzero()` has on the host (row-major, stored
//                          column order) with global columns, the same entries with columns renumbered to
//                          r * n_b + position(node) (what sliceSparseCOO(A[sample], A_idx) keeps: every entry,
//                          since the neighbour set is made of exactly these columns' nodes), and the ascending
//                          neighbour list.
// HBM-bound integer work: a wave walks one row's entries with coalesced 8-byte loads; no atomics (the touched flags
// are idempotent byte-sized facts written as whole words of value 1).
#include <hipcub/hipcub.hpp>

#include "common.hpp"

using namespace mrgcn;

namespace {

constexpr int kTB = 256;
inline unsigned nblocks(int64_t n, int per = kTB) { return (unsigned)((n + per - 1) / per > 0 ? (n + per - 1) / per : 1); }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__global__ void k_fr_len(const int64_t *__restrict__ indptr, const int64_t *__restrict__ sample, int64_t n_sample,
                         int64_t *__restrict__ len) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_sample) return;
  int64_t l = 0;
  if (i < n_sample) {
    const int64_t s = sample[i];
    l = indptr[s + 1] - indptr[s];
  }
  len[i] = l;  // [n_sample] = 0: the exclusive scan leaves the total there
}

// wave per sample row
__global__ void k_fr_mark(const int64_t *__restrict__ indptr, const int64_t *__restrict__ indices,
                          const int64_t *__restrict__ sample, int64_t n_sample, int64_t num_nodes,
                          int32_t *__restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < n_sample; i += nwaves) {
    const int64_t s = sample[i];
    const int64_t e1 = indptr[s + 1];
    for (int64_t e = indptr[s] + lane; e < e1; e += 64) flag[indices[e] % num_nodes] = 1;
  }
}

template <typename VT>
__global__ void k_fr_emit(const int64_t *__restrict__ indptr, const int64_t *__restrict__ indices,
                          const float *__restrict__ data, const int64_t *__restrict__ sample, int64_t n_sample,
                          int64_t num_nodes, const int64_t *__restrict__ row_off,
                          const int32_t *__restrict__ node_pos, int64_t n_nb, int64_t *__restrict__ out_row,
                          int64_t *__restrict__ out_col, VT *__restrict__ out_val,
                          int64_t *__restrict__ out_col_sliced) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave; i < n_sample; i += nwaves) {
    const int64_t s = sample[i];
    const int64_t e0 = indptr[s], e1 = indptr[s + 1];
    const int64_t o0 = row_off[i];
    for (int64_t e = e0 + lane; e < e1; e += 64) {
      const int64_t c = indices[e];
      const int64_t o = o0 + (e - e0);
      out_row[o] = i;
      out_col[o] = c;
      if (out_val) {
        const float v = data[e];
        if constexpr (sizeof(VT) == 1) out_val[o] = (VT)(int32_t)v;  // the boundary cast: truncation toward zero
        else out_val[o] = (VT)v;
      }
      if (out_col_sliced) {
        const int64_t r = c / num_nodes, j = c - r * num_nodes;
        out_col_sliced[o] = r * n_nb + node_pos[j];
      }
    }
  }
}

__global__ void k_fr_neighbours(const int32_t *__restrict__ node_pos, int64_t num_nodes,
                                int64_t *__restrict__ neighbours) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= num_nodes) return;
  const int32_t p = node_pos[j];
  if (node_pos[j + 1] != p) neighbours[p] = j;
}

struct Ws {
  int64_t *len;
  int32_t *flag;
  char *tmp;
  size_t tmp_bytes, total;
};

Ws carve(void *base, int64_t num_nodes, int64_t n_sample) {
  size_t t1 = 0, t2 = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t1, (int64_t *)nullptr, (int64_t *)nullptr, (int)(n_sample + 1));
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, t2, (int32_t *)nullptr, (int32_t *)nullptr, (int)(num_nodes + 1));
  Ws w;
  char *p = (char *)base;
  size_t off = 0;
  w.len = (int64_t *)(p + off);
  off += align256((size_t)(n_sample + 1) * sizeof(int64_t));
  w.flag = (int32_t *)(p + off);
  off += align256((size_t)(num_nodes + 1) * sizeof(int32_t));
  w.tmp = p + off;
  w.tmp_bytes = t1 > t2 ? t1 : t2;
  off += align256(w.tmp_bytes);
  w.total = off;
  return w;
}

}  // namespace

extern "C" {

size_t mrgcn_frontier_workspace_bytes(int64_t num_nodes, int64_t n_sample) {
  if (num_nodes < 0 || n_sample < 0) return 0;
  return carve(nullptr, num_nodes, n_sample).total;
}

int mrgcn_frontier_count(const int64_t *indptr, const int64_t *indices, int64_t num_nodes, const int64_t *sample,
                         int64_t n_sample, int64_t *row_off, int32_t *node_pos, void *ws, size_t ws_bytes,
                         void *stream) {
  MRGCN_REQUIRE(indptr && indices && row_off && node_pos && ws, "NULL");
  MRGCN_REQUIRE(num_nodes > 0 && num_nodes < (int64_t)1 << 31 && n_sample >= 0 && n_sample < (int64_t)1 << 31,
                "num_nodes / n_sample");
  MRGCN_REQUIRE(n_sample == 0 || sample, "sample");
  Ws w = carve(ws, num_nodes, n_sample);
  MRGCN_REQUIRE(ws_bytes >= w.total, "workspace smaller than mrgcn_frontier_workspace_bytes");
  hipStream_t s = (hipStream_t)stream;
  k_fr_len<<<nblocks(n_sample + 1), kTB, 0, s>>>(indptr, sample, n_sample, w.len);
  size_t tb = w.tmp_bytes;
  MRGCN_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.tmp, tb, w.len, row_off, (int)(n_sample + 1), s));
  MRGCN_HIP_TRY(mrgcn::fill_async(w.flag, 0, (size_t)(num_nodes + 1) * sizeof(int32_t), s));
  if (n_sample > 0) {
    const int64_t waves = n_sample;
    int64_t grid = (waves * 64 + kTB - 1) / kTB;
    if (grid > 8192) grid = 8192;
    k_fr_mark<<<dim3((unsigned)grid), kTB, 0, s>>>(indptr, indices, sample, n_sample, num_nodes, w.flag);
  }
  tb = w.tmp_bytes;
  MRGCN_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.tmp, tb, w.flag, node_pos, (int)(num_nodes + 1), s));
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_frontier_emit(const int64_t *indptr, const int64_t *indices, const float *data, int64_t num_nodes,
                        const int64_t *sample, int64_t n_sample, const int64_t *row_off, const int32_t *node_pos,
                        int64_t n_neighbours, int32_t value_dtype, int64_t *out_row, int64_t *out_col,
                        void *out_val, int64_t *out_col_sliced, int64_t *neighbours, void *stream) {
  MRGCN_REQUIRE(indptr && indices && row_off && node_pos, "NULL");  // (outputs of an entry-less slice may be NULL)
  MRGCN_REQUIRE(!out_val || data, "values requested without data");
  MRGCN_REQUIRE(value_dtype == MRGCN_VAL_F32 || value_dtype == MRGCN_VAL_I8, "value_dtype");
  MRGCN_REQUIRE(num_nodes > 0 && n_sample >= 0 && n_neighbours >= 0, "sizes");
  hipStream_t s = (hipStream_t)stream;
  if (n_sample > 0) {
    int64_t grid = (n_sample * 64 + kTB - 1) / kTB;
    if (grid > 8192) grid = 8192;
    if (value_dtype == MRGCN_VAL_I8)
      k_fr_emit<int8_t><<<dim3((unsigned)grid), kTB, 0, s>>>(indptr, indices, data, sample, n_sample, num_nodes,
                                                             row_off, node_pos, n_neighbours, out_row, out_col,
                                                             (int8_t *)out_val, out_col_sliced);
    else
      k_fr_emit<float><<<dim3((unsigned)grid), kTB, 0, s>>>(indptr, indices, data, sample, n_sample, num_nodes,
                                                            row_off, node_pos, n_neighbours, out_row, out_col,
                                                            (float *)out_val, out_col_sliced);
  }
  if (neighbours && n_neighbours > 0)
    k_fr_neighbours<<<nblocks(num_nodes), kTB, 0, s>>>(node_pos, num_nodes, neighbours);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // extern "C"
