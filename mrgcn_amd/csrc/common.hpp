// Internal definitions shared by the translation units of libmrgcn_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/mrgcn_hip.h"

namespace mrgcn {

void set_error(const std::string &msg);

#define MRGCN_HIP_TRY(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ::mrgcn::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" +      \
                         __FILE__ + ":" + std::to_string(__LINE__) + ")");              \
      return MRGCN_ERR_HIP;                                                              \
    }                                                                                    \
  } while (0)

#define MRGCN_REQUIRE(cond, msg)                                     \
  do {                                                               \
    if (!(cond)) {                                                   \
      ::mrgcn::set_error(std::string("invalid argument: ") + (msg)); \
      return MRGCN_ERR_INVALID;                                      \
    }                                                                \
  } while (0)

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kLongThreshold = 32; // rows with more entries go to the split-row path
constexpr int kChunk = 512;        // entries per split-row chunk (one wave each)
constexpr int kWsFeatures = 256;   // split-row workspace is sized for this many features
// k_spmm3 (spmm.hip): rows of <= kShort3Rows entries share a wave 64/G at a time, rows up to kMid3Rows four at a
// time, longer rows are cut into chunks of kChunk3Entries (one gather batch of 8 per lane at G = 4)
constexpr int kShort3Rows = 8;
constexpr int kMid3Rows = 32;
constexpr int kChunk3Entries = 128;
// class S / M rows ordered by length inside windows of this many row ids (0 = plain id order, the default: measured
// neutral to -1 % on the AM shape — the masked filler gathers of mixed-length waves cost nothing)
constexpr int kLenWindowS = 0;
constexpr int kLenWindowM = 0;
constexpr int kChunk3Cap = 64;  // chunks per row at most: longer rows get chunks of several pieces of 128
constexpr int kHotMinRefs = 16;    // columns read by >= this many rows go to the dense hot region of M
constexpr int kNodeBand = 131072;  // source nodes per band of the transform order (see plan.hip)
constexpr int kRelChunk = 1024;    // compact columns of one relation per transform block

// One CSR-shaped view of the adjacency: `rows` output rows, entry e of row i
// multiplies dense row idx[e] by val[e].  Rows longer than kLongThreshold are cut into
// chunks of <= kChunk entries that one wave each reduces into `partials`.
struct SparseView {
  int64_t rows = 0;
  const int32_t *ptr = nullptr;  // [rows+1]
  const int32_t *idx = nullptr;  // [nnz]
  const float *val = nullptr;    // [nnz]
  // split-row path
  int32_t n_long = 0;                  // long rows
  int32_t n_chunks = 0;                // chunks over all long rows
  int32_t n_multi = 0;                 // > 0 iff some long row spans several chunks
  const int32_t *long_row = nullptr;   // [n_long]   row id
  const int32_t *long_cptr = nullptr;  // [n_long+1] chunk range of each long row
  const int32_t *chunk_beg = nullptr;  // [n_chunks] first entry
  const int32_t *chunk_end = nullptr;  // [n_chunks] one past last entry
  const int32_t *chunk_row = nullptr;  // [n_chunks] row id when the row is this single chunk, else -(row + 2);
                                       // [n_chunks .. 2 n_chunks) the position of the chunk's row in long_row
  // [n_long] arrival counters (zero between launches): with them the wave that delivers the LAST partial sum of a row
  // of several chunks adds them all, in chunk order, inside the product — no finalize launch.  NULL: two passes.
  int32_t *ticket = nullptr;
};

}  // namespace mrgcn

struct mrgcn_plan;
namespace mrgcn {
// One relation-major order of the compact columns: (node band, relation, node), cut into chunks of <= kRelChunk
// columns of one (band, relation) group — what the per-relation dense transforms walk.  A plan keeps two: bands of
// kNodeBand nodes for wide inputs (layer 0's X rows) and bands of kNodeBandNarrow nodes for narrow ones (a hidden
// layer's 40-byte rows: a band of them then stays inside one XCD's L2).
struct RelOrder {
  const int32_t *rperm = nullptr;   // [ncols] compact ids in the order
  const int32_t *rnode = nullptr;   // [ncols] source node of each
  const int32_t *rmpos = nullptr;   // [ncols] operand row of each
  const int32_t *relchunk_rel = nullptr, *relchunk_beg = nullptr, *relchunk_end = nullptr;  // [n_relchunks]
  const int32_t *relchunk_ptr = nullptr;  // [R+1] range of each relation inside relchunk_ids
  const int32_t *relchunk_ids = nullptr;  // [n_relchunks] chunk ids grouped by relation
  int32_t n_relchunks = 0, max_relchunks = 0;
};
constexpr int kWorkTickets = 64;       // counters of mrgcn_plan::work_tickets
constexpr int kWorkTicketStride = 64;  // unsigned long longs between two counters (512 bytes)
constexpr int kNarrowInput = 32;       // inputs of up to this many floats per row take the narrow order
constexpr int kNodeBandNarrow = 32768; // source nodes per band of the narrow order
// bf16 <-> f32 (raw uint16_t storage; round to nearest even, NaN kept quiet)
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// operand element store: OT = float or uint16_t (bf16)
template <typename OT> __device__ __forceinline__ void store_operand(OT *p, float v) {
  if constexpr (sizeof(OT) == 4) *p = v; else *p = f32_to_bf16(v);
}
template <typename OT> __device__ __forceinline__ float load_operand(const OT *p) {
  if constexpr (sizeof(OT) == 4) return *p; else return bf16_to_f32(*p);
}

// MFMA relation transforms (xform_mfma.hip)
bool xform_mfma_fwd_supported(int K, int F);
bool xform_mfma_dx_supported(int F, int K);  // Z = dM . W^T: F floats in, K out
bool xform_mfma_dw_supported(int K, int F);
bool xform_mfma_dw_live_supported(int K, int F);
// `o`: the relation-major order to walk; rin_idx / rout_idx (nullable) must belong to the same order
int xform_mfma_fwd(const mrgcn_plan *p, const RelOrder &o, const int32_t *rin_idx, const int32_t *rout_idx,
                   const float *In, int64_t ldIn, int K, const float *W, bool trans_w, int F, void *Out,
                   int64_t ldOut, hipStream_t s, bool out_bf16 = false, const uint8_t *col_live = nullptr);
// (in_bf16: `In` holds bf16 rows, ldIn in elements — the bf16 pipeline's X; the products stay fp32)
int xform_mfma_dw(const mrgcn_plan *p, const RelOrder &o, const int32_t *rin_idx, const void *In, int64_t ldIn,
                  int K, const float *G, int64_t ldG, int F, float *dW, float *workspace,
                  int64_t workspace_floats, hipStream_t s, const uint8_t *col_live = nullptr, bool in_bf16 = false);
// the bf16 pipeline's forward transform (bf16 input rows, v_mfma_f32_16x16x32_bf16), F <= 16, K <= 256, ldOut <= 16
bool xform_bf16_fwd_supported(int K, int F, int64_t ldIn, int64_t ldOut);
int xform_bf16_fwd(const mrgcn_plan *p, const RelOrder &o, const int32_t *rin_idx, const int32_t *rout_idx,
                   const uint16_t *In, int64_t ldIn, int K, const float *W, int F, void *Out, int64_t ldOut,
                   hipStream_t s, bool out_bf16);
int cast_rows_bf16(const float *src, int64_t ldSrc, int64_t rows, int K, uint16_t *dst, int64_t ldDst, hipStream_t s);
int segment_sum(const mrgcn_plan *p, const float *Z, int64_t ldZ, int K, float *dX, int64_t lddX,
                hipStream_t s, const uint8_t *col_live = nullptr, const float *mask_src = nullptr,
                int64_t ldMask = 0, uint8_t *row_live = nullptr, const uint8_t *node_live = nullptr);
// the same over explicit arrays (a gradient support's live columns: nptr = node -> range of Z rows)
int segment_sum_arrays(const int32_t *nptr, int64_t num_nodes, int64_t nz_rows, const float *Z, int64_t ldZ, int K,
                       float *dX, int64_t lddX, hipStream_t s, const float *mask_src, int64_t ldMask,
                       int lane_max = 0);  // > 0: every node is live, a lane sums a node of up to lane_max rows itself

// internal launchers shared with support.hip
// Y = v . D on an arbitrary CSR-shaped view (spmm.hip); `partials`: v.n_chunks * kWsFeatures floats
int spmm_on_view(const SparseView &v, const float *D, int64_t ldD, int F, float *Y, int64_t ldY, float *partials,
                 hipStream_t s, const float *bias = nullptr, int relu = 0);  // (v.ticket set: one launch)
// the basis mix over explicit arrays (rgcn_fused.hip): entry t of a node list owns columns nptr[t] .. nptr[t+1] (their
// relations in urel) and reads V block node_ids[t] (NULL: t); row c of M is column c's
int mix_fwd_arrays(const int32_t *nptr, const int32_t *urel, const int32_t *node_ids, int64_t n_nodes, int64_t ncols,
                   int R, const float *V, const float *comp, int32_t B, int32_t F, float *M, int64_t ldM,
                   hipStream_t s);
// k_mix_bwd_nm over explicit node -> column-range / relation arrays (rgcn_fused.hip); -1: shape outside its limits
int mix_bwd_nm_arrays(const int32_t *nptr, const int32_t *urel, int64_t N, int R, int top_rel, const float *dM,
                      int64_t ldM, const float *V, const float *comp, int32_t B, int32_t F, float *dV, float *dcomp,
                      double *dV_sumsq, hipStream_t s, const uint8_t *col_live, uint8_t *node_cur);
// k_adam_rows_fused over explicit arrays (rgcn_fused.hip)
int adam_rows_fused_arrays(const int32_t *nptr, const int32_t *urel, const uint8_t *col_live, int64_t N, int R,
                           const float *dM, int64_t ldM, const float *comp, int32_t B, int32_t F, float *param,
                           float *exp_avg, float *exp_avg_sq, const uint8_t *row_cur, uint8_t *row_ever, float lr,
                           float beta1, float beta2, float eps, int64_t step, const float *bc_dev,
                           const float *grad_scale, hipStream_t s, const int32_t *lnode = nullptr,
                           const int32_t *lnptr = nullptr, int64_t NL = 0, int ever_outside = 1);
bool xform_use_mfma();
// narrow transform with every relation's weights in LDS, columns in output order (xform_mfma.hip)
bool xform_cols_lds_supported(const mrgcn_plan *p, int K, int F, int64_t ldOut, bool operand_order);
int xform_cols_lds(const mrgcn_plan *p, bool operand_order, const float *In, int64_t ldIn, int K, const float *W, int F,
                   void *Out, int64_t ldOut, hipStream_t s, bool out_bf16);
// hipMemsetAsync for the compute calls (plan.hip).  On a capturing stream the fill is one kernel of this package:
// a captured hipMemsetAsync whose size is not a whole number of 16-byte pieces becomes a memset node that ROCm 7.2
// replays once and then faults on ("write access to a read-only page", second replay; found with the 174 504-byte
// histogram of mrgcn_distmult_orders_counting) — kernel nodes replay fine.
hipError_t fill_async(void *dst, int byte_value, size_t bytes, hipStream_t s);
// dynamic LDS beyond 48 KB needs the kernel's limit raised once — per (device, kernel), whatever call site launches it
// (plan.hip); every launch with more than 48 KB of dynamic LDS goes through it
hipError_t raise_lds_limit(const void *fn, size_t lds);
// the product scratch of `p` for work submitted on stream `s` (see mrgcn_plan::stream_scratch); the first product of a
// second, third ... stream allocates that stream's set and must therefore not run inside a stream capture
int plan_scratch(const mrgcn_plan *p, hipStream_t s, float **partials, int32_t **ticket);
// the literal column of every entry of the COMPACT view (mrgcn_plan::mlcol), built by the first call outside a capture
bool plan_literal_cols(const mrgcn_plan *p, hipStream_t s);
// the compact column of every CSC entry (mrgcn_plan::ecol), built by the first call outside a capture
bool plan_entry_cols(const mrgcn_plan *p, hipStream_t s);
// entries a wave of the entry-sliced transposed product owns; records it leaves: 2 per wave x (1 + 16) floats
constexpr int kSegWaveEntries = 256;
inline int64_t seg_waves(int64_t nnz) { return (nnz + kSegWaveEntries - 1) / kSegWaveEntries; }
inline int64_t seg_scratch_floats(int64_t nnz) { return 2 * seg_waves(nnz) * 17; }
}  // namespace mrgcn

struct mrgcn_plan {
  int64_t num_rows = 0, num_nodes = 0, num_relations = 0, nnz = 0, ncols = 0;
  int64_t max_row_nnz = 0, max_col_nnz = 0;
  int64_t device_bytes = 0;
  hipStream_t build_stream = nullptr;  // the plan's arrays are pool allocations ordered on this stream
  int device = 0;
  // CSR over output rows
  int32_t *rowptr = nullptr, *lcol = nullptr, *ccol = nullptr, *rowidx = nullptr;
  float *val = nullptr;
  // CSC over compact columns, (j, r) order
  int32_t *cptr = nullptr, *crow = nullptr, *urel = nullptr, *unode = nullptr, *nptr = nullptr,
          *ulcol = nullptr;
  float *cval = nullptr;
  // Storage order of the compact dense operand M: row mpos[c] of M holds compact column c.
  // Hot columns (read by >= kHotMinRefs output rows) come first, most referenced first, so that
  // the heavily re-read part of M is small and dense (L2 resident); every other column follows in
  // the order of the FIRST output row that reads it, so that a row's first-touch operand rows
  // are one contiguous run and the forward gather streams them instead of fetching one 128-B
  // line per 40-B row (only the 2nd..kth reads of a lukewarm column stay random).
  // With replicas (MRGCN_PLAN_REPLICATE): M has n_op = n_hot + (entries of non-hot columns) rows; a non-hot
  // column's primary row mpos[c] is the row of its first reader, rep_src / rep_dst list the copies.
  int64_t n_op = 0, n_rep = 0;
  int32_t *rep_src = nullptr, *rep_dst = nullptr;  // [n_rep] operand rows
  int32_t *mpos = nullptr;  // [ncols] compact id -> row of M
  int32_t *mcol = nullptr;  // [nnz]   operand row of each entry, entries of a row sorted by it
  float *mval = nullptr;    // [nnz]   values in the same order (the COMPACT view's arrays)
  // relation-major order of the compact columns (for per-relation dense transforms)
  int32_t *rperm = nullptr;   // [ncols] compact ids sorted by (node band, relation, node)
  int32_t *relptr = nullptr;  // [n_bands*R+1] range of each (band, relation) group in rperm
  int64_t node_band = 0, n_bands = 0;
  int32_t *rnode = nullptr;   // [ncols] unode[rperm[k]]: source node, relation-major
  int32_t *rmpos = nullptr;   // [ncols] mpos[rperm[k]]: operand row, relation-major
  int32_t *relchunk_rel = nullptr, *relchunk_beg = nullptr, *relchunk_end = nullptr;  // [n_relchunks]
  int32_t *relchunk_ptr = nullptr;  // [R+1] range of each relation inside relchunk_ids
  int32_t *relchunk_ids = nullptr;  // [n_relchunks] chunk ids grouped by relation
  int32_t n_relchunks = 0, max_relchunks = 0;
  int32_t top_rel = -1;  // relation with the most compact columns (the identity block in the reference's layout)
  // (source node, relation) of every row of the compact operand, in OPERAND order (rows without a primary column —
  // replicas — hold node -1): what the narrow transform walks, so that its output leaves as one sequential stream
  int32_t *op_node = nullptr, *op_rel = nullptr;  // [n_op]
  // the same order with narrow bands (kNodeBandNarrow) for transforms of narrow inputs; empty when the graph has a
  // single band either way
  int32_t *n_rperm = nullptr, *n_relptr = nullptr, *n_rnode = nullptr, *n_rmpos = nullptr, *n_relchunk_rel = nullptr,
          *n_relchunk_beg = nullptr, *n_relchunk_end = nullptr, *n_relchunk_ptr = nullptr, *n_relchunk_ids = nullptr;
  int32_t n_n_relchunks = 0, n_max_relchunks = 0;
  int64_t n_node_band = 0, n_n_bands = 0;
  mrgcn::RelOrder order_for(int input_width) const {
    mrgcn::RelOrder o;
    if (input_width <= mrgcn::kNarrowInput && n_rperm) {
      o.rperm = n_rperm; o.rnode = n_rnode; o.rmpos = n_rmpos; o.relchunk_rel = n_relchunk_rel;
      o.relchunk_beg = n_relchunk_beg; o.relchunk_end = n_relchunk_end; o.relchunk_ptr = n_relchunk_ptr;
      o.relchunk_ids = n_relchunk_ids; o.n_relchunks = n_n_relchunks; o.max_relchunks = n_max_relchunks;
    } else {
      o.rperm = rperm; o.rnode = rnode; o.rmpos = rmpos; o.relchunk_rel = relchunk_rel;
      o.relchunk_beg = relchunk_beg; o.relchunk_end = relchunk_end; o.relchunk_ptr = relchunk_ptr;
      o.relchunk_ids = relchunk_ids; o.n_relchunks = n_relchunks; o.max_relchunks = max_relchunks;
    }
    return o;
  }
  // split-row descriptors, one set per orientation
  int32_t *r_long_row = nullptr, *r_long_cptr = nullptr, *r_chunk_beg = nullptr, *r_chunk_end = nullptr,
          *r_chunk_row = nullptr;
  int32_t *c_long_row = nullptr, *c_long_cptr = nullptr, *c_chunk_beg = nullptr, *c_chunk_end = nullptr,
          *c_chunk_row = nullptr;
  int32_t r_n_long = 0, r_n_chunks = 0, c_n_long = 0, c_n_chunks = 0;
  // COMPACT view: rows in class-major processing order (rank k = row rowmap[k]; ranks [0, n_short3) have
  // <= kShort3Rows entries, the next n_mid3 <= kMid3Rows, the rest more), ptr3 = row pointer over ranks into
  // mcol / mval; split-row descriptors over ranks for k_spmm (q_*) and k_spmm3 (r3_*)
  int32_t *rowmap = nullptr, *ptr3 = nullptr;
  bool lean = false;  // MRGCN_PLAN_LEAN: ptr3 / mcol / mval / q_* alias rowptr / ccol / val / r_*, rowmap and mpos are identities
  int32_t n_short3 = 0, n_mid3 = 0;
  int32_t *q_long_row = nullptr, *q_long_cptr = nullptr, *q_chunk_beg = nullptr, *q_chunk_end = nullptr,
          *q_chunk_row = nullptr;
  int32_t q_n_long = 0, q_n_chunks = 0;
  int32_t *r3_long_row = nullptr, *r3_long_cptr = nullptr, *r3_chunk_beg = nullptr, *r3_chunk_end = nullptr,
          *r3_chunk_row = nullptr;
  int32_t r3_n_long = 0, r3_n_chunks = 0;
  int32_t *r3_multi = nullptr;  // [r3_n_multi] positions in r3_long_row of the rows that span several chunks
  int32_t r3_n_multi = 0;       // (k_spmm3_finalize runs over these only)
  int32_t *r3_ticket = nullptr;  // [ticket_ints] arrival counters of the in-kernel finalize (zero between launches):
                                 // one per long row of whichever view a product runs on
  int64_t ticket_ints = 1;
  // in-order work tickets of the persistent streaming kernels (k_mix_fwd_mfma's TK form): kWorkTickets counters, one
  // 512-byte slot each (different channels); zeroed by the launcher in front of every launch
  unsigned long long *work_tickets = nullptr;
  // k_spmm3's one-wave rows (kMid3Rows < len <= kChunk3Entries); r3_* above describe the longer, blockwise rows
  int32_t *r3s_long_row = nullptr, *r3s_long_cptr = nullptr, *r3s_chunk_beg = nullptr, *r3s_chunk_end = nullptr,
          *r3s_chunk_row = nullptr;
  int32_t r3s_n_chunks = 0;
  float *partials = nullptr;  // [max(r_n_chunks, c_n_chunks) * kWsFeatures]
  // `partials` and `r3_ticket` are SCRATCH of the products, written by every launch: one set per stream that runs
  // products on the plan, so that products on different streams never share it (mrgcn::plan_scratch).  The set above
  // serves the first stream that asks; another stream gets its own at its first product on the plan.
  struct StreamScratch { hipStream_t stream; float *partials; int32_t *ticket; };
  mutable std::vector<StreamScratch> stream_scratch;
  mutable std::mutex scratch_mu;
  // LITERAL products of narrow layers on the COMPACT view's row classes: the literal column r*N + j of every entry in
  // the compact view's entry order (built by the first such product outside a capture; plan.hip: plan_literal_cols)
  mutable int32_t *mlcol = nullptr;  // [nnz]
  mutable int32_t *ecol = nullptr;   // [nnz] compact column of every CSC entry (the entry-sliced transposed product)
  int64_t partials_floats = 0;

  mrgcn::SparseView view(int which) const {
    mrgcn::SparseView v;
    if (which == MRGCN_VIEW_TRANSPOSED) {
      v.rows = ncols; v.ptr = cptr; v.idx = crow; v.val = cval;
      v.n_long = c_n_long; v.n_chunks = c_n_chunks; v.long_row = c_long_row;
      v.long_cptr = c_long_cptr; v.chunk_beg = c_chunk_beg; v.chunk_end = c_chunk_end;
      v.chunk_row = c_chunk_row; v.n_multi = c_n_chunks - c_n_long;
    } else if (which == MRGCN_VIEW_COMPACT) {  // rows = ranks (class-major); results go to row rowmap[rank]
      v.rows = num_rows; v.ptr = ptr3; v.idx = mcol; v.val = mval;
      v.n_long = q_n_long; v.n_chunks = q_n_chunks; v.long_row = q_long_row;
      v.long_cptr = q_long_cptr; v.chunk_beg = q_chunk_beg; v.chunk_end = q_chunk_end;
      v.chunk_row = q_chunk_row; v.n_multi = q_n_chunks - q_n_long;
    } else {
      v.rows = num_rows; v.ptr = rowptr; v.idx = lcol; v.val = val;
      v.n_long = r_n_long; v.n_chunks = r_n_chunks; v.long_row = r_long_row;
      v.long_cptr = r_long_cptr; v.chunk_beg = r_chunk_beg; v.chunk_end = r_chunk_end;
      v.chunk_row = r_chunk_row; v.n_multi = r_n_chunks - r_n_long;
    }
    return v;
  }
};

// The GRADIENT SUPPORT of a set of output rows on a plan (plan.hip builds it, support.hip computes on it): in a
// semi-supervised epoch the loss touches the labelled rows only (node_classification.py:439-444), so the rows of a
// layer's output gradient that can hold anything are known from the label set and the graph alone — and with them the
// compact columns that receive gradient (the LIVE columns), their source nodes and, among a live column's entries,
// the ones that multiply a live row.  Built once per (plan, row set); the backward of every epoch then runs on dense
// index spaces: live columns numbered 0..L-1 in (node, relation) order, dM / the per-column products stored by that
// number.  Nothing is scanned, marked or skipped per epoch.
struct mrgcn_support {
  const mrgcn_plan *plan = nullptr;
  int device = 0;
  hipStream_t build_stream = nullptr;
  int64_t device_bytes = 0;
  int64_t L = 0, E = 0, NL = 0;  // live columns, live entries (of live columns, in live rows), live nodes
  uint8_t *col_flags = nullptr;   // [ncols]  1 = live
  uint8_t *node_flags = nullptr;  // [N]      1 = the node has a live column (the row set of the layer below)
  uint8_t *node_scratch = nullptr;  // [N]    scratch for kernels that report the nodes they wrote
  int32_t *lcol = nullptr;        // [L]      compact column id, ascending
  int32_t *lrel = nullptr;        // [L]      its relation
  int32_t *nlptr = nullptr;       // [N+1]    node -> range of live columns
  int32_t *lnode = nullptr;       // [NL]     nodes with a live column, ascending
  int32_t *lnptr = nullptr;       // [NL+1]   their ranges
  // the transposed view restricted to live columns x live rows (entries keep the plan's order)
  int32_t *lptr = nullptr, *lrow = nullptr;
  float *lval = nullptr;
  int32_t *t_long_row = nullptr, *t_long_cptr = nullptr, *t_chunk_beg = nullptr, *t_chunk_end = nullptr,
          *t_chunk_row = nullptr;
  int32_t t_n_long = 0, t_n_chunks = 0;
  float *partials = nullptr;
  int32_t *ticket = nullptr;  // [max(t_n_long, f_n_long)] arrival counters of the products' in-kernel finalize
  // FORWARD arrays (MRGCN_SUPPORT_FORWARD: a mini-batch layer as a masked pass over the full plan — the flagged rows
  // are the batch's sample, the live nodes its neighbours): the flagged rows in rising order with their entries in the
  // plan's row order, columns by live number; ranks so that activations / gradients can stay compact
  // ([flagged rows] x F, [live nodes] x K)
  int64_t NR = 0;                  // flagged rows
  int32_t *frow = nullptr;         // [NR]       flagged row ids, rising
  int32_t *rowrank = nullptr;      // [num_rows] rank among the flagged rows, -1 elsewhere
  int32_t *fptr = nullptr, *fcol = nullptr;  // [NR+1], [E]  CSR over the flagged rows, column = live number
  float *fval = nullptr;           // [E]        stored values in that order
  float *ones = nullptr;           // [E]        1.0f (the feature term of a mini-batch multiplies the all-ones slice)
  int32_t *lrow_rank = nullptr;    // [E]        rowrank of lrow (the transposed view over compact gradients)
  int32_t *lnode_ord = nullptr;    // [L]        rank of the live column's node among the live nodes
  int32_t *f_long_row = nullptr, *f_long_cptr = nullptr, *f_chunk_beg = nullptr, *f_chunk_end = nullptr,
          *f_chunk_row = nullptr;
  int32_t f_n_long = 0, f_n_chunks = 0;
  float *f_partials = nullptr;
  bool has_forward = false;
  struct Order {  // live columns in (node band, relation, node) order, cut like common.hpp: RelOrder
    int32_t *lperm = nullptr, *lrin = nullptr;  // [L] live index / source node
    int32_t *lrin_ord = nullptr;                // [L] (forward arrays) rank of that node among the live nodes
    int32_t *chunk_rel = nullptr, *chunk_beg = nullptr, *chunk_end = nullptr, *chunk_ptr = nullptr, *chunk_ids = nullptr;
    int32_t n_chunks = 0, max_chunks = 0;
  } wide, narrow;
  bool has_narrow = false;
  std::vector<void *> owned;
  mrgcn::SparseView tview() const {
    mrgcn::SparseView v;
    v.rows = L; v.ptr = lptr; v.idx = lrow; v.val = lval;
    v.n_long = t_n_long; v.n_chunks = t_n_chunks; v.long_row = t_long_row; v.long_cptr = t_long_cptr;
    v.chunk_beg = t_chunk_beg; v.chunk_end = t_chunk_end; v.chunk_row = t_chunk_row;
    v.n_multi = t_n_chunks - t_n_long;
    v.ticket = ticket;
    return v;
  }
  // the forward view (rows = flagged rows by rank) and the transposed view over compact gradients; `use_values` = the
  // stored values, else all ones
  mrgcn::SparseView fview(bool use_values) const {
    mrgcn::SparseView v;
    v.rows = NR; v.ptr = fptr; v.idx = fcol; v.val = use_values ? fval : ones;
    v.n_long = f_n_long; v.n_chunks = f_n_chunks; v.long_row = f_long_row; v.long_cptr = f_long_cptr;
    v.chunk_beg = f_chunk_beg; v.chunk_end = f_chunk_end; v.chunk_row = f_chunk_row;
    v.n_multi = f_n_chunks - f_n_long;
    v.ticket = ticket;
    return v;
  }
  mrgcn::SparseView tview_ranked(bool use_values) const {
    mrgcn::SparseView v = tview();
    v.idx = lrow_rank;
    v.val = use_values ? lval : ones;
    return v;
  }
  mrgcn::RelOrder order_for(int input_width, bool by_ordinal = false) const {
    const Order &q = (input_width <= mrgcn::kNarrowInput && has_narrow) ? narrow : wide;
    mrgcn::RelOrder o;
    o.rperm = q.lperm; o.rnode = by_ordinal ? q.lrin_ord : q.lrin; o.rmpos = nullptr; o.relchunk_rel = q.chunk_rel; o.relchunk_beg = q.chunk_beg;
    o.relchunk_end = q.chunk_end; o.relchunk_ptr = q.chunk_ptr; o.relchunk_ids = q.chunk_ids;
    o.n_relchunks = q.n_chunks; o.max_relchunks = q.max_chunks;
    return o;
  }
};
