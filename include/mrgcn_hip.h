/*
 * mrgcn_hip.h — C ABI of libmrgcn_hip.so (MI355X / gfx950).
 *
 * The reference (wxwilcke/mrgcn) has no FFI: its hot path leans on PyTorch ATen ops
 * called from Python.  Every entry point below replaces one such call site
 * (file:line relative to the reference tree) and is what a ctypes binding in the
 * reference's `mrgcn/layers/graph.py` would bind.
 *
 * Conventions
 *   - plain C, no C++ / torch types; every pointer is a DEVICE pointer unless the
 *     parameter name starts with `h_` (host);
 *   - every call returns MRGCN_OK (0) or an error code; the message is available
 *     from mrgcn_last_error() (thread local);
 *   - compute calls are asynchronous and stream ordered on `stream` (a hipStream_t,
 *     passed as void*; NULL = the null stream); they never allocate and never
 *     synchronise, so they may be captured into a hipGraph;
 *   - mrgcn_plan_create / _destroy / _export / mrgcn_event_* may allocate and
 *     synchronise;
 *   - buffers are caller owned; a plan owns its index copies, which are immutable after
 *     creation, and the SCRATCH of its products (partial sums of rows cut into chunks, the
 *     arrival counters of the in-kernel finalize), which every product launch writes.  The
 *     plan keeps one scratch set PER STREAM: products of one plan may be in flight on several
 *     streams (and be submitted from several threads) at once; products submitted to the same
 *     stream are ordered by it.  The first product a plan sees on a second, third, ... stream
 *     allocates that stream's set — the one exception to "compute calls never allocate": make
 *     that first call outside a stream capture (it fails with MRGCN_ERR_INVALID inside one).
 *     A gradient support (mrgcn_support_*) keeps ONE scratch set: one product in flight per support.
 */
#ifndef MRGCN_HIP_H
#define MRGCN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRGCN_ABI_VERSION 5

enum mrgcn_status {
  MRGCN_OK = 0,
  MRGCN_ERR_INVALID = 1,     /* bad argument (shape, alignment, NULL, enum) */
  MRGCN_ERR_HIP = 2,         /* a HIP runtime call failed */
  MRGCN_ERR_UNSUPPORTED = 3, /* valid request this build cannot serve */
  MRGCN_ERR_RANGE = 4        /* an index in the input is out of range */
};

enum mrgcn_val_dtype { MRGCN_VAL_I8 = 0, MRGCN_VAL_F32 = 1 };

/* plan flags */
#define MRGCN_PLAN_PRUNE_ZEROS 1u /* drop entries whose value is exactly 0 (legal: they
                                     contribute nothing) */
#define MRGCN_PLAN_REPLICATE 2u   /* compact operand with REPLICAS: every entry of a column read by fewer than
                                     16 rows gets an operand row of its own, placed where its reader streams
                                     (see mrgcn_operand_replicate); hot columns keep one shared row */
#define MRGCN_PLAN_NO_REPLICATE 4u /* never replicate, whatever the default (env MRGCN_REPLICATE) says */
#define MRGCN_PLAN_LEAN 8u        /* a plan for a small, short-lived adjacency (the slices of a re-sampled mini-batch):
                                   * the COMPACT view keeps the rows in their own order and the operand in compact-column
                                   * order (no class-major ranks, no first-touch operand order, no replicas, one
                                   * transform order), its products run on the general row kernel.  Same results up to
                                   * the summation order inside a row; about half the build passes and host round trips. */

/* which sparse view of the plan a product runs on */
enum mrgcn_view {
  MRGCN_VIEW_LITERAL = 0, /* rows = output nodes, cols = r*N + j (the reference's layout) */
  MRGCN_VIEW_COMPACT = 1, /* rows = output nodes, cols = rows of the compact operand M
                             (compact column c is stored at row MPOS[c]) */
  MRGCN_VIEW_TRANSPOSED = 2 /* rows = touched columns in (j, r) order, cols = output nodes */
};

/* arrays retrievable with mrgcn_plan_export (all int32 unless noted) */
enum mrgcn_plan_array {
  MRGCN_ARR_ROWPTR = 0,   /* [num_rows+1]  CSR row pointers                           */
  MRGCN_ARR_LCOL = 1,     /* [nnz]         literal column r*N + j, sorted within a row */
  MRGCN_ARR_CCOL = 2,     /* [nnz]         compact column id                           */
  MRGCN_ARR_VAL = 3,      /* [nnz] float   values (A.float())                          */
  MRGCN_ARR_CPTR = 4,     /* [ncols+1]     CSC pointers over compact columns           */
  MRGCN_ARR_CROW = 5,     /* [nnz]         output row of each CSC entry                */
  MRGCN_ARR_CVAL = 6,     /* [nnz] float   value of each CSC entry                     */
  MRGCN_ARR_UREL = 7,     /* [ncols]       relation r of each compact column           */
  MRGCN_ARR_UNODE = 8,    /* [ncols]       source node j of each compact column        */
  MRGCN_ARR_NPTR = 9,     /* [num_nodes+1] compact-column range of each source node    */
  MRGCN_ARR_ROWIDX = 10,  /* [nnz]         output row of each CSR entry                */
  MRGCN_ARR_ULCOL = 11,   /* [ncols]       literal column r*N + j of each compact column */
  MRGCN_ARR_RPERM = 12,   /* [ncols]       compact ids sorted by (node band, relation, node) */
  MRGCN_ARR_RELPTR = 13,  /* [bands*R+1]   range of each (band, relation) group inside RPERM */
  MRGCN_ARR_MPOS = 14,    /* [ncols]       row of the compact operand M that holds column c */
  MRGCN_ARR_MCOL = 15,    /* [nnz]         operand row per entry of the COMPACT view, whose rows are walked
                                           CLASS-MAJOR (ranks, see ROWMAP / PTR3); a row's entries in
                                           rising operand row                                      */
  MRGCN_ARR_MVAL = 16,    /* [nnz] float   values in MCOL's entry order                */
  MRGCN_ARR_ROWMAP = 17,  /* [num_rows]    COMPACT view: rank -> output row.  Ranks = rows sorted by class
                                           (<= 8 entries, <= 32, more), row order inside a class      */
  MRGCN_ARR_PTR3 = 18     /* [num_rows+1]  COMPACT view: entry range of each rank inside MCOL / MVAL  */
};

typedef struct mrgcn_plan mrgcn_plan_t;

typedef struct mrgcn_plan_info {
  int64_t num_rows;      /* rows of A (output nodes)                     */
  int64_t num_nodes;     /* N: source nodes per relation block           */
  int64_t num_relations; /* R                                            */
  int64_t nnz;           /* stored entries after optional pruning        */
  int64_t ncols;         /* distinct non-empty columns (n_cols_touched)  */
  int64_t max_row_nnz;
  int64_t max_col_nnz;
  int64_t long_rows;     /* rows handled by the split-row path           */
  int64_t long_cols;
  int64_t device_bytes;  /* bytes of device memory the plan owns         */
  int64_t operand_rows;  /* rows of the compact operand M: ncols, or more with MRGCN_PLAN_REPLICATE */
  int64_t replicas;      /* operand rows that are copies (mrgcn_operand_replicate fills them)       */
} mrgcn_plan_info_t;

/* ---- library ---------------------------------------------------------------- */
int mrgcn_abi_version(void);
/* ---- configuration ----------------------------------------------------------------------------------------------
 * Every switch the library's launchers look at lives in ONE table (csrc/config.hip): initialised once, on the first
 * read, from the MRGCN_* environment variables of the same names (upper case, prefixed), changed afterwards only by
 * mrgcn_config_set.  No compute entry point reads the environment or caches a switch: a value set here takes effect
 * at the next call (plans keep what they were built with).  Names: mrgcn_config_name(0 .. count - 1), e.g.
 * "adam_list", "node_band"; the environment spelling ("MRGCN_ADAM_LIST") is accepted too. */
int32_t mrgcn_config_count(void);
const char *mrgcn_config_name(int32_t i);
const char *mrgcn_config_doc(int32_t i);
int mrgcn_config_get(const char *name, int64_t *value);
int mrgcn_config_set(const char *name, int64_t value);
const char *mrgcn_arch(void);       /* "gfx950" */
const char *mrgcn_last_error(void); /* message of the last failing call on this thread */

/* ---- graph plan ---------------------------------------------------------------
 * Built once per adjacency (A is static across epochs).  Input is exactly what the
 * reference hands to GraphConvolution.forward: the uncoalesced COO of
 * mrgcn/data/utils.py:165-170 (int64 indices 2 x nnz, given here as two rows) with
 * int8 values (mrgcn/data/batch.py:144-149) or float32 values.  Replaces the
 * per-call `A.float()` of mrgcn/layers/graph.py:75,:95 (done once here).
 * Requires num_relations*num_nodes < 2^31 and nnz < 2^31. */
int mrgcn_plan_create(mrgcn_plan_t **plan, int64_t num_rows, int64_t num_nodes,
                      int32_t num_relations, int64_t nnz, const int64_t *coo_rows,
                      const int64_t *coo_cols, const void *coo_vals, int32_t val_dtype,
                      uint32_t flags, void *stream);
/* The same plan straight from the CSR arrays the dataset archive holds (`A.npz`:
 * mrgcn/data/io/tarball.py:151-157; device int32 indptr / indices, float32 data), without the
 * host-side `.nonzero()` expansion to an int64 COO (data/utils.py:165-170).
 * boundary_cast_i8 != 0 applies the reference's int8 truncation of the values (batch.py:144-149). */
int mrgcn_plan_create_csr(mrgcn_plan_t **plan, int64_t num_rows, int64_t num_nodes,
                          int32_t num_relations, int64_t nnz, const int32_t *indptr,
                          const int32_t *indices, const float *data, int32_t boundary_cast_i8,
                          uint32_t flags, void *stream);
/* The same two constructors with a hint: the row sizes (bytes) of the compact operands M the plan's COMPACT
 * products will read — 4 * F for the packed rows of a layer of F outputs (F not a multiple of four), 4 * roundup(F, 4)
 * for padded rows, 2 * ld for bf16 rows.  The order of M keeps columns that several rows read off the positions
 * whose row would straddle a 128-byte line for any of these sizes (a re-read then costs one line, not two; AM
 * shape, F = 10: 158 -> 148 us with 40-byte rows).  The hint only permutes the operand rows (MPOS): products are
 * the same up to the order in which a row's terms are added.  At most four sizes are
 * used; without a hint (or through the plain constructors) 48-byte rows are assumed. */
int mrgcn_plan_create_hinted(mrgcn_plan_t **plan, int64_t num_rows, int64_t num_nodes,
                             int32_t num_relations, int64_t nnz, const int64_t *coo_rows,
                             const int64_t *coo_cols, const void *coo_vals, int32_t val_dtype,
                             uint32_t flags, const int32_t *operand_row_bytes, int32_t n_row_bytes,
                             void *stream);
int mrgcn_plan_create_csr_hinted(mrgcn_plan_t **plan, int64_t num_rows, int64_t num_nodes,
                                 int32_t num_relations, int64_t nnz, const int32_t *indptr,
                                 const int32_t *indices, const float *data, int32_t boundary_cast_i8,
                                 uint32_t flags, const int32_t *operand_row_bytes, int32_t n_row_bytes,
                                 void *stream);
int mrgcn_plan_destroy(mrgcn_plan_t *plan);
/* mrgcn_plan_destroy waits for ALL work in flight on the plan's device.  _destroy_after waits only for `event` (a
 * hipEvent_t the caller recorded behind the last work that uses the plan, on whichever stream; NULL: nothing used it)
 * and for the plan's own build: other streams keep running — for callers that drop the plans of finished mini-batch
 * steps while the next batch is being built (mrgcn_amd.data.batch.BatchPrefetcher). */
int mrgcn_plan_destroy_after(mrgcn_plan_t *plan, void *event);
int mrgcn_plan_info(const mrgcn_plan_t *plan, mrgcn_plan_info_t *h_info);
/* copies one plan array to HOST memory (tests: index parity is bit-exact) */
int mrgcn_plan_export(const mrgcn_plan_t *plan, int32_t which, void *h_dst, int64_t capacity_bytes);
/* device pointer + element count of one plan array (owned by the plan, read only) */
int mrgcn_plan_array(const mrgcn_plan_t *plan, int32_t which, const void **d_ptr, int64_t *h_count);

/* ---- sparse x dense ------------------------------------------------------------
 * Y[i, 0:F] = sum_e val[e] * D[col[e], 0:F]  (+ bias[0:F])  (optionally ReLU'd)
 * over the entries e of row i of the chosen view.
 *   view LITERAL    : replaces torch.mm(A.float(), W_I / FW_F), graph.py:75,:95
 *   view COMPACT    : same product on a dense operand that holds only touched rows
 *   view TRANSPOSED : dM[c] = sum val * dY[row]; the autograd of the above
 *                     (SparseAddmmBackward: dDense = A^T dY)
 * `out_index` (nullable, int32 [rows of view]) redirects output row i to
 * Y[out_index[i]] — used to scatter compact rows into the literal (R*N) x F gradient.
 * D, Y row-major with leading dimensions ldD, ldY (in floats).
 * `relu` is a flag word: MRGCN_SPMM_RELU (= 1, what callers always passed for "apply ReLU") and
 * MRGCN_SPMM_PAD_WRITABLE: columns F .. min(ldY, 4*ceil(F/4)) - 1 of every row of Y belong to the
 * caller's buffer and may be overwritten with zeros.  Rows of F = 10 / 11 in a buffer of ldY = 12 then
 * leave as three 16-byte stores and consecutive rows complete their 128-byte lines (COMPACT view,
 * F <= 16: the product runs 5 % faster); without the flag nothing outside [0, F) is touched. */
#define MRGCN_SPMM_RELU 1
#define MRGCN_SPMM_PAD_WRITABLE 2
/* COMPACT view, F <= 16: rows cut into several chunks are by default finished inside the product kernel by the
 * wave that delivers the row's last partial sum (fixed summation order: results are bitwise reproducible and equal
 * to the two-pass form's).  This flag asks for the two-pass form: a second launch adds the partial sums. */
#define MRGCN_SPMM_TWO_PASS 4
int mrgcn_spmm_f32(const mrgcn_plan_t *plan, int32_t view, const float *D, int64_t ldD,
                   int32_t F, float *Y, int64_t ldY, const float *bias, int32_t relu,
                   const int32_t *out_index, void *stream);
/* the same product with the dense operand D stored in bf16 (raw uint16_t, ldD in elements),
 * fp32 values, fp32 accumulation and fp32 Y — the {bf16 dense} half of the SpMM set
 * (the reference has no bf16, tolerance is stated where it is used) */
int mrgcn_spmm_bf16(const mrgcn_plan_t *plan, int32_t view, const uint16_t *D, int64_t ldD,
                    int32_t F, float *Y, int64_t ldY, const float *bias, int32_t relu,
                    const int32_t *out_index, void *stream);

/* Fills the replicas of a compact operand built on a plan with MRGCN_PLAN_REPLICATE: the producers below
 * write each compact column once (row MPOS[c]); this call copies the row of every column read by 2..15
 * output rows to the operand rows of its other readers, so that the forward product reads M front to back
 * instead of fetching one 128-byte line per re-read 40-byte row.  `row_bytes` = leading dimension in bytes
 * (multiple of 4).  A no-op on a plan without replicas.  (AM shape: product 202 -> 189 us, the copies 79 us:
 * not the default.) */
int mrgcn_operand_replicate(const mrgcn_plan_t *plan, void *M, int64_t row_bytes, void *stream);

/* ---- compact dense operand: forward -----------------------------------------------
 * M is [ncols, ldM] row-major with one row per touched column c = (node j_c, relation
 * r_c); column c lives at row MPOS[c] (hot columns first, then single-use columns in the
 * order of the output row that reads them).  Producers write the WHOLE padded row (zeros in
 * [F, ldM)) and never read M: the one scattered pass of the forward is write-only.
 * `addend` (nullable, [ncols, ldA] in plain compact order) is added on the fly — it carries
 * the other term of the layer.  The backward operand dM (output of the TRANSPOSED view) is in
 * plain compact order.
 *
 * basis mix — replaces einsum('rb,bij->rij', weight_I_comp, weight_I.view(B,N,out)) and the
 * view to (R*N, out) of graph.py:69-72, restricted to the rows the product will read:
 *     M[MPOS[c], 0:F] = addend[c, 0:F] + sum_b comp[r_c, b] * V[j_c, b, 0:F]
 *     comp: [R, B];  V: NODE-MAJOR basis table [N][B][F] — the B rows of a node are one contiguous
 *     block of B*F floats.  It is the [B][N][F] -> [N][B][F] transpose of the reference's weight_I
 *     (graph.py:50-51: (B*N, out), row b*N + j): mrgcn_amd.layers.graph.GraphConvolution keeps its
 *     parameter in this layout and converts in its state-dict hooks. */
int mrgcn_basis_mix_fwd_f32(const mrgcn_plan_t *plan, const float *V, const float *comp, int32_t B,
                            int32_t F, const float *addend, int64_t ldA, float *M, int64_t ldM,
                            void *stream);
/* no-bases input term (graph.py:67-68): M[MPOS[c], 0:F] = addend[c, 0:F] + W[r_c*N + j_c, 0:F] */
int mrgcn_gather_rows_f32(const mrgcn_plan_t *plan, const float *W, int32_t F, const float *addend,
                          int64_t ldA, float *M, int64_t ldM, void *stream);
/* Basis contraction of weight_F (graph.py:83-85): W[r, :] = sum_b comp[r, b] * V[b, :] with V = weight_F viewed as
 * (B, X = in * out), W = (R, X) — and its backward: dV[b, :] = sum_r comp[r, b] * dW[r, :], dcomp[r, b] =
 * <dW[r, :], V[b, :]> (either output nullable).  One launch each; fixed summation order. */
int mrgcn_basis_contract_f32(const float *comp, const float *V, int32_t R, int32_t B, int64_t X, float *W,
                             void *stream);
int mrgcn_basis_contract_bwd_f32(const float *comp, const float *V, const float *dW, int32_t R, int32_t B, int64_t X,
                                 float *dcomp, float *dV, void *stream);

/* relation transform — replaces einsum('ij,bjk->bik', X, W_F) + reshape of graph.py:93-94,
 * restricted to touched columns (f32 MFMA 16x16x4 when K <= 256 and F <= 64):
 *     Out[o(c), 0:F] = X[j_c, 0:K] . W[r_c, 0:K, 0:F]          X: [N, ldX], W: [R, K, F]
 * o(c) = c (plain compact order, e.g. to serve as `addend`) or MPOS[c] (`operand_order` != 0) */
int mrgcn_rel_transform_fwd_f32(const mrgcn_plan_t *plan, const float *X, int64_t ldX, int32_t K,
                                const float *W, int32_t F, float *Out, int64_t ldOut,
                                int32_t operand_order, void *stream);
/* bf16-operand forms of the three forward operand builders: M / Out is bf16 (raw uint16_t,
 * leading dimension in elements), every input stays fp32 and every sum is fp32, one rounding
 * at the store.  basis_mix: B <= 64. */
int mrgcn_basis_mix_fwd_bf16(const mrgcn_plan_t *plan, const float *V, const float *comp, int32_t B,
                             int32_t F, const float *addend, int64_t ldA, uint16_t *M, int64_t ldM,
                             void *stream);
int mrgcn_gather_rows_bf16(const mrgcn_plan_t *plan, const float *W, int32_t F, const float *addend,
                           int64_t ldA, uint16_t *M, int64_t ldM, void *stream);
int mrgcn_rel_transform_fwd_bf16(const mrgcn_plan_t *plan, const float *X, int64_t ldX, int32_t K,
                                 const float *W, int32_t F, uint16_t *Out, int64_t ldOut,
                                 int32_t operand_order, void *stream);

/* ---- the bf16 pipeline (BASELINE config 3; SURVEY 8d "dense operands in bf16 with fp32 accumulation") ---------
 * The reference computes graph.py:93-95 in fp32 and has no reduced precision anywhere; in this mode every ACTIVATION
 * the layer reads or writes in bulk is stored in bf16 — the layer input X (rows of ldX elements, ldX % 8 == 0, zeros
 * past K), the feature term's rows (`addend`), the compact operand M — while parameters, every accumulation, the
 * layer output Y and the whole backward's sums stay fp32.
 *   mrgcn_cast_rows_bf16          dst[row, 0:ldDst] = [ bf16(src[row, 0:K]) | 0 ]  (round to nearest even)
 *   mrgcn_rel_transform_fwd_xbf16 mrgcn_rel_transform_fwd_f32 on bf16 input rows, v_mfma_f32_16x16x32_bf16 (W is
 *                                 rounded to bf16 as it is staged): K <= 256, F <= ldOut <= 16; Out fp32 or bf16
 *   mrgcn_basis_mix_fwd_abf16     mrgcn_basis_mix_fwd_f32 / _bf16 with the addend rows in bf16 (ldA elements)
 *   mrgcn_support_rel_transform_bwd_xbf16   (below, with the gradient supports) */
int mrgcn_cast_rows_bf16(const float *src, int64_t ldSrc, int64_t rows, int32_t K, uint16_t *dst, int64_t ldDst,
                         void *stream);
int32_t mrgcn_rel_transform_xbf16_supported(const mrgcn_plan_t *plan, int32_t K, int32_t F, int64_t ldX,
                                            int64_t ldOut);
int mrgcn_rel_transform_fwd_xbf16(const mrgcn_plan_t *plan, const uint16_t *X, int64_t ldX, int32_t K,
                                  const float *W, int32_t F, void *Out, int64_t ldOut, int32_t operand_order,
                                  int32_t out_bf16, void *stream);
int mrgcn_basis_mix_fwd_abf16(const mrgcn_plan_t *plan, const float *V, const float *comp, int32_t B, int32_t F,
                              const uint16_t *addend, int64_t ldA, void *M, int64_t ldM, int32_t out_bf16,
                              void *stream);


/* ---- compact dense operand: backward (autograd of graph.py:69-72, :93-94) -------------
 *     dV[j, b, :]  = sum_{c in node j} comp[r_c, b] * dM[c, :]          (node-major, like V)
 *     dcomp[r, b]  = sum_{c: r_c = r} <dM[c, :], V[j_c, b, :]>           (zeroed inside)
 * col_live (nullable): the liveness byte per compact column that mrgcn_spmm_transposed_live_f32 wrote;
 *     rows of dM flagged 0 are never read — they may be unwritten — and where a kernel has to read every
 *     row they are overwritten with zeros first (hence the non-const dM).  NULL: every column counts.
 * node_cur (nullable): ROW-SPARSE gradient.  A node without a live compact column has a zero gradient
 *     block; with node_cur given such blocks of dV are left UNWRITTEN (and their V blocks unread) and
 *     node_cur[j] = 1 / 0 says which nodes were written — only mrgcn_adam_step_rows_f32 may consume such a
 *     gradient.  NULL: every block of dV is written (zeros where there is no gradient).
 * dV_sumsq (nullable): *dV_sumsq += ||dV||^2 (device double), for the global clip norm.
 * dV may be NULL when node_cur and dV_sumsq are given (norm-only pass in front of mrgcn_adam_step_rows_fused_f32). */
int mrgcn_basis_mix_bwd_f32(const mrgcn_plan_t *plan, float *dM, int64_t ldM, const uint8_t *col_live,
                            const float *V, const float *comp, int32_t B, int32_t F, float *dV,
                            uint8_t *node_cur, float *dcomp, double *dV_sumsq, void *stream);

/* The same two gradients for a WIDE featureless layer with few bases (B <= 4, 16 < F <= 256, F % 4 == 0: the
 * link-prediction encoder of configs/fb15k-237.toml, N x 200 with 2 bases) straight from the layer's output gradient,
 * without the compact operand dM (one 4 F-byte row per touched column, written once and read twice by the pair
 * mrgcn_spmm_f32(TRANSPOSED) + mrgcn_basis_mix_bwd_f32):
 *     dV[j][b][:] = sum_{entries e = (i, (j, r), a)} comp[r][b] * a * dY[i][:]     (every block written)
 *     dcomp[r][b] = sum_{entries of relation r} a * <dY[i][:], V[j][b][:]>
 * over the plan's CSC entries, which are contiguous per source node.  The caller cuts the nodes' entry ranges into
 * UNITS of a few hundred entries (unit_node / unit_beg / unit_end, entry numbers of the CSC order; unit_multi[u] != 0
 * when its node has several units: those add with float atomics, the others store) and provides `erel`, the relation
 * of every CSC entry (mrgcn_plan_entry_relations, once per plan).  dY: [num_rows, ldY], 16-byte aligned rows. */
int mrgcn_plan_entry_relations(const mrgcn_plan_t *plan, int32_t *erel, void *stream);
int32_t mrgcn_wide_input_bwd_supported(const mrgcn_plan_t *plan, int32_t B, int32_t F);
int mrgcn_wide_input_bwd_f32(const mrgcn_plan_t *plan, const int32_t *erel, const int32_t *unit_node,
                             const int32_t *unit_beg, const int32_t *unit_end, const uint8_t *unit_multi,
                             int64_t n_units, const float *dY, int64_t ldY, const float *V, const float *comp,
                             int32_t B, int32_t F, float *dV, float *dcomp, void *stream);
/* The same, BITWISE REPRODUCIBLE (no float atomic): a unit of a node with several units writes its partial sums to
 * slot unit_slot[u] of the workspace (-1: the node's only unit, stored straight into dV) and a second launch adds a
 * node's slots in unit order (hub_node / hub_ptr: the nodes with several units and their slot ranges); dcomp is summed
 * per wave, per block and over the blocks in fixed orders.  What lets the units be small (64 entries: hub nodes stop
 * being the tail of the launch) without the atomics' reordering of cancelling sums. */
int64_t mrgcn_wide_input_bwd_det_workspace(const mrgcn_plan_t *plan, int64_t n_slots, int32_t B, int32_t F);
int mrgcn_wide_input_bwd_det_f32(const mrgcn_plan_t *plan, const int32_t *erel, const int32_t *unit_node,
                                 const int32_t *unit_beg, const int32_t *unit_end, const int32_t *unit_slot,
                                 int64_t n_units, const int32_t *hub_node, const int32_t *hub_ptr, int64_t n_hubs,
                                 int64_t n_slots, const float *dY, int64_t ldY, const float *V, const float *comp,
                                 int32_t B, int32_t F, float *dV, float *dcomp, float *workspace,
                                 int64_t workspace_floats, void *stream);

/*     dX[j, 0:K]  = sum_{c in node j} W[r_c] . dM[c, :]     (nullable; every row written)
 *     dW[r, :, :] = sum_{c: r_c = r} X[j_c, :]^T dM[c, :]    (nullable; zeroed inside)
 * `workspace` (nullable; size from mrgcn_rel_transform_bwd_workspace) lets dX run on the matrix
 * cores (per-column products, then a segmented sum per node) and dW reduce per-chunk partial
 * slabs instead of contended global atomics.                                              */
int64_t mrgcn_rel_transform_bwd_workspace(const mrgcn_plan_t *plan, int32_t K, int32_t F,
                                          int32_t need_dX, int32_t need_dW); /* floats */
int mrgcn_rel_transform_bwd_f32(const mrgcn_plan_t *plan, const float *dM, int64_t ldM,
                                const float *X, int64_t ldX, int32_t K, const float *W, int32_t F,
                                float *dX, int64_t lddX, float *dW, float *workspace,
                                int64_t workspace_floats, void *stream);
/* Same with a liveness byte per compact column (`col_live[c]` = 0: row c of dM is all zeros;
 * nullable = every column live).  In a semi-supervised epoch only the columns that feed a row
 * within reach of a labelled node carry gradient (autograd of graph.py:93-95 multiplies the
 * zeros like everything else); dead columns add exact zeros to dW and dX and are skipped.  Results
 * equal mrgcn_rel_transform_bwd_f32's.  Rows of dM flagged 0 are never read (they may be unwritten);
 * where a fallback kernel has to read every row they are overwritten with zeros first.  `col_live` comes from mrgcn_spmm_transposed_live_f32
 * (or mrgcn_rows_nonzero_f32(dM)). */
int mrgcn_rel_transform_bwd_live_f32(const mrgcn_plan_t *plan, float *dM, int64_t ldM,
                                     const uint8_t *col_live, const float *X, int64_t ldX,
                                     int32_t K, const float *W, int32_t F, float *dX, int64_t lddX,
                                     float *dW, float *workspace, int64_t workspace_floats,
                                     void *stream);
/* mrgcn_rel_transform_bwd_live_f32 with the layer-input gradient finished in the same pass (K <= 16, matrix-core
 * path: ask mrgcn_rel_transform_bwd_masked_supported): `relu_mask_from_x` != 0 multiplies dX by the mask X > 0 — X
 * is then the output of a ReLU (a hidden layer's input, rgcn.py:86-87), and X > 0 is that ReLU's own mask, so the
 * layer that produced X has nothing left to mask; `row_live_out` (nullable, one byte per node) receives "this row
 * of dX holds anything but zeros" — the flags mrgcn_spmm_transposed_live_flagged_f32 takes; `node_live` (nullable;
 * with `col_live`): the per-node flags that call wrote next to `col_live`, read instead of every node's column flags. */
int32_t mrgcn_rel_transform_bwd_masked_supported(const mrgcn_plan_t *plan, int32_t K, int32_t F,
                                                 int64_t workspace_floats);
int mrgcn_rel_transform_bwd_masked_f32(const mrgcn_plan_t *plan, float *dM, int64_t ldM, const uint8_t *col_live,
                                       const float *X, int64_t ldX, int32_t K, const float *W, int32_t F,
                                       float *dX, int64_t lddX, float *dW, float *workspace,
                                       int64_t workspace_floats, int32_t relu_mask_from_x,
                                       uint8_t *row_live_out, const uint8_t *node_live, void *stream);
/* Y[c, 0:F] = sum_i A'[i, c] * D[i, 0:F] — mrgcn_spmm_f32 on MRGCN_VIEW_TRANSPOSED (the autograd
 * of torch.mm(A, .), graph.py:75,:95) for an operand whose rows are mostly zeros, as the output
 * gradient of a layer is when few nodes are labelled: rows of D that hold only zeros are not
 * gathered (their products are zeros; rows of up to 32 entries are bitwise those of mrgcn_spmm_f32).
 * `scratch` (mrgcn_spmm_transposed_live_scratch(plan) bytes, 16-byte aligned) holds the liveness
 * of the rows of D; `col_live` ([ncols] bytes) receives 0 for rows of Y that are certainly all
 * zeros, 1 otherwise — the input of mrgcn_rel_transform_bwd_live_f32.  `live_rows` (nullable,
 * device) receives the number of rows of D that are not all zeros (-1 when F > 16, where the
 * general product runs): with more than about a quarter of the rows live the general
 * mrgcn_spmm_f32 is the faster call (AM shape: 211 us at 0.3 % live rows, 467 us general,
 * 860 us with every row live), so callers keep the last count and choose.
 * `write_dead_rows` = 0 leaves the rows of Y whose flag is 0 UNWRITTEN (their zeros are 90 % of this
 * call's stores): only for consumers that take `col_live` — mrgcn_basis_mix_bwd_f32 and
 * mrgcn_rel_transform_bwd_live_f32. */
int64_t mrgcn_spmm_transposed_live_scratch(const mrgcn_plan_t *plan); /* bytes */
int mrgcn_spmm_transposed_live_f32(const mrgcn_plan_t *plan, const float *D, int64_t ldD, int32_t F,
                                   float *Y, int64_t ldY, uint8_t *scratch, uint8_t *col_live,
                                   int32_t *live_rows, int32_t write_dead_rows, void *stream);
/* The same with the row flags of D given (`row_flags`, nullable: one byte per row of D, 0 = the row is all
 * zeros — written by the producer of D, mrgcn_rel_transform_bwd_masked_f32): D is not scanned for them.
 * `node_live` (nullable, one byte per source node) receives "this node has a live column" — what
 * mrgcn_rel_transform_bwd_masked_f32 takes to skip the nodes without any. */
int mrgcn_spmm_transposed_live_flagged_f32(const mrgcn_plan_t *plan, const float *D, int64_t ldD, int32_t F,
                                           float *Y, int64_t ldY, uint8_t *scratch, uint8_t *col_live,
                                           int32_t *live_rows, int32_t write_dead_rows,
                                           const uint8_t *row_flags, uint8_t *node_live, void *stream);
/* flags[i] = 1 when X[i, 0:F] holds anything but (+-)0 — NaN counts — else 0. */
int mrgcn_rows_nonzero_f32(const float *X, int64_t ld, int32_t F, int64_t nrows, uint8_t *flags,
                           void *stream);

/* ---- gradient support: the backward of a semi-supervised epoch on dense index spaces -------------------------
 * The reference's autograd multiplies the zeros of a layer's output gradient like everything else
 * (node_classification.py:439-444 takes the loss over the labelled rows Y_hat[idx] only; graph.py:75,:95 then run
 * dense).  Which rows of that gradient CAN hold anything follows from the label set and the graph alone: the
 * labelled rows at the last layer, and below it the source nodes of the columns those rows read.  A support fixes
 * that knowledge once per (plan, row set): `row_flags` (device, one byte per output row: 1 = the row of dY may be
 * non-zero) -> the LIVE compact columns (touched by a flagged row), numbered 0..L-1 in (node, relation) order, the
 * entries of those columns that sit in flagged rows, the nodes that own a live column (NODE_FLAGS: the row set of
 * the layer below) and the live columns in the plan's relation-major orders.  The calls below then take dM and
 * every per-column product as [L, ld] arrays indexed by live number: nothing is marked, scanned or skipped per epoch.
 * Results equal those of the plan-level calls with the matching liveness flags (dead entries add exact zeros).
 * A support is immutable after creation and refers to its plan (destroy the support first).  create / destroy
 * allocate and synchronise; the compute calls are stream ordered and capturable. */
typedef struct mrgcn_support mrgcn_support_t;
typedef struct mrgcn_support_info {
  int64_t live_cols;    /* L                                             */
  int64_t live_entries; /* entries of live columns inside flagged rows   */
  int64_t live_nodes;   /* nodes that own a live column                  */
  int64_t device_bytes;
  int64_t chunks_wide, chunks_narrow; /* relation-major chunks of the two transform orders */
  int64_t flagged_rows; /* NR (MRGCN_SUPPORT_FORWARD supports; -1 otherwise) */
} mrgcn_support_info_t;
enum mrgcn_support_array_id {
  MRGCN_SUP_COL_FLAGS = 0,  /* uint8 [ncols]     1 = live                                   */
  MRGCN_SUP_NODE_FLAGS = 1, /* uint8 [num_nodes] 1 = owns a live column                     */
  MRGCN_SUP_LCOL = 2,       /* int32 [L]         compact column of each live column, rising  */
  MRGCN_SUP_LREL = 3,       /* int32 [L]         its relation                                */
  MRGCN_SUP_NLPTR = 4,      /* int32 [num_nodes+1] node -> range of live numbers             */
  MRGCN_SUP_LPTR = 5,       /* int32 [L+1]       entry range of each live column             */
  MRGCN_SUP_LROW = 6,       /* int32 [E]         output row of each kept entry               */
  MRGCN_SUP_LVAL = 7,       /* float [E]         its value                                   */
  MRGCN_SUP_LNODE = 8,      /* int32 [live_nodes] nodes that own a live column, rising       */
  MRGCN_SUP_LPERM = 9,      /* int32 [L]         live numbers in (node band, relation, node) order */
  /* MRGCN_SUPPORT_FORWARD supports only (count 0 otherwise): */
  MRGCN_SUP_FROW = 10,      /* int32 [NR]        flagged rows, rising                        */
  MRGCN_SUP_FPTR = 11,      /* int32 [NR+1]      entry range of each flagged row             */
  MRGCN_SUP_FCOL = 12,      /* int32 [E]         live number of each entry's column (the plan's row order) */
  MRGCN_SUP_FVAL = 13,      /* float [E]         its stored value                            */
  MRGCN_SUP_LNODE_ORD = 14, /* int32 [L]         rank of the live column's node in LNODE     */
  MRGCN_SUP_ROWRANK = 15    /* int32 [num_rows]  rank in FROW, -1 for rows outside the set   */
};
int mrgcn_support_create(mrgcn_support_t **support, const mrgcn_plan_t *plan, const uint8_t *row_flags,
                         void *stream);
/* flags: MRGCN_SUPPORT_FORWARD also keeps the flagged rows as a CSR over (flagged row, live column), the ranks of
 * rows and nodes and the relation-major orders by live-node rank: what the masked-pass calls below need. */
#define MRGCN_SUPPORT_FORWARD 1u
int mrgcn_support_create_ex(mrgcn_support_t **support, const mrgcn_plan_t *plan, const uint8_t *row_flags,
                            uint32_t flags, void *stream);
/* n supports at once: supports[0] on row_flags, supports[i+1] on NODE_FLAGS of supports[i] (the samples of the n layers
 * of a mini-batch, batch.py:222-231).  One host wait for the whole chain.  rows = nodes (the stacked adjacency). */
int mrgcn_support_create_chain(mrgcn_support_t **supports, int32_t n, const mrgcn_plan_t *plan,
                               const uint8_t *row_flags, uint32_t flags, void *stream);
int mrgcn_support_destroy(mrgcn_support_t *support);
/* The same without waiting for the device, for a support that lived for one step (a mini-batch): the caller states
 * that every call that used the support was submitted to the stream it was created on (or is ordered before that
 * stream's tail); its memory is handed to later work on that stream. */
int mrgcn_support_destroy_ordered(mrgcn_support_t *support);
int mrgcn_support_info(const mrgcn_support_t *support, mrgcn_support_info_t *h_info);
int mrgcn_support_array(const mrgcn_support_t *support, int32_t which, const void **d_ptr, int64_t *h_count);
/* dM[k, 0:F] = sum over the kept entries e of live column k of val[e] * dY[row[e], 0:F] — the TRANSPOSED product
 * (autograd of torch.mm(A, .), graph.py:75,:95) restricted to the support; rows of dY outside the row set are never
 * read (they may be unwritten). */
int mrgcn_support_spmm_t_f32(const mrgcn_support_t *support, const float *dY, int64_t ldY, int32_t F, float *dM,
                             int64_t ldM, void *stream);
/* mrgcn_basis_mix_bwd_f32 on the support (dM: [L, ldM] by live number).
 *   dV == NULL : the norm-only pass in front of mrgcn_support_adam_rows_fused_f32: dcomp and *dV_sumsq (both
 *                WRITTEN, not accumulated: no zeroing by the caller, fixed summation order) — one pass over the
 *                live nodes' V blocks that stores each live column's B products, then a relation-major sum of
 *                those rows: no atomics anywhere.  `workspace`: mrgcn_support_mix_bwd_workspace(support, B) floats.
 *   dV != NULL : the gradient itself; `dense` != 0 writes every block (zeros for nodes outside the support),
 *                dense == 0 only the blocks of the support's nodes (row-sparse: mrgcn_adam_step_rows_f32 with
 *                NODE_FLAGS as row_cur).  dcomp written; *dV_sumsq (nullable) ACCUMULATED into. */
int64_t mrgcn_support_mix_bwd_workspace(const mrgcn_support_t *support, int32_t B); /* floats */
int mrgcn_support_mix_bwd_f32(const mrgcn_support_t *support, const float *dM, int64_t ldM, const float *V,
                              const float *comp, int32_t B, int32_t F, float *dV, int32_t dense, float *dcomp,
                              double *dV_sumsq, float *workspace, int64_t workspace_floats, void *stream);
/* mrgcn_adam_step_rows_fused_f32 on the support: row_cur = NODE_FLAGS.  The support's nodes are walked as a list
 * (two node blocks in flight per wave, nontemporal p / m / v); `ever_outside` != 0 adds the pass over the nodes that
 * are flagged in row_ever without being nodes of this support (moments from steps on another row set: they decay,
 * their parameters drift on, as torch.optim.Adam does for a zero gradient).  0: the caller states there are none
 * (every step since row_ever was zeroed ran on this support) and that pass is not launched. */
int mrgcn_support_adam_rows_fused_f32(const mrgcn_support_t *support, const float *dM, int64_t ldM, const float *comp,
                                      int32_t B, int32_t F, float *param, float *exp_avg, float *exp_avg_sq,
                                      uint8_t *row_ever, float lr, float beta1, float beta2, float eps, int64_t step,
                                      const float *bc_dev, const float *grad_scale, int32_t ever_outside,
                                      void *stream);
/* mrgcn_rel_transform_bwd_masked_f32 on the support: dW (nullable; written whole) and dX (nullable; every row
 * written, zeros outside NODE_FLAGS; `relu_mask_from_x` as there).  workspace:
 * mrgcn_support_rel_transform_bwd_workspace floats. */
int64_t mrgcn_support_rel_transform_bwd_workspace(const mrgcn_support_t *support, int32_t K, int32_t F,
                                                  int32_t need_dX, int32_t need_dW);
int mrgcn_support_rel_transform_bwd_f32(const mrgcn_support_t *support, const float *dM, int64_t ldM, const float *X,
                                        int64_t ldX, int32_t K, const float *W, int32_t F, float *dX, int64_t lddX,
                                        float *dW, float *workspace, int64_t workspace_floats,
                                        int32_t relu_mask_from_x, void *stream);
/* the same with the layer input X in bf16 rows (the bf16 pipeline; ldX in elements): dW's products and sums are fp32
 * on the widened elements; dX (= dM . W^T summed per node) never reads X */
int mrgcn_support_rel_transform_bwd_xbf16(const mrgcn_support_t *support, const float *dM, int64_t ldM,
                                          const uint16_t *X, int64_t ldX, int32_t K, const float *W, int32_t F,
                                          float *dX, int64_t lddX, float *dW, float *workspace,
                                          int64_t workspace_floats, void *stream);
/* dlogits[idx[i], 0:C] = *g * drows[i, 0:C] for the n labelled rows ONLY (idx without repeats): the rest of dlogits
 * is not touched — for a consumer that reads the labelled rows only (mrgcn_support_spmm_t_f32).  The 73 MB zero fill
 * of mrgcn_softmax_xent_bwd_f32 (AM shape) goes away. */
int mrgcn_softmax_xent_bwd_rows_f32(const float *drows, const int64_t *idx, int64_t n, int32_t C, const float *g,
                                    float *dlogits, int64_t ldd, void *stream);

/* ---- a mini-batch layer as a masked pass over the full graph's plan ------------------------
 * Replaces, for a re-sampled batch, the slice tensors and the per-batch plans of the frontier path below
 * (mrgcn/data/batch.py:185-263, mrgcn/models/rgcn.py:91-128, mrgcn/layers/graph.py:62-102 with A_idx): the sample of
 * a layer is the row-flag array of a MRGCN_SUPPORT_FORWARD support, its neighbours are the support's live nodes
 * (LNODE = getNeighboursSparse(A, sample)), and every array is compact: activations [NR, F] by FROW rank, the
 * neighbours' features / embeddings [live_nodes, K] by LNODE rank, per-column operands [L, ld] by live number.
 * The sample of the layer below is NODE_FLAGS of this one.
 *   spmm_fwd       Y[q] = act(bias + sum_e v[e] D[FCOL[e]]) over row FROW[q]'s entries; v = the stored values
 *                  (use_values != 0: the input term, graph.py:75) or all ones (the feature term on the sliced
 *                  adjacency, whose values sliceSparseCOO drops: batch.py:258-270).
 *   spmm_t_compact dM[k] = sum over live column k's entries of v[e] dY[ROWRANK[row[e]]]  (dY: [NR, ldY])
 *   mix_fwd        M[k] = sum_b comp[LREL[k], b] V[node of k, b]        (graph.py:66-74 on the live columns)
 *   rel_transform_fwd / _bwd_compact   T[k] = X[LNODE_ORD[k]] . W[LREL[k]] (graph.py:83-95) and its backward: dW
 *                  (nullable, written whole), dX [live_nodes, lddX] (nullable, every row written;
 *                  `relu_mask_from_x` as mrgcn_rel_transform_bwd_masked_f32); workspace:
 *                  mrgcn_support_rel_transform_bwd_workspace floats.  x_by_node != 0: X is the whole feature
 *                  matrix, one row per NODE (the rows X[LNODE] are picked inside the transform instead of by a
 *                  gather in front of it; dX stays by LNODE rank).  Shapes: mrgcn_support_rel_transform_supported
 *                  (need_dX: the input gradient is wanted too — inputs of up to 64 floats per row).
 * The weight_I gradient of the input term is mrgcn_support_mix_bwd_f32 / mrgcn_support_adam_rows_fused_f32 above
 * (dM by live number, V blocks by node id).  One product per support in flight. */
int mrgcn_support_spmm_fwd_f32(const mrgcn_support_t *support, int32_t use_values, const float *D, int64_t ldD,
                               int32_t F, float *Y, int64_t ldY, const float *bias, int32_t relu, void *stream);
int mrgcn_support_spmm_t_compact_f32(const mrgcn_support_t *support, int32_t use_values, const float *dY, int64_t ldY,
                                     int32_t F, float *dM, int64_t ldM, void *stream);
int mrgcn_support_mix_fwd_f32(const mrgcn_support_t *support, const float *V, const float *comp, int32_t B, int32_t F,
                              float *M, int64_t ldM, void *stream);
/* weight_I WITHOUT bases (the reference's literal (R*N) x F table, graph.py:72-74) on the live columns:
 * scatter == 0: M[k] = table[LREL[k] * N + node(k)];  scatter != 0: table is zeroed whole, then those rows = M[k]. */
int mrgcn_support_literal_rows_f32(const mrgcn_support_t *support, int32_t scatter, float *table, int32_t F, float *M,
                                   int64_t ldM, void *stream);
int32_t mrgcn_support_rel_transform_supported(const mrgcn_support_t *support, int32_t K, int32_t F, int32_t need_dX);
int mrgcn_support_rel_transform_fwd_f32(const mrgcn_support_t *support, const float *X, int64_t ldX, int32_t x_by_node,
                                        int32_t K, const float *W, int32_t F, float *T, int64_t ldT, void *stream);
int mrgcn_support_rel_transform_bwd_compact_f32(const mrgcn_support_t *support, const float *dM, int64_t ldM,
                                                const float *X, int64_t ldX, int32_t x_by_node, int32_t K,
                                                const float *W, int32_t F,
                                                float *dX, int64_t lddX, float *dW, float *workspace,
                                                int64_t workspace_floats, int32_t relu_mask_from_x, void *stream);

/* ---- epoch kernels around the layers ----------------------------------------------
 * out = dY * (Y > 0): backward of the nn.ReLU between layers (rgcn.py:86-87) */
int mrgcn_relu_bwd_f32(const float *dY, const float *Y, int64_t n, float *out, void *stream);
/* dst[n_rows, F] (contiguous) = the rows of `src` scattered to their places, zeros everywhere else, in ONE pass:
 * dst[sorted_rows[k], :] = src[perm[k], 0:F] (sorted_rows rising, without repeats).  The dense gradient of the literal
 * (R*N) x out `weight_I` of a layer without bases (autograd of graph.py:75): sorted_rows = the plan's touched literal
 * columns in rising order, perm = their compact ids, src = the compact gradient rows. */
int mrgcn_scatter_rows_zero_fill_f32(const int32_t *sorted_rows, const int32_t *perm, int64_t n_touched,
                                     const float *src, int64_t ldS, int32_t F, float *dst, int64_t n_rows, void *stream);
/* out[0:F] = the column sums of X[M, F] (row stride ld) over the rows with row_flags[m] != 0 (NULL: every row; rows
 * flagged 0 are not read — they may be unwritten).  The bias gradient `dY.sum(0)` of a narrow layer (autograd of
 * `AFW + self.b`, graph.py:98-101) in one pass, summed in a fixed order.  F <= 16; workspace of
 * mrgcn_colsum_rows_workspace(F) floats (-1: F outside the kernel's range). */
int64_t mrgcn_colsum_rows_workspace(int32_t F);
int mrgcn_colsum_rows_f32(const float *X, int64_t ld, int64_t M, int32_t F, const uint8_t *row_flags, float *out,
                          float *workspace, int64_t workspace_floats, void *stream);
/* Streaming yardsticks for measurement (bench.py: extra.device_copy_gbps_hip, extra.triad_gbps): dst = src as a plain
 * float4 copy, and a 3-read / 3-write elementwise pass with Adam's arithmetic (the memory shape of a dense optimizer
 * step).  n % 4 == 0, 16-byte aligned.  Nothing in the package calls them. */
int mrgcn_probe_copy_f32(const float *src, float *dst, int64_t n, void *stream);
int mrgcn_probe_triad_f32(float *p, float *m, float *v, int64_t n, void *stream);
/* the same two as PERSISTENT grids (2 048 resident blocks striding through the arrays): what a kernel with per-block
 * state reaches — 4.8 / 4.9 TB/s on the bench boxes against 6.25 / 6.0 for the one-shot grids above */
int mrgcn_probe_copy_persistent_f32(const float *src, float *dst, int64_t n, void *stream);
int mrgcn_probe_triad_persistent_f32(float *p, float *m, float *v, int64_t n, void *stream);
/* the same with row-strided operands: out[r, 0:F] = dY[r, 0:F] * (Y[r, 0:F] > 0) — the layer output Y of the
 * COMPACT product may live in a buffer with padded rows (MRGCN_SPMM_PAD_WRITABLE) */
int mrgcn_relu_bwd_rows_f32(const float *dY, int64_t ld_dY, const float *Y, int64_t ldY, int64_t rows, int32_t F,
                            float *out, int64_t ld_out, void *stream);
/* loss = mean_i CE(logits[idx[i]], target[i]); dlogits (nullable, [num_rows, ldd]) is zeroed
 * and receives d loss / d logits.  nn.CrossEntropyLoss over Y_hat[idx]
 * (node_classification.py:439-444) and its backward. */
int mrgcn_softmax_xent_f32(const float *logits, int64_t ld, int32_t C, const int64_t *idx,
                           const int64_t *target, int64_t n, float *loss, float *dlogits,
                           int64_t ldd, int64_t num_rows, void *stream);
/* The same loss with its gradient kept compact — drows (nullable, [n, C]) = d loss / d logits[idx[i], :] — and the
 * backward that forms the dense gradient from it: dlogits = 0, dlogits[idx[i], :] += *g * drows[i, :] (g: device
 * float, nullable = 1), row_flags (nullable, [num_rows] bytes) = 1 at the labelled rows, 0 elsewhere (the flags
 * mrgcn_spmm_transposed_live_flagged_f32 takes).  No dense pass besides the zero fill. */
int mrgcn_softmax_xent_rows_f32(const float *logits, int64_t ld, int32_t C, const int64_t *idx, const int64_t *target,
                                int64_t n, float *loss, float *drows, void *stream);
int mrgcn_softmax_xent_bwd_f32(const float *drows, const int64_t *idx, int64_t n, int32_t C, const float *g,
                               float *dlogits, int64_t ldd, int64_t num_rows, uint8_t *row_flags, void *stream);
/* *accum += sum(x^2)  (accum is a device double, zeroed by the caller once per step) */
int mrgcn_sumsq_accum_f32(const float *x, int64_t n, double *accum, void *stream);
/* clip_grad_norm_(…, max_norm) (node_classification.py:192): norm = sqrt(*sumsq),
 * coef = min(1, max_norm / (norm + 1e-6)); both stay on the device */
int mrgcn_clip_coef_f32(const double *sumsq, float max_norm, float *coef, float *norm, void *stream);
/* The clip of the SMALL dense parameters in one launch (<= 16 tensors; host arrays of device pointers / sizes):
 * sum of squares of every gradient, plus `extra` device doubles (squared norms that arrived from elsewhere: the
 * row-sparse node table's), -> *sumsq_out (nullable), the coefficient min(1, max_norm / (norm + 1e-6)) -> *coef
 * (nullable; max_norm <= 0: 1), the norm -> *norm (nullable) and — with `step_dev` — the device step counter and
 * bias corrections of mrgcn_adam_bias_f32.  `accum` (double) and `ticket` (uint32) are scratch words that must be
 * zero at the first call; the kernel leaves them zero.  Replaces n mrgcn_sumsq_accum_f32 + mrgcn_clip_coef_f32 +
 * mrgcn_adam_bias_f32 launches (each ~6 us inside a replayed hipGraph). */
/* *accum += the squared norms of <= 16 tensors in one launch (the launches in front of mrgcn_sumsq_clip_multi_f32 for a
 * model with more small tensors than one call takes; `accum` is the same scratch word, which that call reads and clears) */
int mrgcn_sumsq_accum_multi_f32(int32_t n_tensors, const float *const *grads, const int64_t *numel, double *accum,
                                void *stream);
int mrgcn_sumsq_clip_multi_f32(int32_t n_tensors, const float *const *grads, const int64_t *numel, int32_t n_extra,
                               const double *const *extra, double *accum, uint32_t *ticket, float max_norm,
                               double *sumsq_out, float *coef, float *norm, int64_t *step_dev, float beta1,
                               float beta2, float *bc_dev, void *stream);
/* mrgcn_adam_step_f32 / _dev_f32 for <= 16 small tensors in one launch (per-tensor lr / weight_decay; bc_dev
 * nullable: host-side bias corrections from `step`). */
int mrgcn_adam_step_multi_f32(int32_t n_tensors, float *const *params, const float *const *grads,
                              float *const *exp_avg, float *const *exp_avg_sq, const int64_t *numel, const float *lr,
                              const float *weight_decay, float beta1, float beta2, float eps, int64_t step,
                              const float *bc_dev, const float *grad_scale, void *stream);
/* torch.optim.Adam step (node_classification.py:35-37, :193) on one tensor; the gradient is
 * multiplied by *grad_scale (device float, nullable) first, i.e. the clip is folded in. */
int mrgcn_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                        int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int64_t step, const float *grad_scale, void *stream);

/* ---- link-prediction decoder (DistMult) ---------------------
 * triples: device int64 [n, 3] rows (s, p, o) indexing E (node embeddings = the encoder's
 * output) and Rel (relation embeddings, rgcn.py:40-47 `relations`). */
/* scores[t] = sum_h E[s,h] Rel[p,h] E[o,h] — score_distmult_bc with the 1-D index tensors
 * train_model passes (tasks/link_prediction.py:266-275, :645-665) */
int mrgcn_distmult_score_f32(const float *E, int64_t ldE, const float *Rel, int64_t ldR, int32_t H,
                             const int64_t *triples, int64_t n, float *scores, void *stream);
/* its backward: dE / dRel (nullable) are ACCUMULATED into (zero them first) */
int mrgcn_distmult_score_bwd_f32(const float *E, int64_t ldE, const float *Rel, int64_t ldR,
                                 int32_t H, const int64_t *triples, int64_t n, const float *dscores,
                                 float *dE, int64_t lddE, float *dRel, int64_t lddR, void *stream);
/* the same gradients from triples visited in sorted order: order_s / order_p / order_o are device
 * int64 permutations of 0..n-1 that sort the triples by subject / predicate / object (any order
 * within ties).  Runs of equal targets are summed in registers, so the float atomics of the
 * scatter form (which collide on the few hundred relation rows) shrink by the run length.
 * dE needs order_s and order_o, dRel needs order_p; both are ACCUMULATED into. */
/* ... and those three permutations (any may be NULL): stable radix sorts of the subject / predicate / object columns
 * over the bits ids below num_nodes / num_relations can have (replaces three torch.argsort calls of the run loop's
 * autograd: a merge sort of ~30 launches per column).  workspace: mrgcn_distmult_orders_workspace(n) bytes. */
/* out[i] = pi(i) for i < k, pi a bijection of [0, n) keyed by the device word *seed_dev (a Feistel network with cycle
 * walking): k distinct pseudo-random indices below n in one launch — the corrupted facts of the device-side negative
 * sampler (tasks/link_prediction.py sample_negatives_device; torch.randperm(n)[:k] sorts n keys). */
int mrgcn_random_subset_i64(int64_t n, int64_t k, const int64_t *seed_dev, int64_t *out, void *stream);
/* The in-batch corruption of train_model (tasks/link_prediction.py:239-263) in one launch: out[k] (k < ncorrupt), rows
 * (s, p, o), is a copy of fact pi(k) — ncorrupt DISTINCT facts of the n given, pi keyed by *seed_dev as above — with
 * its head (k < nhead) or its tail replaced by nodes[u], u uniform in [0, n_nodes) (`nodes`: the batch's node set). */
int mrgcn_corrupt_triples_i64(const int64_t *facts, int64_t n, const int64_t *nodes, int64_t n_nodes,
                              const int64_t *seed_dev, int64_t ncorrupt, int64_t nhead, int64_t *out, void *stream);
int64_t mrgcn_distmult_orders_workspace(int64_t n);
int mrgcn_distmult_orders(const int64_t *triples, int64_t n, int64_t num_nodes, int64_t num_relations,
                          int64_t *order_s, int64_t *order_p, int64_t *order_o, void *workspace,
                          int64_t workspace_bytes, void *stream);
/* the same three permutations for a SMALL triple set (the corrupted facts drawn anew every epoch) by a counting sort:
 * a histogram pass, one scan block per column, a fill pass — four launches where the radix sort takes ~25
 * launch-bound ones.  Ties in no particular order.  workspace: mrgcn_distmult_orders_counting_workspace bytes. */
int64_t mrgcn_distmult_orders_counting_workspace(int64_t num_nodes, int64_t num_relations);
int mrgcn_distmult_orders_counting(const int64_t *triples, int64_t n, int64_t num_nodes, int64_t num_relations,
                                   int64_t *order_s, int64_t *order_p, int64_t *order_o, void *workspace,
                                   int64_t workspace_bytes, void *stream);
int mrgcn_distmult_score_bwd_sorted_f32(const float *E, int64_t ldE, const float *Rel, int64_t ldR,
                                        int32_t H, const int64_t *triples, int64_t n,
                                        const float *dscores, const int64_t *order_s,
                                        const int64_t *order_p, const int64_t *order_o, float *dE,
                                        int64_t lddE, float *dRel, int64_t lddR, void *stream);
/* nn.BCEWithLogitsLoss() (link_prediction.py:57, :550-554): *loss = mean; dx (nullable) its
 * gradient w.r.t. x */
int mrgcn_bce_logits_f32(const float *x, const float *y, int64_t n, float *loss, float *dx,
                         void *stream);
/* compute_ranks_fast (link_prediction.py:593-643): ranks[0:nf] tail-corruption, ranks[nf:2nf]
 * head-corruption rank of every fact against all num_nodes candidates;
 * rank = #greater + round_half_even((#ties-1)/2) + 1.  Filtered ranks: per fact a sorted list
 * of candidate nodes to ignore (the -inf entries of filter_scores_, :667-689) in CSR form,
 * tail_* for tail corruption and head_* for head corruption; all four NULL = raw ranks.
 * Scores are never stored.  workspace: device, >= mrgcn_distmult_ranks_workspace() bytes. */
int64_t mrgcn_distmult_ranks_workspace(int64_t num_nodes, int32_t H, int64_t num_facts);
int mrgcn_distmult_ranks(const float *E, int64_t ldE, int64_t num_nodes, const float *Rel,
                         int64_t ldR, int32_t H, const int64_t *triples, int64_t num_facts,
                         const int64_t *tail_ptr, const int32_t *tail_idx, const int64_t *head_ptr,
                         const int32_t *head_idx, void *workspace, int64_t workspace_bytes,
                         int64_t *ranks, void *stream);

/* Graph-capturable Adam (torch.optim.Adam(capturable=True) semantics): the step counter and the
 * bias corrections live in device memory, so a captured epoch replays with the right step.
 * mrgcn_adam_bias_f32: ++*step_dev; bc_dev[0] = 1 - beta1^step, bc_dev[1] = sqrt(1 - beta2^step).
 * mrgcn_adam_step_dev_f32: mrgcn_adam_step_f32 reading the corrections from bc_dev. */
/* mrgcn_adam_step_f32 / _dev_f32 (weight_decay = 0) on a parameter of `nrows` rows of `rowlen` floats
 * (the node-major basis table: nrows = N, rowlen = B*F; float4 accesses when rowlen % 4 == 0) whose gradient was produced
 * row-sparse (mrgcn_basis_mix_bwd_f32 with node_cur): rows with row_ever = 0 and row_cur = 0 are not
 * touched (g = m = v = 0: Adam's update is the identity), rows with row_ever = 1, row_cur = 0 are
 * updated with g = 0 without reading grad, rows with row_cur = 1 take grad and are marked in row_ever
 * (caller keeps row_ever across steps, zero-initialised).  `bc_dev` nullable (then `step` >= 1 is used). */
int mrgcn_adam_step_rows_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                             int64_t nrows, int32_t rowlen, const uint8_t *row_cur, uint8_t *row_ever,
                             float lr, float beta1, float beta2, float eps, int64_t step,
                             const float *bc_dev, const float *grad_scale, void *stream);
/* mrgcn_adam_step_f32 / _dev_f32 (weight_decay = 0) on the rows index[0 .. n_index) of a parameter of rows of `rowlen`
 * floats, the gradient of row index[c] in row c of a COMPACT gradient (`grad`: [n_index, ld_grad]); every other row is
 * left alone.  For the literal (R*N) x F operand of a featureless layer without bases (graph.py:69-75: weight_I is the
 * operand of torch.mm(A, .)): the autograd of that product (SparseAddmmBackward) puts gradient on the rows that are
 * columns of A only — the plan's compact columns (MRGCN_ARR_ULCOL), for every label set — so the other rows keep zero
 * moments and never move.  `index` holds distinct rows; rowlen, ld_grad multiples of 4; 16-byte aligned arrays. */
int mrgcn_adam_step_index_rows_f32(float *param, const float *grad, int64_t ld_grad, float *exp_avg,
                                   float *exp_avg_sq, const int32_t *index, int64_t n_index, int32_t rowlen, float lr,
                                   float beta1, float beta2, float eps, int64_t step, const float *bc_dev,
                                   const float *grad_scale, void *stream);
/* The same update with the gradient formed on the fly: the backward called mrgcn_basis_mix_bwd_f32 with dV = NULL
 * (flags, dcomp and ||dV||^2 only) and this call rebuilds every live node's block from its dM rows,
 *     dV[j][b][f] = sum_{live c of j} comp[r_c][b] * dM[c][f]        (the sums mrgcn_basis_mix_bwd_f32 forms, bit for bit),
 * inside the Adam pass: the gradient tensor of the node table is never written nor read.  `comp` must hold the
 * coefficients the backward saw (a snapshot when the optimizer updates them first).  Shapes:
 * mrgcn_adam_rows_fused_supported (B <= 64, F <= 16, B*F % 4 == 0, R*B*4 <= 64 KB). */
int32_t mrgcn_adam_rows_fused_supported(const mrgcn_plan_t *plan, int32_t B, int32_t F);
int mrgcn_adam_step_rows_fused_f32(const mrgcn_plan_t *plan, const float *dM, int64_t ldM, const uint8_t *col_live,
                                   const float *comp, int32_t B, int32_t F, float *param, float *exp_avg,
                                   float *exp_avg_sq, const uint8_t *row_cur, uint8_t *row_ever, float lr,
                                   float beta1, float beta2, float eps, int64_t step, const float *bc_dev,
                                   const float *grad_scale, void *stream);
int mrgcn_adam_bias_f32(int64_t *step_dev, float beta1, float beta2, float *bc_dev, void *stream);
int mrgcn_adam_step_dev_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                            int64_t n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, const float *bc_dev, const float *grad_scale,
                            void *stream);

/* ---- modality encoders: the dense step in front of the graph path ----------------------
 * Fused literal MLP + gate + masked scatter (mrgcn/models/perceptron.py:6-46 with p_dropout = 0 or in eval
 * mode; mrgcn/models/mrgcn.py:285-303):
 *     XF[rows[i], offset : offset + dims[L]] = gate[0] * MLP(X[i, 0:dims[0]]),   MLP = L x (Linear -> ReLU)
 * dims: HOST array of L + 1 widths (1..16); W / b: HOST arrays of L DEVICE pointers, W[l] is
 * [dims[l+1]][dims[l]] (nn.Linear layout), b[l] nullable; rows: nullable int64 [n] (NULL = row i).
 * The backward ACCUMULATES into dW[l] / db[l] / dgate (caller zeroes them). */
int32_t mrgcn_mlp_fused_supported(int32_t L, const int32_t *dims);
int mrgcn_mlp_gate_scatter_fwd_f32(int32_t L, const int32_t *dims, const float *const *W, const float *const *b,
                                   const float *X, int64_t ldx, int64_t n, const float *gate,
                                   const int64_t *rows, float *XF, int64_t ldxf, int32_t offset, void *stream);
int mrgcn_mlp_gate_scatter_bwd_f32(int32_t L, const int32_t *dims, const float *const *W, const float *const *b,
                                   const float *X, int64_t ldx, int64_t n, const float *gate,
                                   const int64_t *rows, const float *dXF, int64_t ldxf, int32_t offset,
                                   float *const *dW, float *const *db, float *dgate, void *stream);
/* Dense product on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32), the building block of the
 * `pre_fc -> ReLU -> fc` heads (mrgcn/models/imagecnn.py:31-41, transformer.py:29-38) and of the TCNN
 * (temporal_cnn.py:6-156):   C = epilogue(alpha * op(A) . op(B)),  epilogue: + bias[N], ReLU, * (mask > 0).
 *   amode 0: A[m*lda + k]   1: A[k*lda + m]   2: Conv1d im2col of x[b][ci][t] (m = b*Tout + t, k = ci*KW + kw)
 *         3: the transpose of mode 2 (m = ci*KW + kw, k = b*Tout + t)
 *   bmode 0: B[k*ldb + n]   1: B[n*ldb + k]   2: y[b][n][t] read as [k = b*Tout + t][n]
 *   cmode 0: C[m*ldc + n]   2: y[b][n][t], m = b*Tout + t
 *   conv_geom (HOST, modes 2 / 3): {Cin, Tin, KW, pad, Tout, Cout}.  mask: nullable, addressed like C.
 * Products with few output tiles and a long reduction (the convolutions' dW) split K over the grid: C (dense,
 * ldc == N) is zeroed and the partial tiles are added with float atomics — the result is exact to the last bits only
 * up to the order of those adds. */
int mrgcn_gemm_f32(int32_t amode, int32_t bmode, int32_t cmode, int32_t M, int32_t N, int32_t K, const float *A,
                   int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, const float *bias,
                   int32_t relu, const float *mask, float alpha, const int32_t *conv_geom, void *stream);
/* The same product on the bf16 matrix cores (the bf16 pipeline, BASELINE config 3): operands and result stay fp32 in
 * memory; tiles are rounded to bf16 (nearest even) as they are staged, v_mfma_f32_16x16x32_bf16, fp32 accumulation.
 * Shapes outside the tiled form's limits run the exact fp32 element-loader kernel. */
int mrgcn_gemm_bf16mm_f32(int32_t amode, int32_t bmode, int32_t cmode, int32_t M, int32_t N, int32_t K, const float *A,
                          int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, const float *bias,
                          int32_t relu, const float *mask, float alpha, const int32_t *conv_geom, void *stream);
/* out[n] = sum_m X[m*ld + n]  (bias gradients) */
int mrgcn_colsum_f32(const float *X, int64_t ld, int32_t M, int32_t N, float *out, void *stream);

/* The elementwise half of a TCNN block (temporal_cnn.py:6-156: Conv1d -> BatchNorm1d -> ReLU [-> MaxPool1d(k, k) |
 * AdaptiveMaxPool1d(n)]) in one pass over the convolution's output x [B][C][T]:
 *     y[b][c][to] = max_{t in window(to)} relu(gamma[c] (x[b][c][t] - mean[c]) / sqrt(var[c] + eps) + beta[c])
 * pool_kind 0: none (Tout = T), 1: MaxPool1d(pool_arg, stride pool_arg), 2: AdaptiveMaxPool1d(pool_arg).
 * training != 0: mean / var (biased) are the batch statistics, computed and written by the call; otherwise they are
 * read (running statistics).  argmax [B][C][Tout] (needed when pooled) keeps the winning position for the backward.
 * Backward: dz is a [B][C][T] workspace that only adaptive pooling uses (overlapping windows; NULL otherwise);
 * dgamma / dbeta are written; dx = d loss / d x.
 * workspace: mrgcn_bn_workspace_bytes(C) bytes (fp64 partial sums of the per-channel reductions; forward: training only). */
int32_t mrgcn_pool_out_len(int32_t pool_kind, int32_t pool_arg, int32_t T);
size_t mrgcn_bn_workspace_bytes(int32_t C);
int mrgcn_bn_relu_pool_fwd_f32(const float *x, int32_t B, int32_t C, int32_t T, const float *gamma,
                               const float *beta, float eps, int32_t training, float *mean, float *var,
                               int32_t pool_kind, int32_t pool_arg, float *y, int32_t *argmax, void *workspace,
                               void *stream);
/* nn.BatchNorm1d's running statistics after a training-mode forward over n = B * T values per channel:
 * running = (1 - momentum) running + momentum stat, the (biased) batch variance scaled by n / (n - 1). */
int mrgcn_bn_running_stats_f32(const float *mean, const float *var, int32_t C, int64_t n, float momentum,
                               float *running_mean, float *running_var, void *stream);
/* out[c] = sum over (b, t) of x[b][c][t] — the bias gradient of a Conv1d (torch: dy.sum(dim=(0, 2))); fp64 sums per
 * block, one float atomic per block into the zeroed out. */
int mrgcn_channel_sum_f32(const float *x, int32_t B, int32_t C, int32_t T, float *out, void *stream);
int mrgcn_bn_relu_pool_bwd_f32(const float *x, const float *y, const float *dy, const int32_t *argmax, int32_t B,
                               int32_t C, int32_t T, const float *gamma, const float *mean, const float *var,
                               float eps, int32_t training, int32_t pool_kind, int32_t pool_arg, float *dz,
                               float *dx, float *dgamma, float *dbeta, void *workspace, void *stream);
/* the same, and dx_chan_sum[c] (nullable) = sum over (b, t) of dx[b][c][t]: the bias gradient of the Conv1d in front of
 * the block, taken inside the pass that writes dx (float atomics, one per (b, c) row) instead of another pass over it */
int mrgcn_bn_relu_pool_bwd_sum_f32(const float *x, const float *y, const float *dy, const int32_t *argmax, int32_t B,
                               int32_t C, int32_t T, const float *gamma, const float *mean, const float *var,
                               float eps, int32_t training, int32_t pool_kind, int32_t pool_arg, float *dz,
                               float *dx, float *dgamma, float *dbeta, float *dx_chan_sum, void *workspace, void *stream);

/* ---- mini-batch frontier on the resident CSR of the stacked adjacency ---------------------
 * Replaces, per layer of a batch, the host loops of mrgcn/data/batch.py:185-263 (`A[sample_idx]`,
 * getNeighboursSparse :233-249, getAdjacencyNodeColumnIdx :251-256 + sliceSparseCOO :258-270).  indptr / indices:
 * int64 CSR of A (N x R*N) in HBM; sample: int64 [n_sample] row ids.  No allocation inside: the caller provides the
 * workspace and — after reading the two sizes row_off[n_sample] (entries) and node_pos[num_nodes] (neighbours) —
 * the outputs.
 *   count: row_off [n_sample+1] exclusive scan of the rows' lengths; node_pos [num_nodes+1] exclusive scan of
 *          "node j is the source node of some entry of these rows".
 *   emit : COO of the slice in the order A[sample].nonzero() has (row-major, stored column order): out_row = position
 *          in `sample`, out_col = global column, out_val (nullable) = the stored value as float32 or truncated to
 *          int8 (the reference's boundary cast); out_col_sliced (nullable) = r * n_neighbours + node_pos[j]: the
 *          column sliceSparseCOO gives the entry; neighbours (nullable) [n_neighbours] ascending node ids.
 *          (A slice without entries writes nothing: its output pointers are not looked at.) */
size_t mrgcn_frontier_workspace_bytes(int64_t num_nodes, int64_t n_sample);
int mrgcn_frontier_count(const int64_t *indptr, const int64_t *indices, int64_t num_nodes, const int64_t *sample,
                         int64_t n_sample, int64_t *row_off, int32_t *node_pos, void *workspace,
                         size_t workspace_bytes, void *stream);
int mrgcn_frontier_emit(const int64_t *indptr, const int64_t *indices, const float *data, int64_t num_nodes,
                        const int64_t *sample, int64_t n_sample, const int64_t *row_off, const int32_t *node_pos,
                        int64_t n_neighbours, int32_t value_dtype, int64_t *out_row, int64_t *out_col,
                        void *out_val, int64_t *out_col_sliced, int64_t *neighbours, void *stream);

/* ---- timing helpers (HIP events on the caller's stream; used by bench.py) ------ */
int mrgcn_event_create(void **event);
int mrgcn_event_destroy(void *event);
int mrgcn_event_record(void *event, void *stream);
int mrgcn_event_elapsed_ms(void *start, void *stop, float *h_ms); /* synchronises on stop */

#ifdef __cplusplus
}
#endif
#endif /* MRGCN_HIP_H */
