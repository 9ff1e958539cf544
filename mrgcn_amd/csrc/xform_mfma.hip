// Per-relation dense transforms on the matrix cores (f32-in / f32-acc MFMA 16x16x4: exact f32,
// one rounding per product, k-ordered fmaf chain) — the one place on the R-GCN path where the
// contraction really is dense (north_star): for every compact column c = (node j_c, relation r_c)
//
//   fwd   Out[o(c), n]   = sum_k In[i(c), k] * Wm[r_c][k][n]             (graph.py:93-94)
//   dW    dW[r][i][o]    += sum_{c in r} In[i(c), i] * G[c, o]              (its autograd)
//
// Columns are walked relation-major (`rperm`, chunks of <= kRelChunk columns of ONE relation per
// block), so a block stages its relation's weight tile in LDS once and every wave multiplies
// 16 gathered rows at a time.  Replaces the LDS-FMA kernels of rgcn_fused.hip (which stay as the
// fallback for shapes outside the limits below).
//
// MFMA 16x16x4 f32 lane maps (cdna_hip_programming.md §3): A[l&15][k=l>>4], B[k=l>>4][l&15],
// D: col = l&15, row = 4*(l>>4) + reg.  The k slot <-> actual k assignment is free as long as A
// and B agree: slot kq of the s-th MFMA of a 16-wide K step stands for k = k0 + 4*kq + s, so a
// lane feeds four MFMAs from ONE 16-byte load of its gathered row.
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "config.hpp"

namespace mrgcn {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kMaxKSteps = 16;  // K <= 256 for the forward
constexpr int kMaxNT = 4;       // F <= 64

__device__ __forceinline__ f32x4 load4_guarded(const float *row, int k, int K) {
  // the last K step of a row: element-wise, zero past K (never reads past the row)
  f32x4 v;
  v.x = (k + 0 < K) ? row[k + 0] : 0.f;
  v.y = (k + 1 < K) ? row[k + 1] : 0.f;
  v.z = (k + 2 < K) ? row[k + 2] : 0.f;
  v.w = (k + 3 < K) ? row[k + 3] : 0.f;
  return v;
}

__device__ __forceinline__ f32x4 load4_fast(const float *p) {
  // 16-byte load from a dword-aligned address (global loads need dword alignment only)
  return *reinterpret_cast<const f32x4 *>(p);
}
// the same four elements of a bf16 row (the bf16 pipeline's X): one 8-byte load, widened in registers
__device__ __forceinline__ f32x4 load4_guarded(const uint16_t *row, int k, int K) {
  f32x4 v;
  v.x = (k + 0 < K) ? bf16_to_f32(row[k + 0]) : 0.f;
  v.y = (k + 1 < K) ? bf16_to_f32(row[k + 1]) : 0.f;
  v.z = (k + 2 < K) ? bf16_to_f32(row[k + 2]) : 0.f;
  v.w = (k + 3 < K) ? bf16_to_f32(row[k + 3]) : 0.f;
  return v;
}
__device__ __forceinline__ f32x4 load4_fast(const uint16_t *p) {
  using u32x2_ = __attribute__((ext_vector_type(2))) uint32_t;
  const u32x2_ w = *reinterpret_cast<const u32x2_ *>(p);
  f32x4 v;
  v.x = __uint_as_float(w.x << 16);
  v.y = __uint_as_float(w.x & 0xffff0000u);
  v.z = __uint_as_float(w.y << 16);
  v.w = __uint_as_float(w.y & 0xffff0000u);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Live columns of one relation chunk.  In the backward of a semi-supervised epoch most rows of
// dM are exact zeros (only columns that feed a row within reach of a label receive gradient:
// ~10 % in layer 0 of the AM shape, < 1 % in layer 1); `col_live` (one byte per compact column,
// written by the transposed product that made dM) names the others.  The block keeps, in chunk
// order, the compact id and the input row of its live columns in LDS and sweeps only those.
// 256 threads; returns the count (block uniform).
// ---------------------------------------------------------------------------------------------
constexpr int kLiveRounds = kRelChunk / 256;
static_assert(kRelChunk % 256 == 0, "a chunk is compacted in rounds of 256 positions");
constexpr size_t kLiveLds = (2 * (size_t)kRelChunk + 4 * kLiveRounds) * sizeof(int32_t);  // ids | input rows | counts
__device__ __forceinline__ int compact_live_columns(int32_t beg, int32_t end, const int32_t *__restrict__ rperm,
                                                    const int32_t *__restrict__ rin_idx,
                                                    const uint8_t *__restrict__ col_live, int32_t *s_cid,
                                                    int32_t *s_rin, int32_t *s_cnt /* [4 * kLiveRounds] */) {
  // all (<= kRelChunk = 4 x 256) positions of the chunk in one go: the index and flag loads of the
  // four rounds are in flight together, one pair of barriers
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int32_t c[kLiveRounds], e[kLiveRounds];
  bool live[kLiveRounds];
  uint64_t bal[kLiveRounds];
#pragma unroll
  for (int k = 0; k < kLiveRounds; ++k) {
    e[k] = beg + k * 256 + (int32_t)threadIdx.x;
    c[k] = e[k] < end ? rperm[e[k]] : -1;
  }
#pragma unroll
  for (int k = 0; k < kLiveRounds; ++k) live[k] = c[k] >= 0 && col_live[c[k]] != 0;
#pragma unroll
  for (int k = 0; k < kLiveRounds; ++k) {
    bal[k] = __ballot(live[k]);
    if (lane == 0) s_cnt[k * 4 + wv] = __popcll(bal[k]);
  }
  __syncthreads();
  int total = 0;
  int off[kLiveRounds];
#pragma unroll
  for (int k = 0; k < kLiveRounds; ++k) {
    off[k] = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int n = s_cnt[k * 4 + w];
      if (w == wv) off[k] = total;  // chunk order: round k, then wave w, then lane
      total += n;
    }
  }
#pragma unroll
  for (int k = 0; k < kLiveRounds; ++k) {
    if (live[k]) {
      const int pos = off[k] + __popcll(bal[k] & ((1ull << lane) - 1ull));
      s_cid[pos] = c[k];
      s_rin[pos] = rin_idx ? rin_idx[e[k]] : c[k];
    }
  }
  __syncthreads();
  return total;
}

// ---------------------------------------------------------------------------------------------
// forward: Out[o(c), 0:ldOut] = [ In[i(c), 0:K] . Wm[r][0:K][0:F] | 0 ]   (write only)
//   TRANS_W = false: Wm[r][k][n] = W[(r*K + k)*F + n]        (weights stored [R][K][F])
//   TRANS_W = true : Wm[r][k][n] = W[(r*F + n)*K + k]        (weights stored [R][F][K])
//   (a launch may take a SLICE of the output columns: W and Out arrive offset to the slice's first column, `rstride`
//   stays the distance between two relations' weights, `n_store` the columns the slice may write)
//   rin_idx / rout_idx: nullable int32 [ncols] in RELATION-MAJOR order (aligned with rperm):
//   input / output row of each column; null = the compact id rperm[e] itself
// ---------------------------------------------------------------------------------------------
//   LIVE: `col_live` names the columns whose input row is not all zeros; only those are
//   multiplied and only their output rows are written (rin_idx / rout_idx must be null: the
//   backward dX pass, In = dM) — the consumer (k_segment_sum) skips the same columns.
//   KS >= ceil(K / 16), the smallest instantiated (round 6): the loads of a step are straight-line code — no
//   `ks < ksteps` test, no guarded last piece, no nullable index array inside the loop (the compiler had put every such
//   load into its own basic block with an `s_waitcnt vmcnt(0)` behind it).  Every piece starts at min(k, K - 4) and is
//   shifted into place with selects; what then sits in k slots >= K is real row data that meets the zero rows of the
//   staged weights.  K < 4 (the dX pass of a two-class head) keeps the element-wise loads, behind ONE wave-uniform
//   branch around the group.
template <int NT, bool TRANS_W, int KS, typename OT, bool LIVE = false>
__global__ __launch_bounds__(256) void k_xform_mfma_fwd(
    const int32_t *__restrict__ relchunk_rel, const int32_t *__restrict__ relchunk_beg,
    const int32_t *__restrict__ relchunk_end, const int32_t *__restrict__ rperm,
    const int32_t *__restrict__ rin_idx, const int32_t *__restrict__ rout_idx,
    const float *__restrict__ In, int64_t ldIn, int K, const float *__restrict__ W, int F,
    OT *__restrict__ Out, int64_t ldOut, const uint8_t *__restrict__ col_live,
    int64_t rstride /* floats between two relations' weights */, int n_store /* columns of Out this launch writes */) {
  extern __shared__ float WsT[];  // [NT*16][KP]: n-major, k contiguous, zero padded
  constexpr int KP = KS * 16 + 4;  // +4 floats: rows start on different banks
  const int chunk = blockIdx.x;
  const int r = relchunk_rel[chunk];
  int32_t beg = relchunk_beg[chunk], end = relchunk_end[chunk];
  int32_t *s_cid = reinterpret_cast<int32_t *>(WsT + NT * 16 * KP);  // LIVE: [kRelChunk] | [kRelChunk] | [4]
  if constexpr (LIVE) {
    const int n = compact_live_columns(beg, end, rperm, nullptr, col_live, s_cid, s_cid + kRelChunk,
                                       s_cid + 2 * kRelChunk);
    if (n == 0) return;  // block uniform
    beg = 0;
    end = n;
  }
  for (int t = threadIdx.x; t < NT * 16 * KP; t += blockDim.x) {
    const int n = t / KP, k = t - n * KP;
    float w = 0.f;
    if (n < F && k < K)
      w = TRANS_W ? W[(int64_t)r * rstride + (int64_t)n * K + k] : W[(int64_t)r * rstride + (int64_t)k * F + n];
    WsT[t] = w;
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  // relation-major position e -> input row / output row; the indices of the next tile are
  // fetched while the current one is multiplied (no dependent index round trips in the loop)
  const int32_t *pin = rin_idx ? rin_idx : rperm, *pout = rout_idx ? rout_idx : rperm;
  auto fetch = [&](int32_t e, int32_t &valid, int32_t &rin, int32_t &rout) {
    valid = e < end ? 1 : -1;
    const int32_t ee = e < end ? e : beg;
    if constexpr (LIVE) {
      rin = rout = s_cid[ee];
    } else {
      rin = pin[ee];
      rout = pout[ee];
    }
  };
  const bool tiny = K < 4;  // wave uniform
  int32_t n_valid, n_rin, n_rout;
  fetch(beg + wv * 16 + m, n_valid, n_rin, n_rout);
  for (int32_t t0 = beg + wv * 16; t0 < end; t0 += 64) {  // 4 waves x 16 columns per sweep
    const int32_t cid = n_valid, my_rout = n_rout;
    const int64_t rin = n_rin;
    fetch(t0 + 64 + m, n_valid, n_rin, n_rout);
    const float *xrow = In + rin * ldIn;
    // all K steps of the gathered row in flight at once: every piece one 16-byte load whose start is clamped into the
    // row (pieces wholly inside it: unchanged), then shifted into place
    f32x4 a[KS];
    if (tiny) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[ks] = load4_guarded(xrow, ks * 16 + 4 * kq, K);
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[ks] = load4_fast(xrow + min(ks * 16 + 4 * kq, K - 4));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int shift = ks * 16 + 4 * kq - min(ks * 16 + 4 * kq, K - 4);
        const f32x4 w = a[ks];
        a[ks].x = shift == 0 ? w.x : shift == 1 ? w.y : shift == 2 ? w.z : w.w;
        a[ks].y = shift == 0 ? w.y : shift == 1 ? w.z : w.w;
        a[ks].z = shift == 0 ? w.z : w.w;
      }
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f32x4 av = a[ks];
      if (cid < 0) av = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(&WsT[(nt * 16 + m) * KP + ks * 16 + 4 * kq]);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[nt], 0, 0, 0);
      }
    }
    // D: lane (n = l&15, g = l>>4) holds rows 4g + reg of the 16-column tile, feature nt*16 + n
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int32_t ok = __shfl(cid, 4 * kq + reg, 64);
      const int64_t orow = __shfl(my_rout, 4 * kq + reg, 64);
      if (ok < 0) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + m;
        if (n < n_store) store_operand<OT>(Out + orow * ldOut + n, acc[nt][reg]);  // zeros past F: whole padded row
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dW[r][i][o] += sum_{c in chunk} In[i(c), i] * G[c, o]        (K = rows of dW per relation <= 256,
// F <= 16).  M dimension = i, N = o, MFMA k = columns.  Lane m of k-group kq loads 16 bytes
// In[i(c_kq), 64*tq + 4m .. +3]: four M tiles (s = 0..3) where tile (tq, s) row m is i = 64tq+4m+s.
// Per block: every wave stores its tiles to its own LDS region, the block sums the 4 regions and
// writes one partial slab per chunk (k_dw_reduce adds a relation's slabs).
// ---------------------------------------------------------------------------------------------
constexpr int kMaxTQ = 4;  // K <= 256

// The loop holds NO branch around a load (DESIGN "straight-line loads"): until round 6 every gathered piece sat in its
// own conditional — `tq < ntq`, the guarded last piece of a row, nullable index arrays — and the compiler put an
// `s_waitcnt vmcnt(0)` behind each of them: ~10 dependent round trips per step of 8 columns, 0.5 ms alone / 0.93 ms
// beside the mix backward for 0.6 GB of rows at the AM shape (8 % of the roofline, "nobody has yet said why").  Now
// TQ = ceil(K / 64) exactly, every piece is one unconditional load at a clamped address — the row's LAST piece starts
// at min(i, Kr - 4) (Kr = readable elements of a row: K, or the padded row of a bf16 input) and is shifted into place
// with selects; what lands in tile rows i >= K is real row data that no store ever reads — and the indices of the
// next step are fetched before this step's rows.
template <int TQ, int U, bool LIVE, typename IT = float>
__global__ __launch_bounds__(256) void k_xform_mfma_dw(
    const int32_t *__restrict__ relchunk_rel, const int32_t *__restrict__ relchunk_beg,
    const int32_t *__restrict__ relchunk_end, const int32_t *__restrict__ rperm,
    const int32_t *__restrict__ rin_idx /* required */, const IT *__restrict__ In, int64_t ldIn, int K,
    const float *__restrict__ G, int64_t ldG, int F, float *__restrict__ dW,
    float *__restrict__ slab, const uint8_t *__restrict__ col_live, int64_t zero_dw) {
  extern __shared__ float dWs[];  // [4 waves][K*F]: every wave stores its own partial tile set
  const int chunk = blockIdx.x;
  if (slab && zero_dw > 0) {  // k_dw_reduce, the next launch, adds into dW: the zero fill rides along here
    const int64_t per = (zero_dw + gridDim.x - 1) / gridDim.x;
    const int64_t z0 = (int64_t)chunk * per, z1 = min(z0 + per, zero_dw);
    for (int64_t t = z0 + threadIdx.x; t < z1; t += blockDim.x) dW[t] = 0.f;
  }
  const int r = relchunk_rel[chunk];
  int32_t beg = relchunk_beg[chunk], end = relchunk_end[chunk];
  int32_t *s_cid = reinterpret_cast<int32_t *>(dWs + 4 * K * F);  // LIVE: [kRelChunk] | [kRelChunk] | [4]
  int32_t *s_rin = s_cid + kRelChunk;
  if constexpr (LIVE) {  // columns whose dM row is all zeros add nothing: sweep the others only
    end = compact_live_columns(beg, end, rperm, rin_idx, col_live, s_cid, s_rin, s_rin + kRelChunk);
    beg = 0;
    if (end == 0) {  // block uniform: nothing live in this chunk — its partial slab is zero
      if (slab) {
        float *out = slab + (int64_t)chunk * K * F;
        for (int t = threadIdx.x; t < K * F; t += blockDim.x) out[t] = 0.f;
      }
      return;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int m = lane & 15, kq = lane >> 4;
  f32x4 acc[TQ * 4];
#pragma unroll
  for (int t = 0; t < TQ * 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the last piece of a row: start clamped into the row, elements shifted into place
  const int Kr = sizeof(IT) == 2 ? (int)ldIn : K;
  const int i_last = 64 * (TQ - 1) + 4 * m;
  const int i_clamped = min(i_last, Kr - 4);
  const int shift = i_last - i_clamped;  // 0: the piece as loaded; >= 4: nothing of it is real (tile rows >= K)
  const int mG = min(m, F - 1);

  // indices of the next sweep are fetched while the current one is multiplied; positions past the chunk's end
  // read the last valid position's (their G element is zeroed: they add nothing)
  int32_t n_cid[U], n_rin[U];
  auto fetch = [&](int32_t t0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t e = t0 + 4 * u + kq;
      const int32_t ee = min(e, end - 1);
      int32_t c, ri;
      if constexpr (LIVE) {
        c = s_cid[ee];
        ri = s_rin[ee];
      } else {
        c = rperm[ee];
        ri = rin_idx[ee];
      }
      n_cid[u] = e < end ? c : -1 - c;  // (negative: invalid; the id itself stays recoverable for a valid address)
      n_rin[u] = ri;
    }
  };
  fetch(beg + wv * 4 * U);
  for (int32_t t0 = beg + wv * 4 * U; t0 < end; t0 += 16 * U) {  // 4 waves x (U x 4) columns per sweep
    f32x4 a[U][TQ];
    float b[U];
    int32_t cidv[U], rinv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { cidv[u] = n_cid[u]; rinv[u] = n_rin[u]; }
    fetch(t0 + 16 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t cid = cidv[u];
      const IT *xrow = In + (int64_t)rinv[u] * ldIn;
      b[u] = G[(int64_t)(cid >= 0 ? cid : -1 - cid) * ldG + mG];
#pragma unroll
      for (int tq = 0; tq < TQ - 1; ++tq) a[u][tq] = load4_fast(xrow + 64 * tq + 4 * m);
      a[u][TQ - 1] = load4_fast(xrow + i_clamped);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (cidv[u] < 0 || m >= F) b[u] = 0.f;
      const f32x4 w = a[u][TQ - 1];
      f32x4 v;
      v.x = shift == 0 ? w.x : shift == 1 ? w.y : shift == 2 ? w.z : w.w;
      v.y = shift == 0 ? w.y : shift == 1 ? w.z : w.w;
      v.z = shift == 0 ? w.z : w.w;
      v.w = w.w;
      a[u][TQ - 1] = v;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int tq = 0; tq < TQ; ++tq) {
        acc[tq * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][tq].x, b[u], acc[tq * 4 + 0], 0, 0, 0);
        acc[tq * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][tq].y, b[u], acc[tq * 4 + 1], 0, 0, 0);
        acc[tq * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][tq].z, b[u], acc[tq * 4 + 2], 0, 0, 0);
        acc[tq * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][tq].w, b[u], acc[tq * 4 + 3], 0, 0, 0);
      }
    }
  }
  // D of tile (tq, s): lane (o = l&15, g = l>>4), reg -> row m' = 4g + reg -> i = 64tq + 4m' + s
#pragma unroll
  for (int tq = 0; tq < TQ; ++tq) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = 64 * tq + 4 * (4 * kq + reg) + s;
        if (i < K && m < F) dWs[wv * K * F + i * F + m] = acc[tq * 4 + s][reg];  // each (i, o) once per wave
      }
    }
  }
  __syncthreads();
  const int KF = K * F;
  if (slab) {  // per-chunk partial, summed per relation by k_dw_reduce (no contended atomics)
    float *out = slab + (int64_t)chunk * KF;
    for (int t = threadIdx.x; t < KF; t += blockDim.x)
      out[t] = (dWs[t] + dWs[KF + t]) + (dWs[2 * KF + t] + dWs[3 * KF + t]);
  } else {
    float *dWr = dW + (int64_t)r * KF;
    for (int t = threadIdx.x; t < KF; t += blockDim.x) {
      const float x = (dWs[t] + dWs[KF + t]) + (dWs[2 * KF + t] + dWs[3 * KF + t]);
      if (x != 0.f) atomicAdd(&dWr[t], x);
    }
  }
}

// dW[r] += sum of the slab rows of a segment of <= kDwSeg chunks of relation r
constexpr int kDwSeg = 32;
__global__ __launch_bounds__(256) void k_dw_reduce(const int32_t *__restrict__ relchunk_ids,
                                                   const int32_t *__restrict__ relchunk_ptr, int n_seg_max,
                                                   const float *__restrict__ slab, int KF,
                                                   float *__restrict__ dW, int R) {
  // grid = (R * n_seg_max, ceil(KF / 256)): (relation, segment) x a 256-element slice of the tile — one element per
  // thread, the segment's slabs four at a time (the sum order is fixed by the chunk order)
  const int r = blockIdx.x / n_seg_max, seg = blockIdx.x - r * n_seg_max;
  if (r >= R) return;
  const int c0 = relchunk_ptr[r] + seg * kDwSeg;
  const int c1 = min(relchunk_ptr[r + 1], c0 + kDwSeg);
  const int t = blockIdx.y * blockDim.x + threadIdx.x;
  if (c0 >= c1 || t >= KF) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = c0;
  for (; c + 4 <= c1; c += 4) {
    const int64_t i0 = relchunk_ids[c], i1 = relchunk_ids[c + 1], i2 = relchunk_ids[c + 2], i3 = relchunk_ids[c + 3];
    s0 += slab[i0 * KF + t];
    s1 += slab[i1 * KF + t];
    s2 += slab[i2 * KF + t];
    s3 += slab[i3 * KF + t];
  }
  for (; c < c1; ++c) s0 += slab[(int64_t)relchunk_ids[c] * KF + t];
  const float s = (s0 + s1) + (s2 + s3);
  if (relchunk_ptr[r + 1] - relchunk_ptr[r] <= kDwSeg) dW[(int64_t)r * KF + t] = s;  // sole writer
  else if (s != 0.f) atomicAdd(&dW[(int64_t)r * KF + t], s);
}

// dX[j, 0:K] = sum of the rows Z[nptr[j] .. nptr[j+1]) (Z in compact (j, r) order)
template <bool LIVE>
__global__ void k_segment_sum(const int32_t *__restrict__ nptr, const float *__restrict__ Z, int64_t ldZ,
                              int64_t N, int K, float *__restrict__ dX, int64_t lddX,
                              const uint8_t *__restrict__ col_live) {
  const int64_t total = N * K;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / K;
    const int i = (int)(t - j * K);
    const int32_t c0 = nptr[j], c1 = nptr[j + 1];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int32_t c = c0;
    if constexpr (LIVE) {  // rows of dead columns were never written
      for (; c < c1; ++c)
        if (col_live[c]) s0 += Z[(int64_t)c * ldZ + i];
      dX[j * lddX + i] = s0;
      continue;
    }
    for (; c + 4 <= c1; c += 4) {
      s0 += Z[(int64_t)c * ldZ + i];
      s1 += Z[(int64_t)(c + 1) * ldZ + i];
      s2 += Z[(int64_t)(c + 2) * ldZ + i];
      s3 += Z[(int64_t)(c + 3) * ldZ + i];
    }
    for (; c < c1; ++c) s0 += Z[(int64_t)c * ldZ + i];
    dX[j * lddX + i] = (s0 + s1) + (s2 + s3);
  }
}

// The same for WIDE rows (64 < K <= 256: the input gradient of a layer-0 transform whose input has a gradient — the
// encoders' X) as a one-shot grid, a wave per node: lane l sums floats l, l + 64, ... of the node's Z rows (coalesced
// dwords), four rows' loads in flight at clamped addresses; a node without rows stores zeros.  The thread-per-element
// form above walks nptr and one dependent 4-byte load per element and row: 822 us for 3 GB at the AM shape.
template <int NK>
__global__ __launch_bounds__(256) void k_segment_sum_wide(const int32_t *__restrict__ nptr, const float *__restrict__ Z,
                                                          int64_t ldZ, int64_t N, int K, float *__restrict__ dX,
                                                          int64_t lddX) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= N) return;
  const int32_t c0 = nptr[j], c1 = nptr[j + 1];
  int kk[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) kk[k] = (lane + 64 * k < K) ? lane + 64 * k : K - 1;  // (clamped: masked at the store)
  float acc[4][NK];  // (the four partial sums of k_segment_sum, in its order: the same bits)
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int k = 0; k < NK; ++k) acc[u][k] = 0.f;
  for (int32_t c = c0; c < c1; c += 4) {
    float v[4][NK];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float *z = Z + (int64_t)(c + u < c1 ? c + u : c) * ldZ;
#pragma unroll
      for (int k = 0; k < NK; ++k) v[u][k] = z[kk[k]];
    }
    const bool whole = c + 4 <= c1;  // (wave uniform; the rows of a last, partial group all go to the first sum)
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c + u < c1) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          if (whole) acc[u][k] += v[u][k];
          else acc[0][k] += v[u][k];
        }
      }
  }
  float *o = dX + j * lddX;
#pragma unroll
  for (int k = 0; k < NK; ++k)
    if (lane + 64 * k < K) o[lane + 64 * k] = (acc[0][k] + acc[1][k]) + (acc[2][k] + acc[3][k]);
}

// The same with liveness flags, K <= 16: one thread per node.  Nearly every node has no live
// column (layer 1 of a semi-supervised epoch: the 1-hop neighbourhood of the labelled nodes):
// it reads its flag bytes and stores a zero row.
template <int KT>
__global__ void k_segment_sum_live(const int32_t *__restrict__ nptr, const float *__restrict__ Z, int64_t ldZ,
                                   int64_t N, int K, float *__restrict__ dX, int64_t lddX,
                                   const uint8_t *__restrict__ col_live) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < N;
       j += (int64_t)gridDim.x * blockDim.x) {
    float acc[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) acc[i] = 0.f;
    const int32_t c1 = nptr[j + 1];
    for (int32_t c = nptr[j]; c < c1; ++c) {
      if (!col_live[c]) continue;
      const float *z = Z + (int64_t)c * ldZ;
#pragma unroll
      for (int i = 0; i < KT; ++i)
        if (i < K) acc[i] += z[i];
    }
    float *o = dX + j * lddX;
#pragma unroll
    for (int i = 0; i < KT; ++i)
      if (i < K) o[i] = acc[i];
  }
}

// The layer-input gradient of a hidden layer, finished in one pass: dX[j] = sum of node j's live Z rows, times the
// ReLU mask of the layer's input when that input is the output of a fused ReLU (mask_src = the input itself:
// H > 0 is the mask of the ReLU that produced H — the producing layer's backward then has nothing left to mask),
// plus one byte per row: does it hold anything but zeros (what mrgcn_spmm_transposed_live_flagged_f32 of the layer
// below takes instead of scanning dX again).  A thread sums one node; the 256 rows of a block leave through LDS as
// coalesced stores (thread-per-row stores of 40-byte rows were the cost of k_segment_sum_live).  Dead rows are
// written as zeros: the result is a complete gradient tensor.
template <int KT>
__global__ __launch_bounds__(256) void k_segment_sum_mask(const int32_t *__restrict__ nptr, const float *__restrict__ Z,
                                                          int64_t ldZ, int64_t N, int K, float *__restrict__ dX,
                                                          int64_t lddX, const uint8_t *__restrict__ col_live,
                                                          const float *__restrict__ mask_src, int64_t ldMask,
                                                          uint8_t *__restrict__ row_live, int64_t ncols,
                                                          const uint8_t *__restrict__ node_live, int lane_max = 0) {
  // lane_max > 0 (a node list where every node is live — a gradient support's compact arrays): a node of up to
  // lane_max columns is summed by its own lane; the wave-by-wave walk below, made for a few live nodes among many, took
  // 125 us for the 6.8 k nodes of a mini-batch layer
  __shared__ float tile[256][KT + 1];
  const int64_t j0 = (int64_t)blockIdx.x * 256;
  const int64_t j = j0 + threadIdx.x;
  float acc[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i) acc[i] = 0.f;
  bool any_out = false, wide_out = false, own_out = false;
  int32_t c0_out = 0, c1_out = 0;
  if (j < N) {
    const int32_t c0 = nptr[j], c1 = nptr[j + 1];
    // nearly every node of a semi-supervised epoch has no live column: look at its flag bytes eight at a time
    // (aligned 8-byte words; the bytes of neighbouring nodes are masked off, the array's last partial word is read
    // byte by byte) and walk the columns only when one is set
    bool any = col_live == nullptr && c1 > c0;
    // (`node_live`, when the producer of the column flags also flagged their source nodes: one coalesced byte per
    // node instead of dependent looks at its columns' flags — 80 -> 25 us at the AM shape)
    if (node_live) any = c1 > c0 && node_live[j] != 0;
    const bool wide = !node_live && col_live != nullptr && c1 - c0 > 24;  // many columns: the wave looks together (below)
    if (!node_live && col_live && !wide) {
      for (int32_t w0 = c0 & ~7; w0 < c1 && !any; w0 += 8) {
        uint64_t bits = 0;
        if (w0 + 8 <= ncols) {
          bits = *reinterpret_cast<const uint64_t *>(col_live + w0);
        } else {
          for (int b = 0; b < 8 && w0 + b < ncols; ++b) bits |= (uint64_t)col_live[w0 + b] << (8 * b);
        }
        if (w0 < c0) bits &= ~uint64_t(0) << (8 * (c0 - w0));
        if (w0 + 8 > c1) bits &= ~uint64_t(0) >> (8 * (w0 + 8 - c1));
        any = bits != 0;
      }
    }
    if (any && lane_max > 0 && c1 - c0 <= lane_max) {
      for (int32_t c = c0; c < c1; c += 4) {  // four rows' loads in flight (clamped), added in order
        float zz[4][KT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float *z = Z + (int64_t)(c + u < c1 ? c + u : c) * ldZ;
#pragma unroll
          for (int i = 0; i < KT; ++i) zz[u][i] = z[i < K ? i : K - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c + u < c1) {
#pragma unroll
            for (int i = 0; i < KT; ++i)
              if (i < K) acc[i] += zz[u][i];
          }
      }
      any = false;
      own_out = true;
    }
    any_out = any; c0_out = c0; c1_out = c1; wide_out = wide;
  }
  // nodes with many columns (a source node of dozens of relations): their flag bytes are read by the whole wave, 512
  // per step — a lane walking 30 dependent words alone held its block for 30 round trips (85 -> 40 us at the AM shape)
  {
    const int lane = threadIdx.x & 63;
    uint64_t todo = __ballot(wide_out);
    while (todo) {
      const int L = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const int32_t a0 = __shfl(c0_out, L, 64), a1 = __shfl(c1_out, L, 64);
      bool hit = false;
      for (int32_t base = a0 & ~7; base < a1 && !hit; base += 512) {
        const int32_t w0 = base + 8 * lane;
        uint64_t bits = 0;
        if (w0 < a1) {
          if (w0 + 8 <= ncols) {
            bits = *reinterpret_cast<const uint64_t *>(col_live + w0);
          } else {
            for (int b = 0; b < 8 && w0 + b < ncols; ++b) bits |= (uint64_t)col_live[w0 + b] << (8 * b);
          }
          if (w0 < a0) bits &= ~uint64_t(0) << (8 * (a0 - w0));
          if (w0 + 8 > a1) bits &= ~uint64_t(0) >> (8 * (w0 + 8 - a1));
        }
        hit = __ballot(bits != 0) != 0;
      }
      if (lane == L) any_out = hit;
    }
  }
  // the (few) nodes with a live column: the wave sums a node's Z rows together, lane l the columns l, l + 64, ...
  // (a lane walking the 100+ columns of a live hub alone — two dependent round trips per column — held its block, and
  // the kernel's tail, for 70 us at the AM shape)
  {
    const int lane = threadIdx.x & 63;
    uint64_t todo = __ballot(any_out);
    while (todo) {
      const int L = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const int32_t a0 = __shfl(c0_out, L, 64), a1 = __shfl(c1_out, L, 64);
      float part[KT];
#pragma unroll
      for (int i = 0; i < KT; ++i) part[i] = 0.f;
      for (int32_t c = a0 + lane; c < a1; c += 64) {
        if (col_live && !col_live[c]) continue;
        const float *z = Z + (int64_t)c * ldZ;
#pragma unroll
        for (int i = 0; i < KT; ++i)
          if (i < K) part[i] += z[i];
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int i = 0; i < KT; ++i) part[i] += __shfl_xor(part[i], off, 64);
      }
      if (lane == L) {
#pragma unroll
        for (int i = 0; i < KT; ++i) acc[i] = part[i];
      }
    }
  }
  if (j < N) {
    const bool any = any_out || own_out;
    bool nz = false;
    if (any) {
      if (mask_src) {
        const float *m = mask_src + j * ldMask;
#pragma unroll
        for (int i = 0; i < KT; ++i)
          if (i < K && !(m[i] > 0.f)) acc[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < KT; ++i) nz |= acc[i] != 0.f;
    }
    if (row_live) row_live[j] = nz ? 1 : 0;
  }
#pragma unroll
  for (int i = 0; i < KT; ++i) tile[threadIdx.x][i] = acc[i];
  __syncthreads();
  const int64_t rows = min((int64_t)256, N - j0);
  for (int e = threadIdx.x; e < rows * K; e += 256) {
    const int node = e / K, i = e - node * K;
    dX[(j0 + node) * lddX + i] = tile[node][i];
  }
}

// ---------------------------------------------------------------------------------------------
// Narrow transform, columns in OUTPUT order (a hidden layer: K, F <= 16).  Every relation's weight tile fits LDS at
// once (AM layer 1: 267 x 10 x 12 floats = 128 KB), so nothing forces the relation-major walk of the kernels above —
// whose output rows (44-48 bytes each) then land wherever the operand order puts them: 347 us for a 392 MB operand
// at the AM shape, 2.7x its bytes in fabric traffic.  Here position p of the output is computed by the lanes
// 4p .. 4p + 3 (four outputs each): the operand leaves as ONE sequential stream, the input rows (40-48 bytes of a
// 67-80 MB table: cache resident) are gathered by (node, relation) ids stored in output order (plan: op_node /
// op_rel; compact order: unode / urel themselves).
//   Out[p, 0:ldOut] = [ In[node_p, 0:K] . W[rel_p][0:K][0:F] | 0 ]
// LDS: W as [R][K][FP], FP = 4 * ceil(F / 4): lane q of a position reads 16 bytes per k.
// ---------------------------------------------------------------------------------------------
template <int KT, typename OT>
__global__ __launch_bounds__(1024) void k_xform_cols_lds(const int32_t *__restrict__ pnode,
                                                         const int32_t *__restrict__ prel, int64_t npos,
                                                         const float *__restrict__ In, int64_t ldIn, int K,
                                                         const float *__restrict__ W, int R, int F, int FP,
                                                         OT *__restrict__ Out, int64_t ldOut) {
  extern __shared__ __align__(16) float s_w[];  // [R][K][FP]
  const int KF = K * F, KFP = K * FP;
  if (F == FP && (((uintptr_t)W) & 15) == 0) {  // rows of whole 16-byte pieces: the table is copied as it lies
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(W);
    f32x4 *s4 = reinterpret_cast<f32x4 *>(s_w);
    for (int t = threadIdx.x; t < (R * KFP) >> 2; t += blockDim.x) s4[t] = w4[t];
  } else {
    for (int t = threadIdx.x; t < R * KFP; t += blockDim.x) {
      const int r = t / KFP, rem = t - r * KFP;
      const int k = rem / FP, f = rem - k * FP;
      s_w[t] = f < F ? W[(int64_t)r * KF + k * F + f] : 0.f;
    }
  }
  __syncthreads();
  const int q = threadIdx.x & 3;
  const int nq = FP >> 2;  // lanes of a position that hold outputs
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 2);
  // U positions per lane group and step: their ids, then their input rows, are in flight together (one position at a
  // time the walk was two dependent round trips per 16 positions and wave: 277 us at the AM shape)
  constexpr int U = 4;
  for (int64_t p0 = (int64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2); p0 < npos; p0 += U * stride) {
    int32_t j[U], r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pp = min(p0 + u * stride, npos - 1);
      j[u] = pnode[pp];
      r[u] = prel[pp];
    }
    float h[U][KT];
    if (K >= 4) {
      // straight-line loads (no branch around any of them: every piece's waits would drain the others): a piece
      // starts at min(k4, K - 4) — inside the row — and is shifted into place; slots >= K are never multiplied
      f32x4 t[U][KT / 4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float *xrow = In + (int64_t)max(j[u], 0) * ldIn;
#pragma unroll
        for (int k4 = 0; k4 < KT; k4 += 4)   // the four lanes of a position load the same row: one fetch
          t[u][k4 / 4] = *reinterpret_cast<const f32x4 *>(xrow + min(k4, K - 4));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int k4 = 0; k4 < KT; k4 += 4) {
          const int shift = k4 - min(k4, K - 4);   // wave uniform
          const f32x4 w = t[u][k4 / 4];
          h[u][k4] = shift == 0 ? w.x : shift == 1 ? w.y : shift == 2 ? w.z : w.w;
          h[u][k4 + 1] = shift == 0 ? w.y : shift == 1 ? w.z : w.w;
          h[u][k4 + 2] = shift == 0 ? w.z : w.w;
          h[u][k4 + 3] = w.w;
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float *xrow = In + (int64_t)max(j[u], 0) * ldIn;
#pragma unroll
        for (int v = 0; v < KT; ++v) h[u][v] = (v < K) ? xrow[v] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pp = p0 + u * stride;
      if (pp >= npos || j[u] < 0) continue;  // (an operand row without a primary column: a replica, written later)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (q < nq) {
        const float *wr = s_w + r[u] * KFP + 4 * q;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k < K) {
            const f32x4 w = *reinterpret_cast<const f32x4 *>(wr + k * FP);
            acc.x = fmaf(h[u][k], w.x, acc.x);
            acc.y = fmaf(h[u][k], w.y, acc.y);
            acc.z = fmaf(h[u][k], w.z, acc.z);
            acc.w = fmaf(h[u][k], w.w, acc.w);
          }
        }
      }
      // the row's ldOut elements: zeros past F (the whole padded row is written)
      OT *orow = Out + pp * ldOut;
      if constexpr (sizeof(OT) == 4) {
        if (4 * q + 4 <= ldOut) {  // one 16-byte store per lane (dword-aligned addresses: rows may be 44 bytes)
          *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(orow) + 4 * q) = acc;
          continue;
        }
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = 4 * q + v;
        if (n < ldOut) store_operand<OT>(orow + n, acc[v]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// The bf16 pipeline (BASELINE config 3: "bf16 dense operands, fp32 accumulation"; the reference has no reduced
// precision anywhere — graph.py:93-95 is fp32).  Activations (X, H, the feature term's rows, M) are STORED in bf16,
// parameters and every accumulation stay fp32.
//
// k_cast_rows_bf16: dst[row, 0:ldDst] = [ bf16(src[row, 0:K]) | 0 ] — rows padded to whole 16-byte pieces.
// ---------------------------------------------------------------------------------------------
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// VEC: whole 32-byte pieces of 16-byte aligned source rows (K % 8 == 0, ldSrc % 4 == 0) — two 16-byte loads per
// thread, a one-shot grid (the encoders' X, N x 160 every forward: 467 -> ~300 us)
template <bool VEC>
__global__ __launch_bounds__(256) void k_cast_rows_bf16(const float *__restrict__ src, int64_t ldSrc, int64_t rows,
                                                        int K, uint16_t *__restrict__ dst, int64_t ldDst) {
  // one thread per 16-byte piece of dst (8 elements)
  const int pieces = (int)(ldDst >> 3);
  const int64_t total = rows * pieces;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / pieces;
    const int k0 = (int)(t - row * pieces) * 8;
    const float *x = src + row * ldSrc + k0;
    float v[8];
    if constexpr (VEC) {
      const bool in = k0 < K;  // (pieces past K: the row's zero padding)
      const float4 a = *reinterpret_cast<const float4 *>(in ? x : src), b = *reinterpret_cast<const float4 *>(in ? x + 4 : src);
      v[0] = in ? a.x : 0.f; v[1] = in ? a.y : 0.f; v[2] = in ? a.z : 0.f; v[3] = in ? a.w : 0.f;
      v[4] = in ? b.x : 0.f; v[5] = in ? b.y : 0.f; v[6] = in ? b.z : 0.f; v[7] = in ? b.w : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = (k0 + i < K) ? x[i] : 0.f;
    }
    u32x4 o;
    o.x = pack_bf16(v[0], v[1]);
    o.y = pack_bf16(v[2], v[3]);
    o.z = pack_bf16(v[4], v[5]);
    o.w = pack_bf16(v[6], v[7]);
    *reinterpret_cast<u32x4 *>(dst + row * ldDst + k0) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// forward on bf16 input rows, v_mfma_f32_16x16x32_bf16:
//   Out[o(c), 0:ldOut] = [ In[i(c), 0:K] . Wm[r][0:K][0:F] | 0 ],  F <= 16, In rows of ldIn bf16 (ldIn % 8 == 0,
//   zero past K), Wm[r][k][n] = W[(r*K + k)*F + n] in fp32, rounded to bf16 when the block stages it.
// The product is taken TRANSPOSED — A = Wm^T (row = output feature n), B = In^T (column = graph column) — so that the
// accumulator of lane (c = l & 15, g = l >> 4) holds features 4g .. 4g+3 of ITS OWN column c: the row leaves as one
// 16-byte (fp32) or 8-byte (bf16) store per lane, no lane exchange, and the lane that fetched a column's indices is
// the one that stores it.  A lane's B fragment of k-step ks is the 16 bytes In[i(c), 32 ks + 8 g ..+7]: the gather of
// a 320-byte row (K = 155) is five such loads in four lanes, half the bytes of the fp32 kernel's ten.
// U tiles of 16 columns per wave and step (their loads all in flight); the indices of the next step are fetched under
// the products of this one.
// ---------------------------------------------------------------------------------------------
template <int KS, int U, typename OT>
__global__ __launch_bounds__(256) void k_xform_bf16_fwd(
    const int32_t *__restrict__ relchunk_rel, const int32_t *__restrict__ relchunk_beg,
    const int32_t *__restrict__ relchunk_end, const int32_t *__restrict__ rperm,
    const int32_t *__restrict__ rin_idx, const int32_t *__restrict__ rout_idx,
    const uint16_t *__restrict__ In, int64_t ldIn, int K, const float *__restrict__ W, int F,
    OT *__restrict__ Out, int64_t ldOut) {
  extern __shared__ __align__(16) uint16_t WsB[];  // [16][KP] bf16: n-major, k contiguous, zero padded
  constexpr int KP = KS * 32 + 8;  // KS = ceil(K / 32) exactly; +16 bytes: the 16 rows' pieces start in different banks
  const int chunk = blockIdx.x;
  const int r = relchunk_rel[chunk];
  const int32_t beg = relchunk_beg[chunk], end = relchunk_end[chunk];
  for (int t = threadIdx.x; t < 16 * KP / 2; t += blockDim.x) reinterpret_cast<uint32_t *>(WsB)[t] = 0u;
  __syncthreads();
  {
    const float *Wr = W + (int64_t)r * K * F;
    for (int t = threadIdx.x; t < K * F; t += blockDim.x) {  // source order: coalesced
      const int k = t / F, n = t - k * F;
      WsB[n * KP + k] = f32_to_bf16(Wr[t]);
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  // A fragments (the relation's weights): constant over the chunk
  bf16x8 aw[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    aw[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(WsB + c * KP + ks * 32 + 8 * g));
  const int32_t *pin = rin_idx ? rin_idx : rperm, *pout = rout_idx ? rout_idx : rperm;
  auto fetch = [&](int32_t e, int32_t &rin, int32_t &rout) {  // (no branch around a load: straight-line steps)
    const int32_t ee = e < end ? e : beg;
    rin = pin[ee];
    rout = pout[ee];
  };
  int32_t n_rin[U], n_rout[U];
#pragma unroll
  for (int u = 0; u < U; ++u) fetch(beg + (wv * U + u) * 16 + c, n_rin[u], n_rout[u]);
  for (int32_t t0 = beg + wv * U * 16; t0 < end; t0 += 4 * U * 16) {  // 4 waves x U tiles x 16 columns per sweep
    int32_t rout[U];
    u32x4 x[U][KS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rout[u] = n_rout[u];
      // (a piece that starts past the padded row is read from the row's last piece instead: its k slots meet zero
      // weights, and no load leaves the row)
      const uint16_t *xrow = In + (int64_t)n_rin[u] * ldIn;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        x[u][ks] = *reinterpret_cast<const u32x4 *>(xrow + min(32 * ks + 8 * g, (int)ldIn - 8));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) fetch(t0 + (4 * U + u) * 16 + c, n_rin[u], n_rout[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[ks], __builtin_bit_cast(bf16x8, x[u][ks]), acc, 0, 0, 0);
      if (t0 + u * 16 + c >= end) continue;
      OT *orow = Out + (int64_t)rout[u] * ldOut + 4 * g;
      if (4 * g + 4 <= ldOut) {  // whole pieces: zeros past F come out of the zero rows of WsB
        if constexpr (sizeof(OT) == 4) {
          *reinterpret_cast<f32x4 *>(orow) = acc;
        } else {
          u32x2 o;
          o.x = pack_bf16(acc.x, acc.y);
          o.y = pack_bf16(acc.z, acc.w);
          *reinterpret_cast<u32x2 *>(orow) = o;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * g + i < ldOut) store_operand<OT>(orow + i, acc[i]);
      }
    }
  }
}

}  // namespace

bool xform_cols_lds_supported(const mrgcn_plan *p, int K, int F, int64_t ldOut, bool operand_order) {
  const bool on = cfg(CFG_XFORM_COLS_LDS) != 0;
  if (!on || !p || K > 16 || F > 16 || ldOut > 16) return false;
  if (operand_order && !p->op_node) return false;
  const int FP = (F + 3) / 4 * 4;
  return (size_t)p->num_relations * K * FP * sizeof(float) <= 150 * 1024;
}

int xform_cols_lds(const mrgcn_plan *p, bool operand_order, const float *In, int64_t ldIn, int K, const float *W, int F,
                   void *Out, int64_t ldOut, hipStream_t s, bool out_bf16) {
  const int64_t npos = operand_order ? p->n_op : p->ncols;
  if (npos == 0) return MRGCN_OK;
  const int32_t *pnode = operand_order ? p->op_node : p->unode;
  const int32_t *prel = operand_order ? p->op_rel : p->urel;
  const int R = (int)p->num_relations, FP = (F + 3) / 4 * 4;
  const size_t lds = (size_t)R * K * FP * sizeof(float);
  int64_t grid = (npos + 255) / 256;
  if (grid > 256) grid = 256;  // one block of 16 waves per CU (LDS)
#define XC_GO(KT_, O_)                                                                                             \
  do {                                                                                                             \
    auto kfn = k_xform_cols_lds<KT_, O_>;                                                                          \
    MRGCN_HIP_TRY(mrgcn::raise_lds_limit((const void *)kfn, lds));                                                 \
    kfn<<<dim3((unsigned)grid), dim3(1024), lds, s>>>(pnode, prel, npos, In, ldIn, K, W, R, F, FP, (O_ *)Out, ldOut); \
  } while (0)
  const int KT = K <= 4 ? 4 : K <= 8 ? 8 : K <= 12 ? 12 : 16;
  if (out_bf16) {
    if (KT == 4) XC_GO(4, uint16_t); else if (KT == 8) XC_GO(8, uint16_t); else if (KT == 12) XC_GO(12, uint16_t); else XC_GO(16, uint16_t);
  } else {
    if (KT == 4) XC_GO(4, float); else if (KT == 8) XC_GO(8, float); else if (KT == 12) XC_GO(12, float); else XC_GO(16, float);
  }
#undef XC_GO
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

// ---- launchers used by the C ABI entry points in rgcn_fused.hip -------------------------------
bool xform_mfma_fwd_supported(int K, int F) { return K <= kMaxKSteps * 16 && F <= kMaxNT * 16; }
// the dX pass Z = dM . W^T: F floats in, K out (sliced by 64 columns)
bool xform_mfma_dx_supported(int F, int K) { return F <= kMaxKSteps * 16 && K <= 256; }
bool xform_mfma_dw_supported(int K, int F) {  // (K >= 4: a row's last piece is a whole 16-byte load inside the row)
  return K >= 4 && K <= kMaxTQ * 64 && F <= 16 && (size_t)4 * K * F * 4 <= 64 * 1024;
}
// ... and with room in LDS for the list of live columns
bool xform_mfma_dw_live_supported(int K, int F) {
  return xform_mfma_dw_supported(K, F) && (size_t)4 * K * F * 4 + kLiveLds <= 64 * 1024;
}


// one launch: output columns [0, F) of a slice whose weights start at W (relations `rstride` floats apart) and whose
// rows start at Out; `n_store` columns of a row may be written (zeros past F)
static int xform_mfma_fwd_one(const mrgcn_plan *p, const RelOrder &o, const int32_t *rin_idx, const int32_t *rout_idx,
                              const float *In, int64_t ldIn, int K, const float *W, bool trans_w, int F, void *Out,
                              int64_t ldOut, hipStream_t s, bool out_bf16, const uint8_t *col_live, int64_t rstride,
                              int n_store) {
  if (o.n_relchunks == 0) return MRGCN_OK;
  int NT = (n_store + 15) / 16;  // tiles that cover the columns to write
  if (NT < (F + 15) / 16) NT = (F + 15) / 16;
  const int ksteps = (K + 15) / 16;
  // (the staged weight tile is sized by the INSTANTIATED k-step count: its row stride is a compile-time constant)
  auto lds_for = [&](int ks_inst) { return (size_t)NT * 16 * (ks_inst * 16 + 4) * sizeof(float); };
  if (col_live) {  // backward dX pass over the live columns only
    if (!trans_w || rin_idx || rout_idx || out_bf16 || ksteps > 4) {
      set_error("xform_mfma_fwd: col_live is for the dX pass (transposed weights, K <= 64)");
      return MRGCN_ERR_UNSUPPORTED;
    }
#define XF_LIVE(N_)                                                                                        \
  do {                                                                                                     \
    if (ksteps <= 1)                                                                                       \
      k_xform_mfma_fwd<N_, true, 1, float, true><<<dim3(o.n_relchunks), dim3(256), lds_for(1) + kLiveLds, s>>>( \
          o.relchunk_rel, o.relchunk_beg, o.relchunk_end, o.rperm, nullptr, nullptr, In, ldIn, K, W,   \
          F, (float *)Out, ldOut, col_live, rstride, n_store);                                             \
    else                                                                                                   \
      k_xform_mfma_fwd<N_, true, 4, float, true><<<dim3(o.n_relchunks), dim3(256), lds_for(4) + kLiveLds, s>>>( \
          o.relchunk_rel, o.relchunk_beg, o.relchunk_end, o.rperm, nullptr, nullptr, In, ldIn, K, W,   \
          F, (float *)Out, ldOut, col_live, rstride, n_store);                                             \
  } while (0)
    switch (NT) { case 1: XF_LIVE(1); break; case 2: XF_LIVE(2); break;
                  case 3: XF_LIVE(3); break; default: XF_LIVE(4); break; }
#undef XF_LIVE
    MRGCN_HIP_TRY(hipGetLastError());
    return MRGCN_OK;
  }
#define XF_K(N_, T_, O_, KS_)                                                                          \
  k_xform_mfma_fwd<N_, T_, KS_, O_><<<dim3(o.n_relchunks), dim3(256), lds_for(KS_), s>>>(              \
      o.relchunk_rel, o.relchunk_beg, o.relchunk_end, o.rperm, rin_idx, rout_idx, In, ldIn, K, W, F,  \
      (O_ *)Out, ldOut, nullptr, rstride, n_store)
#define XF_GO3(N_, T_, O_)                                                                              \
  do {                                                                                                  \
    if (N_ == 1) { /* narrow outputs (the layer transforms): the k-step count in steps of one or two */ \
      if (ksteps <= 1) XF_K(1, T_, O_, 1);                                                              \
      else if (ksteps <= 2) XF_K(1, T_, O_, 2);                                                         \
      else if (ksteps <= 3) XF_K(1, T_, O_, 3);                                                         \
      else if (ksteps <= 4) XF_K(1, T_, O_, 4);                                                         \
      else if (ksteps <= 6) XF_K(1, T_, O_, 6);                                                         \
      else if (ksteps <= 8) XF_K(1, T_, O_, 8);                                                         \
      else if (ksteps <= 10) XF_K(1, T_, O_, 10);                                                       \
      else if (ksteps <= 12) XF_K(1, T_, O_, 12);                                                       \
      else XF_K(1, T_, O_, kMaxKSteps);                                                                 \
    } else if (ksteps <= 1) XF_K(N_, T_, O_, 1);                                                        \
    else if (ksteps <= 4) XF_K(N_, T_, O_, 4);                                                          \
    else XF_K(N_, T_, O_, kMaxKSteps);                                                                  \
  } while (0)
#define XF_GO(N_, T_)                                                  \
  do {                                                                 \
    if (out_bf16) XF_GO3(N_, T_, uint16_t); else XF_GO3(N_, T_, float); \
  } while (0)
  if (out_bf16 && trans_w) { set_error("xform_mfma_fwd: bf16 output only for the forward operand"); return MRGCN_ERR_UNSUPPORTED; }
  if (trans_w) {
    switch (NT) { case 1: XF_GO3(1, true, float); break; case 2: XF_GO3(2, true, float); break;
                  case 3: XF_GO3(3, true, float); break; default: XF_GO3(4, true, float); break; }
  } else {
    switch (NT) { case 1: XF_GO(1, false); break; case 2: XF_GO(2, false); break;
                  case 3: XF_GO(3, false); break; default: XF_GO(4, false); break; }
  }
#undef XF_GO
#undef XF_GO3
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int xform_mfma_fwd(const mrgcn_plan *p, const RelOrder &o, const int32_t *rin_idx, const int32_t *rout_idx,
                   const float *In, int64_t ldIn, int K, const float *W, bool trans_w, int F, void *Out,
                   int64_t ldOut, hipStream_t s, bool out_bf16, const uint8_t *col_live) {
  const int64_t rstride = (int64_t)K * F;
  // columns a row may receive: zeros past F up to the end of the padded row (at most the launch's 64 / the last tile)
  const int width = (int)std::min<int64_t>(ldOut, F <= kMaxNT * 16 ? kMaxNT * 16 : (F + 15) / 16 * 16);
  if (F <= kMaxNT * 16)
    return xform_mfma_fwd_one(p, o, rin_idx, rout_idx, In, ldIn, K, W, trans_w, F, Out, ldOut, s, out_bf16, col_live,
                              rstride, std::min(width, kMaxNT * 16));
  // wide outputs (the dX pass of a wide layer: Z[c, 0:K_x] = dM[c] . W[r_c]^T with K_x up to 256): slices of 64
  // columns, one launch each — the input rows are 40 bytes, every slice reads them again
  if (!trans_w || out_bf16) {
    set_error("xform_mfma_fwd: more than 64 output columns only with transposed weights (the dX pass)");
    return MRGCN_ERR_UNSUPPORTED;
  }
  for (int n0 = 0; n0 < F; n0 += kMaxNT * 16) {
    const int nw = std::min(kMaxNT * 16, F - n0);
    const int rc = xform_mfma_fwd_one(p, o, rin_idx, rout_idx, In, ldIn, K, W + (int64_t)n0 * K, true, nw,
                                      (float *)Out + n0, ldOut, s, false, col_live, rstride,
                                      std::min(width - n0, kMaxNT * 16));
    if (rc != MRGCN_OK) return rc;
  }
  return MRGCN_OK;
}

bool xform_bf16_fwd_supported(int K, int F, int64_t ldIn, int64_t ldOut) {
  return K >= 1 && K <= 256 && F >= 1 && F <= 16 && ldOut <= 16 && ldOut >= F && (ldIn & 7) == 0 && ldIn >= K;
}

int xform_bf16_fwd(const mrgcn_plan *p, const RelOrder &o, const int32_t *rin_idx, const int32_t *rout_idx,
                   const uint16_t *In, int64_t ldIn, int K, const float *W, int F, void *Out, int64_t ldOut,
                   hipStream_t s, bool out_bf16) {
  if (!xform_bf16_fwd_supported(K, F, ldIn, ldOut)) {
    set_error("xform_bf16_fwd: K <= 256, F <= ldOut <= 16, input rows of whole 16-byte pieces");
    return MRGCN_ERR_UNSUPPORTED;
  }
  if (o.n_relchunks == 0) return MRGCN_OK;
  const int ksteps = (K + 31) / 32;
  const size_t lds = (size_t)16 * (ksteps * 32 + 8) * sizeof(uint16_t);
#define XB_GO(KS_, U_)                                                                                          \
  do {                                                                                                          \
    if (out_bf16)                                                                                               \
      k_xform_bf16_fwd<KS_, U_, uint16_t><<<dim3(o.n_relchunks), dim3(256), lds, s>>>(                         \
          o.relchunk_rel, o.relchunk_beg, o.relchunk_end, o.rperm, rin_idx, rout_idx, In, ldIn, K, W, F,      \
          (uint16_t *)Out, ldOut);                                                                              \
    else                                                                                                        \
      k_xform_bf16_fwd<KS_, U_, float><<<dim3(o.n_relchunks), dim3(256), lds, s>>>(                            \
          o.relchunk_rel, o.relchunk_beg, o.relchunk_end, o.rperm, rin_idx, rout_idx, In, ldIn, K, W, F,      \
          (float *)Out, ldOut);                                                                                 \
  } while (0)
  // (U: tiles of 16 columns in flight per wave; rows of up to two k-steps are short, four tiles keep as many bytes
  // in flight as two tiles of the long ones)
  switch (ksteps) {  // (exact: the kernel reads KS pieces of every row)
    case 1: XB_GO(1, 4); break;
    case 2: XB_GO(2, 4); break;
    case 3: XB_GO(3, 2); break;
    case 4: XB_GO(4, 2); break;
    case 5: XB_GO(5, 2); break;
    case 6: XB_GO(6, 2); break;
    case 7: XB_GO(7, 1); break;
    default: XB_GO(8, 1); break;
  }
#undef XB_GO
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int cast_rows_bf16(const float *src, int64_t ldSrc, int64_t rows, int K, uint16_t *dst, int64_t ldDst, hipStream_t s) {
  if (rows == 0) return MRGCN_OK;
  const int64_t total = rows * (ldDst >> 3);
  int64_t grid = (total + 255) / 256;
  const bool vec = K % 8 == 0 && ldSrc % 4 == 0 && (((uintptr_t)src) & 15) == 0 && grid <= 0x7fffffff;
  if (vec) {
    k_cast_rows_bf16<true><<<dim3((unsigned)grid), dim3(256), 0, s>>>(src, ldSrc, rows, K, dst, ldDst);
    MRGCN_HIP_TRY(hipGetLastError());
    return MRGCN_OK;
  }
  if (grid > 256 * 32) grid = 256 * 32;
  k_cast_rows_bf16<false><<<dim3((unsigned)grid), dim3(256), 0, s>>>(src, ldSrc, rows, K, dst, ldDst);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int xform_mfma_dw(const mrgcn_plan *p, const RelOrder &o, const int32_t *rin_idx, const void *In_, int64_t ldIn,
                  int K, const float *G, int64_t ldG, int F, float *dW, float *workspace,
                  int64_t workspace_floats, hipStream_t s, const uint8_t *col_live, bool in_bf16) {
  const float *In = (const float *)In_;
  if (o.n_relchunks == 0) {
    MRGCN_HIP_TRY(mrgcn::fill_async(dW, 0, (size_t)p->num_relations * K * F * sizeof(float), s));
    return MRGCN_OK;
  }
  size_t lds = (size_t)4 * K * F * sizeof(float);
  if (in_bf16) col_live = nullptr;  // (the bf16 pipeline's backward runs on gradient supports: dense index spaces)
  if (col_live && lds + kLiveLds <= 64 * 1024) lds += kLiveLds;
  else col_live = nullptr;  // no room for the list: every column is swept (same result)
  float *slab = (workspace && workspace_floats >= (int64_t)o.n_relchunks * K * F) ? workspace : nullptr;
  // dW is zeroed inside the first launch in the slab form (no fill node per call); the atomic form adds into a dW
  // the caller zeroed
  const int64_t zero_dw = slab ? (int64_t)p->num_relations * K * F : 0;
  // Unroll U (columns in flight per wave = 4 U) and tile count are chosen for registers, i.e. for
  // waves per SIMD — the pass waits on the gathered input rows (PMC: profiles/r01_xform_pmc.md).
  // AM shape, same run: K = 10: U = 8 / 4 / 2 -> 1.42 / 1.20 / 1.11 ms (with the dX half);
  // K = 155: <4,4> 1.92 ms, <3,4> 1.65 ms, <3,2> 1.62 ms.
#define DW_GO(TQ_, U_)                                                                                   \
  do {                                                                                                   \
    if (in_bf16)                                                                                         \
      k_xform_mfma_dw<TQ_, U_, false, uint16_t><<<dim3(o.n_relchunks), dim3(256), lds, s>>>(            \
          o.relchunk_rel, o.relchunk_beg, o.relchunk_end, o.rperm, rin_idx, (const uint16_t *)In_, ldIn, \
          K, G, ldG, F, dW, slab, nullptr, zero_dw);                                                     \
    else if (col_live)                                                                                   \
      k_xform_mfma_dw<TQ_, U_, true><<<dim3(o.n_relchunks), dim3(256), lds, s>>>(                       \
          o.relchunk_rel, o.relchunk_beg, o.relchunk_end, o.rperm, rin_idx, In, ldIn, K, G, ldG, F,  \
          dW, slab, col_live, zero_dw);                                                                  \
    else                                                                                                 \
      k_xform_mfma_dw<TQ_, U_, false><<<dim3(o.n_relchunks), dim3(256), lds, s>>>(                      \
          o.relchunk_rel, o.relchunk_beg, o.relchunk_end, o.rperm, rin_idx, In, ldIn, K, G, ldG, F,  \
          dW, slab, nullptr, zero_dw);                                                                   \
  } while (0)
  // (live form, K = 155, columns in flight per wave 4 / 8 / 16: 591 / 463 / 485 us)
  if (!rin_idx || K < 4 || (in_bf16 && (ldIn < 4 || (ldIn & 3)))) {
    set_error("xform_mfma_dw: needs the input-row index of every column, K >= 4 (bf16 rows: ld a multiple of 4)");
    return MRGCN_ERR_UNSUPPORTED;
  }
  if (K <= 64) DW_GO(1, 2);
  else if (K <= 128) DW_GO(2, 2);
  else if (K <= 192) DW_GO(3, 2);
  else DW_GO(4, 2);
#undef DW_GO
  MRGCN_HIP_TRY(hipGetLastError());
  if (slab) {
    const int n_seg_max = (o.max_relchunks + kDwSeg - 1) / kDwSeg;
    if (n_seg_max > 0) {
      k_dw_reduce<<<dim3((unsigned)(p->num_relations * n_seg_max), (unsigned)((K * F + 255) / 256)), dim3(256), 0, s>>>(
          o.relchunk_ids, o.relchunk_ptr, n_seg_max, slab, K * F, dW, (int)p->num_relations);
      MRGCN_HIP_TRY(hipGetLastError());
    }
  }
  return MRGCN_OK;
}

int segment_sum(const mrgcn_plan *p, const float *Z, int64_t ldZ, int K, float *dX, int64_t lddX,
                hipStream_t s, const uint8_t *col_live, const float *mask_src, int64_t ldMask, uint8_t *row_live,
                const uint8_t *node_live) {
  if (K <= 16 && (mask_src || row_live || col_live) && p->num_nodes > 0) {  // the one-pass form
    const dim3 grid((unsigned)((p->num_nodes + 255) / 256));
    if (K <= 8)
      k_segment_sum_mask<8><<<grid, dim3(256), 0, s>>>(p->nptr, Z, ldZ, p->num_nodes, K, dX, lddX, col_live, mask_src,
                                                       ldMask, row_live, p->ncols, node_live);
    else
      k_segment_sum_mask<16><<<grid, dim3(256), 0, s>>>(p->nptr, Z, ldZ, p->num_nodes, K, dX, lddX, col_live, mask_src,
                                                        ldMask, row_live, p->ncols, node_live);
    MRGCN_HIP_TRY(hipGetLastError());
    return MRGCN_OK;
  }
  if (mask_src || row_live) {
    set_error("segment_sum: the masked / flagged form needs K <= 16");
    return MRGCN_ERR_UNSUPPORTED;
  }
  int64_t work = p->num_nodes * K;
  int64_t blocks = (work + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (col_live && K <= 16) {
    int64_t nb = (p->num_nodes + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (K <= 8)
      k_segment_sum_live<8><<<dim3((unsigned)nb), dim3(256), 0, s>>>(p->nptr, Z, ldZ, p->num_nodes, K, dX, lddX, col_live);
    else
      k_segment_sum_live<16><<<dim3((unsigned)nb), dim3(256), 0, s>>>(p->nptr, Z, ldZ, p->num_nodes, K, dX, lddX, col_live);
  } else if (col_live)
    k_segment_sum<true><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p->nptr, Z, ldZ, p->num_nodes, K, dX, lddX, col_live);
  else
    k_segment_sum<false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p->nptr, Z, ldZ, p->num_nodes, K, dX, lddX, nullptr);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}


int segment_sum_arrays(const int32_t *nptr, int64_t num_nodes, int64_t nz_rows, const float *Z, int64_t ldZ, int K,
                       float *dX, int64_t lddX, hipStream_t s, const float *mask_src, int64_t ldMask, int lane_max) {
  if (num_nodes == 0) return MRGCN_OK;
  if (K <= 16) {  // every row of dX written (zeros for nodes without a row of Z), masked in the same pass
    const dim3 grid((unsigned)((num_nodes + 255) / 256));
    if (K <= 8)
      k_segment_sum_mask<8><<<grid, dim3(256), 0, s>>>(nptr, Z, ldZ, num_nodes, K, dX, lddX, nullptr, mask_src, ldMask,
                                                       nullptr, nz_rows, nullptr, lane_max);
    else
      k_segment_sum_mask<16><<<grid, dim3(256), 0, s>>>(nptr, Z, ldZ, num_nodes, K, dX, lddX, nullptr, mask_src, ldMask,
                                                        nullptr, nz_rows, nullptr, lane_max);
    MRGCN_HIP_TRY(hipGetLastError());
    return MRGCN_OK;
  }
  if (mask_src) {
    set_error("segment_sum: the masked form needs K <= 16");
    return MRGCN_ERR_UNSUPPORTED;
  }
  if (K > 64 && K <= 256 && (num_nodes + 3) / 4 <= 0x7fffffff) {  // wide rows: a wave per node, one-shot
    const dim3 grid((unsigned)((num_nodes + 3) / 4));
    if (K <= 128) k_segment_sum_wide<2><<<grid, dim3(256), 0, s>>>(nptr, Z, ldZ, num_nodes, K, dX, lddX);
    else if (K <= 192) k_segment_sum_wide<3><<<grid, dim3(256), 0, s>>>(nptr, Z, ldZ, num_nodes, K, dX, lddX);
    else k_segment_sum_wide<4><<<grid, dim3(256), 0, s>>>(nptr, Z, ldZ, num_nodes, K, dX, lddX);
    MRGCN_HIP_TRY(hipGetLastError());
    return MRGCN_OK;
  }
  int64_t blocks = (num_nodes * K + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  k_segment_sum<false><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(nptr, Z, ldZ, num_nodes, K, dX, lddX, nullptr);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // namespace mrgcn
