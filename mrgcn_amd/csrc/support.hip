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
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.hpp"
#include "config.hpp"

namespace mrgcn {
namespace {

constexpr int kSupTB = 512;
constexpr int kSqParts = 2048;  // per-block partial sums of ||dV||^2 (doubles), added in block order by k_dcomp_final

// F floats of one basis row (rows 4 F bytes apart: 8-byte aligned when F is even).  EXACT: F == FT, known at compile
// time — the loads are then straight-line code (a run-time F puts every piece behind a branch, and the waitcnt pass
// drains all outstanding loads at the first use behind a merge: DESIGN §3, "straight-line loads")
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
template <int FT, bool EXACT>
__device__ __forceinline__ void load_basis_row(const float *__restrict__ p, int F, float (&v)[FT]) {
  if constexpr (EXACT) {
    if constexpr ((FT & 1) == 0) {
#pragma unroll
      for (int o = 0; o + 4 <= FT; o += 4) {
        const f32x4_a8 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_a8 *>(p + o));  // (read once per epoch)
        v[o] = t.x; v[o + 1] = t.y; v[o + 2] = t.z; v[o + 3] = t.w;
      }
      if constexpr ((FT & 3) != 0) {
        typedef float f32x2_nt __attribute__((ext_vector_type(2)));
        const f32x2_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x2_nt *>(p + (FT & ~3)));
        v[FT & ~3] = t.x;
        v[(FT & ~3) + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int o = 0; o < FT; ++o) v[o] = p[o];
    }
    return;
  }
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
// range) pairs of 64 live nodes with one coalesced load each and walks them NB at a time: the loads of NB nodes — lane
// b its 4 F-byte row of each node's V block, the 16-lane group q row klo + q of each node's dM rows and its relation —
// are issued back to back at clamped addresses in one basic block, then the nodes are finished one after the other
// (the pass is bound by bytes in flight per wave: one node at a time it ran at 2.9 TB/s of its own traffic).  `comp`
// lives in LDS (the relation ids arrive with the loads; a dependent global load per column would put a second round
// trip on every node).  v_readlane hands a dM row and its relation round; lane b stores D[k][b] (a live column's B
// products: one 4 B-byte row per column, coalesced) and keeps the column's share of its dV row for the squared norm.
// Nodes with more than four live columns (9 % at the AM shape) walk the rest of their columns with direct loads.
// GC (round 6): `comp` is read from the (L2-resident) global table — lane b loads comp[r][b] of the node's first four
// live columns right behind their relation ids — instead of from an LDS image: the kernel then has no workgroup state
// and runs as a ONE-SHOT grid (a wave per 64 list entries) without re-staging 43 KB per block (`sup_mix_once = 2`).
template <int FT, int NB, int TB, bool EXACT, bool GC = false>
__global__ __launch_bounds__(TB) void k_mix_bwd_sup(const int32_t *__restrict__ lnode,
                                                        const int32_t *__restrict__ lnptr,
                                                        const int32_t *__restrict__ lrel,
                                                        const float *__restrict__ dM, int64_t ldM,
                                                        const float *__restrict__ V, const float *__restrict__ comp,
                                                        int64_t NL, int R, int B, int F_, float *__restrict__ D,
                                                        double *__restrict__ sq_part, int epw) {
  const int F = EXACT ? FT : F_;  // (a compile-time constant in the shapes that matter)
  extern __shared__ __align__(16) float s_comp[];  // [R][B]  (not with GC)
  if constexpr (!GC) {
    for (int t = threadIdx.x; t < R * B; t += blockDim.x) s_comp[t] = comp[t];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = blockDim.x >> 6;
  const bool on = lane < B;
  const int b = on ? lane : 0;
  const int kq = lane >> 4, oq = lane & 15;
  const int oqc = oq < F ? oq : F - 1;
  const int64_t nwaves = (int64_t)gridDim.x * nw;
  float sq = 0.f;
  // (`epw` <= 64 list entries per wave and round: 64 on a large support; a small one spreads its few entries over many
  // waves instead of walking 64 of them as one serial chain)
  for (int64_t base = ((int64_t)blockIdx.x * nw + wv) * epw; base < NL; base += nwaves * epw) {
    const int64_t me = (base + lane < NL) ? base + lane : NL - 1;
    const int32_t jn = lnode[me], k0v = lnptr[me], k1v = lnptr[me + 1];
    const int cnt = (int)((NL - base < epw) ? NL - base : epw);
    for (int i0 = 0; i0 < cnt; i0 += NB) {
      int32_t klo[NB], khi[NB], rmine[NB];
      float v[NB][FT], dmine[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {  // every load of the step, unconditional at clamped addresses
        const int ii = (i0 + u < cnt) ? i0 + u : cnt - 1;
        const int32_t j = __builtin_amdgcn_readlane(jn, ii);
        klo[u] = __builtin_amdgcn_readlane(k0v, ii);
        khi[u] = __builtin_amdgcn_readlane(k1v, ii);
        load_basis_row<FT, EXACT>(V + ((int64_t)j * B + b) * F, F, v[u]);
        const int32_t kk = (klo[u] + kq < khi[u]) ? klo[u] + kq : klo[u];
        dmine[u] = dM[(int64_t)kk * ldM + oqc];
        rmine[u] = lrel[kk];
      }
      float w4[NB][4];
      if constexpr (GC) {  // the comp rows of the first four live columns of every node of the step: unconditional loads
#pragma unroll
        for (int u = 0; u < NB; ++u)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            w4[u][t] = comp[(int64_t)__builtin_amdgcn_readlane(rmine[u], 16 * t) * B + b];
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if (i0 + u < cnt) {  // wave uniform
          float acc[FT];
#pragma unroll
          for (int o = 0; o < FT; ++o) acc[o] = 0.f;
          const int nc = (khi[u] - klo[u] < 4) ? khi[u] - klo[u] : 4;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (t < nc) {  // wave uniform
              const int r = __builtin_amdgcn_readlane(rmine[u], 16 * t);
              const float w = GC ? w4[u][t] : s_comp[r * B + b];
              float d[FT];
#pragma unroll
              for (int o = 0; o < FT; ++o)
                d[o] = (o < F) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dmine[u]),
                                                                                      16 * t + o))
                               : 0.f;
              float dot = 0.f;
#pragma unroll
              for (int o = 0; o < FT; ++o) dot = fmaf(d[o], v[u][o], dot);
              if (on) D[(int64_t)(klo[u] + t) * B + b] = dot;
#pragma unroll
              for (int o = 0; o < FT; ++o) acc[o] = fmaf(w, d[o], acc[o]);
            }
          }
          for (int32_t kb = klo[u] + 4; kb < khi[u]; kb += 4) {  // (few nodes have more than four live columns)
            const int32_t kk = (kb + kq < khi[u]) ? kb + kq : kb;
            const float dm = dM[(int64_t)kk * ldM + oqc];
            const int32_t rm = lrel[kk];
            const int nc2 = (khi[u] - kb < 4) ? khi[u] - kb : 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if (t < nc2) {
                const int r = __builtin_amdgcn_readlane(rm, 16 * t);
                const float w = GC ? comp[(int64_t)r * B + b] : s_comp[r * B + b];
                float d[FT];
#pragma unroll
                for (int o = 0; o < FT; ++o)
                  d[o] = (o < F) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dm), 16 * t + o))
                                 : 0.f;
                float dot = 0.f;
#pragma unroll
                for (int o = 0; o < FT; ++o) dot = fmaf(d[o], v[u][o], dot);
                if (on) D[(int64_t)(kb + t) * B + b] = dot;
#pragma unroll
                for (int o = 0; o < FT; ++o) acc[o] = fmaf(w, d[o], acc[o]);
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
    }
  }
  __shared__ float s_sq[TB / 64];
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
// (node band, relation) group).  A row of D is B floats = P = B / 4 16-byte pieces: lane l takes piece l % P of row
// slot l / P, so one load instruction brings 64 / P rows (six at B = 40) and eight of them are in flight per wave; the
// four waves take a quarter of the chunk each.  The row slots meet in LDS, added in a fixed order.
using f32x4s = __attribute__((ext_vector_type(4))) float;
template <int P>
__global__ __launch_bounds__(256) void k_dcomp_chunks(const int32_t *__restrict__ chunk_beg,
                                                      const int32_t *__restrict__ chunk_end,
                                                      const int32_t *__restrict__ lperm, const float *__restrict__ D,
                                                      float *__restrict__ slab) {
  constexpr int RPI = 64 / P;  // rows per load instruction
  constexpr int B = 4 * P;
  __shared__ f32x4s s_acc[4][RPI][P];
  const int chunk = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int slot = lane / P, piece = lane - slot * P;
  const bool on = slot < RPI;
  const int32_t beg = chunk_beg[chunk], end = chunk_end[chunk];
  const int32_t per = (end - beg + 3) >> 2;
  const int32_t a0 = beg + wv * per, a1 = (a0 + per < end) ? a0 + per : end;
  const f32x4s *D4 = reinterpret_cast<const f32x4s *>(D);
  constexpr int U = (64 / RPI) < 8 ? (64 / RPI) : 8;  // load instructions per step (U * RPI <= 64 row numbers)
  f32x4s acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int32_t e0 = a0; e0 < a1; e0 += U * RPI) {
    const int32_t mine = (e0 + lane < a1) ? lperm[e0 + lane] : 0;
    f32x4s x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = u * RPI + (on ? slot : 0);
      const int32_t k = __shfl(mine, t, 64);
      x[u] = D4[(int64_t)k * P + piece];  // (row 0 where the chunk has ended: masked below)
    }
#pragma unroll
    for (int u = 0; u < U; u += 2) {
      if (on && e0 + u * RPI + slot < a1) acc0 += x[u];
      if (u + 1 < U && on && e0 + (u + 1) * RPI + slot < a1) acc1 += x[u + 1 < U ? u + 1 : u];
    }
  }
  if (on) s_acc[wv][slot][piece] = acc0 + acc1;
  __syncthreads();
  if (threadIdx.x < (unsigned)B) {
    const int pc = threadIdx.x >> 2, el = threadIdx.x & 3;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int sl = 0; sl < RPI; ++sl) t += s_acc[w][sl][pc][el];
    slab[(int64_t)chunk * B + threadIdx.x] = t;
  }
}

// any B: lane = basis, one row per load instruction
__global__ __launch_bounds__(256) void k_dcomp_chunks_any(const int32_t *__restrict__ chunk_beg,
                                                          const int32_t *__restrict__ chunk_end,
                                                          const int32_t *__restrict__ lperm,
                                                          const float *__restrict__ D, int B,
                                                          float *__restrict__ slab) {
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

// dcomp[r][b] = sum of relation r's chunk slabs (wave w adds chunks w, w + 4, ... in order, the four partial sums
// meet in LDS; zeros for a relation without live columns); the block after the last relation adds the per-block parts
// of ||dV||^2
__global__ __launch_bounds__(256) void k_dcomp_final(const int32_t *__restrict__ chunk_ptr,
                                                     const int32_t *__restrict__ chunk_ids,
                                                     const float *__restrict__ slab, int R, int B,
                                                     float *__restrict__ dcomp, const double *__restrict__ sq_part,
                                                     int n_parts, double *__restrict__ sumsq) {
  __shared__ float s_part[4][64];
  const int r = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (r == R) {
    if (!sumsq || wv != 0) return;
    double t = 0.0;
    for (int i = lane; i < n_parts; i += 64) t += sq_part[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) *sumsq = t;
    return;
  }
  // wave w adds the relation's chunks w, w + 4, w + 8, ... — alternately into s0 and s1, each in rising order.  Sixteen
  // of them per step: their ids with one coalesced load, then sixteen slab rows in flight (a chunk at a time was two
  // dependent round trips each: 21 us for the 1 600 chunks of AM's self-loop relation)
  float s0 = 0.f, s1 = 0.f;
  {
    const int c0 = chunk_ptr[r] + wv, c1 = chunk_ptr[r + 1];
    const int K = c1 > c0 ? (c1 - c0 + 3) / 4 : 0;  // this wave's chunks
    const int bl = lane < B ? lane : B - 1;
    for (int k0 = 0; k0 < K; k0 += 16) {
      const int kk = k0 + (lane & 15);
      const int32_t idv = chunk_ids[c0 + 4 * (kk < K ? kk : K - 1)];
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) x[u] = slab[(int64_t)__builtin_amdgcn_readlane(idv, u) * B + bl];
#pragma unroll
      for (int u = 0; u < 16; u += 2) {
        if (k0 + u < K) s0 += x[u];
        if (k0 + u + 1 < K) s1 += x[u + 1];
      }
    }
  }
  s_part[wv][lane] = s0 + s1;
  __syncthreads();
  if (threadIdx.x < (unsigned)B)
    dcomp[(int64_t)r * B + threadIdx.x] =
        (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
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
  // D [L][B] (16-byte aligned rows when B % 4 == 0) | slab [chunks][B] | parts of ||dV||^2 (doubles, 8-byte aligned)
  return ((q->L + q->wide.n_chunks) * (int64_t)B + 3) / 4 * 4 + 2 * (int64_t)kSqParts;
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
    MRGCN_HIP_TRY(mrgcn::fill_async(dcomp, 0, (size_t)R * B * sizeof(float), s));
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
  const mrgcn_support::Order &o = q->wide;
  float *D = workspace;
  float *slab = D + q->L * (int64_t)B;
  double *sq_part = reinterpret_cast<double *>(workspace + ((q->L + o.n_chunks) * (int64_t)B + 3) / 4 * 4);
  const size_t lds = (size_t)R * B * sizeof(float);
  MRGCN_REQUIRE(lds <= 64 * 1024, "R * B * 4 must fit 64 KB of LDS");
  // nodes in flight per wave (NB) x waves per CU.  512-thread blocks with two nodes per step (62 VGPRs at F = 10: three
  // blocks of eight waves per CU, bounded by the 42 KB of comp in LDS); MRGCN_SUP_MIX = "<threads>x<nodes>" picks
  // another shape (experiments)
  // AM shape, kernel alone on one box: 512x1 598 us, 512x2 512, 512x4 557, 1024x2 501
  int tb = (int)cfg(CFG_SUP_MIX_TB), nb = (int)cfg(CFG_SUP_MIX_NB);
  if (nb < 1 || nb > 4) nb = 2;
  // (a small support: a wave's 64 list entries are one serial chain of 64 / NB steps — 122 us at MUTAG's 23 k nodes with
  // two per step; four per step halve the chain where occupancy does not matter)
  if (q->NL < 262144 && nb < 4) nb = 4;
  if (tb != 1024) tb = 512;
  if (tb == 1024 && nb > 2) nb = 2;
  const int nw = tb / 64;
  int epw = 64;
  if (q->NL < 262144) {  // ~4096 waves' worth of entries each, a multiple of the step
    epw = (int)((q->NL / 4096 + nb - 1) / nb * nb);
    epw = std::min(64, std::max(epw, nb));
  }
  int64_t grid = (q->NL + (int64_t)epw * nw - 1) / ((int64_t)epw * nw);
  int per_cu = (int)std::min<size_t>(tb == 1024 ? 2 : 4, (160 * 1024) / (lds + 1024));
  if (per_cu < 1) per_cu = 1;
  // (`sup_mix_once`: a ONE-SHOT grid — a wave takes one group of 64 list entries and ends; the block dispatcher hands the
  // groups out in address order: tools/lab/copy_lab.hip.  0: a resident grid striding through the list)
  const int once = (int)cfg(CFG_SUP_MIX_ONCE);  // 0: resident grid; 1: one-shot, LDS comp; 2: one-shot, comp from global
  if (once == 0 && grid > 256 * per_cu) grid = 256 * per_cu;
  if (grid > kSqParts) grid = kSqParts;
  if (grid < 1) grid = 1;
  const int FT = (F == 10 || F == 11) ? F : (F + 3) / 4 * 4;
#define SUP_GO3(T, NB_, TB_)                                                                                        \
  do {                                                                                                              \
    if (F == T) SUP_GO4(T, NB_, TB_, true);                                                                         \
    else SUP_GO4(T, 1, 512, false);                                                                                 \
  } while (0)
#define SUP_GO4(T, NB_, TB_, EX_)                                                                                   \
  do {                                                                                                              \
    if (once == 2 && EX_) {                                                                                         \
      k_mix_bwd_sup<T, NB_, TB_, EX_, true><<<dim3((unsigned)grid), dim3(TB_), 0, s>>>(                             \
          q->lnode, q->lnptr, q->lrel, dM, ldM, V, comp, q->NL, R, B, F, D, sq_part, epw);                          \
    } else {                                                                                                        \
      auto kfn = k_mix_bwd_sup<T, NB_, TB_, EX_>;                                                                   \
      MRGCN_HIP_TRY(mrgcn::raise_lds_limit((const void *)kfn, lds));                                               \
      kfn<<<dim3((unsigned)grid), dim3(TB_), lds, s>>>(q->lnode, q->lnptr, q->lrel, dM, ldM, V, comp, q->NL, R, B,  \
                                                       F, D, sq_part, epw);                                         \
    }                                                                                                               \
  } while (0)
#define SUP_GO(T)                                  \
  do {                                             \
    if (tb == 1024 && nb >= 2) SUP_GO3(T, 2, 1024); \
    else if (tb == 1024) SUP_GO3(T, 1, 1024);      \
    else if (nb >= 4) SUP_GO3(T, 4, 512);          \
    else if (nb >= 2) SUP_GO3(T, 2, 512);          \
    else SUP_GO3(T, 1, 512);                       \
  } while (0)
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
#undef SUP_GO3
#undef SUP_GO4
  MRGCN_HIP_TRY(hipGetLastError());
  if (o.n_chunks > 0) {
    const dim3 cg((unsigned)o.n_chunks);
    const bool vec = (B % 4 == 0) && (((uintptr_t)D) % 16 == 0);
    if (vec && B == 40) k_dcomp_chunks<10><<<cg, dim3(256), 0, s>>>(o.chunk_beg, o.chunk_end, o.lperm, D, slab);
    else if (vec && B == 32) k_dcomp_chunks<8><<<cg, dim3(256), 0, s>>>(o.chunk_beg, o.chunk_end, o.lperm, D, slab);
    else if (vec && B == 16) k_dcomp_chunks<4><<<cg, dim3(256), 0, s>>>(o.chunk_beg, o.chunk_end, o.lperm, D, slab);
    else if (vec && B == 8) k_dcomp_chunks<2><<<cg, dim3(256), 0, s>>>(o.chunk_beg, o.chunk_end, o.lperm, D, slab);
    else k_dcomp_chunks_any<<<cg, dim3(256), 0, s>>>(o.chunk_beg, o.chunk_end, o.lperm, D, B, slab);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  k_dcomp_final<<<dim3((unsigned)(R + 1)), dim3(256), 0, s>>>(o.chunk_ptr, o.chunk_ids, slab, R, B, dcomp, sq_part,
                                                             q->NL > 0 ? (int)grid : 0, dV_sumsq);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_support_adam_rows_fused_f32(const mrgcn_support_t *q, const float *dM, int64_t ldM, const float *comp,
                                      int32_t B, int32_t F, float *param, float *exp_avg, float *exp_avg_sq,
                                      uint8_t *row_ever, float lr, float beta1, float beta2, float eps, int64_t step,
                                      const float *bc_dev, const float *grad_scale, int32_t ever_outside,
                                      void *stream) {
  MRGCN_REQUIRE(q && dM && comp && param && exp_avg && exp_avg_sq && row_ever, "NULL");
  MRGCN_REQUIRE(mrgcn_adam_rows_fused_supported(q->plan, B, F), "shape outside mrgcn_adam_rows_fused_supported");
  MRGCN_REQUIRE(ldM >= F, "ldM");
  MRGCN_REQUIRE(((((uintptr_t)param) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15) == 0,
                "param / moments must be 16-byte aligned");
  MRGCN_REQUIRE(bc_dev || step >= 1, "step");
  const mrgcn_plan *p = q->plan;
  return adam_rows_fused_arrays(q->nlptr, q->lrel, nullptr, p->num_nodes, (int)p->num_relations, dM, ldM, comp, B, F,
                                param, exp_avg, exp_avg_sq, q->node_flags, row_ever, lr, beta1, beta2, eps, step,
                                bc_dev, grad_scale, (hipStream_t)stream, q->lnode, q->lnptr, q->NL, ever_outside);
}

int64_t mrgcn_support_rel_transform_bwd_workspace(const mrgcn_support_t *q, int32_t K, int32_t F, int32_t need_dX,
                                                  int32_t need_dW) {
  if (!q) return -1;
  if (!xform_use_mfma()) return -1;
  if (need_dW && !xform_mfma_dw_supported(K, F)) return -1;
  if (need_dX && !(xform_mfma_dx_supported(F, K))) return -1;
  const int64_t a = need_dX ? q->L * (((int64_t)K + 3) / 4 * 4) : 0;
  const int64_t b = need_dW ? (int64_t)q->order_for(K).n_relchunks * K * F : 0;
  const int64_t m = a > b ? a : b;
  return m > 0 ? m : 1;
}

static int support_rel_transform_bwd(const mrgcn_support_t *q, const float *dM, int64_t ldM, const void *X_, bool x_bf16,
                                     int64_t ldX, int32_t K, const float *W, int32_t F, float *dX, int64_t lddX,
                                     float *dW, float *workspace, int64_t workspace_floats,
                                     int32_t relu_mask_from_x, void *stream) {
  const float *X = (const float *)X_;
  MRGCN_REQUIRE(q && dM && X && W, "NULL");
  MRGCN_REQUIRE(!(x_bf16 && relu_mask_from_x), "the ReLU mask is read from fp32 rows");
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
    int rc = xform_mfma_dw(p, o, o.rnode, X_, ldX, K, dM, ldM, F, dW, workspace, workspace_floats, s, nullptr, x_bf16);
    if (rc != MRGCN_OK) return rc;
  }
  if (dX) {
    MRGCN_REQUIRE(lddX >= K, "lddX");
    const int64_t ldZ = ((int64_t)K + 3) / 4 * 4;
    // Z[k, 0:K] = dM[k, 0:F] . W[r_k]^T on the matrix cores, then dX[j] = sum of node j's Z rows (every row written)
    int rc = xform_mfma_fwd(p, q->order_for(F), nullptr, nullptr, dM, ldM, F, W, true, K, workspace, ldZ, s, false,
                            nullptr);
    if (rc != MRGCN_OK) return rc;
    // (a node of up to 8 live columns is summed by its own lane: the wave-cooperative walk takes the live nodes of a
    // wave one after the other — 41 us at the AM shape and at AIFB's 8 k nodes alike)
    rc = segment_sum_arrays(q->nlptr, p->num_nodes, q->L, workspace, ldZ, K, dX, lddX, s,
                            relu_mask_from_x ? X : nullptr, ldX, 8);
    if (rc != MRGCN_OK) return rc;
  }
  return MRGCN_OK;
}

int mrgcn_support_rel_transform_bwd_f32(const mrgcn_support_t *q, const float *dM, int64_t ldM, const float *X,
                                        int64_t ldX, int32_t K, const float *W, int32_t F, float *dX, int64_t lddX,
                                        float *dW, float *workspace, int64_t workspace_floats,
                                        int32_t relu_mask_from_x, void *stream) {
  return support_rel_transform_bwd(q, dM, ldM, X, false, ldX, K, W, F, dX, lddX, dW, workspace, workspace_floats,
                                   relu_mask_from_x, stream);
}

// the same with the layer's input in bf16 rows (the bf16 pipeline's X: ldX in elements); dW's sums stay fp32
int mrgcn_support_rel_transform_bwd_xbf16(const mrgcn_support_t *q, const float *dM, int64_t ldM, const uint16_t *X,
                                          int64_t ldX, int32_t K, const float *W, int32_t F, float *dX, int64_t lddX,
                                          float *dW, float *workspace, int64_t workspace_floats, void *stream) {
  return support_rel_transform_bwd(q, dM, ldM, X, true, ldX, K, W, F, dX, lddX, dW, workspace, workspace_floats, 0,
                                   stream);
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
