// EXPERIMENT RECORD (round 5, second attempt at the layer-0 transform) — NOT part of the library, not compiled by
// mrgcn_amd/build.py.  Needs, in the plan, the relations ranked by column count (`rel_rank`, chunks grouped by rank) and
// in k_xform_mfma_fwd a `chunk_ids` indirection so that the relation-major kernel takes the other ranks only; both were
// in the tree with it (see the commit that adds this file) and went out with it.
//
// Idea (the round-4 verdict's 3d): the dozen relations that own half of the compact columns (the identity block alone
// a fifth) are taken NODE-TILE-MAJOR — a persistent workgroup keeps their transposed weight tiles in LDS (13 x 6.6 KB),
// streams tiles of 64-96 consecutive X rows through registers into LDS, deals the tile's columns into one list per hot
// relation and multiplies 16 listed columns at a time out of LDS — so that an X row is read once for all of them.  The
// other relations stay relation-major.  Bit-equal to the relation-major kernel on every column (same MFMA, same k
// order: tests/test_gpu_xform_hot.py of that commit, 7 shapes x both output orders x f32 / bf16 rows).
//
// Measured at the AM shape (8.17 M columns, X 1.03 GB; relation-major kernel alone: 1 020-1 060 us):
//
//   NH = 13, tiles of 96 nodes, 1 workgroup / CU     hot 560 us + cold 615 us = 1 175 us
//   NH = 13, tiles of 64                             1 318 us          NH = 8: 1 254 us     NH = 4: hot 418 + cold 825
//   NH = 8, tiles of 32, 2 workgroups / CU           1 201 us          NH = 4, tiles of 64, 2 / CU: 1 204 us
//   prefetch depth 1 / 2 / 3 tiles                   no difference beyond noise
//
// Why it does not pay: the relation-major kernel's 5.9 GB of gathers are served at 5.7 TB/s — mostly by the Infinity
// Cache: a band of 131 072 nodes is 81 MB of X, the identity block's chunks stream it in and the band's other relations
// hit it.  The hot kernel replaces 2.4 GB of those cache-served gathers by ONE 1.03 GB stream from HBM (4.7 TB/s at
// best: >= 220 us, 380-560 us as built) and takes the identity block away from the cold kernel, whose first touch of
// every X row is then a random HBM access instead of a cache hit (cold: 2.65 GB in 615 us = 4.3 TB/s).  Best case on
// paper: 0.27 + 0.5 ms.  X rows padded to whole 128-byte lines (ld = 160) leave the relation-major kernel at the same
// 1 016 us: it is bound by 64-byte sectors through the fabric, not by lines.
#include "common.hpp"
#include "config.hpp"

namespace mrgcn {
namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ f32x4 load4_fast(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

// ---------------------------------------------------------------------------------------------
// Node-tile-major forward of the NH relations with the most columns (wide inputs: layer 0's X rows of 620 bytes).
//
// The relation-major kernel above gathers a node's input row once per relation of the node: 8.17 M gathers of
// 620 bytes for a 1.03 GB X at the AM shape, every one of them through the fabric (the rows of a 131 072-node band
// do not fit an L2).  Half of those gathers belong to a dozen relations (the identity block alone is a fifth).  Here
// a persistent workgroup keeps the transposed weight tiles of those NH relations in LDS for its whole life and walks
// tiles of TILE consecutive source nodes: the tile's input rows arrive as ONE contiguous stream (registers, D tiles
// ahead, then LDS), the tile's compact columns — a contiguous range of (node, relation) pairs — are dealt into one
// list per hot relation with LDS counters (a node has at most one column per relation: TILE slots per list), and
// every wave multiplies 16 listed columns at a time out of LDS.  An input row is read from memory once for all hot
// relations; the columns of the other relations stay with the relation-major kernel (chunk list by rank).
//
// Same MFMA, same k-slot assignment and the same instruction order per column as k_xform_mfma_fwd: the same bits.
// LDS: Ws [NH][F][KP] | Xs [TILE][KP] | l_o int32 [NH][TILE] | cnt int32 [2][16] | hrel int32 [16] |
//      l_n uint8 [NH][TILE] | rank uint8 [R]
// ---------------------------------------------------------------------------------------------
constexpr int kHotTB = 512;
constexpr int kHotMaxNH = 16;

template <int NP, int D, typename OT>
__global__ __launch_bounds__(kHotTB) void k_xform_hot_fwd(
    const int32_t *__restrict__ nptr, const int32_t *__restrict__ urel, const int32_t *__restrict__ unode,
    const int32_t *__restrict__ mpos /* nullable: output row of column c (null: c) */,
    const int32_t *__restrict__ rel_rank, const float *__restrict__ In, int64_t N, int R, int K,
    const float *__restrict__ W, int F, int NH, int TILE, int n_tiles, OT *__restrict__ Out, int64_t ldOut) {
  extern __shared__ __align__(16) float s_hot[];
  const int ksteps = (K + 15) >> 4;
  const int KP = ksteps * 16 + 4;
  float *Ws = s_hot;
  float *Xs = Ws + NH * F * KP;
  int32_t *l_o = reinterpret_cast<int32_t *>(Xs + TILE * KP);
  int32_t *s_cnt = l_o + NH * TILE;  // [2][16]
  int32_t *s_hrel = s_cnt + 32;      // [16]
  uint8_t *l_n = reinterpret_cast<uint8_t *>(s_hrel + 16);
  uint8_t *s_rank = l_n + NH * TILE;  // [R]: hot slot of a relation, 255 = not hot
  const int tid = threadIdx.x;
  const int G = gridDim.x;

  for (int r = tid; r < R; r += kHotTB) {
    const int k = rel_rank[r];
    s_rank[r] = k < NH ? (uint8_t)k : (uint8_t)255;
    if (k < NH) s_hrel[k] = r;
  }
  if (tid < 32) s_cnt[tid] = 0;
  for (int t = tid; t < TILE * (KP - K); t += kHotTB) {  // the k pad of the input tile: zeros, never rewritten
    const int row = t / (KP - K), k = K + (t - row * (KP - K));
    Xs[row * KP + k] = 0.f;
  }
  __syncthreads();
  for (int t = tid; t < NH * F * KP; t += kHotTB) {
    const int h = t / (F * KP), rem = t - h * (F * KP);
    const int n = rem / KP, k = rem - n * KP;
    Ws[t] = k < K ? W[((int64_t)s_hrel[h] * K + k) * F + n] : 0.f;
  }

  const int64_t total = N * (int64_t)K;
  const int tile_f4 = (TILE * K) >> 2;  // (TILE % 4 == 0)
  f32x4 xr[D][NP];
  auto load_tile = [&](int t, f32x4 (&dst)[NP]) {
    if (t >= n_tiles) return;
    const int64_t base = (int64_t)t * TILE * K;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int pc = tid + kHotTB * q;
      if (pc < tile_f4) {
        const int64_t e = base + 4 * (int64_t)pc;
        if (e + 4 <= total) dst[q] = load4_fast(In + e);
        else {
          dst[q].x = e + 0 < total ? In[e + 0] : 0.f;
          dst[q].y = e + 1 < total ? In[e + 1] : 0.f;
          dst[q].z = e + 2 < total ? In[e + 2] : 0.f;
          dst[q].w = e + 3 < total ? In[e + 3] : 0.f;
        }
      }
    }
  };
  // the tile's first 2 x 512 columns: relation, source node, output row — one tile ahead, in registers
  int32_t ir[2], in_[2], io[2];
  auto load_idx = [&](int32_t cA, int32_t cB) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int32_t c = cA + tid + kHotTB * i;
      const int32_t cc = c < cB ? c : (cB > cA ? cB - 1 : 0);
      ir[i] = urel[cc];
      in_[i] = unode[cc];
      io[i] = mpos ? mpos[cc] : cc;
    }
  };
  auto range_of = [&](int t, int32_t &cA, int32_t &cB) {
    if (t < n_tiles) {
      const int64_t j0 = (int64_t)t * TILE, j1 = min(j0 + TILE, N);
      cA = nptr[j0];
      cB = nptr[j1];
    } else {
      cA = cB = 0;
    }
  };
  int t_cur = blockIdx.x;
#pragma unroll
  for (int d = 0; d < D; ++d) load_tile(t_cur + d * G, xr[d]);
  int32_t cA, cB, cA1, cB1;
  range_of(t_cur, cA, cB);
  range_of(t_cur + G, cA1, cB1);
  load_idx(cA, cB);

  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, kq = lane >> 4;
  const int mw = m < F ? m : F - 1;
  int cur = 0;
  for (; t_cur < n_tiles; t_cur += G, cur ^= 1) {
    __syncthreads();  // the previous tile's products are done with Xs and the lists (first pass: Ws is complete)
    // ---- this tile's input rows: registers -> Xs[row][k] ----
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int pc = tid + kHotTB * q;
      if (pc < tile_f4) {
        const int e = 4 * pc;
        int row = e / K, k = e - row * K;
        const f32x4 v = xr[0][q];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          Xs[row * KP + k] = v[i];
          if (++k == K) { k = 0; ++row; }
        }
      }
    }
    // ---- the tile's columns -> one list per hot relation ----
    const int32_t j0 = (int32_t)((int64_t)t_cur * TILE);
    int32_t *cnt = s_cnt + 16 * cur;
    auto add = [&](int32_t r, int32_t node, int32_t orow) {
      const int h = s_rank[r];
      if (h != 255) {
        const int pos = atomicAdd(&cnt[h], 1);
        l_o[h * TILE + pos] = orow;
        l_n[h * TILE + pos] = (uint8_t)(node - j0);
      }
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (cA + tid + kHotTB * i < cB) add(ir[i], in_[i], io[i]);
    for (int32_t c = cA + 2 * kHotTB + tid; c < cB; c += kHotTB) add(urel[c], unode[c], mpos ? mpos[c] : c);
    __syncthreads();
    // ---- loads of the tiles to come (they fly under the products) ----
    if (tid < 16) s_cnt[16 * (cur ^ 1) + tid] = 0;
#pragma unroll
    for (int d = 0; d + 1 < D; ++d)
#pragma unroll
      for (int q = 0; q < NP; ++q) xr[d][q] = xr[d + 1][q];
    load_tile(t_cur + D * G, xr[D - 1]);
    cA = cA1;
    cB = cB1;
    load_idx(cA, cB);
    range_of(t_cur + 2 * G, cA1, cB1);
    // ---- products: work item = 16 listed columns of one hot relation; the waves take items round robin ----
    const int my_cnt = lane < NH ? cnt[lane] : 0;
    const int my_tiles = (my_cnt + 15) >> 4;
    int incl = my_tiles;
#pragma unroll
    for (int off = 1; off < kHotMaxNH; off <<= 1) {
      const int v = __shfl_up(incl, off, 64);
      if (lane >= off) incl += v;
    }
    const int start = incl - my_tiles;
    const int items = __shfl(incl, kHotMaxNH - 1, 64);
    for (int it = wv; it < items; it += kHotTB / 64) {
      const uint64_t mask = __ballot(lane < NH && start <= it);
      const int h = __builtin_amdgcn_readfirstlane(63 - __clzll(mask));
      const int q16 = it - __shfl(start, h, 64);
      const int n_h = __shfl(my_cnt, h, 64);
      const int ci = 16 * q16 + m;
      const bool valid = ci < n_h;
      const int32_t orow = valid ? l_o[h * TILE + ci] : -1;
      const int nl = valid ? (int)l_n[h * TILE + ci] : 0;
      const float *xa = Xs + nl * KP + 4 * kq;
      const float *wb = Ws + (h * F + mw) * KP + 4 * kq;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int ks = 0; ks < ksteps; ++ks) {
        const f32x4 av = *reinterpret_cast<const f32x4 *>(xa + ks * 16);
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(wb + ks * 16);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t o_r = __shfl(orow, 4 * kq + reg, 64);
        if (o_r >= 0 && m < ldOut) store_operand<OT>(Out + o_r * ldOut + m, m < F ? acc[reg] : 0.f);
      }
    }
  }
}

}  // namespace

// ---- node-tile-major forward of the hot relations ------------------------------------------------
namespace {
size_t hot_lds_bytes(int NH, int TILE, int K, int F, int R) {
  const int KP = (K + 15) / 16 * 16 + 4;
  return (size_t)(NH * F * KP + TILE * KP) * 4 + (size_t)NH * TILE * 5 + (32 + 16) * 4 + (size_t)R + 16;
}
}  // namespace

// > 0: the number of hot relations to run node-tile-major (and *tile / *depth); 0: not applicable
int xform_hot_plan(const mrgcn_plan *p, int K, int F, const float *In, int64_t ldIn, int *tile, int *depth) {
  const int64_t mode = cfg(CFG_XFORM_HOT);  // 1: large graphs only; 2: whenever the shape allows
  if (!mode || !p || !p->rel_rank || p->h_rank_cols.empty()) return 0;
  if (K < 64 || K > kMaxKSteps * 16 || F > 16 || ldIn != K || ((uintptr_t)In & 15)) return 0;
  const int64_t N = p->num_nodes, R = p->num_relations;
  if (R > 4096) return 0;
  if (mode == 1 && N < 65536) return 0;
  int T = (int)cfg(CFG_XFORM_HOT_TILE);
  T = std::max(16, std::min(T, 256)) & ~3;
  while (T > 16 && (T * K / 4 + kHotTB - 1) / kHotTB > 8) T -= 4;  // <= 8 16-byte pieces per thread and tile
  if ((T * K / 4 + kHotTB - 1) / kHotTB > 8) return 0;
  int NH = (int)std::min<int64_t>(kHotMaxNH, R);
  const int64_t want = cfg(CFG_XFORM_HOT_NH);
  if (want > 0) NH = (int)std::min<int64_t>(NH, want);
  while (NH > 0 && hot_lds_bytes(NH, T, K, F, (int)R) > 160 * 1024) --NH;
  // a list of fewer than ~3 columns per tile costs more matrix-core time than its gathers cost the fabric
  while (NH > 0 && mode == 1 && p->h_rank_cols[NH - 1] * T < 3 * N) --NH;
  while (NH > 0 && p->h_rank_cols[NH - 1] == 0) --NH;
  if (NH == 0) return 0;
  int64_t hot = 0;
  for (int k = 0; k < NH; ++k) hot += p->h_rank_cols[k];
  if (mode == 1 && hot * 4 < p->ncols) return 0;
  int Dp = (int)cfg(CFG_XFORM_HOT_DEPTH);
  *tile = T;
  *depth = std::max(1, std::min(Dp, 3));
  return NH;
}

int xform_hot_fwd(const mrgcn_plan *p, int NH, int tile, int depth, bool operand_order, const float *In, int K,
                  const float *W, int F, void *Out, int64_t ldOut, hipStream_t s, bool out_bf16) {
  const int64_t N = p->num_nodes;
  if (N == 0 || p->ncols == 0) return MRGCN_OK;
  if (ldOut > 16) { set_error("xform_hot_fwd: rows of at most 16 elements"); return MRGCN_ERR_UNSUPPORTED; }
  const int R = (int)p->num_relations;
  const int n_tiles = (int)((N + tile - 1) / tile);
  const size_t lds = hot_lds_bytes(NH, tile, K, F, R);
  const int per_cu = std::max<int>(1, std::min<int>(4, (int)((160 * 1024) / (lds + 512))));  // workgroups a CU's LDS holds
  const int grid = std::min(n_tiles, 256 * per_cu);
  const int np = (tile * K / 4 + kHotTB - 1) / kHotTB;
  const int32_t *mp = operand_order ? p->mpos : nullptr;
#define HOT_GO(NP_, D_, O_)                                                                                        \
  do {                                                                                                             \
    auto kfn = k_xform_hot_fwd<NP_, D_, O_>;                                                                       \
    static size_t lds_allowed = 48 * 1024;                                                                         \
    if (lds > lds_allowed) {                                                                                       \
      MRGCN_HIP_TRY(hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      lds_allowed = lds;                                                                                           \
    }                                                                                                              \
    kfn<<<dim3((unsigned)grid), dim3(kHotTB), lds, s>>>(p->nptr, p->urel, p->unode, mp, p->rel_rank, In, N, R, K,  \
                                                        W, F, NH, tile, n_tiles, (O_ *)Out, ldOut);                \
  } while (0)
#define HOT_D(NP_, O_)                                                                       \
  do {                                                                                       \
    if (depth <= 1) HOT_GO(NP_, 1, O_); else if (depth == 2) HOT_GO(NP_, 2, O_); else HOT_GO(NP_, 3, O_); \
  } while (0)
#define HOT_NP(O_)                                                                           \
  do {                                                                                       \
    if (np <= 4) HOT_D(4, O_); else if (np <= 6) HOT_D(6, O_); else HOT_D(8, O_);            \
  } while (0)
  if (out_bf16) HOT_NP(uint16_t); else HOT_NP(float);
#undef HOT_NP
#undef HOT_D
#undef HOT_GO
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // namespace mrgcn
