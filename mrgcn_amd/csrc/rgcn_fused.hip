// Fused construction of the compact dense operand M (one row per *touched* column of A)
// and its backward — the MI355X replacement for the reference's materialisation of the
// (R*N) x out operands W_I and FW_F (mrgcn/layers/graph.py:69-72, :83-85, :93-94) and of
// their dense gradients (autograd of the same lines).
//
//   M[mpos[c], :] = sum_b comp_I[r_c, b] * V_I[b, j_c, :]      basis mix   (input term)
//                 + X[j_c, :] . W_F[r_c]                        relation transform (feature term)
//
// Compact columns are numbered in (source node j, relation r) order, so everything that
// belongs to one node is contiguous.  The basis table is NODE-MAJOR here, V[j, b, :] ([N][B][F]: the
// B rows of a node are one contiguous block of B*F floats) — the internal layout of the layer's
// weight_I parameter; the reference's (B*N, out) tensor (graph.py:50-51, :69-72) is its [B][N][F]
// transpose, produced / consumed by the layer's state-dict hooks.  V is streamed exactly once, a node
// without gradient is skipped as one block (backward, Adam) and dV needs no atomics.  The per-relation dense
// transforms run relation-major on the matrix cores (xform_mfma.hip).  When both terms are
// present the transform writes its rows in compact order (sequential) and the mix pass adds
// them while it emits the final rows in the operand's storage order `mpos` — the only
// scattered traffic of the forward is that one write-only pass of whole padded rows.
#include <cmath>
#include <cstdlib>

#include <mutex>
#include <unordered_map>

#include "common.hpp"
#include "config.hpp"

namespace mrgcn {
namespace {

constexpr int kTB = 256;
constexpr int kMixTB = 512;  // mix kernels: 8 waves share one LDS copy of comp
// forward mix: 1024-thread blocks, i.e. 128 registers per thread and one 16-wave block per CU.
// Measured 4 % faster (1.24 vs 1.30 ms, AM shape) than 512-thread blocks at 64 registers / 24 waves:
// the 40 basis values per thread plus a chunk of prefetched indices want the registers more than
// the pass wants waves (forcing 64 registers at 32 waves spills: 6 ms).
constexpr int kMixFwdTB = 1024;
constexpr int kPre = 8;

// =====================================================================================
// basis mix, forward.  thread = (node j, padded feature o < FW); V[j, ., o] in registers.
//   M[mpos[c], o] = (addend ? addend[c, o] : 0) (+ old value if accumulate)
//                   + sum_b comp[r_c, b] * V[j, b, o]          for o < F;  0 for F <= o < FW
// =====================================================================================
// LDS row stride of the staged comp slice: multiple of 4 floats (16-byte ds_read_b128) and
// = 12 mod 32 banks when BT = 40, so that the 4-6 relations a wave touches at once land on
// different banks
__host__ __device__ constexpr int comp_stride(int BT) { return BT >= 4 ? BT + 4 : BT; }

template <int BT>
__device__ __forceinline__ void load_comp_row(const float *row, float (&w)[BT]) {
  if constexpr (BT % 4 == 0) {
#pragma unroll
    for (int b = 0; b < BT; b += 4) {
      const float4 q = *reinterpret_cast<const float4 *>(row + b);
      w[b] = q.x; w[b + 1] = q.y; w[b + 2] = q.z; w[b + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int b = 0; b < BT; ++b) w[b] = row[b];
  }
}

template <int BT, typename OT>
__global__ __launch_bounds__(kMixFwdTB) void k_mix_fwd(const int32_t *__restrict__ nptr,
                                                    const int32_t *__restrict__ urel,
                                                    const int32_t *__restrict__ mpos,
                                                    const float *__restrict__ V,
                                                    const float *__restrict__ comp, int64_t N, int R,
                                                    int B, int b0, int F, int FW,
                                                    const float *__restrict__ addend, int64_t ldA,
                                                    OT *__restrict__ M, int64_t ldM, int accumulate,
                                                    int comp_in_lds,
                                                    const int32_t *__restrict__ node_ids = nullptr) {
  extern __shared__ __align__(16) float s_comp[];  // [R][CS] slice b0..b0+BT of comp when it fits
  constexpr int CS = comp_stride(BT);
  const int nb = min(BT, B - b0);
  if (comp_in_lds) {
    for (int t = threadIdx.x; t < R * BT; t += blockDim.x) {
      int r = t / BT, b = t - r * BT;
      s_comp[r * CS + b] = (b < nb) ? comp[(int64_t)r * B + b0 + b] : 0.f;
    }
    __syncthreads();
  }
  const int64_t total = N * FW;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = t / FW;
    const int o = (int)(t - j * FW);
    const bool live = o < F;
    const int32_t c0 = nptr[j], c1 = nptr[j + 1];
    if (c0 == c1) continue;
    float v[BT];
    const int64_t jv = node_ids ? (int64_t)node_ids[j] : j;  // (a node list: entry j of it owns columns nptr[j] ..)
#pragma unroll
    for (int b = 0; b < BT; ++b)
      v[b] = (live && b < nb) ? V[(jv * B + (b0 + b)) * F + o] : 0.f;

    // the node's columns in chunks of kPre: everything a chunk needs (relation id, operand row,
    // addend) is requested up front, so a chunk costs one round trip however long the node is
    for (int32_t cb = c0; cb < c1; cb += kPre) {
      int rr[kPre];
      int32_t pp[kPre];
      float aa[kPre];
#pragma unroll
      for (int i = 0; i < kPre; ++i) {
        const int32_t c = (cb + i < c1) ? cb + i : cb;
        rr[i] = urel[c];
        pp[i] = mpos ? mpos[c] : c;
        aa[i] = (addend && live) ? addend[(int64_t)c * ldA + o] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < kPre; ++i) {
        if (cb + i < c1) {
          float s = aa[i];
          if (comp_in_lds) {
            float w[BT];
            load_comp_row<BT>(s_comp + rr[i] * CS, w);
#pragma unroll
            for (int b = 0; b < BT; ++b) s = fmaf(w[b], v[b], s);
          } else {
            const float *cr = comp + (int64_t)rr[i] * B + b0;
#pragma unroll
            for (int b = 0; b < BT; ++b)
              if (b < nb) s = fmaf(cr[b], v[b], s);
          }
          OT *m = M + (int64_t)pp[i] * ldM + o;
          if (accumulate) s += load_operand<OT>(m);
          if constexpr (sizeof(OT) == 2) {
            // bf16 rows: two features per 4-byte store (2-byte scattered stores were measured
            // 30 % slower than the fp32 rows they replace).  Lanes o and o + 1 belong to the same
            // node (FW is even), so they take this branch together.
            const float mine = live ? s : 0.f;
            const float next = __shfl_down(mine, 1, kWave);
            if ((o & 1) == 0 && (FW & 1) == 0) {
              *reinterpret_cast<uint32_t *>(m) = (uint32_t)f32_to_bf16(mine) | ((uint32_t)f32_to_bf16(next) << 16);
            } else if (FW & 1) {
              store_operand<OT>(m, mine);
            }
          } else {
            store_operand<OT>(m, live ? s : 0.f);
          }
        }
      }
    }
  }
}

// =====================================================================================
// basis mix, forward, ON THE MATRIX CORES (B <= 64, F <= 16, B*F % 4 == 0): per source node j the rows of
// its columns are a small dense product
//     M_j [ncols_j x F] = C_j [ncols_j x B] . V_j [B x F],      C_j[m, :] = comp[r of the node's m-th column, :]
// PMC of the scalar form above (thread = (node, feature), 40 strided loads + 40 FMAs per column) showed the
// texture addresser 67 % and the VALU 52 % busy for 3.5 GB of traffic at 3.0 TB/s.  Here a wave takes kFwdTN
// nodes per step: their V blocks enter a per-wave LDS tile as 16-byte pieces in lane order, the comp table
// lives transposed-free in LDS ([R][K padded to 16s], k contiguous), and v_mfma_f32_16x16x4_f32 (exact fp32)
// multiplies up to 16 columns of a node at a time: lane (m, kq) feeds A = comp[r_m][16 ks + 4 kq ..] (one
// 16-byte LDS read per K step) and B = V_j[16 ks + 4 kq + s][n] (the same for every column tile of the node).
// The result lane (n, q) holds columns 4 q + reg: addend added, row stored at its operand position.
// =====================================================================================
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

using f32x4m = __attribute__((ext_vector_type(4))) float;
constexpr int kFwdTB = 1024;  // 16 waves share one LDS copy of comp (R x 52 floats is most of a CU's LDS at R ~ 270)

// the tiles of one node: columns [c0, c1), 16 per MFMA tile.  NEAR: the relation / position words of the
// step's columns sit in the lanes of `ur` / `mp` (column `base` + lane) and are fetched with lane permutes,
// and the addend words of the node's (single) tile sit in `pa`; the products then hold no global load at all,
// so the waits the compiler places never drain the loads of the NEXT step (a load merged into the same
// registers would: the wait sits at the use, after the merge).
// ADD: 0 no addend; 1 the addend words of a NEAR node's tile in `pa`; 2 the addend rows of the step's columns staged
// in the wave's LDS piece `s_add` (row of column `base` first, ldA floats per row)
// AT: element type of the addend rows (float, or uint16_t = bf16: the bf16 pipeline's feature term)
template <int KS, bool NEAR, int ADD, typename OT, typename AT = float>
__device__ __forceinline__ void mix_node_tiles(int32_t c0, int32_t c1, int32_t base, int32_t ur, int32_t mp,
                                               const float (&pa)[4], const int32_t *__restrict__ urel,
                                               const int32_t *__restrict__ mpos, const float *s_comp,
                                               const float *s_v, int B, int F, const AT *__restrict__ addend,
                                               int64_t ldA, OT *__restrict__ M, int64_t ldM, int m, int kq,
                                               const AT *s_add = nullptr) {
  constexpr int KP = KS * 16 + 4;
  // B operand of this node: V_j[16 ks + 4 kq + s][m]
  f32x4m bv[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int k0 = ks * 16 + 4 * kq;
    const float *t = s_v + k0 * F + m;
    bv[ks].x = (m < F && k0 + 0 < B) ? t[0] : 0.f;
    bv[ks].y = (m < F && k0 + 1 < B) ? t[F] : 0.f;
    bv[ks].z = (m < F && k0 + 2 < B) ? t[2 * F] : 0.f;
    bv[ks].w = (m < F && k0 + 3 < B) ? t[3 * F] : 0.f;
  }
  for (int32_t cb = c0; cb < c1; cb += 16) {
    const int32_t cm = cb + m;  // A side: this lane's column
    int32_t r;
    int32_t pos[4];  // D side: this lane's four columns 4 kq + reg
    f32x4m acc = f32x4m{0.f, 0.f, 0.f, 0.f};
    if (NEAR) {
      r = __shfl(ur, (cm - base) & 63);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        pos[reg] = __shfl(mp, (cb - base + 4 * kq + reg) & 63);
        if (ADD == 1) acc[reg] = pa[reg];
        if (ADD == 2) acc[reg] = load_operand<AT>(s_add + (min(cb + 4 * kq + reg, c1 - 1) - base) * (int)ldA + min(m, F - 1));
      }
    } else {
      r = urel[min(cm, c1 - 1)];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int32_t c = min(cb + 4 * kq + reg, c1 - 1);
        pos[reg] = mpos ? mpos[c] : c;
        if (ADD) acc[reg] = load_operand<AT>(addend + (int64_t)c * ldA + min(m, F - 1));
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f32x4m av = *reinterpret_cast<const f32x4m *>(s_comp + r * KP + ks * 16 + 4 * kq);
      if (cm >= c1) av = f32x4m{0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[ks].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[ks].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[ks].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[ks].w, acc, 0, 0, 0);
    }
    if (m < F) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        if (cb + 4 * kq + reg < c1) store_operand<OT>(M + (int64_t)pos[reg] * ldM + m, acc[reg]);
    }
    if (NEAR && ADD == 1) break;  // (such a node has one tile: `pa` holds one tile's words)
  }
}

// Wide rows with few bases (the link-prediction encoder: F = 200, B = 2): a wave per node, lane = four features.  The
// node's B blocks of V stay in registers, the relation / operand-row words of 64 columns are loaded together and read
// back lane by lane, and every column leaves as ONE 4F-byte row store (16 bytes per lane) — the scalar form stored
// four bytes per lane and spent 135 us on the 260 MB of M at the FB15k-237 shape.
template <int BT>
__global__ __launch_bounds__(256) void k_mix_fwd_wide(const int32_t *__restrict__ nptr, const int32_t *__restrict__ urel,
                                                      const int32_t *__restrict__ mpos, const float *__restrict__ V,
                                                      const float *__restrict__ comp, int64_t N, int R, int B, int F,
                                                      float *__restrict__ M, int64_t ldM,
                                                      const int32_t *__restrict__ node_ids) {
  extern __shared__ __align__(16) float s_comp[];  // [R][B]
  for (int t = threadIdx.x; t < R * B; t += blockDim.x) s_comp[t] = comp[t];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int f0 = 4 * lane;
  const bool active = f0 < F;
  const f32x4m zero = {0.f, 0.f, 0.f, 0.f};
  for (int64_t j = (int64_t)blockIdx.x * 4 + wv; j < N; j += (int64_t)gridDim.x * 4) {
    const int32_t c0 = nptr[j], c1 = nptr[j + 1];
    if (c0 == c1) continue;
    const int64_t jv = node_ids ? (int64_t)node_ids[j] : j;
    f32x4m v[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b)
      v[b] = (active && b < B) ? *reinterpret_cast<const f32x4m *>(V + (jv * B + b) * F + f0) : zero;
    for (int32_t cb = c0; cb < c1; cb += 64) {
      const int32_t my = (cb + lane < c1) ? cb + lane : c1 - 1;
      const int32_t mr = urel[my], mp = mpos ? mpos[my] : my;
      const int cnt = (c1 - cb < 64) ? c1 - cb : 64;
      for (int t = 0; t < cnt; ++t) {
        const int r = __builtin_amdgcn_readlane(mr, t);
        const int64_t pos = __builtin_amdgcn_readlane(mp, t);
        f32x4m sum = zero;
#pragma unroll
        for (int b = 0; b < BT; ++b)
          if (b < B) sum += v[b] * s_comp[r * B + b];
        if (active) *reinterpret_cast<f32x4m *>(M + pos * ldM + f0) = sum;
      }
    }
  }
}

// IDS: the nodes are a list (a gradient support's live nodes): entry t owns columns nptr[t] .. nptr[t+1] and its V
// block is node_ids[t]'s; the ids travel with the node pointers, two steps ahead of their use.
#ifndef MIX_V_NT
#define MIX_V_NT 1  // V is read once per epoch: nontemporal loads (A/B: -DMIX_V_NT=0)
#endif
// ADD = 2: the addend rows of a step's columns are ONE contiguous run (a node's columns are consecutive, rows of ldA
// <= 16 floats, ldA % 4 == 0): they come in as one or two 16-byte loads per lane, a step ahead like the V blocks, and
// reach the accumulator layout through a per-wave LDS piece — instead of four 4-byte loads per node whose lanes
// address four different rows (ADD = 1, kept for other row strides).
constexpr int kAddPieces = 128;  // 16-byte pieces of addend rows per step (two loads per lane): 42 columns at ldA = 12
// TK (round 6): the waves take their steps IN ORDER from ticket counters instead of striding through the node range.
// tools/lab/copy_lab.hip: a persistent grid-stride stream runs at 4.8-4.9 TB/s on these boxes, a one-shot grid — work
// handed out in address order as blocks retire — at 6.0-6.25; resident waves drift apart and what is in flight stops
// being one narrow window of DRAM pages.  This kernel cannot be a one-shot grid (its comp table is most of a CU's LDS),
// so a wave draws TILES of kMixTile consecutive steps from one of kWorkTickets counters (counter c serves the c-th
// contiguous slice of the node range; one agent-scope atomic costs ~12 ns at a single address: per-step tickets from
// one counter would take longer than the kernel), the next tile's ticket one tile ahead.
constexpr int kMixTile = 4;  // default steps (of TN nodes) per ticket (`mix_ticket_tile`)
template <int KS, int NQ, int TN, int ADD, typename OT, bool IDS = false, typename AT = float, bool TK = false>
__global__ __launch_bounds__(kFwdTB) void k_mix_fwd_mfma(
    const int32_t *__restrict__ nptr, const int32_t *__restrict__ urel, const int32_t *__restrict__ mpos,
    const float *__restrict__ V, const float *__restrict__ comp, int64_t N, int R, int B, int F,
    const AT *__restrict__ addend, int64_t ldA, OT *__restrict__ M, int64_t ldM,
    const int32_t *__restrict__ node_ids = nullptr, unsigned long long *__restrict__ tickets = nullptr,
    int n_counters = 0, int tile = kMixTile) {
  extern __shared__ __align__(16) float s_mem[];
  constexpr int KP = KS * 16 + 4;  // padded comp row: rows start on different banks
  float *s_comp = s_mem;           // [R][KP], zero beyond B
  const int BF = B * F, nf4 = BF >> 2;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = blockDim.x >> 6;
  float *s_tile = s_mem + ((R * KP + 3) & ~3) + wv * (TN * BF);
  float *s_add = s_mem + ((R * KP + 3) & ~3) + nw * (TN * BF) + wv * (kAddPieces * 4);  // (ADD == 2 only)
  const int q4 = (int)((ldA * (int64_t)sizeof(AT)) >> 4);  // 16-byte pieces of an addend row
  for (int t = threadIdx.x; t < R * KP; t += blockDim.x) {
    const int r = t / KP, k = t - r * KP;
    s_comp[t] = k < B ? comp[(int64_t)r * B + k] : 0.f;
  }
  __syncthreads();
  const int m = lane & 15, kq = lane >> 4;  // A: column m of the tile, k quarter kq;  D: feature m, column quarter kq
  const int64_t ngroups = (N + TN - 1) / TN;
  const int64_t nwaves = (int64_t)gridDim.x * nw;
  // the wave's sequence of steps g, g1 (next), g2 (the one after): a stride of `nwaves`, or (TK) consecutive steps of
  // tiles drawn from the wave's ticket counter
  int64_t g = (int64_t)blockIdx.x * nw + wv, g1 = g + nwaves, g2 = g + 2 * nwaves;
  unsigned long long *ctr = nullptr;
  int64_t seg_lo = 0, seg_hi = 0;      // TK: the steps [seg_lo, seg_hi) this wave's counter serves
  unsigned long long tk_ahead = 0;     // TK: the ticket after the tile g2 is in (lane 0 holds the atomic's answer)
  auto tile_first = [&](unsigned long long t) {  // first step of ticket t of this wave's segment (>= ngroups: none)
    const int64_t st = seg_lo + (int64_t)t * tile;
    return st < seg_hi ? st : ngroups;
  };
  auto next_step = [&](int64_t x) {    // the step after x in this wave's sequence (TK)
    if (x >= ngroups) return x;
    if ((x - seg_lo + 1) % tile != 0 && x + 1 < seg_hi) return x + 1;
    const unsigned long long t = __shfl(tk_ahead, 0, 64);
    if (lane == 0) tk_ahead = atomicAdd(ctr, 1ull);
    return tile_first(t);
  };
  if (TK) {
    const int64_t wid = g;  // global wave id
    const int c = (int)(wid % n_counters);
    ctr = tickets + (int64_t)c * kWorkTicketStride;
    const int64_t per = ((ngroups + n_counters - 1) / n_counters + tile - 1) / tile * tile;
    seg_lo = (int64_t)c * per;
    seg_hi = min(seg_lo + per, ngroups);
    unsigned long long t0 = 0;
    if (lane == 0) {
      t0 = atomicAdd(ctr, 1ull);
      tk_ahead = atomicAdd(ctr, 1ull);
    }
    g = tile_first(__shfl(t0, 0, 64));
    g1 = next_step(g);
    g2 = next_step(g1);
  }
  if (g >= ngroups) return;
  // software pipeline over the wave's steps (TN nodes each): the node pointers run two steps ahead; the V
  // blocks (registers: NQ = ceil(B F / 256) 16-byte pieces per lane and node), the relation / position words
  // of the step's columns (one contiguous range: lane l holds column cp[0] + l) and the addend words of every
  // node's first tile run one step ahead, so the loads of step s+1 fly under the products of step s.  All of
  // them are unconditional at clamped addresses: predicated pieces would send the registers to scratch.
#define MIX_LOAD_NP(gg) nptr[min(min((gg), ngroups - 1) * TN + min(lane, TN), N)]
#define MIX_LOAD_ID(gg) (IDS ? node_ids[min(min((gg), ngroups - 1) * TN + min(lane, TN - 1), N - 1)] : 0)
#define MIX_LOAD_V(gg, idreg)                                                                               \
  _Pragma("unroll") for (int i = 0; i < TN; ++i) {                                                          \
    const int64_t nid = IDS ? (int64_t)__builtin_amdgcn_readlane(idreg, i) : min((gg) * TN + i, N - 1);     \
    const f32x4m *src = reinterpret_cast<const f32x4m *>(V) + nid * (int64_t)nf4;                           \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                                          \
      pv[i][q] = MIX_V_NT ? __builtin_nontemporal_load(src + min(lane + 64 * q, nf4 - 1)) : src[min(lane + 64 * q, nf4 - 1)]; \
  }
#define MIX_LOAD_IDX(np, ur, mp)                                                                            \
  {                                                                                                         \
    const int32_t a0 = __builtin_amdgcn_readlane(np, 0), a1 = __builtin_amdgcn_readlane(np, TN);            \
    const int32_t ci = max(min(a0 + lane, a1 - 1), 0);                                                      \
    ur = urel[ci];                                                                                          \
    mp = mpos ? mpos[ci] : ci;                                                                              \
    if (ADD == 2) {                                                                                         \
      const int64_t p0 = (int64_t)a0 * q4, p1 = (int64_t)a1 * q4;                                           \
      const f32x4m *a4 = reinterpret_cast<const f32x4m *>(addend);                                          \
      pa4_n1[0] = a4[max(min(p0 + lane, p1 - 1), (int64_t)0)];                                              \
      pa4_n1[1] = a4[max(min(p0 + 64 + lane, p1 - 1), (int64_t)0)];                                         \
    }                                                                                                       \
    if (ADD == 1) {                                                                                         \
      _Pragma("unroll") for (int i = 0; i < TN; ++i) {                                                      \
        const int32_t b0 = __builtin_amdgcn_readlane(np, i), b1 = __builtin_amdgcn_readlane(np, i + 1);     \
        _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) pa_n1[i][reg] =                                 \
            load_operand<AT>(addend + (int64_t)max(min(b0 + 4 * kq + reg, b1 - 1), 0) * ldA + min(m, F - 1)); \
      }                                                                                                     \
    }                                                                                                       \
  }
  f32x4m pv[TN][NQ];
  f32x4m pa4_n1[2];
  float pa_n1[TN][4], pa_cur[TN][4];
  int32_t np_cur = MIX_LOAD_NP(g), np_n1 = MIX_LOAD_NP(g1);
  int32_t id_cur = MIX_LOAD_ID(g), id_n1 = MIX_LOAD_ID(g1);
  int32_t ur_cur, mp_cur, ur_n1, mp_n1;
  MIX_LOAD_IDX(np_cur, ur_n1, mp_n1)
  MIX_LOAD_V(g, id_cur)
  for (; g < ngroups; g = g1, g1 = g2, g2 = TK ? next_step(g2) : g2 + nwaves) {
    const int32_t np_now = np_cur;
    const int32_t cbase = __builtin_amdgcn_readlane(np_now, 0), cend = __builtin_amdgcn_readlane(np_now, TN);
    const int32_t np_n2 = MIX_LOAD_NP(g2);
    const int32_t id_n2 = MIX_LOAD_ID(g2);
    ur_cur = ur_n1;
    mp_cur = mp_n1;
    bool near = cend - cbase <= 64;  // wave uniform
    // ADD == 2: the step's addend rows must fit the wave's LDS piece (nodes of any size: 42 columns at ldA = 12 — at
    // the AM shape 1.7 % of the steps / 12 % of the columns leave the prefetched path; with one tile per node, the
    // 4-byte form's limit, 6.7 % / 26 %)
    if (ADD == 2) near = near && (cend - cbase) * q4 <= kAddPieces;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      if (ADD == 1) {  // (one tile's addend words per node in `pa`)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) pa_cur[i][reg] = pa_n1[i][reg];
        near = near && __builtin_amdgcn_readlane(np_now, i + 1) - __builtin_amdgcn_readlane(np_now, i) <= 16;
      }
      // this step's V blocks: registers -> the wave's tile
      f32x4m *dst = reinterpret_cast<f32x4m *>(s_tile + i * BF);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (lane + 64 * q < nf4) dst[lane + 64 * q] = pv[i][q];
    }
    if (ADD == 2) {  // this step's addend rows (used by a NEAR step: at most kAddPieces 16-byte pieces)
      reinterpret_cast<f32x4m *>(s_add)[lane] = pa4_n1[0];
      reinterpret_cast<f32x4m *>(s_add)[64 + lane] = pa4_n1[1];
    }
    wave_lds_fence();
    MIX_LOAD_V(g1, id_n1)  // next step's blocks (the last step re-reads its own)
    MIX_LOAD_IDX(np_n1, ur_n1, mp_n1)
    np_cur = np_n1;
    np_n1 = np_n2;
    id_n1 = id_n2;
    if (near) {
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int32_t c0 = __builtin_amdgcn_readlane(np_now, i), c1 = __builtin_amdgcn_readlane(np_now, i + 1);
        if (c1 > c0)
          mix_node_tiles<KS, true, ADD, OT, AT>(c0, c1, cbase, ur_cur, mp_cur, pa_cur[i], urel, mpos, s_comp,
                                                s_tile + i * BF, B, F, addend, ldA, M, ldM, m, kq,
                                                reinterpret_cast<const AT *>(s_add));
      }
    } else {
#pragma unroll 1
      for (int i = 0; i < TN; ++i)
        mix_node_tiles<KS, false, ADD, OT, AT>(__builtin_amdgcn_readlane(np_now, i),
                                               __builtin_amdgcn_readlane(np_now, i + 1), cbase, ur_cur, mp_cur,
                                               pa_cur[0], urel, mpos, s_comp, s_tile + i * BF, B, F, addend, ldA, M,
                                               ldM, m, kq);
    }
    wave_lds_fence();  // the tile is rewritten by the next step
  }
#undef MIX_LOAD_NP
#undef MIX_LOAD_ID
#undef MIX_LOAD_V
#undef MIX_LOAD_IDX
}

// =====================================================================================
// basis mix, forward, column-parallel form (B <= 64): thread = compact column c, lanes =
// consecutive columns.  Index loads (urel / unode / mpos) are coalesced and independent, there
// is no per-node loop (no divergence under degree skew, no dependent load chain); the V rows of
// a node are re-read by its ~5 columns out of L1/L2, so HBM still streams V once.
//   M[mpos[c], 0:FW] = [ addend[c, 0:F] + sum_b comp[r_c, b] * V[b, j_c, 0:F] | 0 ]
// comp lives in LDS with an odd row stride (lanes hold different relations: conflict free).
// =====================================================================================
template <int FT, bool VEC2>
__global__ __launch_bounds__(kMixTB) void k_mix_fwd_cols(const int32_t *__restrict__ urel,
                                                         const int32_t *__restrict__ unode,
                                                         const int32_t *__restrict__ mpos,
                                                         const float *__restrict__ V,
                                                         const float *__restrict__ comp, int64_t N, int R,
                                                         int B, int F, int FW,
                                                         const float *__restrict__ addend, int64_t ldA,
                                                         float *__restrict__ M, int64_t ldM, int64_t ncols,
                                                         int comp_in_lds) {
  extern __shared__ float s_comp[];  // [R][BS], BS = B | 1
  const int BS = B | 1;
  if (comp_in_lds) {
    for (int t = threadIdx.x; t < R * B; t += blockDim.x) {
      const int r = t / B, b = t - r * B;
      s_comp[r * BS + b] = comp[t];
    }
    __syncthreads();
  }
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols;
       c += (int64_t)gridDim.x * blockDim.x) {
    const int r = urel[c];
    const int64_t j = unode[c];
    const int64_t pos = mpos ? mpos[c] : c;
    float acc[FT];
#pragma unroll
    for (int o = 0; o < FT; ++o) acc[o] = (addend && o < F) ? addend[c * ldA + o] : 0.f;
    const float *cr = comp_in_lds ? (s_comp + r * BS) : (comp + (int64_t)r * B);
#pragma unroll 4
    for (int b = 0; b < B; ++b) {
      const float w = cr[b];
      const float *vp = V + ((int64_t)j * B + b) * F;
      if (VEC2) {
#pragma unroll
        for (int o = 0; o < FT; o += 2)
          if (o < F) {
            const float2 vv = *reinterpret_cast<const float2 *>(vp + o);
            acc[o] = fmaf(w, vv.x, acc[o]);
            acc[o + 1] = fmaf(w, vv.y, acc[o + 1]);
          }
      } else {
#pragma unroll
        for (int o = 0; o < FT; ++o)
          if (o < F) acc[o] = fmaf(w, vp[o], acc[o]);
      }
    }
    float *m = M + pos * ldM;
    if ((ldM & 3) == 0 && (FW & 3) == 0) {
#pragma unroll
      for (int o = 0; o < FT; o += 4)
        if (o < FW) {
          float4 q;
          q.x = (o + 0 < F) ? acc[o + 0] : 0.f;
          q.y = (o + 1 < F) ? acc[o + 1] : 0.f;
          q.z = (o + 2 < F) ? acc[o + 2] : 0.f;
          q.w = (o + 3 < F) ? acc[o + 3] : 0.f;
          *reinterpret_cast<float4 *>(m + o) = q;
        }
    } else {
#pragma unroll
      for (int o = 0; o < FT; ++o)
        if (o < FW) m[o] = (o < F) ? acc[o] : 0.f;
    }
  }
}

// =====================================================================================
// basis mix, backward.
//
//   k_mix_bwd_nm    (F <= 16, B <= 64) WAVE STEPS OVER NODES, LANE = BASIS: dV and dcomp in one pass.
//                     dV[j, b, :]    = sum_{c in node j} comp[r_c, b] * dM[c, :]
//                     dcomp[r_c, b] += <dM[c, :], V[j, b, :]>
//                   Everything per column is wave-uniform (relation id and dM row travel through
//                   SGPRs via v_readlane), the dcomp atomics of a column go to B consecutive LDS
//                   words, and a node without any live column costs neither its V block nor its dV
//                   block: in the node-major layout both are one contiguous run of B*F floats that
//                   lane b reads / writes F floats of.
//   k_mix_bwd_dv + k_mix_bwd_dcomp   the same as two kernels for any F, B (thread = (node, feature)
//                   with accumulators over the bases; thread = compact column with LDS atomics).
// =====================================================================================
template <int BT>
__global__ __launch_bounds__(kMixTB) void k_mix_bwd_dv(const int32_t *__restrict__ nptr,
                                                       const int32_t *__restrict__ urel,
                                                       const float *__restrict__ dM, int64_t ldM,
                                                       const float *__restrict__ comp, int64_t N, int R,
                                                       int B, int b0, int F, float *__restrict__ dV,
                                                       int comp_in_lds, double *__restrict__ sumsq) {
  extern __shared__ __align__(16) float s_comp[];  // [R][CS]
  float sq = 0.f;
  constexpr int CS = comp_stride(BT);
  const int nb = min(BT, B - b0);
  if (comp_in_lds) {
    for (int t = threadIdx.x; t < R * BT; t += blockDim.x) {
      int r = t / BT, b = t - r * BT;
      s_comp[r * CS + b] = (b < nb) ? comp[(int64_t)r * B + b0 + b] : 0.f;
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
    for (int32_t cb = c0; cb < c1; cb += kPre) {  // one round trip per chunk of kPre columns
      int rr[kPre];
      float dd[kPre];
#pragma unroll
      for (int i = 0; i < kPre; ++i) {
        const bool in = cb + i < c1;
        rr[i] = in ? urel[cb + i] : 0;
        dd[i] = in ? dM[(int64_t)(cb + i) * ldM + o] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < kPre; ++i) {
        if (cb + i < c1) {
          if (comp_in_lds) {
            float w[BT];
            load_comp_row<BT>(s_comp + rr[i] * CS, w);
#pragma unroll
            for (int b = 0; b < BT; ++b) acc[b] = fmaf(w[b], dd[i], acc[b]);
          } else {
            const float *cr = comp + (int64_t)rr[i] * B + b0;
#pragma unroll
            for (int b = 0; b < BT; ++b)
              if (b < nb) acc[b] = fmaf(cr[b], dd[i], acc[b]);
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < BT; ++b)
      if (b < nb) {
        dV[((int64_t)j * B + (b0 + b)) * F + o] = acc[b];
        sq = fmaf(acc[b], acc[b], sq);
      }
  }
  if (sumsq) {  // ||dV||^2 for clip_grad_norm_: saves a separate pass over the gradient
    __shared__ float s_sq[kMixTB / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    if ((threadIdx.x & 63) == 0) s_sq[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_sq[i];
      atomicAdd(sumsq, (double)t);
    }
  }
}

// ---- wave over nodes, lane = basis ---------------------------------------------------------------
constexpr int kNodeTB = 1024;  // launch bound; the block size is picked at launch (MRGCN_MIX_BWD_TB): its waves share one LDS copy of the dcomp accumulators
constexpr int kGroup = 4;     // consecutive nodes whose pointers / flags / relation ids a wave fetches at once

// F floats of one basis row, rows only 4 * F bytes apart: 8-byte aligned when F is even.  The vector memory
// path only needs dword alignment, so such a row goes as 16-byte pieces plus an 8-byte tail (F = 10: three
// instructions instead of five — the kernel is bound by the texture addresser, which walks the same 13 lines of
// a node's block for every one of them)
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
template <int FT>
__device__ __forceinline__ void load_row(const float *__restrict__ p, int F, float (&v)[FT]) {
  if ((F & 1) == 0) {
#pragma unroll
    for (int o = 0; o < FT; o += 4) {
      if (o + 4 <= F && o + 4 <= FT) {
        const f32x4_a8 t = *reinterpret_cast<const f32x4_a8 *>(p + o);
        v[o] = t.x;
        v[o + 1] = t.y;
        v[o + 2] = t.z;
        v[o + 3] = t.w;
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
template <int FT>
__device__ __forceinline__ void store_row_f(float *__restrict__ p, int F, const float (&v)[FT]) {
  if ((F & 1) == 0) {
#pragma unroll
    for (int o = 0; o < FT; o += 2)
      if (o < F) *reinterpret_cast<float2 *>(p + o) = make_float2(v[o], (o + 1 < FT) ? v[o + 1] : 0.f);
  } else {
#pragma unroll
    for (int o = 0; o < FT; ++o)
      if (o < F) p[o] = v[o];
  }
}

// col_live  (nullable) one byte per compact column: does its dM row carry anything (dead rows of dM may
//           be unwritten); NULL = every column counts
// node_cur  (nullable) ROW-SPARSE gradient: only the blocks of nodes with a live column are written and
//           node_cur[j] says which; NULL = every block is written (zeros for nodes without gradient)
template <int FT>
__global__ __launch_bounds__(kNodeTB) void k_mix_bwd_nm(const int32_t *__restrict__ nptr,
                                                        const int32_t *__restrict__ urel,
                                                        const float *__restrict__ dM, int64_t ldM,
                                                        const float *__restrict__ V,
                                                        const float *__restrict__ comp, int64_t N, int R,
                                                        int B, int F, float *__restrict__ dV,
                                                        float *__restrict__ dcomp,
                                                        double *__restrict__ sumsq, int top_rel,
                                                        const uint8_t *__restrict__ col_live,
                                                        uint8_t *__restrict__ node_cur, int dc_in_lds) {
  extern __shared__ __align__(16) float s_dc[];  // dcomp accumulators [R*B] of the block
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = blockDim.x >> 6;
  if (dc_in_lds) {
    for (int t = threadIdx.x; t < R * B; t += blockDim.x) s_dc[t] = 0.f;
    __syncthreads();
  }
  // (the two address spaces are kept apart in the code: through one generic pointer the adds become flat atomics)
  const bool on = lane < B;
  const int b = on ? lane : 0;
  const int64_t ngroups = (N + kGroup - 1) / kGroup;
  const int64_t nwaves = (int64_t)gridDim.x * nw;
  float sq = 0.f;
  float hid = 0.f;  // dcomp row of the most frequent relation (the identity block) stays in a register
  for (int64_t g = (int64_t)blockIdx.x * nw + wv; g < ngroups; g += nwaves) {
    const int64_t j0 = g * kGroup;
    // wave-uniform values travel through SGPRs: one lane loads, v_readlane hands them out
    int32_t cp[kGroup + 1];
    {
      const int32_t mine = nptr[(j0 + lane < N) ? j0 + lane : N];
#pragma unroll
      for (int i = 0; i <= kGroup; ++i) cp[i] = __builtin_amdgcn_readlane(mine, i);
    }
    const int32_t ncg = cp[kGroup] - cp[0];  // columns of the group
    // relation ids and liveness of the group's first 64 columns, requested before anything depends on them
    const int32_t rl = (lane < ncg) ? urel[cp[0] + lane] : 0;
    const uint64_t amask = col_live ? __builtin_amdgcn_ballot_w64(lane < ncg && col_live[cp[0] + lane] != 0)
                                    : __builtin_amdgcn_ballot_w64(lane < ncg);
#pragma unroll
    for (int i = 0; i < kGroup; ++i) {
      const int64_t j = j0 + i;
      if (j >= N) break;
      const int32_t c_lo = cp[i], c_hi = cp[i + 1];
      // does the node have a live column?  (columns beyond the 64 the mask covers: look the flags up)
      bool any;
      {
        const int32_t lo = c_lo - cp[0], hi = c_hi - cp[0];
        if (hi <= 64) {
          const uint64_t m = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((lo >= 64) ? ~0ull : ((1ull << lo) - 1ull));
          any = (amask & m) != 0;
        } else if (!col_live) {
          any = c_hi > c_lo;
        } else {
          bool f = false;
          for (int32_t c = c_lo + lane; c < c_hi; c += 64) f |= col_live[c] != 0;
          any = __builtin_amdgcn_ballot_w64(f) != 0;
        }
      }
      float *drow = dV ? dV + ((int64_t)j * B + b) * F : nullptr;  // NULL: norm-only pass (node_cur given)
      if (!any) {  // wave uniform
        if (node_cur) {
          if (lane == 0) node_cur[j] = 0;
        } else if (on) {
          float z[FT];
#pragma unroll
          for (int o = 0; o < FT; ++o) z[o] = 0.f;
          store_row_f<FT>(drow, F, z);
        }
        continue;
      }
      float v[FT];
      load_row<FT>(V + ((int64_t)j * B + b) * F, F, v);
      float acc[FT];
#pragma unroll
      for (int o = 0; o < FT; ++o) acc[o] = 0.f;
      for (int32_t cb = c_lo; cb < c_hi; cb += 4) {
        // one load fetches the dM rows of four columns: the 16-lane group k reads column cb + k
        // (lane o of the group its feature o); v_readlane turns them into scalars
        const int32_t off = cb - cp[0];  // position among the group's columns (uniform)
        const int kq = lane >> 4, oq = lane & 15;
        const int32_t cc = cb + kq;
        const bool cin = cc < c_hi;
        uint32_t live = 0xFu;
        if (off + 4 <= 64) {
          live = (uint32_t)(amask >> off) & 0xFu;
        } else if (col_live) {  // beyond the 64 columns the mask covers: ask the flags
          const uint64_t bl = __builtin_amdgcn_ballot_w64(cin && col_live[cc] != 0);
          live = (uint32_t)((bl & 1u) | ((bl >> 15) & 2u) | ((bl >> 30) & 4u) | ((bl >> 45) & 8u));
        }
        if (c_hi - cb < 4) live &= (1u << (c_hi - cb)) - 1u;  // the chunk's tail belongs to the next node
        if (live == 0) continue;  // four columns without gradient
        const float dmine = (cin && oq < F && ((live >> kq) & 1u)) ? dM[(int64_t)cc * ldM + oq] : 0.f;
        int r[4];
        if (off + 4 <= 64) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) r[kk] = __builtin_amdgcn_readlane(rl, off + kk);
        } else {  // a group with more than 64 columns: fetch the ids of this chunk
          const int32_t rmine = cin ? urel[cc] : 0;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) r[kk] = __builtin_amdgcn_readlane(rmine, 16 * kk);
        }
        float w[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w[kk] = comp[(int64_t)r[kk] * B + b];  // R*B floats: cache resident
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if ((live >> kk) & 1u) {  // wave-uniform
            float d[FT];
#pragma unroll
            for (int o = 0; o < FT; ++o)
              d[o] = __builtin_bit_cast(
                  float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dmine), 16 * kk + o));
            float dot = 0.f;
#pragma unroll
            for (int o = 0; o < FT; ++o) dot = fmaf(d[o], v[o], dot);
            // LDS float atomics cost ~3 cycles per lane whatever the addresses: the most frequent
            // relation (a fifth of all columns on the AM shape) adds up in a register instead
            if (r[kk] == top_rel) hid += dot;
            else if (on) {
              if (dc_in_lds) atomicAdd(&s_dc[r[kk] * B + b], dot);  // ds_add_f32
              else atomicAdd(&dcomp[r[kk] * B + b], dot);
            }
#pragma unroll
            for (int o = 0; o < FT; ++o) acc[o] = fmaf(w[kk], d[o], acc[o]);
          }
        }
      }
      if (on) {
#pragma unroll
        for (int o = 0; o < FT; ++o)
          if (o < F) sq = fmaf(acc[o], acc[o], sq);
        if (drow) store_row_f<FT>(drow, F, acc);
      }
      if (node_cur && lane == 0) node_cur[j] = 1;
    }
  }
  if (sumsq) {
    __shared__ float s_sq[kNodeTB / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    if (lane == 0) s_sq[wv] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int i = 0; i < nw; ++i) t += s_sq[i];
      atomicAdd(sumsq, (double)t);
    }
  }
  if (top_rel >= 0 && on && hid != 0.f) {
    if (dc_in_lds) atomicAdd(&s_dc[top_rel * B + b], hid);
    else atomicAdd(&dcomp[top_rel * B + b], hid);
  }
  if (dc_in_lds) {
    __syncthreads();
    for (int t = threadIdx.x; t < R * B; t += blockDim.x) {
      const float x = s_dc[t];
      if (x != 0.f) atomicAdd(&dcomp[t], x);
    }
  }
}

// =====================================================================================
// Adam on the node-major basis table with the gradient FORMED ON THE FLY (row-sparse steps): the backward ran
// k_mix_bwd_nm without a dV buffer (dcomp, ||dV||^2 and the per-node flags only), and once the clip coefficient
// is known this kernel rebuilds every live node's block from its few dM rows,
//     dV[j][b][f] = sum_{live c of j} comp[r_c][b] * dM[c][f]        (same order, same fmaf chain: same bits),
// and applies torch.optim.Adam to it in the same pass.  The 2.67 GB gradient tensor (AM) is never written nor
// read back: traffic is p / m / v of the touched blocks only.  Wave per node, lane = 16-byte pieces of the block
// (P = ceil(B F / 256) per lane); comp (a snapshot taken before the optimizer moves it) lives in LDS.
// =====================================================================================
constexpr int kFusedTB = 1024;  // 16 waves share one LDS copy of comp; two blocks per CU at <= 64 VGPRs

// A wave takes a node's block 64 16-byte pieces at a time (AM: 100 pieces = two rounds); each lane forms the
// gradient of its own four elements.  Few registers (one p / m / v piece per lane) -> 8 waves per SIMD.
__global__ __launch_bounds__(kFusedTB, 8) void k_adam_rows_fused(
    const int32_t *__restrict__ nptr, const int32_t *__restrict__ urel, const uint8_t *__restrict__ col_live,
    const float *__restrict__ dM, int64_t ldM, const float *__restrict__ comp, int64_t N, int R, int B, int F,
    float *__restrict__ p, float *__restrict__ m, float *__restrict__ v, const uint8_t *__restrict__ cur,
    uint8_t *__restrict__ ever, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
    const float *__restrict__ scale, const float *__restrict__ bc_dev, int skip_cur) {
  extern __shared__ __align__(16) float s_comp[];  // [R][B]
  for (int t = threadIdx.x; t < R * B; t += blockDim.x) s_comp[t] = comp[t];
  __syncthreads();
  if (bc_dev) {
    bc1 = bc_dev[0];
    bc2_sqrt = bc_dev[1];
  }
  const float sc = scale ? *scale : 1.f;
  const float step = lr / bc1;
  auto upd = [&](float &pp, float gg, float &mm, float &vv) {  // == k_adam with wd = 0
    gg *= sc;
    mm = fmaf(b1, mm, (1.f - b1) * gg);
    vv = fmaf(b2, vv, (1.f - b2) * gg * gg);
    float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step * (mm / denom);
  };
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nw = blockDim.x >> 6;
  const int nv = (B * F) >> 2;       // 16-byte pieces of a block
  const int nh = (nv + 63) >> 6;     // 64-piece tasks per node
  const unsigned magicF = 65536u / (unsigned)F + 1u;  // e / F == (e * magicF) >> 16 for e * F < 2^16 (B F <= 1024)
  using f4 = __attribute__((ext_vector_type(4))) float;
  const int kq = lane >> 4, oq = lane & 15;
  const int64_t stride = (int64_t)gridDim.x * nw;  // the waves of the grid walk the nodes side by side
  // 64 of the wave's nodes at a time: lane l fetches flags and column range of node jb + l * stride with one
  // load each, the loop below visits only the nodes that have (or ever had) gradient — a node that never had any
  // costs nothing, not even a dependent flag load
  for (int64_t jb = (int64_t)blockIdx.x * nw + wv; jb < N; jb += 64 * stride) {
    const int64_t jl = jb + lane * stride;
    const int64_t jc = min(jl, N - 1);
    int32_t flv = jl < N ? ((int32_t)cur[jc] | ((int32_t)ever[jc] << 1)) : 0;
    // skip_cur: the nodes with gradient this step were updated by k_adam_rows_list; what is left are nodes that hold
    // moments from earlier steps without being in the list (their moments decay, their parameters drift on)
    if (skip_cur && (flv & 1)) flv = 0;
    const int32_t n0v = nptr[jc], n1v = nptr[jc + 1];
    uint64_t act = __builtin_amdgcn_ballot_w64(flv != 0);
    while (act) {
      const int l = __builtin_ctzll(act);
      act &= act - 1;
      const int64_t j = jb + l * stride;
      const bool c = (__builtin_amdgcn_readlane(flv, l) & 1) != 0;
      const int32_t n0 = __builtin_amdgcn_readlane(n0v, l), n1 = __builtin_amdgcn_readlane(n1v, l);
      f4 *p4 = reinterpret_cast<f4 *>(p) + j * (int64_t)nv;
      f4 *m4 = reinterpret_cast<f4 *>(m) + j * (int64_t)nv;
      f4 *v4 = reinterpret_cast<f4 *>(v) + j * (int64_t)nv;
      // the node's first four columns: the 16-lane group kq reads column n0 + kq (clamped: also issued for a node
      // without gradient this step, whose columns all count as dead)
      const int32_t cc0 = max(min(n0 + kq, n1 - 1), 0);
      const bool lv0 = c && n0 + kq < n1 && (!col_live || col_live[cc0] != 0);
      const int32_t r0 = urel[cc0];
      const float d0 = dM[(int64_t)cc0 * ldM + min(oq, F - 1)];
      for (int half = 0; half < nh; ++half) {
        const int q = lane + 64 * half;  // this lane's piece
        const int qc = min(q, nv - 1);
        // unconditional at clamped addresses (see the note on straight-line loads)
        f4 Pr = p4[qc], Mr = m4[qc], Vr = v4[qc];
        int bf[4];  // (basis, feature) of the piece's four elements
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned e = 4u * (unsigned)qc + (unsigned)k;
          const unsigned bb = (e * magicF) >> 16;
          bf[k] = (int)((bb << 8) | (e - bb * (unsigned)F));
        }
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        int32_t rmine = r0;
        float dmine = d0;
        bool lv = lv0;
        for (int32_t cb = n0;;) {
          const uint64_t bl = __builtin_amdgcn_ballot_w64(lv && oq == 0);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            if (!((bl >> (16 * kk)) & 1ull)) continue;  // wave uniform: a column without gradient (or past the node)
            const int r = __builtin_amdgcn_readlane(rmine, 16 * kk);
            const float *crow = s_comp + r * B;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              g[k] = fmaf(crow[bf[k] >> 8], __shfl(dmine, 16 * kk + (bf[k] & 255)), g[k]);
          }
          cb += 4;
          if (!c || cb >= n1) break;  // (few nodes have more than four columns)
          const int32_t cc = min(cb + kq, n1 - 1);
          lv = cb + kq < n1 && (!col_live || col_live[cc] != 0);
          rmine = urel[cc];
          dmine = dM[(int64_t)cc * ldM + min(oq, F - 1)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float pp = Pr[k], mm = Mr[k], vv = Vr[k];
          upd(pp, g[k], mm, vv);
          Pr[k] = pp;
          Mr[k] = mm;
          Vr[k] = vv;
        }
        if (q < nv) {
          p4[q] = Pr;
          m4[q] = Mr;
          v4[q] = Vr;
        }
      }
      if (c && lane == 0) ever[j] = 1;
    }
  }
}

// The same update over a LIST of nodes (a gradient support's live nodes: `lnode`, with `lnptr` their live column ranges
// by list position) — what the replayed epoch runs.  Against k_adam_rows_fused (stream_lab, AM shape, 828 703 live nodes of
// 1 600 bytes x 3 arrays in and out: 2 151 -> 1 693 us = the rate of a plain triad over the same blocks, 4.7 TB/s on the
// box where a float4 copy moves 4.7 TB/s):
//   * all NH = ceil(B F / 256) 16-byte pieces of a lane are loaded in ONE round (the round-4 kernel took them 64 pieces
//     at a time, each round behind the previous round's arithmetic);
//   * the NEXT node's p / m / v pieces are in flight while this node's gradient is formed and its update stored
//     (twice the registers: 4 waves per SIMD);
//   * p / m / v are touched once per epoch: nontemporal loads and stores (3.95 -> 4.70 TB/s for the bare triad over the
//     same blocks — they do not push the rest of the epoch's working set out of the Infinity Cache).
// Same fmaf chain per element as k_adam_rows_fused: the same bits.
template <int NH, bool PIPE>
__global__ __launch_bounds__(kFusedTB, PIPE ? 4 : 8) void k_adam_rows_list(
    const int32_t *__restrict__ lnode, const int32_t *__restrict__ lnptr, const int32_t *__restrict__ lrel,
    const float *__restrict__ dM, int64_t ldM, const float *__restrict__ comp, int64_t NL, int R, int B, int F,
    float *__restrict__ p, float *__restrict__ m, float *__restrict__ v, uint8_t *__restrict__ ever, float lr, float b1,
    float b2, float eps, float bc1, float bc2_sqrt, const float *__restrict__ scale, const float *__restrict__ bc_dev) {
  extern __shared__ __align__(16) float s_comp[];  // [R][B]
  for (int t = threadIdx.x; t < R * B; t += blockDim.x) s_comp[t] = comp[t];
  __syncthreads();
  if (bc_dev) {
    bc1 = bc_dev[0];
    bc2_sqrt = bc_dev[1];
  }
  const float sc = scale ? *scale : 1.f;
  const float step = lr / bc1;
  auto upd = [&](float &pp, float gg, float &mm, float &vv) {  // == k_adam with wd = 0
    gg *= sc;
    mm = fmaf(b1, mm, (1.f - b1) * gg);
    vv = fmaf(b2, vv, (1.f - b2) * gg * gg);
    float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step * (mm / denom);
  };
  using f4 = __attribute__((ext_vector_type(4))) float;
  const int lane = threadIdx.x & 63;
  const int nv = (B * F) >> 2;  // 16-byte pieces of a block
  const unsigned magicF = 65536u / (unsigned)F + 1u;  // e / F == (e * magicF) >> 16 for e * F < 2^16 (B F <= 1024)
  const int kq = lane >> 4, oq = lane & 15;
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int bf[NH][4];  // (basis, feature) of the lane's elements
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned e = 4u * (unsigned)min(lane + 64 * h, nv - 1) + (unsigned)k;
      const unsigned bb = (e * magicF) >> 16;
      bf[h][k] = (int)((bb << 8) | (e - bb * (unsigned)F));
    }
  f4 P[NH], M[NH], V[NH];
  // unconditional at clamped addresses (see the note on straight-line loads)
  auto load_block = [&](int64_t j, f4 *Pd, f4 *Md, f4 *Vd) {
    const f4 *p4 = reinterpret_cast<const f4 *>(p) + j * (int64_t)nv;
    const f4 *m4 = reinterpret_cast<const f4 *>(m) + j * (int64_t)nv;
    const f4 *v4 = reinterpret_cast<const f4 *>(v) + j * (int64_t)nv;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int q = min(lane + 64 * h, nv - 1);
      Pd[h] = __builtin_nontemporal_load(p4 + q);
      Md[h] = __builtin_nontemporal_load(m4 + q);
      Vd[h] = __builtin_nontemporal_load(v4 + q);
    }
  };
  int64_t i = w;
  if (i >= NL) return;
  int64_t j = lnode[i];
  int32_t n0 = lnptr[i], n1 = lnptr[i + 1];
  if (PIPE) load_block(j, P, M, V);
  for (; i < NL; i += nw) {
    // the node's first four live columns: the 16-lane group kq reads column n0 + kq
    const int32_t cc0 = max(min(n0 + kq, n1 - 1), 0);
    bool lv = n0 + kq < n1;
    int32_t rmine = lrel[cc0];
    float dmine = dM[(int64_t)cc0 * ldM + min(oq, F - 1)];
    // the next node's ids and (pipelined) block, issued before this node's arithmetic
    const int64_t inext = min(i + nw, NL - 1);
    const int64_t jn = lnode[inext];
    const int32_t n0n = lnptr[inext], n1n = lnptr[inext + 1];
    f4 Pn[NH], Mn[NH], Vn[NH];
    if (PIPE) load_block(jn, Pn, Mn, Vn);
    else load_block(j, P, M, V);
    float g[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int k = 0; k < 4; ++k) g[h][k] = 0.f;
    for (int32_t cb = n0;;) {
      const uint64_t bl = __builtin_amdgcn_ballot_w64(lv && oq == 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (!((bl >> (16 * kk)) & 1ull)) continue;  // wave uniform: past the node's columns
        const int r = __builtin_amdgcn_readlane(rmine, 16 * kk);
        const float *crow = s_comp + r * B;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            g[h][k] = fmaf(crow[bf[h][k] >> 8], __shfl(dmine, 16 * kk + (bf[h][k] & 255)), g[h][k]);
      }
      cb += 4;
      if (cb >= n1) break;  // (few nodes have more than four live columns)
      const int32_t cc = min(cb + kq, n1 - 1);
      lv = cb + kq < n1;
      rmine = lrel[cc];
      dmine = dM[(int64_t)cc * ldM + min(oq, F - 1)];
    }
    f4 *p4 = reinterpret_cast<f4 *>(p) + j * (int64_t)nv;
    f4 *m4 = reinterpret_cast<f4 *>(m) + j * (int64_t)nv;
    f4 *v4 = reinterpret_cast<f4 *>(v) + j * (int64_t)nv;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pp = P[h][k], mm = M[h][k], vv = V[h][k];
        upd(pp, g[h][k], mm, vv);
        P[h][k] = pp;
        M[h][k] = mm;
        V[h][k] = vv;
      }
      const int q = lane + 64 * h;
      if (q < nv) {
        __builtin_nontemporal_store(P[h], p4 + q);
        __builtin_nontemporal_store(M[h], m4 + q);
        __builtin_nontemporal_store(V[h], v4 + q);
      }
    }
    if (lane == 0) ever[j] = 1;
    if (PIPE) {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        P[h] = Pn[h];
        M[h] = Mn[h];
        V[h] = Vn[h];
      }
    }
    j = jn;
    n0 = n0n;
    n1 = n1n;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_adam_rows_once (round 6): the same update as k_adam_rows_list as a ONE-SHOT grid — a wave owns NPW consecutive list
// entries and ends; there is no persistent loop and no workgroup state.  Why: tools/lab/copy_lab.hip (profiles/
// r06_copy_lab.txt).  On these boxes a float4 copy moves 4.8 TB/s as a persistent grid-stride loop (round 5's yardstick, and
// the shape of every streaming kernel of the epoch) and 6.25 TB/s as a one-shot grid (MI355X_MICROARCH.md's 6.29); a
// 3-read / 3-write triad 4.9 against 6.0.  The hardware's block dispatcher hands out work in address order as blocks
// retire, so what is in flight stays one narrow, advancing window of DRAM pages; resident waves striding through the
// array drift apart.  (In-order ticket counters reproduce part of it — 5.5 TB/s with 16 counters — but one agent-scope
// atomic costs 12 ns; the dispatcher is the cheap in-order counter.)
// What kept the Adam pass persistent was the 43 KB `comp` table in LDS.  Here a wave reads the `comp` ROW of each live
// column of its nodes straight from the (L2-resident) table — lane b holds comp[r][b], the products fetch it with a
// lane permute — so a wave needs nothing but its own registers.  Same fmaf chain per element as k_adam_rows_list and
// k_adam_rows_fused: the same bits.  Measured in the AM epoch, alternating processes on one box (rocprofv3): the
// persistent list kernel 1 485 us, one entry per wave 1 374 / 1 385 us (5.9 TB/s), two per wave 1 444 / 1 550 us.
// ---------------------------------------------------------------------------------------------------------------------
template <int NH, int NPW>
__global__ __launch_bounds__(256) void k_adam_rows_once(
    const int32_t *__restrict__ lnode, const int32_t *__restrict__ lnptr, const int32_t *__restrict__ lrel,
    const float *__restrict__ dM, int64_t ldM, const float *__restrict__ comp, int64_t NL, int R, int B, int F,
    float *__restrict__ p, float *__restrict__ m, float *__restrict__ v, uint8_t *__restrict__ ever, float lr, float b1,
    float b2, float eps, float bc1, float bc2_sqrt, const float *__restrict__ scale, const float *__restrict__ bc_dev) {
  using f4 = __attribute__((ext_vector_type(4))) float;
  const int lane = threadIdx.x & 63;
  const int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t i0 = w * NPW;
  if (i0 >= NL) return;  // wave uniform
  const int nv = (B * F) >> 2;  // 16-byte pieces of a block
  const unsigned magicF = 65536u / (unsigned)F + 1u;  // e / F == (e * magicF) >> 16 for e * F < 2^16 (B F <= 1024)
  const int kq = lane >> 4, oq = lane & 15;
  // round 1: the list entries (clamped: a wave past the end repeats the last entry and stores nothing for it)
  int64_t j[NPW];
  int32_t n0[NPW], n1[NPW];
#pragma unroll
  for (int t = 0; t < NPW; ++t) {
    const int64_t i = min(i0 + t, NL - 1);
    j[t] = lnode[i];
    n0[t] = lnptr[i];
    n1[t] = lnptr[i + 1];
  }
  // round 2: the nodes' blocks, and relation / gradient row of their first four live columns
  f4 P[NPW][NH], M[NPW][NH], V[NPW][NH];
  int32_t rmine[NPW];
  float dmine[NPW];
#pragma unroll
  for (int t = 0; t < NPW; ++t) {
    const f4 *p4 = reinterpret_cast<const f4 *>(p) + j[t] * (int64_t)nv;
    const f4 *m4 = reinterpret_cast<const f4 *>(m) + j[t] * (int64_t)nv;
    const f4 *v4 = reinterpret_cast<const f4 *>(v) + j[t] * (int64_t)nv;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int q = min(lane + 64 * h, nv - 1);
      P[t][h] = __builtin_nontemporal_load(p4 + q);
      M[t][h] = __builtin_nontemporal_load(m4 + q);
      V[t][h] = __builtin_nontemporal_load(v4 + q);
    }
    const int32_t cc0 = max(min(n0[t] + kq, n1[t] - 1), 0);
    rmine[t] = lrel[cc0];
    dmine[t] = dM[(int64_t)cc0 * ldM + min(oq, F - 1)];
  }
  if (bc_dev) {
    bc1 = bc_dev[0];
    bc2_sqrt = bc_dev[1];
  }
  const float sc = scale ? *scale : 1.f;
  const float step = lr / bc1;
  int bf[NH][4];  // (basis, feature) of the lane's elements
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned e = 4u * (unsigned)min(lane + 64 * h, nv - 1) + (unsigned)k;
      const unsigned bb = (e * magicF) >> 16;
      bf[h][k] = (int)((bb << 8) | (e - bb * (unsigned)F));
    }
  // round 3: the comp rows of those columns (lane b: comp[r][b]) — four unconditional loads per node
  float cv[NPW][4];
#pragma unroll
  for (int t = 0; t < NPW; ++t)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int r = __builtin_amdgcn_readlane(rmine[t], 16 * kk);
      cv[t][kk] = comp[(int64_t)r * B + min(lane, B - 1)];
    }
#pragma unroll
  for (int t = 0; t < NPW; ++t) {
    float g[NH][4];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int k = 0; k < 4; ++k) g[h][k] = 0.f;
    bool lv = n0[t] + kq < n1[t];
    int32_t rm = rmine[t];
    float dm = dmine[t];
    float c4[4] = {cv[t][0], cv[t][1], cv[t][2], cv[t][3]};
    for (int32_t cb = n0[t];;) {
      const uint64_t bl = __builtin_amdgcn_ballot_w64(lv && oq == 0);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (!((bl >> (16 * kk)) & 1ull)) continue;  // wave uniform: past the node's columns
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            g[h][k] = fmaf(__shfl(c4[kk], bf[h][k] >> 8), __shfl(dm, 16 * kk + (bf[h][k] & 255)), g[h][k]);
      }
      cb += 4;
      if (cb >= n1[t]) break;  // (few nodes have more than four live columns)
      const int32_t cc = min(cb + kq, n1[t] - 1);
      lv = cb + kq < n1[t];
      rm = lrel[cc];
      dm = dM[(int64_t)cc * ldM + min(oq, F - 1)];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        c4[kk] = comp[(int64_t)__builtin_amdgcn_readlane(rm, 16 * kk) * B + min(lane, B - 1)];
    }
    if (i0 + t >= NL) continue;  // wave uniform
    f4 *p4 = reinterpret_cast<f4 *>(p) + j[t] * (int64_t)nv;
    f4 *m4 = reinterpret_cast<f4 *>(m) + j[t] * (int64_t)nv;
    f4 *v4 = reinterpret_cast<f4 *>(v) + j[t] * (int64_t)nv;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pp = P[t][h][k], mm = M[t][h][k], vv = V[t][h][k];
        float gg = g[h][k] * sc;  // == k_adam with wd = 0
        mm = fmaf(b1, mm, (1.f - b1) * gg);
        vv = fmaf(b2, vv, (1.f - b2) * gg * gg);
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp -= step * (mm / denom);
        P[t][h][k] = pp;
        M[t][h][k] = mm;
        V[t][h][k] = vv;
      }
      const int q = lane + 64 * h;
      if (q < nv) {
        __builtin_nontemporal_store(P[t][h], p4 + q);
        __builtin_nontemporal_store(M[t][h], m4 + q);
        __builtin_nontemporal_store(V[t][h], v4 + q);
      }
    }
    if (lane == 0) ever[j[t]] = 1;
  }
}

template <int FT, bool VEC2>
__global__ __launch_bounds__(kMixTB) void k_mix_bwd_dcomp(const int32_t *__restrict__ urel,
                                                          const int32_t *__restrict__ unode,
                                                          const float *__restrict__ dM, int64_t ldM,
                                                          const float *__restrict__ V, int64_t ldV,
                                                          int64_t N, int R, int B, int F, int64_t ncols,
                                                          float *__restrict__ dcomp, int dcomp_in_lds) {
  extern __shared__ float s_dcomp[];  // [R][BS], BS = B | 1: lanes hold different relations
  const int BS = B | 1;
  if (dcomp_in_lds) {
    for (int t = threadIdx.x; t < R * BS; t += blockDim.x) s_dcomp[t] = 0.f;
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
      const float *vp = V + ((int64_t)j * B + b) * ldV;
      float dot = 0.f;
      if (VEC2) {  // F even: rows of V are 8-byte aligned; 16-byte loads need dword alignment only
#pragma unroll
        for (int o = 0; o < FT; o += 4) {
          if (o + 4 <= F) {
            const float4 vv = *reinterpret_cast<const float4 *>(vp + o);
            dot = fmaf(dm[o], vv.x, dot);
            dot = fmaf(dm[o + 1], vv.y, dot);
            dot = fmaf(dm[o + 2], vv.z, dot);
            dot = fmaf(dm[o + 3], vv.w, dot);
          } else if (o < F) {
            const float2 vv = *reinterpret_cast<const float2 *>(vp + o);
            dot = fmaf(dm[o], vv.x, dot);
            dot = fmaf(dm[o + 1], vv.y, dot);
            if (o + 2 < F) dot = fmaf(dm[o + 2], vp[o + 2], dot);
          }
        }
      } else {
#pragma unroll
        for (int o = 0; o < FT; ++o)
          if (o < F) dot = fmaf(dm[o], vp[o], dot);
      }
      if (dcomp_in_lds) atomicAdd(&s_dcomp[r * BS + b], dot);
      else atomicAdd(&dcomp[(int64_t)r * B + b], dot);
    }
  }
  if (dcomp_in_lds) {
    __syncthreads();
    for (int t = threadIdx.x; t < R * B; t += blockDim.x) {
      const int r = t / B, b = t - r * B;
      const float x = s_dcomp[r * BS + b];
      if (x != 0.f) atomicAdd(&dcomp[t], x);
    }
  }
}

// the same for wide layers (F > 64): a WAVE per compact column, lanes along the features — the rows of dM and V are
// read as whole contiguous rows (a thread per column reads 64 different lines per load: 3 x 110 us at the FB15k-237
// shape, F = 200), the dot products meet by shuffles, one LDS (or global) atomic per (column, basis)
__global__ __launch_bounds__(256) void k_mix_bwd_dcomp_wide(const int32_t *__restrict__ urel,
                                                            const int32_t *__restrict__ unode,
                                                            const float *__restrict__ dM, int64_t ldM,
                                                            const float *__restrict__ V, int64_t ldV, int R, int B,
                                                            int F, int64_t ncols, float *__restrict__ dcomp,
                                                            int dcomp_in_lds) {
  extern __shared__ float s_dcomp[];  // [R][B]
  if (dcomp_in_lds) {
    for (int t = threadIdx.x; t < R * B; t += blockDim.x) s_dcomp[t] = 0.f;
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  for (int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; c < ncols;
       c += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const int r = urel[c];
    const int64_t j = unode[c];
    const float *dp = dM + c * ldM;
    for (int b = 0; b < B; ++b) {
      const float *vp = V + ((int64_t)j * B + b) * ldV;
      float dot = 0.f;
      for (int o = lane; o < F; o += 64) dot = fmaf(dp[o], vp[o], dot);
#pragma unroll
      for (int sh = 32; sh > 0; sh >>= 1) dot += __shfl_xor(dot, sh, 64);
      if (lane == 0 && dot != 0.f) {
        if (dcomp_in_lds) atomicAdd(&s_dcomp[r * B + b], dot);
        else atomicAdd(&dcomp[(int64_t)r * B + b], dot);
      }
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
// no-bases input term: M[mpos[c], :] = W[ulcol[c], :] (+ addend[c, :])
// =====================================================================================
template <typename OT>
__global__ void k_gather_rows(const int32_t *__restrict__ ulcol, const int32_t *__restrict__ mpos,
                              int64_t ncols, const float *__restrict__ W, int F, int FW,
                              const float *__restrict__ addend, int64_t ldA, OT *__restrict__ M,
                              int64_t ldM) {
  const int64_t total = ncols * FW;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = t / FW;
    const int o = (int)(t - c * FW);
    float x = 0.f;
    if (o < F) {
      x = W[(int64_t)ulcol[c] * F + o];
      if (addend) x += addend[c * ldA + o];
    }
    store_operand<OT>(M + (int64_t)mpos[c] * ldM + o, x);
  }
}

// =====================================================================================
// relation transform fallbacks (shapes outside the MFMA kernels' limits): LDS tiles + FMA.
// One block per relation chunk (<= kRelChunk compact columns of one relation, relation-major
// order `rperm`); W[r] (K x F) and a gathered X tile (TK columns x K) staged in LDS.
// =====================================================================================
constexpr int kTK = 32;   // columns per LDS tile
constexpr int kKS = 160;  // K slab held in LDS at once

template <typename OT>
__global__ __launch_bounds__(kTB) void k_xform_fwd(const int32_t *__restrict__ relchunk_rel,
                                                   const int32_t *__restrict__ relchunk_beg,
                                                   const int32_t *__restrict__ relchunk_end,
                                                   const int32_t *__restrict__ rperm,
                                                   const int32_t *__restrict__ unode,
                                                   const int32_t *__restrict__ out_index,
                                                   const float *__restrict__ X, int64_t ldX, int K,
                                                   const float *__restrict__ W, int F, int FW,
                                                   OT *__restrict__ M, int64_t ldM) {
  extern __shared__ float smem[];
  float *Ws = smem;            // [KS][F]
  float *Xs = smem + kKS * F;  // [TK][KS+1]
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
      s_c[threadIdx.x] = out_index ? out_index[c] : c;
      s_j[threadIdx.x] = unode[c];
    }
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
        store_operand<OT>(M + (int64_t)s_c[kk] * ldM + o, acc[q]);
      }
    }
    // zero the padding of the rows just written
    if (FW > F)
      for (int p = threadIdx.x; p < nk * (FW - F); p += kTB) {
        const int kk = p / (FW - F), o = F + p - kk * (FW - F);
        store_operand<OT>(M + (int64_t)s_c[kk] * ldM + o, 0.f);
      }
  }
}

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
  float *Xs = smem;             // [TK][KS+1]
  float *Ds = smem + kTK * XS;  // [TK][F]
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

// dX[j, i] = sum_{c in node j} sum_o dM[c, o] * W[r_c, i, o]   (node-major; no atomics)
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

int grid_for(int64_t work_items, int tb = kTB, int max_blocks = 256 * 8) {
  int64_t b = (work_items + tb - 1) / tb;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

constexpr size_t kLdsBudget = 64 * 1024;  // dynamic LDS these kernels may take

bool use_mfma() {
  const bool v = cfg(CFG_XFORM_MFMA) != 0;
  return v;
}

// persistent grid of 512-thread blocks for a kernel that takes `lds` bytes of LDS
int mix_grid(size_t lds, int64_t work_items) {
  int per_cu = lds > 0 ? (int)((160 * 1024) / (lds + 256)) : 4;
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  int64_t want = (work_items + kMixTB - 1) / kMixTB;
  int64_t grid = 256 * per_cu * 2;  // two rounds: evens out the tail
  if (grid > want) grid = want;
  return (int)(grid < 1 ? 1 : grid);
}

}  // namespace
}  // namespace mrgcn

using namespace mrgcn;

namespace {

// the columns a basis mix walks: a plan's (every node) or a gradient support's (a node list, rows by live number)
struct MixCols {
  const int32_t *nptr, *urel, *mpos, *unode, *node_ids;
  int64_t N, ncols;
  int R;
  unsigned long long *tickets = nullptr;  // the plan's work-ticket counters (k_mix_fwd_mfma's TK form), or NULL
};

template <typename OT, typename AT = float>
int mix_fwd_cols(const MixCols *p, const float *V, const float *comp, int32_t B, int32_t F,
                 const AT *addend, int64_t ldA, OT *M, int64_t ldM, void *stream) {
  MRGCN_REQUIRE(V && comp && M, "NULL");
  MRGCN_REQUIRE(B > 0 && F > 0 && ldM >= F, "B / F / ldM");
  MRGCN_REQUIRE(!addend || ldA >= F, "ldA");
  if (p->ncols == 0 || p->N == 0) return MRGCN_OK;
  hipStream_t s = (hipStream_t)stream;
  const int R = p->R;
  const int64_t N = p->N;
  const int32_t *node_ids = p->node_ids;
  // Columns F..ldM of M are padding that only the SpMM's 16-byte gathers touch, and their
  // products land in accumulator lanes that are never stored (test: padding set to 1e30 does
  // not leak): by default only the F real features are computed and written (a sixth fewer
  // lanes and stores at F = 10, ld = 12); MRGCN_MIX_PAD=1 writes zeros there.
  const bool write_pad = cfg(CFG_MIX_PAD) != 0;
  const int FW = write_pad ? (int)ldM : F;
  // node-major is the default: measured 2.7 ms vs 4.95 ms for the column-parallel form (AM shape)
  const bool by_cols = cfg(CFG_MIX_COLS) != 0;
  const int32_t *mpos_arg = p->mpos;
  if constexpr (sizeof(OT) == 4 && sizeof(AT) == 4) if (by_cols && !node_ids && B <= 64 && F <= 64 && FW <= 64) {
    size_t lds = (size_t)R * (B | 1) * sizeof(float);
    int in_lds = lds <= kLdsBudget;
    if (!in_lds) lds = 0;
    int grid = mix_grid(lds, p->ncols);
    const bool v2 = (F % 2 == 0) && (((uintptr_t)V) % 8 == 0);
    const int need = FW > F ? FW : F;
#define MIXC_GO(T)                                                                                      \
  do {                                                                                                  \
    if (v2)                                                                                             \
      k_mix_fwd_cols<T, true><<<dim3(grid), dim3(kMixTB), lds, s>>>(                                    \
          p->urel, p->unode, mpos_arg, V, comp, N, R, B, F, FW, addend, ldA, M, ldM, p->ncols, in_lds);  \
    else                                                                                                \
      k_mix_fwd_cols<T, false><<<dim3(grid), dim3(kMixTB), lds, s>>>(                                   \
          p->urel, p->unode, mpos_arg, V, comp, N, R, B, F, FW, addend, ldA, M, ldM, p->ncols, in_lds);  \
  } while (0)
    if (need <= 4) MIXC_GO(4);
    else if (need <= 8) MIXC_GO(8);
    else if (need <= 12) MIXC_GO(12);
    else if (need <= 16) MIXC_GO(16);
    else if (need <= 32) MIXC_GO(32);
    else MIXC_GO(64);
#undef MIXC_GO
    MRGCN_HIP_TRY(hipGetLastError());
    return MRGCN_OK;
  }
  if constexpr (sizeof(OT) == 4) {
    const bool wide_on = cfg(CFG_MIX_WIDE) != 0;
    if (wide_on && !addend && F > 16 && F <= 256 && F % 4 == 0 && B <= 4 && ldM % 4 == 0 &&
        ((((uintptr_t)V) | ((uintptr_t)M)) & 15) == 0 && (size_t)R * B * sizeof(float) <= 64 * 1024) {
      const size_t lds = (size_t)R * B * sizeof(float);
      int64_t grid = (N + 3) / 4;
      if (grid > 256 * 8) grid = 256 * 8;
#define MIXW_GO(BT_)                                                                                            \
  k_mix_fwd_wide<BT_><<<dim3((unsigned)grid), dim3(256), lds, s>>>(p->nptr, p->urel, mpos_arg, V, comp, N, R, B, F, \
                                                                    (float *)M, ldM, node_ids)
      if (B == 1) MIXW_GO(1);
      else if (B == 2) MIXW_GO(2);
      else MIXW_GO(4);
#undef MIXW_GO
      MRGCN_HIP_TRY(hipGetLastError());
      return MRGCN_OK;
    }
  }
  {
    const bool mfma_on = cfg(CFG_MIX_MFMA) != 0;
    constexpr int tn = 2;  // nodes per wave step
    const int KS = (B + 15) / 16;
    const int NQ = (B * F + 255) / 256;  // 16-byte pieces of a V block per lane (<= KS)
    // the addend as 16-byte pieces through LDS (k_mix_fwd_mfma, ADD = 2) when its rows allow it
    const size_t lds_plain =
        ((size_t)((R * (KS * 16 + 4) + 3) & ~3) + (size_t)(kFwdTB / 64) * tn * B * F) * sizeof(float);
    const size_t lds_add = (size_t)(kFwdTB / 64) * kAddPieces * 4 * sizeof(float);
    const bool add_vec = addend && (ldA * sizeof(AT)) % 16 == 0 && ldA <= 16 && (((uintptr_t)addend) & 15) == 0 &&
                         !node_ids && cfg(CFG_MIX_ADD_VEC) != 0 &&
                         lds_plain + lds_add <= 150 * 1024;  // (else the element-wise form)
    const size_t lds = lds_plain + (add_vec ? lds_add : 0);
    if (mfma_on && B <= 64 && F <= 16 && (B * F) % 4 == 0 && (((uintptr_t)V) & 15) == 0 &&
        lds <= 150 * 1024 && !(node_ids && (addend || sizeof(OT) != 4))) {
      const int64_t want = ((N + tn - 1) / tn + (kFwdTB / 64) - 1) / (kFwdTB / 64);
      int64_t grid = 256;  // one block of 16 waves per CU (LDS and the 128-register budget allow no second)
      if (grid > want) grid = want;
      // in-order tickets (see k_mix_fwd_mfma, TK): a full grid over a plan that carries the counters
      const int n_ctr = (int)std::min<int64_t>(std::max<int64_t>(cfg(CFG_MIX_TICKETS), 0), kWorkTickets);
      const int tk_tile = (int)std::min<int64_t>(std::max<int64_t>(cfg(CFG_MIX_TICKET_TILE), 1), 64);
      // (a full grid with at least four tiles per resident wave: on a small graph the plain stride keeps every wave busy —
      // with tiles of four MUTAG's 11.8 k steps occupied 2 956 of the 4 096 waves with four serial steps each)
      const int64_t mix_steps = (N + tn - 1) / tn;
      const bool tk = p->tickets && !node_ids && n_ctr > 0 && grid == 256 &&
                      mix_steps >= (int64_t)4 * tk_tile * grid * (kFwdTB / 64);
      if (tk)
        MRGCN_HIP_TRY(mrgcn::fill_async(p->tickets, 0, (size_t)kWorkTickets * kWorkTicketStride * sizeof(unsigned long long), s));
#define MIXM_GO(KS_, NQ_, TN_)                                                                              \
  do {                                                                                                      \
    auto kfn = addend ? (add_vec ? k_mix_fwd_mfma<KS_, NQ_, TN_, 2, OT, false, AT>                          \
                                 : k_mix_fwd_mfma<KS_, NQ_, TN_, 1, OT, false, AT>)                         \
                      : k_mix_fwd_mfma<KS_, NQ_, TN_, 0, OT, false, AT>;                                    \
    if (tk) kfn = addend ? (add_vec ? k_mix_fwd_mfma<KS_, NQ_, TN_, 2, OT, false, AT, true>                 \
                                    : k_mix_fwd_mfma<KS_, NQ_, TN_, 1, OT, false, AT, true>)                \
                         : k_mix_fwd_mfma<KS_, NQ_, TN_, 0, OT, false, AT, true>;                           \
    if constexpr (sizeof(OT) == 4 && sizeof(AT) == 4) if (node_ids) kfn = k_mix_fwd_mfma<KS_, NQ_, TN_, 0, OT, true>; \
    MRGCN_HIP_TRY(raise_lds_limit((const void *)kfn, lds)); /* (per kernel: the variants share this site) */ \
    kfn<<<dim3((unsigned)grid), dim3(kFwdTB), lds, s>>>(p->nptr, p->urel, mpos_arg, V, comp, N, R, B, F,     \
                                                        addend, ldA, M, ldM, node_ids, p->tickets,          \
                                                        n_ctr, tk_tile);                                    \
  } while (0)
#define MIXM_TN(KS_, NQ_) MIXM_GO(KS_, NQ_, tn)
      switch (KS * 8 + NQ) {
        case 1 * 8 + 1: MIXM_TN(1, 1); break;
        case 2 * 8 + 1: MIXM_TN(2, 1); break;
        case 2 * 8 + 2: MIXM_TN(2, 2); break;
        case 3 * 8 + 1: MIXM_TN(3, 1); break;
        case 3 * 8 + 2: MIXM_TN(3, 2); break;
        case 3 * 8 + 3: MIXM_TN(3, 3); break;
        case 4 * 8 + 1: MIXM_TN(4, 1); break;
        case 4 * 8 + 2: MIXM_TN(4, 2); break;
        case 4 * 8 + 3: MIXM_TN(4, 3); break;
        default: MIXM_TN(4, 4); break;
      }
#undef MIXM_TN
#undef MIXM_GO
      MRGCN_HIP_TRY(hipGetLastError());
      return MRGCN_OK;
    }
  }
  if constexpr (sizeof(AT) != 4) {
    set_error("basis mix with a bf16 addend: outside the matrix-core form's limits (B <= 64, F <= 16, B*F % 4 == 0)");
    return MRGCN_ERR_UNSUPPORTED;
  } else {
  int acc = 0;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int nb = (B - b0 < 64) ? (B - b0) : 64;
    int BT = nb <= 2 ? 2 : nb <= 4 ? 4 : nb <= 8 ? 8 : nb <= 16 ? 16 : nb <= 32 ? 32 : nb <= 40 ? 40 : 64;
    size_t lds = (size_t)R * comp_stride(BT) * sizeof(float);
    int in_lds = lds <= kLdsBudget;
    if (!in_lds) lds = 0;
    const int fwd_tb = (int)cfg(CFG_MIX_FWD_TB);
    int grid = mix_grid(lds, N * FW);
    const float *add = acc ? nullptr : addend;
#define MIX_GO(T)                                                                                    \
  k_mix_fwd<T, OT><<<dim3(grid), dim3(fwd_tb), lds, s>>>(p->nptr, p->urel, mpos_arg, V, comp, N, R, B, b0, \
                                                     F, FW, add, ldA, M, ldM, acc, in_lds, node_ids)
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
}

template <typename OT>
int mix_fwd_impl(const mrgcn_plan_t *p, const float *V, const float *comp, int32_t B, int32_t F,
                 const float *addend, int64_t ldA, OT *M, int64_t ldM, void *stream) {
  MRGCN_REQUIRE(p, "NULL");
  const MixCols c{p->nptr, p->urel, p->mpos, p->unode, nullptr, p->num_nodes, p->ncols, (int)p->num_relations,
                  p->work_tickets};
  return mix_fwd_cols<OT>(&c, V, comp, B, F, addend, ldA, M, ldM, stream);
}
}  // namespace

namespace mrgcn {
int mix_fwd_arrays(const int32_t *nptr, const int32_t *urel, const int32_t *node_ids, int64_t n_nodes, int64_t ncols,
                   int R, const float *V, const float *comp, int32_t B, int32_t F, float *M, int64_t ldM,
                   hipStream_t s) {
  const MixCols c{nptr, urel, nullptr, nullptr, node_ids, n_nodes, ncols, R};
  return mix_fwd_cols<float, float>(&c, V, comp, B, F, (const float *)nullptr, 0, M, ldM, (void *)s);
}
}  // namespace mrgcn

namespace {

// ---- basis contraction of weight_F (graph.py:83-85): W[r] = sum_b comp[r, b] V[b] --------------------------------
// (R x B) . (B x X), X = in * out: a few hundred thousand outputs of B terms each — one small launch of this
// package instead of a library GEMM; backward: dV[b] = sum_r comp[r, b] dW[r] and dcomp[r, b] = <dW[r], V[b]> in one
// launch (two block ranges).  Sums in a fixed order: bitwise reproducible.
__global__ __launch_bounds__(kTB) void k_basis_contract(const float *__restrict__ comp, const float *__restrict__ V,
                                                        int R, int B, int64_t X, float *__restrict__ W) {
  const int64_t total = (int64_t)R * X;
  for (int64_t t = (int64_t)blockIdx.x * kTB + threadIdx.x; t < total; t += (int64_t)gridDim.x * kTB) {
    const int64_t r = t / X, x = t - r * X;
    const float *c = comp + r * B;
    float s0 = 0.f, s1 = 0.f;
    int b = 0;
    for (; b + 2 <= B; b += 2) {
      s0 = fmaf(c[b], V[(int64_t)b * X + x], s0);
      s1 = fmaf(c[b + 1], V[(int64_t)(b + 1) * X + x], s1);
    }
    if (b < B) s0 = fmaf(c[b], V[(int64_t)b * X + x], s0);
    W[t] = s0 + s1;
  }
}

// backward, dV (dcomp: k_basis_contract_dcomp below).
//   dV[b, x] = sum_r comp[r, b] dW[r, x]: a block owns 64 columns x and every basis b; comp lives in LDS (rows padded
//   to BT floats), thread (x, q = wave) sums the relations r = q mod 4 into B register accumulators (one coalesced
//   load of dW per relation, eight in flight; broadcast 16-byte LDS reads of the comp row), the four partial sums meet
//   in LDS in a fixed order.  B <= kContractMaxB and R * BT * 4 + 4 * BT * 256 bytes of LDS; otherwise the plain
//   thread-per-output walk.  (Measured on the way: comp through 4-byte loads in a loop 120 us, through scalar loads per
//   relation 150 us — both a chain of dependent round trips.)
//   NW waves per block (round 6: 16 where the launch has few blocks — X / 64: 25 and 2 at the AM shape — and is a chain
//   of comp-staging and relation rounds: 24 -> 17 us; waves 4 .. NW - 1 add their partial sums into the four LDS slots
//   in three further phases, a fixed order).
constexpr int kContractMaxB = 64;
template <int BT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_basis_contract_bwd(const float *__restrict__ comp, const float *__restrict__ V,
                                                            const float *__restrict__ dW, int R, int B, int64_t X,
                                                            float *__restrict__ dcomp, float *__restrict__ dV,
                                                            int dv_blocks, int tiled) {
  constexpr int kTB = 64 * NW;  // (shadows the file's 256)
  extern __shared__ float s_mem[];
  {
    if (!tiled) {  // dV[b, x]: a thread per output
      const int64_t total = (int64_t)B * X;
      for (int64_t t = (int64_t)blockIdx.x * kTB + threadIdx.x; t < total; t += (int64_t)dv_blocks * kTB) {
        const int64_t b = t / X, x = t - b * X;
        float s0 = 0.f, s1 = 0.f;
        int r = 0;
        for (; r + 2 <= R; r += 2) {
          s0 = fmaf(comp[(int64_t)r * B + b], dW[(int64_t)r * X + x], s0);
          s1 = fmaf(comp[(int64_t)(r + 1) * B + b], dW[(int64_t)(r + 1) * X + x], s1);
        }
        if (r < R) s0 = fmaf(comp[(int64_t)r * B + b], dW[(int64_t)r * X + x], s0);
        dV[t] = s0 + s1;
      }
      return;
    }
    float *s_comp = s_mem;                         // [R][BT], rows padded to BT (a multiple of four) floats
    float *s_part = s_mem + (size_t)R * BT;        // [4][BT][64]
    // comp into LDS: every thread's loads in flight at once (a loop of dependent 4-byte loads cost 40 us here)
    for (int t0 = 0; t0 < R * BT; t0 += kTB * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u * kTB + threadIdx.x;
        const int r = t / BT, b = t - r * BT;
        v[u] = (t < R * BT && b < B) ? comp[(int64_t)r * B + b] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u * kTB + threadIdx.x;
        if (t < R * BT) s_comp[t] = v[u];
      }
    }
    __syncthreads();
    const int xl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t x = (int64_t)blockIdx.x * 64 + xl;
    const int64_t xc = x < X ? x : X - 1;  // (clamped: unconditional loads, eight relations in flight)
    float acc[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[b] = 0.f;
    int r = q;
    for (; r + 7 * NW < R; r += 8 * NW) {
      float d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) d[u] = dW[(int64_t)(r + NW * u) * X + xc];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 *c4 = reinterpret_cast<const float4 *>(s_comp + (r + NW * u) * BT);  // broadcast 16-byte LDS reads
#pragma unroll
        for (int b = 0; b < BT; b += 4) {
          const float4 c = c4[b >> 2];
          acc[b] = fmaf(c.x, d[u], acc[b]);
          acc[b + 1] = fmaf(c.y, d[u], acc[b + 1]);
          acc[b + 2] = fmaf(c.z, d[u], acc[b + 2]);
          acc[b + 3] = fmaf(c.w, d[u], acc[b + 3]);
        }
      }
    }
    for (; r < R; r += NW) {
      const float d = dW[(int64_t)r * X + xc];
      const float4 *c4 = reinterpret_cast<const float4 *>(s_comp + r * BT);
#pragma unroll
      for (int b = 0; b < BT; b += 4) {
        const float4 c = c4[b >> 2];
        acc[b] = fmaf(c.x, d, acc[b]);
        acc[b + 1] = fmaf(c.y, d, acc[b + 1]);
        acc[b + 2] = fmaf(c.z, d, acc[b + 2]);
        acc[b + 3] = fmaf(c.w, d, acc[b + 3]);
      }
    }
    // waves 0 .. 3 store, then waves 4 .. 7, 8 .. 11, 12 .. 15 add into slot q % 4 one group after the other
#pragma unroll
    for (int ph = 0; ph < NW / 4; ++ph) {
      if ((q >> 2) == ph) {
#pragma unroll
        for (int b = 0; b < BT; ++b)
          if (b < B) {
            float *slot = s_part + ((q & 3) * BT + b) * 64 + xl;
            *slot = ph == 0 ? acc[b] : *slot + acc[b];
          }
      }
      __syncthreads();
    }
    for (int t = threadIdx.x; t < B * 64; t += kTB) {
      const int b = t >> 6, xx = t & 63;
      const int64_t xo = (int64_t)blockIdx.x * 64 + xx;
      if (xo < X)
        dV[(int64_t)b * X + xo] = (s_part[(0 * BT + b) * 64 + xx] + s_part[(1 * BT + b) * 64 + xx]) +
                                  (s_part[(2 * BT + b) * 64 + xx] + s_part[(3 * BT + b) * 64 + xx]);
    }
  }
}

// dcomp[r, b] = <dW[r, :], V[b, :]>: a wave per output, lanes over x, eight loads of each operand in flight.  Its own
// launch: sharing one with the dV blocks would give every block their 90 KB of LDS (one block per CU).
__global__ __launch_bounds__(kTB) void k_basis_contract_dcomp(const float *__restrict__ V, const float *__restrict__ dW,
                                                              int R, int B, int64_t X, float *__restrict__ dcomp) {
  const int lane = threadIdx.x & 63;
  const int64_t w = ((int64_t)blockIdx.x * kTB + threadIdx.x) >> 6;
  if (w >= (int64_t)R * B) return;
  const int64_t r = w / B, b = w - r * B;
  const float *a = dW + r * X, *v = V + b * X;
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  int64_t x = lane;
  for (; x + 7 * 64 < X; x += 8 * 64) {
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = fmaf(a[x + u * 64], v[x + u * 64], s[u]);
  }
  for (; x < X; x += 64) s[0] = fmaf(a[x], v[x], s[0]);
  float t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if (lane == 0) dcomp[w] = t;
}

template <typename OT>
int gather_rows_impl(const mrgcn_plan_t *p, const float *W, int32_t F, const float *addend, int64_t ldA,
                     OT *M, int64_t ldM, void *stream) {
  MRGCN_REQUIRE(p && W && M, "NULL");
  MRGCN_REQUIRE(F > 0 && ldM >= F, "F / ldM");
  MRGCN_REQUIRE(!addend || ldA >= F, "ldA");
  if (p->ncols == 0) return MRGCN_OK;
  k_gather_rows<OT><<<dim3(grid_for(p->ncols * ldM)), dim3(kTB), 0, (hipStream_t)stream>>>(
      p->ulcol, p->mpos, p->ncols, W, F, (int)ldM, addend, ldA, M, ldM);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

template <typename OT>
int rel_transform_fwd_impl(const mrgcn_plan_t *p, const float *X, int64_t ldX, int32_t K, const float *W,
                           int32_t F, OT *Out, int64_t ldOut, int32_t operand_order, void *stream) {
  MRGCN_REQUIRE(p && X && W && Out, "NULL");
  MRGCN_REQUIRE(K > 0 && F > 0 && ldX >= K && ldOut >= F, "K / F / leading dimensions");
  MRGCN_REQUIRE(F <= 64, "rel_transform supports F <= 64 (tile the feature dimension)");
  const RelOrder o = p->order_for(K);  // narrow inputs: the order with narrow node bands
  if (o.n_relchunks == 0) return MRGCN_OK;
  const int32_t *oidx = operand_order ? p->mpos : nullptr;
  // narrow rows with every relation's weights in LDS: columns in OUTPUT order, the operand leaves as one stream
  if (xform_cols_lds_supported(p, K, F, ldOut, operand_order != 0))
    return xform_cols_lds(p, operand_order != 0, X, ldX, K, W, F, Out, ldOut, (hipStream_t)stream, sizeof(OT) == 2);
  if (use_mfma() && xform_mfma_fwd_supported(K, F))
    return xform_mfma_fwd(p, o, o.rnode, operand_order ? o.rmpos : nullptr, X, ldX, K, W, false, F, Out,
                          ldOut, (hipStream_t)stream, sizeof(OT) == 2);
  size_t lds = ((size_t)kKS * F + (size_t)kTK * (kKS + 1)) * sizeof(float);
  k_xform_fwd<OT><<<dim3(o.n_relchunks), dim3(kTB), lds, (hipStream_t)stream>>>(
      o.relchunk_rel, o.relchunk_beg, o.relchunk_end, o.rperm, p->unode, oidx, X, ldX, K, W, F,
      (int)ldOut, Out, ldOut);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // namespace

extern "C" {

int mrgcn_basis_mix_fwd_f32(const mrgcn_plan_t *p, const float *V, const float *comp, int32_t B,
                            int32_t F, const float *addend, int64_t ldA, float *M, int64_t ldM,
                            void *stream) {
  return mix_fwd_impl<float>(p, V, comp, B, F, addend, ldA, M, ldM, stream);
}

int mrgcn_basis_mix_fwd_bf16(const mrgcn_plan_t *p, const float *V, const float *comp, int32_t B,
                             int32_t F, const float *addend, int64_t ldA, uint16_t *M, int64_t ldM,
                             void *stream) {
  MRGCN_REQUIRE(B <= 64, "basis_mix_fwd_bf16: more than 64 bases would re-round the operand per pass");
  return mix_fwd_impl<uint16_t>(p, V, comp, B, F, addend, ldA, M, ldM, stream);
}

int mrgcn_gather_rows_bf16(const mrgcn_plan_t *p, const float *W, int32_t F, const float *addend,
                           int64_t ldA, uint16_t *M, int64_t ldM, void *stream) {
  return gather_rows_impl<uint16_t>(p, W, F, addend, ldA, M, ldM, stream);
}

int mrgcn_rel_transform_fwd_bf16(const mrgcn_plan_t *p, const float *X, int64_t ldX, int32_t K,
                                 const float *W, int32_t F, uint16_t *Out, int64_t ldOut,
                                 int32_t operand_order, void *stream) {
  return rel_transform_fwd_impl<uint16_t>(p, X, ldX, K, W, F, Out, ldOut, operand_order, stream);
}

// ---- the bf16 pipeline (activations stored in bf16; parameters, accumulation and the backward's sums fp32) ----------
int mrgcn_cast_rows_bf16(const float *src, int64_t ldSrc, int64_t rows, int32_t K, uint16_t *dst, int64_t ldDst,
                         void *stream) {
  MRGCN_REQUIRE(src && dst, "NULL");
  MRGCN_REQUIRE(K > 0 && ldSrc >= K && ldDst >= K && ldDst % 8 == 0, "K / leading dimensions (ldDst: whole 16-byte pieces)");
  MRGCN_REQUIRE((((uintptr_t)dst) & 15) == 0, "dst must be 16-byte aligned");
  return cast_rows_bf16(src, ldSrc, rows, K, dst, ldDst, (hipStream_t)stream);
}

int32_t mrgcn_rel_transform_xbf16_supported(const mrgcn_plan_t *p, int32_t K, int32_t F, int64_t ldX, int64_t ldOut) {
  return p && use_mfma() && xform_bf16_fwd_supported(K, F, ldX, ldOut) ? 1 : 0;
}

int mrgcn_rel_transform_fwd_xbf16(const mrgcn_plan_t *p, const uint16_t *X, int64_t ldX, int32_t K, const float *W,
                                  int32_t F, void *Out, int64_t ldOut, int32_t operand_order, int32_t out_bf16,
                                  void *stream) {
  MRGCN_REQUIRE(p && X && W && Out, "NULL");
  MRGCN_REQUIRE((((uintptr_t)X) & 15) == 0, "X must be 16-byte aligned");
  if (!mrgcn_rel_transform_xbf16_supported(p, K, F, ldX, ldOut)) {
    set_error("mrgcn_rel_transform_fwd_xbf16: K <= 256, F <= ldOut <= 16, ldX a multiple of 8 elements");
    return MRGCN_ERR_UNSUPPORTED;
  }
  const RelOrder o = p->order_for(K);
  if (o.n_relchunks == 0) return MRGCN_OK;
  return xform_bf16_fwd(p, o, o.rnode, operand_order ? o.rmpos : nullptr, X, ldX, K, W, F, Out, ldOut,
                        (hipStream_t)stream, out_bf16 != 0);
}

int mrgcn_basis_mix_fwd_abf16(const mrgcn_plan_t *p, const float *V, const float *comp, int32_t B, int32_t F,
                              const uint16_t *addend, int64_t ldA, void *M, int64_t ldM, int32_t out_bf16,
                              void *stream) {
  MRGCN_REQUIRE(p && addend, "NULL");
  MRGCN_REQUIRE(B <= 64 && F <= 16 && (B * F) % 4 == 0, "basis_mix_fwd_abf16: B <= 64, F <= 16, B*F % 4 == 0");
  const MixCols c{p->nptr, p->urel, p->mpos, p->unode, nullptr, p->num_nodes, p->ncols, (int)p->num_relations,
                  p->work_tickets};
  if (out_bf16) return mix_fwd_cols<uint16_t, uint16_t>(&c, V, comp, B, F, addend, ldA, (uint16_t *)M, ldM, stream);
  return mix_fwd_cols<float, uint16_t>(&c, V, comp, B, F, addend, ldA, (float *)M, ldM, stream);
}

}  // extern "C"

namespace {
// rows of dM whose liveness flag is 0 were possibly never written (mrgcn_spmm_transposed_live_f32 with
// write_dead_rows = 0): kernels that read every row get zeros there first
__global__ void k_zero_dead_rows(float *__restrict__ dM, int64_t ldM, int F, const uint8_t *__restrict__ col_live,
                                 int64_t ncols) {
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols;
       c += (int64_t)gridDim.x * blockDim.x) {
    if (col_live[c]) continue;
    float *row = dM + c * ldM;
    for (int o = 0; o < F; ++o) row[o] = 0.f;
  }
}

static int zero_dead_rows(float *dM, int64_t ldM, int F, const uint8_t *col_live, int64_t ncols, hipStream_t s) {
  if (ncols == 0) return MRGCN_OK;
  int64_t blocks = (ncols + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  k_zero_dead_rows<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(dM, ldM, F, col_live, ncols);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

// wave-over-nodes form (k_mix_bwd_nm); MRGCN_OK when it ran, -1 when the shape is outside its limits
int mix_bwd_nm_launch_arrays(const int32_t *nptr, const int32_t *urel, int64_t N, int R, int top_rel, const float *dM,
                             int64_t ldM, const float *V, const float *comp, int32_t B, int32_t F, float *dV,
                             float *dcomp, double *dV_sumsq, hipStream_t s, const uint8_t *col_live,
                             uint8_t *node_cur) {
  const bool node_on = cfg(CFG_MIX_NODE) != 0;
  bool ok = node_on && F <= 16 && B <= 64 && N > 0;
  if ((F & 1) == 0) ok = ok && (((uintptr_t)V | (uintptr_t)dV) & 7) == 0;  // 8-byte row accesses
  if (!ok) return -1;
  size_t lds = (size_t)R * B * sizeof(float);
  const int dc_in_lds = lds <= 96 * 1024;
  if (!dc_in_lds) lds = 0;
  // register arrays of exactly F features for the hidden sizes of the BASELINE configs (10, 11)
  const int FT = (F == 10 || F == 11) ? F : (F + 3) / 4 * 4;
  const int tb_cfg = (int)cfg(CFG_MIX_BWD_TB);
  const int tb = (tb_cfg >= 64 && tb_cfg <= kNodeTB) ? tb_cfg / 64 * 64 : 512;
  int per_cu = lds > 0 ? (int)((160 * 1024) / (lds + 1024)) : 4;
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 4) per_cu = 4;
  if (per_cu * tb > 2048) per_cu = 2048 / tb;  // 32 waves per CU at most
  const int per_cu_cap = (int)cfg(CFG_MIX_BWD_PER_CU);  // experiments
  if (per_cu_cap > 0 && per_cu > per_cu_cap) per_cu = per_cu_cap;
  const int64_t want = ((N + kGroup - 1) / kGroup + (tb / 64) - 1) / (tb / 64);
  int64_t grid = (int64_t)256 * per_cu;
  if (grid > want) grid = want;
#define NODE_GO(T)                                                                                        \
  do {                                                                                                    \
    auto kfn = k_mix_bwd_nm<T>;                                                                           \
    MRGCN_HIP_TRY(mrgcn::raise_lds_limit((const void *)kfn, lds));                                                 \
    kfn<<<dim3((unsigned)grid), dim3(tb), lds, s>>>(nptr, urel, dM, ldM, V, comp, N, R, B, F, dV,        \
                                                         dcomp, dV_sumsq, top_rel, col_live,              \
                                                         node_cur, dc_in_lds);                           \
  } while (0)
  switch (FT) {
    case 4: NODE_GO(4); break;
    case 8: NODE_GO(8); break;
    case 10: NODE_GO(10); break;
    case 11: NODE_GO(11); break;
    case 12: NODE_GO(12); break;
    default: NODE_GO(16); break;
  }
#undef NODE_GO
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mix_bwd_nm_launch(const mrgcn_plan_t *p, const float *dM, int64_t ldM, const float *V, const float *comp,
                      int32_t B, int32_t F, float *dV, float *dcomp, double *dV_sumsq, hipStream_t s,
                      const uint8_t *col_live, uint8_t *node_cur) {
  return mix_bwd_nm_launch_arrays(p->nptr, p->urel, p->num_nodes, (int)p->num_relations, (int)p->top_rel, dM, ldM, V,
                                  comp, B, F, dV, dcomp, dV_sumsq, s, col_live, node_cur);
}

int mix_bwd_dv_launch(const mrgcn_plan_t *p, const float *dM, int64_t ldM, const float *comp, int32_t B,
                      int32_t F, float *dV, double *dV_sumsq, hipStream_t s) {
  const int R = (int)p->num_relations;
  const int64_t N = p->num_nodes;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int nb = (B - b0 < 64) ? (B - b0) : 64;
    int BT = nb <= 2 ? 2 : nb <= 4 ? 4 : nb <= 8 ? 8 : nb <= 16 ? 16 : nb <= 32 ? 32 : nb <= 40 ? 40 : 64;
    size_t lds = (size_t)R * comp_stride(BT) * sizeof(float);
    int in_lds = lds <= kLdsBudget;
    if (!in_lds) lds = 0;
    int grid = mix_grid(lds, N * F);
#define MIXDV_GO(T)                                                                                       \
  k_mix_bwd_dv<T><<<dim3(grid), dim3(kMixTB), lds, s>>>(p->nptr, p->urel, dM, ldM, comp, N, R, B, b0, F, \
                                                        dV, in_lds, dV_sumsq)
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
  return MRGCN_OK;
}
}  // namespace

extern "C" {

int32_t mrgcn_adam_rows_fused_supported(const mrgcn_plan_t *p, int32_t B, int32_t F) {
  const bool on = cfg(CFG_FUSED_ADAM) != 0;
  const bool node_on = cfg(CFG_MIX_NODE) != 0;
  if (!p || !on || !node_on) return 0;
  const int64_t R = p->num_relations;
  return (B > 0 && B <= 64 && F > 0 && F <= 16 && (B * F) % 4 == 0 && p->num_nodes > 0 &&
          (size_t)R * B * sizeof(float) <= 64 * 1024) ? 1 : 0;
}

int mrgcn_adam_step_rows_fused_f32(const mrgcn_plan_t *p, const float *dM, int64_t ldM, const uint8_t *col_live,
                                   const float *comp, int32_t B, int32_t F, float *param, float *exp_avg,
                                   float *exp_avg_sq, const uint8_t *row_cur, uint8_t *row_ever, float lr,
                                   float beta1, float beta2, float eps, int64_t step, const float *bc_dev,
                                   const float *grad_scale, void *stream) {
  MRGCN_REQUIRE(p && dM && comp && param && exp_avg && exp_avg_sq && row_cur && row_ever, "NULL");
  MRGCN_REQUIRE(mrgcn_adam_rows_fused_supported(p, B, F), "shape outside mrgcn_adam_rows_fused_supported");
  MRGCN_REQUIRE(ldM >= F, "ldM");
  MRGCN_REQUIRE(((((uintptr_t)param) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15) == 0,
                "param / moments must be 16-byte aligned");
  MRGCN_REQUIRE(bc_dev || step >= 1, "step");
  if (p->ncols == 0)  // no column, no gradient: moments of `ever` nodes decay (the gradient pointer is never read)
    return mrgcn_adam_step_rows_f32(param, param, exp_avg, exp_avg_sq, p->num_nodes, B * F, row_cur, row_ever, lr,
                                    beta1, beta2, eps, step, bc_dev, grad_scale, stream);
  return mrgcn::adam_rows_fused_arrays(p->nptr, p->urel, col_live, p->num_nodes, (int)p->num_relations, dM, ldM, comp,
                                       B, F, param, exp_avg, exp_avg_sq, row_cur, row_ever, lr, beta1, beta2, eps,
                                       step, bc_dev, grad_scale, (hipStream_t)stream);
}

}  // extern "C"

namespace mrgcn {
bool xform_use_mfma() { return use_mfma(); }

int mix_bwd_nm_arrays(const int32_t *nptr, const int32_t *urel, int64_t N, int R, int top_rel, const float *dM,
                      int64_t ldM, const float *V, const float *comp, int32_t B, int32_t F, float *dV, float *dcomp,
                      double *dV_sumsq, hipStream_t s, const uint8_t *col_live, uint8_t *node_cur) {
  return mix_bwd_nm_launch_arrays(nptr, urel, N, R, top_rel, dM, ldM, V, comp, B, F, dV, dcomp, dV_sumsq, s, col_live,
                                  node_cur);
}

static bool adam_list_enabled() {
  const bool on = cfg(CFG_ADAM_LIST) != 0;
  return on;
}

int adam_rows_fused_arrays(const int32_t *nptr, const int32_t *urel, const uint8_t *col_live, int64_t N, int R,
                           const float *dM, int64_t ldM, const float *comp, int32_t B, int32_t F, float *param,
                           float *exp_avg, float *exp_avg_sq, const uint8_t *row_cur, uint8_t *row_ever, float lr,
                           float beta1, float beta2, float eps, int64_t step, const float *bc_dev,
                           const float *grad_scale, hipStream_t s, const int32_t *lnode, const int32_t *lnptr, int64_t NL,
                           int ever_outside) {
  float bc1 = 1.f, bc2s = 1.f;
  if (!bc_dev) {
    bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  }
  const size_t lds = (size_t)R * B * sizeof(float);
  const int nv = (B * F) / 4;
  int per_cu = (int)((150 * 1024) / (lds + 512));
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  int64_t grid = (int64_t)256 * per_cu;
  const int64_t want = (N + kFusedTB / 64 - 1) / (kFusedTB / 64);
  if (grid > want) grid = want;
  // a list of the nodes with gradient (a gradient support) and blocks of at most two 16-byte pieces per lane: the
  // pipelined kernel takes the list, the flag scan below then only visits what the list left out
  const bool listed = lnode && lnptr && !col_live && nv <= 128 && adam_list_enabled();
  if (listed && NL > 0) {
    const int64_t lwant = (NL + kFusedTB / 64 - 1) / (kFusedTB / 64);
    int64_t lgrid = 256;  // 4 waves per SIMD (two node blocks in flight per wave): one block of 16 waves per CU
    if (lgrid > lwant) lgrid = lwant;
    const int once = (int)cfg(CFG_ADAM_ONCE);  // nodes per wave of the one-shot form (0: the persistent list kernel)
    if (once > 0 && B <= 64) {
      // one-shot grid: a wave owns `once` consecutive list entries (see k_adam_rows_once)
      const int npw = once >= 4 ? 4 : once >= 2 ? 2 : 1;
      const int64_t waves = (NL + npw - 1) / npw;
      const dim3 ogrid((unsigned)((waves + 3) / 4));
#define ADAM_ONCE_GO(NH_, NPW_)                                                                                    \
  k_adam_rows_once<NH_, NPW_><<<ogrid, dim3(256), 0, s>>>(lnode, lnptr, urel, dM, ldM, comp, NL, R, B, F, param,   \
                                                          exp_avg, exp_avg_sq, row_ever, lr, beta1, beta2, eps,    \
                                                          bc1, bc2s, grad_scale, bc_dev)
      if (nv <= 64) { if (npw == 4) ADAM_ONCE_GO(1, 4); else if (npw == 2) ADAM_ONCE_GO(1, 2); else ADAM_ONCE_GO(1, 1); }
      else { if (npw == 4) ADAM_ONCE_GO(2, 4); else if (npw == 2) ADAM_ONCE_GO(2, 2); else ADAM_ONCE_GO(2, 1); }
#undef ADAM_ONCE_GO
      MRGCN_HIP_TRY(hipGetLastError());
    } else {
#define ADAM_LIST_GO(NH_)                                                                                          \
  do {                                                                                                             \
    auto kfn = k_adam_rows_list<NH_, true>;                                                                        \
    MRGCN_HIP_TRY(mrgcn::raise_lds_limit((const void *)kfn, lds));                                                 \
    kfn<<<dim3((unsigned)lgrid), dim3(kFusedTB), lds, s>>>(lnode, lnptr, urel, dM, ldM, comp, NL, R, B, F, param,  \
                                                           exp_avg, exp_avg_sq, row_ever, lr, beta1, beta2, eps,   \
                                                           bc1, bc2s, grad_scale, bc_dev);                         \
  } while (0)
    if (nv <= 64) ADAM_LIST_GO(1);
    else ADAM_LIST_GO(2);
#undef ADAM_LIST_GO
    MRGCN_HIP_TRY(hipGetLastError());
    }
  }
  if (!listed || ever_outside) {
    auto kfn = k_adam_rows_fused;
    MRGCN_HIP_TRY(mrgcn::raise_lds_limit((const void *)kfn, lds));
    kfn<<<dim3((unsigned)grid), dim3(kFusedTB), lds, s>>>(nptr, urel, col_live, dM, ldM, comp, N, R, B, F,
                                                          param, exp_avg, exp_avg_sq, row_cur, row_ever, lr, beta1,
                                                          beta2, eps, bc1, bc2s, grad_scale, bc_dev, listed ? 1 : 0);
  }
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}
}  // namespace mrgcn

extern "C" {

int mrgcn_basis_mix_bwd_f32(const mrgcn_plan_t *p, float *dM, int64_t ldM, const uint8_t *col_live,
                            const float *V, const float *comp, int32_t B, int32_t F, float *dV,
                            uint8_t *node_cur, float *dcomp, double *dV_sumsq, void *stream) {
  MRGCN_REQUIRE(p && dM && V && comp && dcomp, "NULL");
  MRGCN_REQUIRE(B > 0 && F > 0 && ldM >= F, "B / F / ldM");
  // dV == NULL: the norm-only pass in front of mrgcn_adam_step_rows_fused_f32 (flags, dcomp and ||dV||^2 only)
  MRGCN_REQUIRE(dV || (node_cur && dV_sumsq && mrgcn_adam_rows_fused_supported(p, B, F)),
                "dV may only be NULL with node_cur and dV_sumsq, on shapes mrgcn_adam_rows_fused_supported accepts");
  hipStream_t s = (hipStream_t)stream;
  const int R = (int)p->num_relations;
  const int64_t N = p->num_nodes;
  MRGCN_HIP_TRY(mrgcn::fill_async(dcomp, 0, (size_t)R * B * sizeof(float), s));
  {  // one pass for dV and dcomp when the shape allows it
    int rc = mix_bwd_nm_launch(p, dM, ldM, V, comp, B, F, dV, dcomp, dV_sumsq, s, col_live, node_cur);
    if (rc >= 0) return rc;
  }
  MRGCN_REQUIRE(dV, "the norm-only pass needs the wave-over-nodes kernel");
  // the two-kernel form reads every row of dM (rows flagged dead may be unwritten) and writes every
  // block of dV: every node counts as written
  if (col_live) {
    int rc = zero_dead_rows(dM, ldM, F, col_live, p->ncols, s);
    if (rc != MRGCN_OK) return rc;
  }
  if (node_cur && N > 0) MRGCN_HIP_TRY(mrgcn::fill_async(node_cur, 1, (size_t)N, s));
  {
    int rc = mix_bwd_dv_launch(p, dM, ldM, comp, B, F, dV, dV_sumsq, s);
    if (rc != MRGCN_OK) return rc;
  }
  // dcomp, the feature dimension in tiles of <= 64 (dcomp accumulates over the tiles)
  if (p->ncols > 0) {
    size_t lds = (size_t)R * (B | 1) * sizeof(float);
    int in_lds = lds <= kLdsBudget;
    if (!in_lds) lds = 0;
    int grid = mix_grid(lds, p->ncols);
    const bool wide_on = cfg(CFG_DCOMP_WIDE) != 0;
    if (F > 64 && wide_on) {  // a wave per column over whole rows
      const size_t lds_w = in_lds ? (size_t)R * B * sizeof(float) : 0;
      int64_t blocks = (p->ncols + 3) / 4;
      if (blocks > 8192) blocks = 8192;
      k_mix_bwd_dcomp_wide<<<dim3((unsigned)blocks), dim3(256), lds_w, s>>>(p->urel, p->unode, dM, ldM, V, F, R, B, F,
                                                                         p->ncols, dcomp, in_lds);
      MRGCN_HIP_TRY(hipGetLastError());
    } else
    for (int f0 = 0; f0 < F; f0 += 64) {
      const int Ft = (F - f0 < 64) ? (F - f0) : 64;
      const float *dMt = dM + f0, *Vt = V + f0;
      const bool v2 = (Ft % 2 == 0) && (F % 2 == 0) && (((uintptr_t)Vt) % 8 == 0);
#define MIXDC_GO(T)                                                                                    \
  do {                                                                                                 \
    if (v2)                                                                                            \
      k_mix_bwd_dcomp<T, true><<<dim3(grid), dim3(kMixTB), lds, s>>>(p->urel, p->unode, dMt, ldM, Vt,  \
                                                                     F, N, R, B, Ft, p->ncols, dcomp,  \
                                                                     in_lds);                          \
    else                                                                                               \
      k_mix_bwd_dcomp<T, false><<<dim3(grid), dim3(kMixTB), lds, s>>>(p->urel, p->unode, dMt, ldM, Vt, \
                                                                      F, N, R, B, Ft, p->ncols, dcomp, \
                                                                      in_lds);                         \
  } while (0)
      if (Ft <= 4) MIXDC_GO(4);
      else if (Ft <= 8) MIXDC_GO(8);
      else if (Ft <= 12) MIXDC_GO(12);
      else if (Ft <= 16) MIXDC_GO(16);
      else if (Ft <= 32) MIXDC_GO(32);
      else MIXDC_GO(64);
#undef MIXDC_GO
      MRGCN_HIP_TRY(hipGetLastError());
    }
  }
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_basis_contract_f32(const float *comp, const float *V, int32_t R, int32_t B, int64_t X, float *W,
                             void *stream) {
  MRGCN_REQUIRE(comp && V && W, "NULL");
  MRGCN_REQUIRE(R > 0 && B > 0 && X > 0, "R / B / X");
  k_basis_contract<<<dim3(grid_for((int64_t)R * X)), dim3(kTB), 0, (hipStream_t)stream>>>(comp, V, R, B, X, W);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_basis_contract_bwd_f32(const float *comp, const float *V, const float *dW, int32_t R, int32_t B, int64_t X,
                                 float *dcomp, float *dV, void *stream) {
  MRGCN_REQUIRE(comp && V && dW, "NULL");
  MRGCN_REQUIRE(R > 0 && B > 0 && X > 0, "R / B / X");
  const int BT = B <= 16 ? 16 : (B <= 32 ? 32 : (B <= 48 ? 48 : 64));
  const size_t lds = ((size_t)R * BT + (size_t)4 * BT * 64) * sizeof(float);
  const int tiled = B <= kContractMaxB && lds <= 150 * 1024;
  const int dv_blocks = !dV ? 0 : (tiled ? (int)((X + 63) / 64) : grid_for((int64_t)B * X));
  const int dc_blocks = dcomp ? (int)(((int64_t)R * B * 64 + kTB - 1) / kTB) : 0;
  if (dc_blocks > 0) {
    k_basis_contract_dcomp<<<dim3(dc_blocks), dim3(kTB), 0, (hipStream_t)stream>>>(V, dW, R, B, X, dcomp);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  if (dv_blocks == 0) return MRGCN_OK;
  const dim3 grid(dv_blocks);
  const size_t sh = tiled ? lds : 0;
  // few blocks (X / 64) and many relations: sixteen waves each; otherwise four (more blocks per CU, shorter chains)
  const bool wide_blocks = tiled && dv_blocks <= 256 && R >= 128;
#define CONTRACT_BWD(BT_)                                                                                          \
  do {                                                                                                             \
    if (wide_blocks) {                                                                                             \
      auto kfn = k_basis_contract_bwd<BT_, 16>;                                                                    \
      MRGCN_HIP_TRY(mrgcn::raise_lds_limit((const void *)kfn, sh));                                                \
      kfn<<<grid, dim3(1024), sh, (hipStream_t)stream>>>(comp, V, dW, R, B, X, dcomp, dV, dv_blocks, tiled);       \
    } else {                                                                                                       \
      auto kfn = k_basis_contract_bwd<BT_>;                                                                        \
      MRGCN_HIP_TRY(mrgcn::raise_lds_limit((const void *)kfn, sh));                                                \
      kfn<<<grid, dim3(kTB), sh, (hipStream_t)stream>>>(comp, V, dW, R, B, X, dcomp, dV, dv_blocks, tiled);        \
    }                                                                                                              \
  } while (0)
  if (BT == 16) CONTRACT_BWD(16);
  else if (BT == 32) CONTRACT_BWD(32);
  else if (BT == 48) CONTRACT_BWD(48);
  else CONTRACT_BWD(64);
#undef CONTRACT_BWD
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_gather_rows_f32(const mrgcn_plan_t *p, const float *W, int32_t F, const float *addend,
                          int64_t ldA, float *M, int64_t ldM, void *stream) {
  return gather_rows_impl<float>(p, W, F, addend, ldA, M, ldM, stream);
}

int mrgcn_rel_transform_fwd_f32(const mrgcn_plan_t *p, const float *X, int64_t ldX, int32_t K,
                                const float *W, int32_t F, float *Out, int64_t ldOut,
                                int32_t operand_order, void *stream) {
  return rel_transform_fwd_impl<float>(p, X, ldX, K, W, F, Out, ldOut, operand_order, stream);
}

int64_t mrgcn_rel_transform_bwd_workspace(const mrgcn_plan_t *p, int32_t K, int32_t F, int32_t need_dX,
                                          int32_t need_dW) {
  if (!p) return 0;
  int64_t a = need_dX ? p->ncols * (((int64_t)K + 3) / 4 * 4) : 0;
  int64_t b = need_dW ? (int64_t)p->order_for(K).n_relchunks * K * F : 0;
  return a > b ? a : b;
}

int mrgcn_rel_transform_bwd_f32(const mrgcn_plan_t *p, const float *dM, int64_t ldM, const float *X,
                                int64_t ldX, int32_t K, const float *W, int32_t F, float *dX,
                                int64_t lddX, float *dW, float *workspace, int64_t workspace_floats,
                                void *stream) {
  return mrgcn_rel_transform_bwd_live_f32(p, const_cast<float *>(dM), ldM, nullptr, X, ldX, K, W, F, dX, lddX,
                                          dW, workspace, workspace_floats, stream);
}

int mrgcn_rel_transform_bwd_live_f32(const mrgcn_plan_t *p, float *dM, int64_t ldM,
                                     const uint8_t *col_live, const float *X, int64_t ldX, int32_t K,
                                     const float *W, int32_t F, float *dX, int64_t lddX, float *dW,
                                     float *workspace, int64_t workspace_floats, void *stream) {
  return mrgcn_rel_transform_bwd_masked_f32(p, dM, ldM, col_live, X, ldX, K, W, F, dX, lddX, dW, workspace,
                                            workspace_floats, 0, nullptr, nullptr, stream);
}

int32_t mrgcn_rel_transform_bwd_masked_supported(const mrgcn_plan_t *p, int32_t K, int32_t F, int64_t workspace_floats) {
  const int64_t ldZ = ((int64_t)K + 3) / 4 * 4;
  return p && K <= 16 && use_mfma() && workspace_floats >= p->ncols * ldZ && xform_mfma_fwd_supported(F, K);
}

int mrgcn_rel_transform_bwd_masked_f32(const mrgcn_plan_t *p, float *dM, int64_t ldM,
                                       const uint8_t *col_live, const float *X, int64_t ldX, int32_t K,
                                       const float *W, int32_t F, float *dX, int64_t lddX, float *dW,
                                       float *workspace, int64_t workspace_floats, int32_t relu_mask_from_x,
                                       uint8_t *row_live_out, const uint8_t *node_live, void *stream) {
  MRGCN_REQUIRE(p && dM && X && W, "NULL");
  MRGCN_REQUIRE(!col_live || ((uintptr_t)col_live & 7) == 0, "col_live must be 8-byte aligned");
  MRGCN_REQUIRE(!(relu_mask_from_x || row_live_out) ||
                    (dX && mrgcn_rel_transform_bwd_masked_supported(p, K, F, workspace ? workspace_floats : 0)),
                "the masked / flagged dX needs K <= 16 and the matrix-core path (see ..._masked_supported)");
  MRGCN_REQUIRE(K > 0 && F > 0 && ldX >= K && ldM >= F, "K / F / leading dimensions");
  MRGCN_REQUIRE(F <= 64, "rel_transform supports F <= 64 (tile the feature dimension)");
  hipStream_t s = (hipStream_t)stream;
  if (col_live) {
    // every kernel below that sweeps all columns reads every row of dM: rows flagged dead may be
    // unwritten, so unless both halves take the live-column form they are zeroed first
    const bool dw_live = !dW || (use_mfma() && xform_mfma_dw_live_supported(K, F));
    const int64_t ldZ_ = ((int64_t)K + 3) / 4 * 4;
    const bool dx_live = !dX || (use_mfma() && workspace && workspace_floats >= p->ncols * ldZ_ &&
                                 xform_mfma_dx_supported(F, K));
    if (!(dw_live && dx_live)) {
      int rc = zero_dead_rows(dM, ldM, F, col_live, p->ncols, s);
      if (rc != MRGCN_OK) return rc;
    }
  }
  if (dW) {
    const RelOrder o = p->order_for(K);
    // (the matrix-core form with a slab workspace zeroes dW inside its first launch)
    const bool self_zero = use_mfma() && xform_mfma_dw_supported(K, F) && workspace &&
                           workspace_floats >= (int64_t)o.n_relchunks * K * F;
    if (!self_zero) MRGCN_HIP_TRY(mrgcn::fill_async(dW, 0, (size_t)p->num_relations * K * F * sizeof(float), s));
    if (use_mfma() && xform_mfma_dw_supported(K, F)) {
      int rc = xform_mfma_dw(p, o, o.rnode, X, ldX, K, dM, ldM, F, dW, workspace, workspace_floats, s, col_live);
      if (rc != MRGCN_OK) return rc;
    } else if (p->n_relchunks > 0) {
      size_t lds = ((size_t)kTK * (kKS + 1) + (size_t)kTK * F) * sizeof(float);
      k_xform_bwd_dw<<<dim3(p->n_relchunks), dim3(kTB), lds, s>>>(
          p->relchunk_rel, p->relchunk_beg, p->relchunk_end, p->rperm, p->unode, X, ldX, K, dM, ldM, F, dW);
      MRGCN_HIP_TRY(hipGetLastError());
    }
  }
  const int64_t ldZ = ((int64_t)K + 3) / 4 * 4;
  if (dX && use_mfma() && workspace && workspace_floats >= p->ncols * ldZ && xform_mfma_dx_supported(F, K)) {
    // Z[c, 0:K] = dM[c, 0:F] . W[r_c]^T on the matrix cores, then dX[j] = sum of node j's Z rows
    MRGCN_REQUIRE(lddX >= K, "lddX");
    // (rows of dM and of Z in plain compact order; the walk follows the order of the narrower of the two)
    int rc = xform_mfma_fwd(p, p->order_for(F), nullptr, nullptr, dM, ldM, F, W, true, K, workspace, ldZ, s, false,
                            col_live);
    if (rc != MRGCN_OK) return rc;
    rc = segment_sum(p, workspace, ldZ, K, dX, lddX, s, col_live, relu_mask_from_x ? X : nullptr, ldX, row_live_out,
                     col_live ? node_live : nullptr);
    if (rc != MRGCN_OK) return rc;
  } else if (dX) {
    MRGCN_REQUIRE(lddX >= K, "lddX");
    k_xform_bwd_dx<<<dim3(grid_for(p->num_nodes * K)), dim3(kTB), 0, s>>>(
        p->nptr, p->urel, dM, ldM, W, p->num_nodes, K, F, dX, lddX);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  return MRGCN_OK;
}

}  // extern "C"
