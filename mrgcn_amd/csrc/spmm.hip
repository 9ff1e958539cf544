// Stacked-CSR sparse x dense product  Y[i,:] = sum_e val[e] * D[idx[e],:]  for gfx950.
//
// Replaces torch.mm(A.float(), dense) of mrgcn/layers/graph.py:75,:95 and its autograd
// (dDense = A^T dY) when run on the transposed view.
//
// HBM-bound gather: a dense row is only F*4 bytes (40-64 B at the R-GCN hidden sizes), so
// the design goal is many independent row gathers in flight per wave, vector loads
// (VEC floats per lane) and coalesced index streaming — not FLOPs.
//
// One launch covers every row class (block ranges of the same grid):
//   short rows (<= kLongThreshold entries): a subgroup of G lanes owns one row; a 64-lane
//       wave walks 64/G adjacent rows at once, so every load instruction carries 64/G
//       independent row gathers, 4 deep per lane.
//   long rows: cut at plan time into chunks of <= kChunk entries, one wave per chunk: all
//       64/G slots stride over the chunk, a butterfly over the slots reduces them; a row
//       that is one chunk stores its result directly, a row of several chunks stores
//       partials that a second tiny launch adds in chunk order.
// No atomics anywhere: results are bitwise reproducible and Y needs no pre-zeroing.
#include <cstdlib>

#include "common.hpp"
#include "config.hpp"

namespace mrgcn {
namespace {

template <int VEC> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };

// VEC consecutive operand elements -> floats.  DT = float, or bf16 (raw uint16_t; the forward
// operand may be stored in bf16, accumulation stays fp32): 2*VEC bytes per lane, VEC <= 8.
template <int VEC, typename DT> __device__ __forceinline__ void load_vec(const DT *p, float (&x)[VEC]) {
  if constexpr (sizeof(DT) == 4) {
    using T = typename Vec<VEC>::T;
    T t = *reinterpret_cast<const T *>(p);
    const float *f = reinterpret_cast<const float *>(&t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) x[i] = f[i];
  } else if constexpr (VEC == 1) {
    x[0] = bf16_to_f32(p[0]);
  } else {
    uint32_t w[VEC / 2];
    if constexpr (VEC == 2) w[0] = *reinterpret_cast<const uint32_t *>(p);
    if constexpr (VEC == 4) { const uint2 t = *reinterpret_cast<const uint2 *>(p); w[0] = t.x; w[1] = t.y; }
    if constexpr (VEC == 8) { const uint4 t = *reinterpret_cast<const uint4 *>(p); w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w; }
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i) {
      x[2 * i] = __uint_as_float(w[i] << 16);
      x[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
}

template <int VEC>
__device__ __forceinline__ void store_row(float *y, const float (&acc)[VEC], int f0, int F,
                                          const float *bias, int relu, bool vec_ok, int room = 0) {
  float o[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    float v = acc[i];
    if (bias && f0 + i < F) v += bias[f0 + i];
    if (relu) v = fmaxf(v, 0.f);
    o[i] = (f0 + i < F) ? v : 0.f;
  }
  if (vec_ok && f0 + VEC <= (room > F ? room : F)) {
    if constexpr (VEC == 8) {
      *reinterpret_cast<float4 *>(y + f0) = *reinterpret_cast<const float4 *>(o);
      *reinterpret_cast<float4 *>(y + f0 + 4) = *reinterpret_cast<const float4 *>(o + 4);
    } else {
      using T = typename Vec<VEC>::T;
      *reinterpret_cast<T *>(y + f0) = *reinterpret_cast<const T *>(o);
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i)
      if (f0 + i < F) y[f0 + i] = o[i];
  }
}

// One gather + multiply-add of entry (c, a); `on` masks the contribution.
// TAIL: rows are only dword aligned and not padded (ld = F not a multiple of 4): 16-byte loads
// from dword-aligned addresses (legal for global loads on gfx950), scalar loads for a last
// partial vector so that nothing past the row is touched.
template <int VEC, bool TAIL, typename DT>
__device__ __forceinline__ void gather_fma(const DT *Dq, int64_t ldD, int32_t c, float a, bool on,
                                           float (&acc)[VEC], int nvalid) {
  // branch-free: the load is unconditional (masked-off entries carry c = 0, a valid row) so that
  // the compiler can issue a whole batch of gathers before the first use; the select keeps a
  // NaN/Inf in row 0 from leaking into rows that do not reference it
  float x[VEC];
  if constexpr (TAIL) {
    if (nvalid < VEC) {
      const DT *p = Dq + (int64_t)c * ldD;
#pragma unroll
      for (int i = 0; i < VEC; ++i) x[i] = (i < nvalid) ? load_operand<DT>(p + i) : 0.f;
    } else {
      load_vec<VEC, DT>(Dq + (int64_t)c * ldD, x);
    }
  } else {
    load_vec<VEC, DT>(Dq + (int64_t)c * ldD, x);
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = fmaf(a, on ? x[i] : 0.f, acc[i]);
}

// The sum of a split row's partial sums (chunks c0 .. c1 of one row), by one wave, in a FIXED order — the arithmetic of
// the two-pass form (k_spmm_finalize) and of the in-kernel finalize (the last arriving chunk of k_spmm) alike, so that
// both give the same bits.  AGENT: the partial sums were written by other CUs / XCDs a moment ago: agent-scope loads.
template <bool AGENT>
__device__ __forceinline__ float load_partial(const float *p) {
  if constexpr (AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
// WIDE: the caller's feature tile can hold 32 or more features (compile time: the wide path's 64 registers of loads in
// flight would otherwise set the register count — and the occupancy — of the narrow instantiations too: 54 -> 96 VGPRs,
// eight -> five waves per SIMD, the TRANSPOSED product at F = 10 443 -> 626 us)
template <bool AGENT, int NFT = 4>  // NFT: 64-feature groups whose loads fly together (0: the narrow path only)
__device__ __forceinline__ void finish_split_row(int32_t c0, int32_t c1, int64_t row, const float *partials, int ldP,
                                                 int F, float *__restrict__ Y, int64_t ldY,
                                                 const float *__restrict__ bias, int relu, int lane) {
  if constexpr (NFT > 0) if (F >= 32) {
    // wide rows: lanes over features (coalesced: lane l owns features l, l + 64, l + 128, l + 192 of a tile of up to
    // 256); chunk c adds into accumulator c mod 8 of its feature, and the loads of eight chunks x four features are
    // in flight together (a hub row of the FB15k-237 shape has 300+ partial sums of 200 floats: with two accumulators
    // and one feature at a time this wave's chain of round trips was the last 17 us of the product)
    for (int fb = 0; fb < F; fb += NFT * kWave) {
      float a[NFT][8];
#pragma unroll
      for (int t = 0; t < NFT; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u) a[t][u] = 0.f;
      int fo[NFT];
#pragma unroll
      for (int t = 0; t < NFT; ++t) fo[t] = min(fb + t * kWave + lane, F - 1);  // (clamped: lanes past F read feature F-1)
      for (int32_t c = c0; c < c1; c += 8) {
        float x[NFT][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float *pr = partials + (int64_t)min(c + u, c1 - 1) * ldP;
#pragma unroll
          for (int t = 0; t < NFT; ++t) x[t][u] = load_partial<AGENT>(pr + fo[t]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (c + u < c1) {  // wave uniform
#pragma unroll
            for (int t = 0; t < NFT; ++t) a[t][u] += x[t][u];
          }
      }
#pragma unroll
      for (int t = 0; t < NFT; ++t) {
        const int f = fb + t * kWave + lane;
        if (f < F) {
          float s = ((a[t][0] + a[t][1]) + (a[t][2] + a[t][3])) + ((a[t][4] + a[t][5]) + (a[t][6] + a[t][7]));
          if (bias) s += bias[f];
          if (relu) s = fmaxf(s, 0.f);
          Y[row * ldY + f] = s;
        }
      }
    }
    return;
  }
  for (int f = 0; f < F; ++f) {
    float s = 0.f;
    for (int32_t c = c0 + lane; c < c1; c += kWave) s += load_partial<AGENT>(partials + (int64_t)c * ldP + f);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
    if (lane == 0) {
      if (bias) s += bias[f];
      if (relu) s = fmaxf(s, 0.f);
      Y[row * ldY + f] = s;
    }
  }
}

// Latency, not bandwidth, bounds a naive row walk (pointer -> index -> gather is three dependent
// round trips per few entries).  Both paths therefore first pull *all* indices and values of
// their row / chunk into registers with coalesced loads (one round trip), then hand them to the
// gathering lanes with cross-lane reads and issue the gathers back to back.
// LIVE (split-row blocks only): `op_live` flags the operand rows that hold anything but zeros and
// `out_live` the output rows that receive any of them; dead entries are not gathered (they
// add a * 0), chunks of dead output rows store zeros.
template <int G, int VEC, bool TAIL, typename DT, bool LIVE = false>
__global__ __launch_bounds__(256) void k_spmm(SparseView v, const DT *__restrict__ D, int64_t ldD,
                                              int F, float *__restrict__ Y, int64_t ldY,
                                              const float *__restrict__ bias, int relu,
                                              const int32_t *__restrict__ out_index,
                                              int store_vec_ok, float *__restrict__ partials,
                                              int ldP, int chunk_blocks, int64_t short_blocks,
                                              int64_t xcd_per, int min_len,
                                              const uint8_t *__restrict__ op_live = nullptr,
                                              const uint8_t *__restrict__ out_live = nullptr) {
  constexpr int SLOTS = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int slot = lane / G, q = lane % G;
  const int f0 = q * VEC;
  const bool active = f0 < F;
  const DT *Dq = D + (active ? f0 : 0);  // idle feature lanes shadow lane 0 (always in bounds)
  const int nvalid = active ? min(VEC, F - f0) : VEC;  // floats of this lane's vector inside the row
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;

  if ((int)blockIdx.x < chunk_blocks) {
    // ---- long rows: one wave per chunk of <= kChunk entries ------------------------------
    const int64_t chunk = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    if (chunk >= v.n_chunks) return;
    const int32_t b = v.chunk_beg[chunk];
    int32_t n = v.chunk_end[chunk] - b;
    if constexpr (LIVE) {
      const int32_t cr = v.chunk_row[chunk];
      if (!out_live[cr >= 0 ? cr : -cr - 2]) n = 0;  // wave uniform: nothing live reaches this row
    }
    constexpr int T = kChunk / kWave;  // staged registers per lane
    int32_t ci[T];
    float ca[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {  // entry b + 64 t + lane: fully coalesced
      const int32_t m = t * kWave + lane;
      ci[t] = (m < n) ? v.idx[b + m] : 0;
      ca[t] = (m < n) ? v.val[b + m] : 0.f;
    }
    constexpr int UPT = kWave / SLOTS;  // = G steps of SLOTS entries per staged register
    if constexpr (LIVE) {
      // the flags of the staged entries, one look-up per register; only live entries are gathered
      // (same slots, same order as below: identical sums)
#pragma unroll
      for (int t = 0; t < T; ++t) {
        if (t * kWave >= n) break;  // wave uniform
        const int32_t m = t * kWave + lane;
        const int lv = (m < n) ? (int)op_live[ci[t]] : 0;
        if (!__any(lv)) continue;
        for (int u = 0; u < UPT; ++u) {
          const int src = u * SLOTS + slot;
          const int32_t c = __shfl(ci[t], src, kWave);
          const float a = __shfl(ca[t], src, kWave);
          const int on = __shfl(lv, src, kWave);
          if (on) gather_fma<VEC, TAIL, DT>(Dq, ldD, c, a, active, acc, nvalid);
        }
      }
    } else
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t * kWave >= n) break;  // wave uniform
      if (UPT <= 8) {
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
          const int src = u * SLOTS + slot;
          const int32_t c = __shfl(ci[t], src, kWave);
          const float a = __shfl(ca[t], src, kWave);
          gather_fma<VEC, TAIL, DT>(Dq, ldD, c, a, active && (t * kWave + src < n), acc, nvalid);
        }
      } else {
        // few slots per wave (wide rows: G >= 16): eight gathers in flight per slot — one at a time the wave waited a
        // full round trip per 800-byte row (FB15k-237 shape, F = 200: 128 -> see DESIGN §5)
        for (int u0 = 0; u0 < UPT; u0 += 8) {
          if (t * kWave + u0 * SLOTS >= n) break;  // wave uniform
#pragma unroll
          for (int u = u0; u < u0 + 8; ++u) {
            const int src = u * SLOTS + slot;
            const int32_t c = __shfl(ci[t], src, kWave);
            const float a = __shfl(ca[t], src, kWave);
            gather_fma<VEC, TAIL, DT>(Dq, ldD, c, a, active && (t * kWave + src < n), acc, nvalid);
          }
        }
      }
    }
    // butterfly over the slots (lanes with equal q); every lane takes part
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
    }
    const int32_t row = v.chunk_row[chunk];  // (wave uniform)
    if (row >= 0) {  // the whole row was this chunk: finished
      if (slot == 0 && active) {
        const int64_t orow = out_index ? (int64_t)out_index[row] : (int64_t)row;
        store_row<VEC>(Y + orow * ldY, acc, f0, F, bias, relu, store_vec_ok != 0);
      }
      return;
    }
    float *p = partials + chunk * (int64_t)ldP + f0;
    if (LIVE || !v.ticket) {  // two-pass form: k_spmm_finalize adds the chunks' sums
      if (slot == 0 && active) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          if (f0 + i < F) p[i] = acc[i];
      }
      return;
    }
    // ---- in-kernel finalize (as in k_spmm3): the wave that brings a row's LAST partial sum adds them all, in chunk
    // order — the result does not depend on which wave that is.  Hand-off across CUs / XCDs (MI355X_MICROARCH.md,
    // "inter-workgroup visibility"): agent-scope stores, wait, agent-scope add to the row's arrival counter; the last
    // arriver reads the sums with agent-scope loads and puts the counter back to zero for the next launch.
    if (slot == 0 && active) {
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        if (f0 + i < F) __hip_atomic_store(p + i, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // position of the row among the long rows (second half of chunk_row: plan.hip k_long_fill)
    const int32_t ck = __builtin_amdgcn_readfirstlane((int32_t)chunk);
    const int32_t li = v.chunk_row[v.n_chunks + ck], c0 = v.long_cptr[li], c1 = v.long_cptr[li + 1];
    int32_t old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(v.ticket + li, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    asm volatile("" ::: "memory");
    if (old != c1 - c0 - 1) return;
    const int32_t lrow = -row - 2;
    const int64_t orow = out_index ? (int64_t)out_index[lrow] : (int64_t)lrow;
    // (loads in flight sized to the tile: the finishing path must not set the register count of the product)
    constexpr int NFT = G * VEC < 32 ? 0 : G * VEC <= 64 ? 1 : G * VEC <= 128 ? 2 : 4;
    finish_split_row<true, NFT>(c0, c1, orow, partials, ldP, F, Y, ldY, bias, relu, lane);
    if (lane == 0) __hip_atomic_store(v.ticket + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  // ---- short rows ----------------------------------------------------------------------
  // blocks b and b+8 share an XCD (round-robin dispatch): give every XCD one contiguous run
  // of rows so that a row block and its successor meet in the same L2
  int64_t sb = (int64_t)blockIdx.x - chunk_blocks;
  if (xcd_per > 0) {
    sb = (sb & 7) * xcd_per + (sb >> 3);
    if (sb >= short_blocks) return;
  }
  const int64_t wave = (sb * blockDim.x + threadIdx.x) / kWave;
  const int64_t row = wave * SLOTS + slot;
  // no early return: every lane takes part in the cross-lane reads below
  int32_t b = 0, n = 0;
  if (row < v.rows) {
    b = v.ptr[row];
    n = v.ptr[row + 1] - b;
  }
  // rows longer than kLongThreshold belong to the chunk path; with the tiny-row pre-pass
  // (min_len > 0) rows of <= min_len entries (empty rows included) were already written
  const bool mine = row < v.rows && n <= kLongThreshold && !(min_len > 0 && n <= min_len);
  if (!mine) n = 0;
  if (!__any(mine)) return;  // wave uniform
  if (G <= 8) {
    // the G lanes of a row stage its <= 32 entries: lane q holds entries q, q+G, q+2G, ...
    constexpr int T = (kLongThreshold + G - 1) / G;
    int32_t ci[T];
    float ca[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int32_t m = t * G + q;
      ci[t] = (m < n) ? v.idx[b + m] : 0;
      ca[t] = (m < n) ? v.val[b + m] : 0.f;
    }
    const int sbase = slot * G;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (!__any(t * G < n)) break;  // wave uniform: nobody has entries left
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int32_t c = __shfl(ci[t], sbase + u, kWave);
        const float a = __shfl(ca[t], sbase + u, kWave);
        gather_fma<VEC, TAIL, DT>(Dq, ldD, c, a, active && (t * G + u < n), acc, nvalid);
      }
    }
  } else {
    // wide rows (G >= 16): the slot's <= 32 entries are staged with coalesced loads (lane q of the slot holds
    // entries q, q + G), then gathered eight at a time — index and gather are no longer two dependent round trips
    constexpr int T2 = (kLongThreshold + G - 1) / G;
    int32_t ci[T2];
    float ca[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t) {
      const int32_t m = t * G + q;
      ci[t] = (m < n) ? v.idx[b + m] : 0;
      ca[t] = (m < n) ? v.val[b + m] : 0.f;
    }
    const int sbase = slot * G;
    constexpr int GB = G < kLongThreshold ? G : kLongThreshold;  // entries a register of the slot can hold
#pragma unroll
    for (int t = 0; t < T2; ++t) {
      for (int u0 = 0; u0 < GB; u0 += 8) {
        if (!__any(t * G + u0 < n)) break;  // wave uniform
#pragma unroll
        for (int u = u0; u < u0 + 8; ++u) {
          const int32_t c = __shfl(ci[t], sbase + u, kWave);
          const float a = __shfl(ca[t], sbase + u, kWave);
          gather_fma<VEC, TAIL, DT>(Dq, ldD, c, a, active && (t * G + u < n), acc, nvalid);
        }
      }
    }
  }
  if (mine && active) {
    const int64_t orow = out_index ? (int64_t)out_index[row] : row;
    store_row<VEC>(Y + orow * ldY, acc, f0, F, bias, relu, store_vec_ok != 0);
  }
}

// =====================================================================================================
// k_spmm3 — the forward product for narrow layers (G <= 4 lanes of float4 per row: F <= 16).
//
// What bounds k_spmm above at F = 10 is not bandwidth but dependent round trips: it gathers four entries
// per row, waits, looks whether any row has entries left, gathers four more (rows of <= 32 entries: up to
// eight rounds; chunks of 512: 32 rounds) — measured 118 us with every operand row cache resident.  Here
// every wave issues ALL its gathers (at most eight per lane) in one basic block, after one round of staging
// loads, whatever it works on:
//   S  rows of <= 8 entries: 64/G consecutive rows per wave, a slot of G lanes each
//   M  rows of 9..32 entries (plan list `mid_rows`): four rows per wave, 16 lanes = 16/G sub-slots each,
//      sub-slot s takes entries s, s + 16/G, ...; a 16-lane butterfly joins them
//   L  rows of > 32 entries: plan chunks of <= 128 entries, one wave each, slot s takes entries s, s + 64/G, ...;
//      a wave butterfly joins the slots; rows of several chunks leave partials for k_spmm3_finalize
// The number of gathers is wave uniform, so the batch is picked by a switch over straight-line code: the
// compiler's wait-count pass serialises loads that sit behind exec-dependent branches.  No atomics: bitwise
// reproducible.  AM shape, F = 10: S 75 us / M 36 us / L 60 us on their own, each at the HBM rate of the
// bytes it touches.
// =====================================================================================================
constexpr int kShort3 = kShort3Rows;    // S: rows of at most this many entries
constexpr int kMid3 = kMid3Rows;        // M: up to this many
constexpr int kChunk3 = kChunk3Entries; // L: entries per chunk (8 gathers per lane at G = 4)

// OFF32: the operand is smaller than 4 GB — a gather address is the uniform base (a scalar register pair) plus a
// 32-bit byte offset (one vector register instead of two per gather in flight)
template <int NT, int VEC, bool TAIL, typename DT, bool OFF32, typename GetFn, typename OnFn>
__device__ __forceinline__ void gather_batch(const DT *__restrict__ Dq, int64_t ldD, bool active, int nvalid,
                                             float (&acc)[VEC], GetFn get, OnFn on, const DT *__restrict__ Dbase,
                                             uint32_t ldb, uint32_t qoff) {
  float x[NT][VEC];
  float a[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int32_t c;
    get(t, c, a[t]);
    const DT *p;
    if constexpr (OFF32)
      p = reinterpret_cast<const DT *>(reinterpret_cast<const char *>(Dbase) + ((uint32_t)c * ldb + qoff));
    else
      p = Dq + (int64_t)c * ldD;
    if constexpr (TAIL) {
      if (nvalid < VEC) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) x[t][i] = (i < nvalid) ? (float)p[i] : 0.f;
      } else {
        load_vec<VEC, DT>(p, x[t]);
      }
    } else {
      load_vec<VEC, DT>(p, x[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const bool o = active && on(t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = fmaf(a[t], o ? x[t][i] : 0.f, acc[i]);
  }
}

template <int VEC, bool TAIL, typename DT, bool OFF32, typename GetFn, typename OnFn>
__device__ __forceinline__ void gather_rounds(int nr, const DT *__restrict__ Dq, int64_t ldD, bool active,
                                              int nvalid, float (&acc)[VEC], GetFn get, OnFn on,
                                              const DT *__restrict__ Dbase, uint32_t ldb, uint32_t qoff) {
  switch (nr) {  // wave uniform
    case 1: gather_batch<1, VEC, TAIL, DT, OFF32>(Dq, ldD, active, nvalid, acc, get, on, Dbase, ldb, qoff); break;
    case 2: gather_batch<2, VEC, TAIL, DT, OFF32>(Dq, ldD, active, nvalid, acc, get, on, Dbase, ldb, qoff); break;
    case 3: gather_batch<3, VEC, TAIL, DT, OFF32>(Dq, ldD, active, nvalid, acc, get, on, Dbase, ldb, qoff); break;
    case 4: gather_batch<4, VEC, TAIL, DT, OFF32>(Dq, ldD, active, nvalid, acc, get, on, Dbase, ldb, qoff); break;
    case 5: gather_batch<5, VEC, TAIL, DT, OFF32>(Dq, ldD, active, nvalid, acc, get, on, Dbase, ldb, qoff); break;
    case 6: gather_batch<6, VEC, TAIL, DT, OFF32>(Dq, ldD, active, nvalid, acc, get, on, Dbase, ldb, qoff); break;
    case 7: gather_batch<7, VEC, TAIL, DT, OFF32>(Dq, ldD, active, nvalid, acc, get, on, Dbase, ldb, qoff); break;
    case 8: gather_batch<8, VEC, TAIL, DT, OFF32>(Dq, ldD, active, nvalid, acc, get, on, Dbase, ldb, qoff); break;
    default: break;
  }
}

struct View3 {  // what k_spmm3 needs beyond the COMPACT SparseView (whose rows are class-major ranks)
  int32_t n_short = 0, n_mid = 0;      // ranks [0, n_short): S rows, the next n_mid: M rows, the rest: L rows
  int32_t n_chunks = 0, n_long = 0, n_multi = 0;
  const int32_t *rowmap = nullptr;     // [rows] rank -> output row
  const int32_t *chunk_beg = nullptr, *chunk_end = nullptr, *chunk_row = nullptr;  // [n_chunks], rows = ranks
  const int32_t *long_row = nullptr, *long_cptr = nullptr;                   // [n_long] ranks, [n_long + 1]
  int64_t op_rows = 0;                 // rows of the operand the indices point into
  int32_t pad_ok = 0;                  // columns F..ldY-1 of Y belong to the caller's buffer and may be zeroed
  const int32_t *multi = nullptr;      // [n_multi_rows] positions in long_row of the rows of several chunks
  int32_t n_multi_rows = 0;
  int32_t s_n = 0;                     // rows of one wave each (kMid3 < n <= kChunk3): first entry, end, rank
  const int32_t *s_beg = nullptr, *s_end = nullptr, *s_row = nullptr;
  int32_t *ticket = nullptr;           // [n_long] arrival counters (zero between launches)
  int32_t fold = 0;                    // rows of several chunks are finished inside k_spmm3 by their last chunk's wave
};

// sum of a blockwise row's block sums, lane (k, f): x[u] = sum of block j0 + 4u (j0 = first block + k); blocks are
// added in a fixed order — u first, then the butterfly over k — by k_spmm3's last arriver and by k_spmm3_finalize alike
__device__ __forceinline__ float xl_row_sum(const float (&x)[4], int32_t j0, int32_t b1) {
  float s = j0 < b1 ? x[0] : 0.f;
#pragma unroll
  for (int u = 1; u < 4; ++u) s += (j0 + 4 * u < b1) ? x[u] : 0.f;
  s += __shfl_xor(s, 16, kWave);
  s += __shfl_xor(s, 32, kWave);
  return s;
}

// OFF32: gathers address the operand with 32-bit byte offsets (operand < 4 GB).  WPE: waves per SIMD the register
// allocation must leave room for (7 at F = 10 / 16: 72 registers instead of 76, a seventh wave of gathers in flight
// per SIMD: -4 %; 8 would take the batch of eight gathers apart: no gain).  Three-lane slots for the S rows at
// F <= 12 (21 rows per wave) gain the same 3-4 % on their own and nothing on top: measured, not kept.
template <int G, int VEC, bool TAIL, typename DT, bool OFF32 = false, int WPE = 1>
__global__ __launch_bounds__(256, WPE) void k_spmm3(SparseView v, View3 w, const DT *__restrict__ D, int64_t ldD,
                                               int F, float *__restrict__ Y, int64_t ldY,
                                               const float *__restrict__ bias, int relu, int store_vec_ok,
                                               float *__restrict__ partials, int ldP, int chunk_blocks,
                                               int single_blocks, int mid_blocks, int64_t short_blocks,
                                               int64_t xcd_per, int pack) {
  constexpr int SLOTS = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int slot = lane / G, q = lane % G;
  const int f0 = q * VEC;
  const bool active = f0 < F;
  // pack: operand rows of exactly F floats (no pad, only dword aligned; F >= VEC).  Lane q loads the floats
  // [lo, lo + VEC) with lo = min(f0, F - VEC): the row's last vector overlaps its neighbour's instead of reaching past
  // the row (F = 10: bytes 0-15, 16-31, 24-39 of a 40-byte row) — every gather stays one 16-byte load inside its row.
  // The sums are kept in the loaded layout and moved to the lane's own features (f0 ..) once, before they leave.
  const int lo = pack ? min(f0, F - VEC) : f0;
  const int shift = f0 - lo;
  const DT *Dq = D + (active ? lo : 0);  // idle feature lanes shadow lane 0 (always in bounds)
  const int nvalid = active ? (pack ? VEC : min(VEC, F - f0)) : VEC;
  const uint32_t ldb = (uint32_t)ldD * (uint32_t)sizeof(DT), qoff = (uint32_t)(active ? lo : 0) * (uint32_t)sizeof(DT);
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  auto unpack = [&]() {  // acc[i] <- the sum of feature f0 + i (zero past the row)
    if (shift) {
      float r[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        r[i] = 0.f;
#pragma unroll
        for (int j = 1; j < VEC; ++j)
          if (shift == j && i + j < VEC) r[i] = acc[i + j];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = r[i];
    }
  };

  if ((int)blockIdx.x < chunk_blocks) {  // ---- XL: a block = four equal chunks of ONE row of > kChunk3 entries
    __shared__ float s_part[4][16];
    const int wv = threadIdx.x >> 6;
    const int c = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);  // wave uniform: scalar loads below
    const int32_t cb = w.chunk_beg[c], ce = w.chunk_end[c];
    const int32_t li = -w.chunk_row[c] - 2;     // position of the row among the blockwise rows
    const int32_t lc0 = w.long_cptr[li], lc1 = w.long_cptr[li + 1];
    constexpr int T = kChunk3 / kWave;
    constexpr int PER = kWave / SLOTS;  // = G gather rounds per staged register
    for (int32_t b = cb; b < ce; b += kChunk3) {  // (chunks of more than kChunk3 entries: several pieces)
      const int32_t n = min(ce - b, kChunk3);
      int32_t ci[T];
      float ca[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {  // entry b + 64 t + lane: fully coalesced
        const int32_t m = t * kWave + lane;
        ci[t] = (m < n) ? v.idx[b + m] : 0;
        ca[t] = (m < n) ? v.val[b + m] : 0.f;
      }
      const int nr = __builtin_amdgcn_readfirstlane((n + SLOTS - 1) / SLOTS);
      gather_rounds<VEC, TAIL, DT, OFF32>(
          nr, Dq, ldD, active, nvalid, acc,
          [&](int t, int32_t &cc, float &aa) {
            const int src = (t % PER) * SLOTS + slot;
            cc = __shfl(ci[t / PER], src, kWave);
            aa = __shfl(ca[t / PER], src, kWave);
          },
          [&](int t) { return t * SLOTS + slot < n; }, D, ldb, qoff);
    }
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
    }
    // the block's four sums meet in LDS; wave 0 adds them in wave order
    unpack();
    if (slot == 0 && active) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) s_part[wv][f0 + i] = acc[i];
    }
    __syncthreads();
    if (wv != 0) return;
    const int f = lane & 15, k = lane >> 4;
    float t = ((s_part[0][f] + s_part[1][f]) + s_part[2][f]) + s_part[3][f];
    const int32_t b0 = lc0 >> 2, b1 = lc1 >> 2;  // the row's blocks (blockwise rows start on a multiple of four chunks)
    if (b1 - b0 > 1) {
      float *pp = partials + (int64_t)blockIdx.x * ldP + f;
      if (!w.fold) {  // two-pass form: k_spmm3_finalize adds the blocks' sums
        if (k == 0 && f < F) *pp = t;
        return;
      }
      // ---- in-kernel finalize: the block that brings a row's LAST partial sum adds them all, in block order (the
      // result does not depend on which block that is: bitwise reproducible).  Hand-off across CUs / XCDs
      // (MI355X_MICROARCH.md, "inter-workgroup visibility"): the block's sum leaves as agent-scope (sc1,
      // write-through) stores, the storing wave waits for them, then adds to the row's arrival counter with an
      // agent-scope atomic; the wave whose add returns "all others were here" reads the sums back with
      // agent-scope (sc1) loads, which bypass its CU's L1 — and puts the counter back to zero for the next launch.
      if (k == 0 && f < F) __hip_atomic_store(pp, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int32_t old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add(w.ticket + li, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      asm volatile("" ::: "memory");
      if (old != b1 - b0 - 1) return;
      float x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // unconditional loads at clamped addresses: all in flight at once
        const int32_t j = min(b0 + k + 4 * u, b1 - 1);
        x[u] = __hip_atomic_load(partials + (int64_t)j * ldP + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      t = xl_row_sum(x, b0 + k, b1);
      if (lane == 0) __hip_atomic_store(w.ticket + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int64_t orow = w.rowmap[w.long_row[li]];
    if (k == 0 && f < F) {
      if (bias) t += bias[f];
      if (relu) t = fmaxf(t, 0.f);
      Y[orow * ldY + f] = t;
    } else if (k == 0 && w.pad_ok && f < ldY && f < (F + 3) / 4 * 4) {
      Y[orow * ldY + f] = 0.f;
    }
    return;
  }
  if ((int)blockIdx.x < chunk_blocks + single_blocks) {  // ---- L: rows of kMid3 < n <= kChunk3 entries, one wave each
    const int c = __builtin_amdgcn_readfirstlane((blockIdx.x - chunk_blocks) * 4 + (threadIdx.x >> 6));
    if (c >= w.s_n) return;
    const int32_t b = w.s_beg[c], n = w.s_end[c] - b;
    const int32_t rk = w.s_row[c];
    constexpr int T = kChunk3 / kWave;
    constexpr int PER = kWave / SLOTS;
    int32_t ci[T];
    float ca[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int32_t m = t * kWave + lane;
      ci[t] = (m < n) ? v.idx[b + m] : 0;
      ca[t] = (m < n) ? v.val[b + m] : 0.f;
    }
    const int nr = __builtin_amdgcn_readfirstlane((n + SLOTS - 1) / SLOTS);
    gather_rounds<VEC, TAIL, DT, OFF32>(
        nr, Dq, ldD, active, nvalid, acc,
        [&](int t, int32_t &cc, float &aa) {
          const int src = (t % PER) * SLOTS + slot;
          cc = __shfl(ci[t / PER], src, kWave);
          aa = __shfl(ca[t / PER], src, kWave);
        },
        [&](int t) { return t * SLOTS + slot < n; }, D, ldb, qoff);
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
    }
    unpack();
    if (slot == 0 && active)
      store_row<VEC>(Y + (int64_t)w.rowmap[rk] * ldY, acc, f0, F, bias, relu, (store_vec_ok & 1) != 0, (store_vec_ok & 2) ? (int)ldY : 0);
    return;
  }
  if ((int)blockIdx.x < chunk_blocks + single_blocks + mid_blocks) {  // ---- M: four rows per wave, 16 lanes each
    constexpr int SS = 16 / G;                          // sub-slots per row
    const int wv = (blockIdx.x - chunk_blocks - single_blocks) * 4 + (threadIdx.x >> 6);
    const int rsel = lane >> 4, l16 = lane & 15, ss = slot % SS;
    const int mi = wv * 4 + rsel;
    int32_t b = 0, n = 0, row = -1;
    if (mi < w.n_mid) {
      const int32_t rk = w.n_short + mi;
      row = w.rowmap[rk];
      b = v.ptr[rk];
      n = v.ptr[rk + 1] - b;
    }
    if (!__any(row >= 0)) return;
    constexpr int T = kMid3 / 16;
    int32_t ci[T];
    float ca[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int32_t m = t * 16 + l16;
      ci[t] = (m < n) ? v.idx[b + m] : 0;
      ca[t] = (m < n) ? v.val[b + m] : 0.f;
    }
    int nr = 0;
#pragma unroll
    for (int t = 0; t < kMid3 / SS; ++t)
      if (t < 8 && __any(t * SS < n)) nr = t + 1;
    const int base = rsel * 16;
    gather_rounds<VEC, TAIL, DT, OFF32>(
        nr, Dq, ldD, active, nvalid, acc,
        [&](int t, int32_t &cc, float &aa) {  // sub-slot ss takes entry t*SS + ss of its row
          const int e = t * SS + ss;
          cc = __shfl(ci[e / 16 < T ? e / 16 : T - 1], base + (e & 15), kWave);
          aa = __shfl(ca[e / 16 < T ? e / 16 : T - 1], base + (e & 15), kWave);
        },
        [&](int t) { return t * SS + ss < n; }, D, ldb, qoff);
#pragma unroll
    for (int off = G; off < 16; off <<= 1) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
    }
    unpack();
    if (ss == 0 && row >= 0 && active) store_row<VEC>(Y + (int64_t)row * ldY, acc, f0, F, bias, relu, (store_vec_ok & 1) != 0, (store_vec_ok & 2) ? (int)ldY : 0);
    return;
  }
  // ---- S: 64/G consecutive ranks per wave, all of them rows of <= kShort3 entries
  // blocks b and b+8 share an XCD (round-robin dispatch): every XCD gets one contiguous run of ranks
  int64_t sb = (int64_t)blockIdx.x - chunk_blocks - single_blocks - mid_blocks;
  if (xcd_per > 0) {
    sb = (sb & 7) * xcd_per + (sb >> 3);
    if (sb >= short_blocks) return;
  }
  const int64_t rk = (sb * 4 + (threadIdx.x >> 6)) * SLOTS + slot;
  int32_t b = 0, n = 0;
  int64_t row = 0;
  const bool mine = rk < w.n_short;
  if (mine) {
    b = v.ptr[rk];
    n = v.ptr[rk + 1] - b;
    row = w.rowmap[rk];
  }
  if (!__any(mine)) return;  // wave uniform
  constexpr int T = (kShort3 + G - 1) / G;
  int32_t ci[T];
  float ca[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int32_t m = t * G + q;
    ci[t] = (m < n) ? v.idx[b + m] : 0;
    ca[t] = (m < n) ? v.val[b + m] : 0.f;
  }
  int nr = 0;
#pragma unroll
  for (int t = 0; t < kShort3; ++t)
    if (__any(t < n)) nr = t + 1;
  const int sbase = slot * G;
  gather_rounds<VEC, TAIL, DT, OFF32>(
      nr, Dq, ldD, active, nvalid, acc,
      [&](int t, int32_t &cc, float &aa) {
        cc = __shfl(ci[t / G], sbase + (t % G), kWave);
        aa = __shfl(ca[t / G], sbase + (t % G), kWave);
      },
      [&](int t) { return t < n; }, D, ldb, qoff);
  unpack();
  if (mine && active) store_row<VEC>(Y + row * ldY, acc, f0, F, bias, relu, (store_vec_ok & 1) != 0, (store_vec_ok & 2) ? (int)ldY : 0);
}

// two-pass form: rows of several blocks — one wave per row adds the blocks' sums exactly as the last arriver would
__global__ __launch_bounds__(256) void k_spmm3_finalize(View3 w, const float *__restrict__ partials, int ldP, int F,
                                                        float *__restrict__ Y, int64_t ldY,
                                                        const float *__restrict__ bias, int relu) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wi = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  if (wi >= w.n_multi_rows) return;
  const int64_t li = w.multi[wi];  // (rows of one block were stored by that block: they are not in the list)
  const int32_t b0 = w.long_cptr[li] >> 2, b1 = w.long_cptr[li + 1] >> 2;
  const int f = lane & 15, k = lane >> 4;  // F <= 16 here; at most kChunk3Cap / 4 = 16 blocks per row
  float x[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) x[u] = partials[(int64_t)min(b0 + k + 4 * u, b1 - 1) * ldP + (f < F ? f : 0)];
  float t = xl_row_sum(x, b0 + k, b1);
  if (k == 0 && f < F) {
    const int64_t row = w.rowmap[w.long_row[li]];
    if (bias) t += bias[f];
    if (relu) t = fmaxf(t, 0.f);
    Y[row * ldY + f] = t;
  } else if (k == 0 && w.pad_ok && f < ldY && f < (F + 3) / 4 * 4) {
    Y[(int64_t)w.rowmap[w.long_row[li]] * ldY + f] = 0.f;
  }
}

// ---- tiny rows (<= kTiny entries) ------------------------------------------------------------
// The transposed view has millions of rows of 1-2 entries (AM: 8.2 M rows, 87 % single entry):
// a wave that owns only 64/G such rows is pure latency (pointer -> index -> gather, three
// dependent round trips for 16 entries).  Here every G-lane group owns kRpg rows at once and
// walks their dependency chains side by side: 4x the rows, the same three round trips.
// Rows are 4-byte aligned only (ld = F): 16-byte loads from dword-aligned addresses, scalar tail.
constexpr int kTiny = 4;
constexpr int kRpg = 4;

__device__ __forceinline__ void load4_tail_safe(const float *row, int f0, int F, float (&x)[4]) {
  if (f0 + 4 <= F) {
    const float4 t = *reinterpret_cast<const float4 *>(row + f0);
    x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = (f0 + i < F) ? row[f0 + i] : 0.f;
  }
}

template <int G>
__global__ __launch_bounds__(256) void k_spmm_tiny(SparseView v, const float *__restrict__ D, int64_t ldD,
                                                   int F, float *__restrict__ Y, int64_t ldY,
                                                   const float *__restrict__ bias, int relu,
                                                   const int32_t *__restrict__ out_index) {
  constexpr int SLOTS = kWave / G;
  const int lane = threadIdx.x & (kWave - 1);
  const int slot = lane / G, q = lane % G;
  const int f0 = q * 4;
  const bool active = f0 < F;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int64_t row0 = wave * (SLOTS * kRpg) + slot;  // rows row0 + k*SLOTS, k < kRpg
  int32_t b[kRpg], n[kRpg];
#pragma unroll
  for (int k = 0; k < kRpg; ++k) {
    const int64_t row = row0 + (int64_t)k * SLOTS;
    b[k] = 0; n[k] = -1;  // -1: not this kernel's row
    if (row < v.rows) {
      b[k] = v.ptr[row];
      n[k] = v.ptr[row + 1] - b[k];
      if (n[k] > kTiny) n[k] = -1;  // the general kernel owns this row
    }
  }
  int32_t ci[kRpg][kTiny];
  float ca[kRpg][kTiny];
#pragma unroll
  for (int k = 0; k < kRpg; ++k)
#pragma unroll
    for (int e = 0; e < kTiny; ++e) {
      const bool on = e < n[k];
      ci[k][e] = on ? v.idx[b[k] + e] : 0;
      ca[k][e] = on ? v.val[b[k] + e] : 0.f;
    }
  float x[kRpg][kTiny][4];
#pragma unroll
  for (int k = 0; k < kRpg; ++k)
#pragma unroll
    for (int e = 0; e < kTiny; ++e) {
      if (active && e < n[k]) load4_tail_safe(D + (int64_t)ci[k][e] * ldD, f0, F, x[k][e]);
      else { x[k][e][0] = x[k][e][1] = x[k][e][2] = x[k][e][3] = 0.f; }
    }
#pragma unroll
  for (int k = 0; k < kRpg; ++k) {
    if (n[k] < 0 || !active) continue;  // empty rows (n = 0) are written (bias / zeros)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < kTiny; ++e)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fmaf(ca[k][e], x[k][e][i], acc[i]);
    const int64_t row = row0 + (int64_t)k * SLOTS;
    const int64_t orow = out_index ? (int64_t)out_index[row] : row;
    float *y = Y + orow * ldY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o = acc[i];
      if (bias && f0 + i < F) o += bias[f0 + i];
      if (relu) o = fmaxf(o, 0.f);
      acc[i] = o;
    }
    if ((ldY & 3) == 0 && f0 + 4 <= F && (((uintptr_t)Y) & 15) == 0) {
      *reinterpret_cast<float4 *>(y + f0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (f0 + i < F) y[f0 + i] = acc[i];
    }
  }
}

// ---- transposed product with mostly-zero operand rows ----------------------------------------
// dM = A'^T dY in the backward of a semi-supervised epoch: dY is the gradient at a layer's output,
// and only rows within reach of a labelled node hold anything (AM shape: 5565 of 1.67 M rows in
// layer 0, the 1000 labelled rows in layer 1).  One thread per output row (= compact column;
// 87 % have a single entry): it looks its entries' rows up in `row_live` (a byte per operand
// row: 1.7 MB, cache resident) and gathers only the live ones — for nine columns in ten that is
// no gather at all, just the zero row and the flag.  Entries are added in the order k_spmm adds
// them, the skipped ones would have added a * 0: bitwise the same result.  Rows longer than
// kLongThreshold stay with the split-row path of k_spmm (flagged live without looking).
// Pass 1 — which operand rows are live, and which compact columns do they touch.  One thread per
// row of D: the flag; then the wave walks the (few) live rows among its 64 and sets the byte of
// every compact column in them (`ccol`: the compact column of each entry, row-major order).
// Plain byte stores of the same value: no atomics.  Rows longer than kLongThreshold are left to
// k_long_rows_mark.  `col_live` was zeroed before.
// `flags_in` (nullable): the row flags are known already (the producer of D wrote them) — D is not scanned
__global__ __launch_bounds__(256) void k_rows_live_mark(const float *__restrict__ D, int64_t ldD, int F,
                                                        int64_t nrows, const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ ccol,
                                                        uint8_t *__restrict__ row_live,
                                                        uint8_t *__restrict__ col_live,
                                                        int32_t *__restrict__ n_live,
                                                        const uint8_t *__restrict__ flags_in) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool nz = false;
  if (i < nrows) {
    if (flags_in) {
      nz = flags_in[i] != 0;
    } else {
      const float *row = D + i * ldD;
      for (int q = 0; q < F; ++q) nz |= row[q] != 0.f;
    }
    row_live[i] = nz ? 1 : 0;
  }
  // (the count of live rows is taken by k_count_flags afterwards: one atomic per live wave on ONE address cost
  // ~10 ns each — 56 us for the 5 565 live rows of the AM epoch's layer 0)
  int32_t b = 0, n = 0;
  if (nz) {
    b = rowptr[i];
    n = rowptr[i + 1] - b;
  }
  uint64_t todo = __ballot(nz && n <= kLongThreshold);
  while (todo) {
    const int L = __ffsll((unsigned long long)todo) - 1;
    todo &= todo - 1;
    const int32_t bb = __shfl(b, L, kWave), nn = __shfl(n, L, kWave);
    if (lane < nn) col_live[ccol[bb + lane]] = 1;
  }
}

// number of set flag bytes: a few blocks, one atomic each
__global__ __launch_bounds__(256) void k_count_flags(const uint8_t *__restrict__ flags, int64_t n,
                                                     int32_t *__restrict__ out) {
  int32_t c = 0;
  const int64_t nw = n >> 3;  // eight flags per load (the array is 16-byte aligned scratch)
  const uint64_t *w = reinterpret_cast<const uint64_t *>(flags);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t v = w[i];
    v |= v >> 4; v |= v >> 2; v |= v >> 1;             // any bit of a byte -> its lowest bit
    c += __popcll(v & 0x0101010101010101ull);
  }
  if (blockIdx.x == 0 && (int64_t)threadIdx.x < n - (nw << 3)) c += flags[(nw << 3) + threadIdx.x] != 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, kWave);
  __shared__ int32_t s_c[4];
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int32_t t = s_c[0] + s_c[1] + s_c[2] + s_c[3];
    if (t) atomicAdd(out, t);
  }
}

// the same for the split rows: one wave per chunk (<= kChunk entries) of a live long row
__global__ __launch_bounds__(256) void k_long_rows_mark(SparseView rv, const int32_t *__restrict__ ccol,
                                                        const uint8_t *__restrict__ row_live,
                                                        uint8_t *__restrict__ col_live) {
  const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int lane = threadIdx.x & 63;
  if (c >= rv.n_chunks) return;
  const int32_t cr = rv.chunk_row[c];
  if (!row_live[cr >= 0 ? cr : -cr - 2]) return;
  for (int32_t e = rv.chunk_beg[c] + lane; e < rv.chunk_end[c]; e += kWave) col_live[ccol[e]] = 1;
}

// Pass 2 — the live output rows (= compact columns).  A block looks at kLiveTB consecutive columns,
// packs the ids of the live ones into LDS (ballot + popcount, order kept) and lets its first threads
// take one each: the waves that walk the pointer -> index -> flag -> gather chain have dense lanes
// (with one thread per column, live or not, every wave paid for the chain with ~7 of 64 lanes busy).
// Entries are added in the order k_spmm adds them (the skipped ones would have added a * 0).  Dead
// rows are not touched here (zeroed beforehand, or left to the consumers' flags).
constexpr int kLiveTB = 256;  // AM epoch, layer 0 / layer 1: 1024 -> 258 / 37 us, 512 -> 229 / 31, 256 -> 206 / 27
template <int FT>
__global__ __launch_bounds__(kLiveTB) void k_spmm_t_live(SparseView v, const float *__restrict__ D,
                                                         int64_t ldD, int F, float *__restrict__ Y,
                                                         int64_t ldY, const uint8_t *__restrict__ row_live,
                                                         const uint8_t *__restrict__ col_live,
                                                         int store_vec_ok, const int32_t *__restrict__ unode,
                                                         uint8_t *__restrict__ node_live) {
  __shared__ int32_t s_cid[kLiveTB];
  __shared__ int32_t s_cnt[kLiveTB / 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * kLiveTB + threadIdx.x;
  const bool live = c < v.rows && col_live[c] != 0;
  const uint64_t bal = __ballot(live);
  if (lane == 0) s_cnt[wv] = __popcll(bal);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kLiveTB / 64; ++w) {
    const int n = s_cnt[w];
    if (w < wv) off += n;
    tot += n;
  }
  if (live) s_cid[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)c;
  __syncthreads();
  if ((int)threadIdx.x >= tot) return;
  const int64_t row = s_cid[threadIdx.x];
  if (node_live) node_live[unode[row]] = 1;  // the source node of a live column (same value from every writer)
  const int32_t b = v.ptr[row];
  const int32_t n = v.ptr[row + 1] - b;
  if (n > kLongThreshold) return;  // the split-row blocks of k_spmm write it
  float acc[FT];
#pragma unroll
  for (int o = 0; o < FT; ++o) acc[o] = 0.f;
  // a wave runs as long as its longest row: look eight entries up per round trip
  constexpr int kLook = 8;
  for (int32_t e0 = b; e0 < b + n; e0 += kLook) {
    int32_t ii[kLook];
    bool lv[kLook];
#pragma unroll
    for (int k = 0; k < kLook; ++k) ii[k] = (e0 + k < b + n) ? v.idx[e0 + k] : -1;
#pragma unroll
    for (int k = 0; k < kLook; ++k) lv[k] = (ii[k] >= 0) ? row_live[ii[k]] != 0 : false;
#pragma unroll
    for (int k = 0; k < kLook; ++k) {
      if (!lv[k]) continue;
      const float a = v.val[e0 + k];
      const float *d = D + (int64_t)ii[k] * ldD;
#pragma unroll
      for (int f0 = 0; f0 < FT; f0 += 4) {
        if (f0 < F) {
          float x[4];
          load4_tail_safe(d, f0, F, x);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[f0 + q] = fmaf(a, x[q], acc[f0 + q]);
        }
      }
    }
  }
  float *y = Y + row * ldY;
  if (store_vec_ok) {  // ldY >= roundup(F, 4), rows 16-byte aligned
#pragma unroll
    for (int f0 = 0; f0 < FT; f0 += 4)
      if (f0 < F) *reinterpret_cast<float4 *>(y + f0) = make_float4(acc[f0], acc[f0 + 1], acc[f0 + 2], acc[f0 + 3]);
  } else {
#pragma unroll
    for (int o = 0; o < FT; ++o)
      if (o < F) y[o] = acc[o];
  }
}

// ---- rows of several chunks: one wave per long row adds its partials in a fixed order ----
__global__ __launch_bounds__(256) void k_spmm_finalize(SparseView v, const float *__restrict__ partials,
                                                       int ldP, int F, float *__restrict__ Y,
                                                       int64_t ldY, const float *__restrict__ bias,
                                                       int relu, const int32_t *__restrict__ out_index) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t li = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  if (li >= v.n_long) return;
  const int32_t c0 = v.long_cptr[li], c1 = v.long_cptr[li + 1];
  if (c1 - c0 <= 1) return;  // single-chunk rows were stored by the chunk wave
  int64_t row = v.long_row[li];
  if (out_index) row = out_index[row];
  finish_split_row<false>(c0, c1, row, partials, ldP, F, Y, ldY, bias, relu, lane);
}

template <int G, int VEC, bool TAIL = false, typename DT = float>
int launch(const SparseView &v, const DT *D, int64_t ldD, int F, float *Y, int64_t ldY,
           const float *bias, int relu, const int32_t *out_index, float *partials, bool use_tiny,
           hipStream_t s, const View3 *w3 = nullptr) {
  constexpr int SLOTS = kWave / G;
  constexpr int SV = VEC > 4 ? 4 : VEC;  // widest single store
  const bool store_vec_ok = (ldY % SV == 0) && (((uintptr_t)Y) % (SV * 4) == 0);
  if constexpr (G <= 4 && VEC == 4) {
    const bool v3_on = cfg(CFG_SPMM_V3) != 0;
    if (w3 && v3_on && v.rows > 0) {  // one gather batch per wave (k_spmm3); rows = class-major ranks
      // the caller owns the pad of Y's rows: whole 16-byte pieces are stored (zeros past F) — rows of ld = 12 at
      // F = 10 / 11 then leave as three vector stores and consecutive rows fill their lines (-5 % on the product)
      const int padw = (w3->pad_ok && store_vec_ok) ? 2 : 0;
      const int wpe = (int)cfg(CFG_SPMM_WPE);  // 0: the plain form
      const bool off32 = !TAIL && w3->op_rows > 0 && wpe > 0 &&
                         (uint64_t)w3->op_rows * (uint64_t)ldD * sizeof(DT) < ((uint64_t)1 << 32);
      const int64_t short_waves = ((int64_t)w3->n_short + SLOTS - 1) / SLOTS;
      const int64_t short_blocks = (short_waves + 3) / 4;
      const int chunk_blocks = w3->n_chunks / 4;  // blockwise rows: four chunks of one row per block
      const int single_blocks = (w3->s_n + 3) / 4;
      const int mid_blocks = ((w3->n_mid + 3) / 4 + 3) / 4;
      const int64_t xcd_per = (short_blocks + 7) / 8;
      if (xcd_per * 8 + chunk_blocks + single_blocks + mid_blocks > 0) {
        const dim3 grid((unsigned)(xcd_per * 8 + chunk_blocks + single_blocks + mid_blocks));
        // rows of exactly F >= 4 floats (TAIL): the padded-row instantiation with overlapping last vectors (`pack`)
        constexpr bool PACKED = TAIL;  // (fp32 rows of 4 F bytes, bf16 rows of 2 F bytes: F = 10 -> 40 / 20 bytes)
        const int pack = (PACKED && F >= VEC) ? 1 : 0;
#define SPMM3_GO(T_, O_, W_)                                                                                       \
  k_spmm3<G, VEC, T_, DT, O_, W_><<<grid, dim3(256), 0, s>>>(v, *w3, D, ldD, F, Y, ldY, bias, relu,                \
                                                               (store_vec_ok ? 1 : 0) | padw, partials, 16, chunk_blocks,   \
                                                               single_blocks, mid_blocks, short_blocks, xcd_per, pack)
        bool done = false;
        const bool off32p = (off32 || (pack && w3->op_rows > 0 && wpe > 0 &&
                                       (uint64_t)w3->op_rows * (uint64_t)ldD * sizeof(DT) < ((uint64_t)1 << 32)));
        if constexpr (G == 4 && sizeof(DT) == 4) {
          if (off32p && (!TAIL || pack)) {
            done = true;
            if (wpe == 7) SPMM3_GO(false, true, 7);
            else SPMM3_GO(false, true, 1);
          }
        }
        if (!done) {
          if (pack) SPMM3_GO(false, false, 1);
          else SPMM3_GO(TAIL, false, 1);
        }
#undef SPMM3_GO
        MRGCN_HIP_TRY(hipGetLastError());
      }
      if (w3->n_multi > 0 && !w3->fold) {
        k_spmm3_finalize<<<dim3((unsigned)((w3->n_multi_rows + 3) / 4)), dim3(256), 0, s>>>(*w3, partials, 16, F, Y, ldY,
                                                                                      bias, relu);
        MRGCN_HIP_TRY(hipGetLastError());
      }
      return MRGCN_OK;
    }
  }
  const int64_t short_waves = (v.rows + SLOTS - 1) / SLOTS;
  const int64_t short_blocks = (short_waves + 3) / 4;
  const int64_t chunk_blocks = ((int64_t)v.n_chunks + 3) / 4;
  const bool xcd_map = cfg(CFG_SPMM_XCD) != 0;
  const int64_t xcd_per = xcd_map ? (short_blocks + 7) / 8 : 0;
  const int64_t launch_short = xcd_map ? xcd_per * 8 : short_blocks;
  int min_len = 0;
  if constexpr (sizeof(DT) == 4) {
  if (use_tiny && v.rows > 0) {  // G*4 >= F guaranteed by the caller
    constexpr int TG = (G * VEC + 3) / 4 < 1 ? 1 : (G * VEC + 3) / 4;  // lanes per row at 4 floats each
    constexpr int TSLOTS = kWave / TG;
    const int64_t waves = (v.rows + (int64_t)TSLOTS * kRpg - 1) / ((int64_t)TSLOTS * kRpg);
    k_spmm_tiny<TG><<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s>>>(v, D, ldD, F, Y, ldY, bias, relu,
                                                                            out_index);
    MRGCN_HIP_TRY(hipGetLastError());
    min_len = kTiny;
  }
  }
  if (launch_short + chunk_blocks > 0) {
    k_spmm<G, VEC, TAIL, DT><<<dim3((unsigned)(launch_short + chunk_blocks)), dim3(256), 0, s>>>(
        v, D, ldD, F, Y, ldY, bias, relu, out_index, store_vec_ok ? 1 : 0, partials, kWsFeatures,
        (int)chunk_blocks, short_blocks, xcd_per, min_len);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  if (v.n_multi > 0 && !v.ticket) {  // (with arrival counters the product finished its split rows itself)
    const int64_t blocks = ((int64_t)v.n_long + 3) / 4;
    k_spmm_finalize<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(v, partials, kWsFeatures, F, Y, ldY, bias,
                                                                 relu, out_index);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  return MRGCN_OK;
}

// picks lanes-per-row G and vector width VEC for one feature tile of width F <= 256
int dispatch(const SparseView &v, const float *D, int64_t ldD, int64_t avail, int F, float *Y,
             int64_t ldY, const float *bias, int relu, const int32_t *out_index, float *partials,
             bool use_tiny, bool operand_cached, hipStream_t s, const View3 *w3 = nullptr) {
  // widest vector the operand layout allows; loads past F must stay inside the row
  // (`avail` = floats left in a row of D from this tile's first column)
  int vec = 1;
  auto ok = [&](int w) {
    int64_t padded = ((int64_t)F + w - 1) / w * w;
    return ldD % w == 0 && ((uintptr_t)D) % (w * 4) == 0 && avail >= padded;
  };
  // MRGCN_SPMM_UNALIGNED=1: 16-byte loads on rows that are only 4/8-byte aligned (gfx950 global
  // loads need dword alignment only); the caller guarantees 12 readable bytes past the operand
  if (ok(4)) vec = 4; else if (ok(2)) vec = 2;
  // unpadded rows that are only 4/8-byte aligned (e.g. the dY of a 10- or 11-class layer, ld = F):
  // 16-byte loads from dword-aligned addresses with a scalar tail instead of 4- or 8-byte lanes
  const bool no_tail = cfg(CFG_SPMM_TAIL) == 0;
  // (only for operands that stay cache resident: on a table far larger than the Infinity Cache a
  // 16-byte load that straddles two 128-B lines costs two HBM line fetches — measured 463 vs 349 us
  // on the 17.8 GB literal operand, 453 vs 496 / 472 vs 725 us on the 67-73 MB dY)
  if (vec < 4 && F <= 32 && !no_tail && operand_cached) {
    const int l4 = (F + 3) / 4;
    if (l4 <= 1) return launch<1, 4, true>(v, D, ldD, F, Y, ldY, bias, relu, out_index, partials, use_tiny, s, w3);
    if (l4 <= 2) return launch<2, 4, true>(v, D, ldD, F, Y, ldY, bias, relu, out_index, partials, use_tiny, s, w3);
    if (l4 <= 4) return launch<4, 4, true>(v, D, ldD, F, Y, ldY, bias, relu, out_index, partials, use_tiny, s, w3);
    return launch<8, 4, true>(v, D, ldD, F, Y, ldY, bias, relu, out_index, partials, use_tiny, s);
  }
  const int lanes = (F + vec - 1) / vec;  // lanes needed per row
#define MRGCN_GO(G, V) return launch<G, V>(v, D, ldD, F, Y, ldY, bias, relu, out_index, partials, use_tiny, s, w3)
  if (vec == 4) {
    if (lanes <= 1) MRGCN_GO(1, 4);
    if (lanes <= 2) MRGCN_GO(2, 4);
    if (lanes <= 4) MRGCN_GO(4, 4);
    if (lanes <= 8) MRGCN_GO(8, 4);
    if (lanes <= 16) MRGCN_GO(16, 4);
    if (lanes <= 32) MRGCN_GO(32, 4);
    MRGCN_GO(64, 4);
  } else if (vec == 2) {
    if (lanes <= 1) MRGCN_GO(1, 2);
    if (lanes <= 2) MRGCN_GO(2, 2);
    if (lanes <= 4) MRGCN_GO(4, 2);
    if (lanes <= 8) MRGCN_GO(8, 2);
    if (lanes <= 16) MRGCN_GO(16, 2);
    if (lanes <= 32) MRGCN_GO(32, 2);
    if (lanes <= 64) MRGCN_GO(64, 2);
  } else {
    if (lanes <= 1) MRGCN_GO(1, 1);
    if (lanes <= 2) MRGCN_GO(2, 1);
    if (lanes <= 4) MRGCN_GO(4, 1);
    if (lanes <= 8) MRGCN_GO(8, 1);
    if (lanes <= 16) MRGCN_GO(16, 1);
    if (lanes <= 32) MRGCN_GO(32, 1);
    if (lanes <= 64) MRGCN_GO(64, 1);
  }
#undef MRGCN_GO
  set_error("internal: no SpMM instantiation for this feature tile");
  return MRGCN_ERR_UNSUPPORTED;
}

// bf16 operand: 2-byte elements, up to 8 per lane (16-byte loads)
int dispatch_bf16(const SparseView &v, const uint16_t *D, int64_t ldD, int64_t avail, int F, float *Y,
                  int64_t ldY, const float *bias, int relu, const int32_t *out_index, float *partials,
                  hipStream_t s, const View3 *w3 = nullptr) {
  auto ok = [&](int w) {
    int64_t padded = ((int64_t)F + w - 1) / w * w;
    return ldD % w == 0 && ((uintptr_t)D) % (w * 2) == 0 && avail >= padded;
  };
  // packed bf16 rows of a narrow layer (ld = F, F not a multiple of four: 20-byte rows at F = 10): k_spmm3 with 8-byte
  // gathers whose last vector overlaps its neighbour's (`pack`) — rows are only 4-byte aligned, which is all a global
  // load needs.  COMPACT view only (w3): the general kernel keeps its 4-byte lanes for such rows.
  if (w3 && ldD == F && F >= 4 && F <= 16 && (F & 3) && (F & 1) == 0 && ((uintptr_t)D) % 4 == 0 && avail >= F) {
    const int l4 = (F + 3) / 4;
    if (l4 <= 2) return launch<2, 4, true, uint16_t>(v, D, ldD, F, Y, ldY, bias, relu, out_index, partials, false, s, w3);
    return launch<4, 4, true, uint16_t>(v, D, ldD, F, Y, ldY, bias, relu, out_index, partials, false, s, w3);
  }
  const int vec = ok(8) ? 8 : ok(4) ? 4 : ok(2) ? 2 : 1;
  const int lanes = (F + vec - 1) / vec;
#define MRGCN_GO(G, V) \
  return launch<G, V, false, uint16_t>(v, D, ldD, F, Y, ldY, bias, relu, out_index, partials, false, s, w3)
#define MRGCN_LANES(V)            \
  if (lanes <= 1) MRGCN_GO(1, V); \
  if (lanes <= 2) MRGCN_GO(2, V); \
  if (lanes <= 4) MRGCN_GO(4, V); \
  if (lanes <= 8) MRGCN_GO(8, V); \
  if (lanes <= 16) MRGCN_GO(16, V); \
  if (lanes <= 32) MRGCN_GO(32, V); \
  if (lanes <= 64) MRGCN_GO(64, V);
  if (vec == 8) { MRGCN_LANES(8) }
  else if (vec == 4) { MRGCN_LANES(4) }
  else if (vec == 2) { MRGCN_LANES(2) }
  else { MRGCN_LANES(1) }
#undef MRGCN_LANES
#undef MRGCN_GO
  set_error("internal: no bf16 SpMM instantiation for this feature tile");
  return MRGCN_ERR_UNSUPPORTED;
}

// MRGCN_SPMM_FOLD=0: the separate finalize launch for every call (A/B)
bool spmm3_fold_default() {
  const bool on = cfg(CFG_SPMM_FOLD) != 0;
  return on;
}

View3 view3_of(const mrgcn_plan *p) {
  View3 w;
  w.n_short = p->n_short3; w.n_mid = p->n_mid3; w.rowmap = p->rowmap;
  w.n_chunks = p->r3_n_chunks; w.n_long = p->r3_n_long; w.n_multi = p->r3_n_multi;
  w.chunk_beg = p->r3_chunk_beg; w.chunk_end = p->r3_chunk_end; w.chunk_row = p->r3_chunk_row;
  w.long_row = p->r3_long_row; w.long_cptr = p->r3_long_cptr;
  w.op_rows = p->n_op;
  w.multi = p->r3_multi; w.n_multi_rows = p->r3_n_multi;
  w.ticket = p->r3_ticket;
  w.s_n = p->r3s_n_chunks; w.s_beg = p->r3s_chunk_beg; w.s_end = p->r3s_chunk_end; w.s_row = p->r3s_chunk_row;
  return w;
}

// The LITERAL product of a narrow layer (the reference's own operand layout, `(R N) x out`: graph.py:75 /:95) on the
// COMPACT view's machinery: the same class-major rows, split-row descriptors and k_spmm3 — only the entry -> operand
// row array differs (literal columns instead of compact operand rows).  AM shape, F = 10 / 11: 335 / 415 us with the
// general kernel -> 307 / 315 us; every touched operand row is still its own random 64-byte access to a 17.8 GB table
// (F = 16, rows that ARE one aligned 64-byte piece: 265 us), which is what bounds this view.
// Returns true and fills v / w3 when the route applies; the index array is built by the first call (not inside a
// capture: that call takes the general kernel).
bool literal_on_compact(const mrgcn_plan *p, int F, const int32_t *out_index, hipStream_t s, SparseView *v, View3 *w3) {
  if (!cfg(CFG_SPMM_LITERAL_V3) || F > 16 || p->lean || out_index || !p->op_node || p->n_rep > 0 || p->nnz == 0)
    return false;
  if (!plan_literal_cols(p, s)) return false;
  *v = p->view(MRGCN_VIEW_COMPACT);
  v->idx = p->mlcol;
  *w3 = view3_of(p);
  w3->op_rows = p->num_relations * p->num_nodes;
  return true;
}

}  // namespace

// Y = v . D for any CSR-shaped view (the filtered transposed view of a gradient support): mrgcn_spmm_f32's tiling
// without bias / ReLU / redirection.  The operand rows such a view reads are few (the live rows of a layer's output
// gradient) and stay cache resident.
int spmm_on_view(const SparseView &v, const float *D, int64_t ldD, int F, float *Y, int64_t ldY, float *partials,
                 hipStream_t s, const float *bias, int relu) {
  if (v.rows == 0) return MRGCN_OK;
  int tile = 64;
  if (ldD % 4 == 0 && ((uintptr_t)D) % 16 == 0) tile = 256;
  else if (ldD % 2 == 0 && ((uintptr_t)D) % 8 == 0) tile = 128;
  for (int f = 0; f < F; f += tile) {
    const int w = (F - f < tile) ? (F - f) : tile;
    int rc = dispatch(v, D + f, ldD, ldD - f, w, Y + f, ldY, bias ? bias + f : nullptr, relu, nullptr, partials, false,
                      true, s, nullptr);
    if (rc != MRGCN_OK) return rc;
  }
  return MRGCN_OK;
}
}  // namespace mrgcn

namespace mrgcn {
namespace {
// replicas of the compact operand: row rep_dst[k] = row rep_src[k], rows of `w4` 4-byte words; a group of
// lanes per replica so that a row moves with one load and one store instruction
__global__ void k_replicate(const int32_t *__restrict__ src, const int32_t *__restrict__ dst, int64_t n_rep,
                            uint32_t *__restrict__ M, int w4, int lanes) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t k = t / lanes;
  const int q = (int)(t - k * lanes);
  if (k >= n_rep) return;
  const uint32_t *a = M + (int64_t)src[k] * w4;
  uint32_t *b = M + (int64_t)dst[k] * w4;
  if ((w4 & 3) == 0 && (((uintptr_t)M) & 15) == 0) {
    for (int i = q; i * 4 < w4; i += lanes)
      reinterpret_cast<uint4 *>(b)[i] = reinterpret_cast<const uint4 *>(a)[i];
  } else {
    for (int i = q; i < w4; i += lanes) b[i] = a[i];
  }
}
}  // namespace
}  // namespace mrgcn

extern "C" int mrgcn_operand_replicate(const mrgcn_plan_t *plan, void *M, int64_t row_bytes, void *stream) {
  using namespace mrgcn;
  MRGCN_REQUIRE(plan && M, "NULL");
  MRGCN_REQUIRE(row_bytes > 0 && row_bytes % 4 == 0, "row_bytes must be a multiple of 4");
  if (plan->n_rep == 0) return MRGCN_OK;
  const int w4 = (int)(row_bytes / 4);
  int lanes = ((w4 & 3) == 0) ? w4 / 4 : w4;  // one 16-byte (or 4-byte) piece per lane
  int lp = 1;
  while (lp < lanes && lp < 16) lp <<= 1;
  const int64_t threads = plan->n_rep * lp;
  k_replicate<<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      plan->rep_src, plan->rep_dst, plan->n_rep, (uint32_t *)M, w4, lp);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

// flags[i] = 1 when X[i, 0:F] holds anything but zeros (optim.hip)
extern "C" int mrgcn_rows_nonzero_f32(const float *X, int64_t ld, int32_t F, int64_t nrows, uint8_t *flags,
                                      void *stream);

extern "C" int64_t mrgcn_spmm_transposed_live_scratch(const mrgcn_plan_t *plan) {
  return plan ? (plan->num_rows + 15) / 16 * 16 : 0;
}

extern "C" int mrgcn_spmm_transposed_live_f32(const mrgcn_plan_t *plan, const float *D, int64_t ldD,
                                              int32_t F, float *Y, int64_t ldY, uint8_t *scratch,
                                              uint8_t *col_live, int32_t *live_rows,
                                              int32_t write_dead_rows, void *stream) {
  return mrgcn_spmm_transposed_live_flagged_f32(plan, D, ldD, F, Y, ldY, scratch, col_live, live_rows,
                                                write_dead_rows, nullptr, nullptr, stream);
}

extern "C" int mrgcn_spmm_transposed_live_flagged_f32(const mrgcn_plan_t *plan, const float *D, int64_t ldD,
                                                      int32_t F, float *Y, int64_t ldY, uint8_t *scratch,
                                                      uint8_t *col_live, int32_t *live_rows,
                                                      int32_t write_dead_rows, const uint8_t *row_flags,
                                                      uint8_t *node_live, void *stream) {
  using namespace mrgcn;
  MRGCN_REQUIRE(plan, "plan is NULL");
  MRGCN_REQUIRE(F > 0 && ldD >= F && ldY >= F, "F / leading dimensions");
  MRGCN_REQUIRE(D && Y && scratch && col_live, "NULL operand");
  MRGCN_REQUIRE(((uintptr_t)scratch & 15) == 0, "scratch must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  SparseView v = plan->view(MRGCN_VIEW_TRANSPOSED);
  float *partials;
  int32_t *ticket;
  {
    int rc = plan_scratch(plan, s, &partials, &ticket);
    if (rc != MRGCN_OK) return rc;
  }
  if (live_rows) MRGCN_HIP_TRY(mrgcn::fill_async(live_rows, 0, sizeof(int32_t), s));
  if (F > 16) {  // wide layers: the general product, then the flags from its result
    int rc = mrgcn_spmm_f32(plan, MRGCN_VIEW_TRANSPOSED, D, ldD, F, Y, ldY, nullptr, 0, nullptr, stream);
    if (rc != MRGCN_OK) return rc;
    if (live_rows) MRGCN_HIP_TRY(mrgcn::fill_async(live_rows, 0xff, sizeof(int32_t), s));  // -1: not counted
    if (node_live) MRGCN_HIP_TRY(mrgcn::fill_async(node_live, 1, (size_t)plan->num_nodes, s));  // (not looked at: all)
    return mrgcn_rows_nonzero_f32(Y, ldY, F, v.rows, col_live, stream);
  }
  uint8_t *row_live = scratch;
  MRGCN_HIP_TRY(mrgcn::fill_async(col_live, 0, (size_t)v.rows, s));
  if (node_live && F <= 16) MRGCN_HIP_TRY(mrgcn::fill_async(node_live, 0, (size_t)plan->num_nodes, s));
  if (write_dead_rows && v.rows > 0) MRGCN_HIP_TRY(mrgcn::fill_async(Y, 0, (size_t)v.rows * ldY * sizeof(float), s));
  if (plan->num_rows > 0 && v.rows > 0) {
    k_rows_live_mark<<<dim3((unsigned)((plan->num_rows + 255) / 256)), dim3(256), 0, s>>>(
        D, ldD, F, plan->num_rows, plan->rowptr, plan->ccol, row_live, col_live, live_rows,
        F <= 16 ? row_flags : nullptr);
    MRGCN_HIP_TRY(hipGetLastError());
    if (live_rows) {
      k_count_flags<<<dim3(128), dim3(256), 0, s>>>(row_live, plan->num_rows, live_rows);
      MRGCN_HIP_TRY(hipGetLastError());
    }
    SparseView rv = plan->view(MRGCN_VIEW_LITERAL);  // row-major entry coordinates, as `ccol`
    if (rv.n_chunks > 0) {
      const int64_t waves = rv.n_chunks;
      k_long_rows_mark<<<dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s>>>(rv, plan->ccol, row_live, col_live);
      MRGCN_HIP_TRY(hipGetLastError());
    }
    const int vec_ok = (ldY % 4 == 0) && (ldY >= (F + 3) / 4 * 4) && (((uintptr_t)Y) % 16 == 0);
    const dim3 grid((unsigned)((v.rows + kLiveTB - 1) / kLiveTB));
    if (F <= 4) k_spmm_t_live<4><<<grid, dim3(kLiveTB), 0, s>>>(v, D, ldD, F, Y, ldY, row_live, col_live, vec_ok, plan->unode, node_live);
    else if (F <= 8) k_spmm_t_live<8><<<grid, dim3(kLiveTB), 0, s>>>(v, D, ldD, F, Y, ldY, row_live, col_live, vec_ok, plan->unode, node_live);
    else if (F <= 12) k_spmm_t_live<12><<<grid, dim3(kLiveTB), 0, s>>>(v, D, ldD, F, Y, ldY, row_live, col_live, vec_ok, plan->unode, node_live);
    else k_spmm_t_live<16><<<grid, dim3(kLiveTB), 0, s>>>(v, D, ldD, F, Y, ldY, row_live, col_live, vec_ok, plan->unode, node_live);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  // long rows: the split-row blocks of the general kernel only (its short-row blocks are not launched)
  const int64_t chunk_blocks = ((int64_t)v.n_chunks + 3) / 4;
  if (chunk_blocks > 0) {
    const bool store_vec_ok = (ldY % 4 == 0) && (((uintptr_t)Y) % 16 == 0);
    // lanes per row as mrgcn_spmm_f32 picks them for 16-byte loads: the same summation order
#define LIVE_LONG(G_)                                                                               \
  k_spmm<G_, 4, true, float, true><<<dim3((unsigned)chunk_blocks), dim3(256), 0, s>>>(              \
      v, D, ldD, F, Y, ldY, nullptr, 0, nullptr, store_vec_ok ? 1 : 0, partials, kWsFeatures,       \
      (int)chunk_blocks, 0, 0, 0, row_live, col_live)
    if (F <= 4) LIVE_LONG(1);
    else if (F <= 8) LIVE_LONG(2);
    else LIVE_LONG(4);
#undef LIVE_LONG
    MRGCN_HIP_TRY(hipGetLastError());
    if (v.n_multi > 0) {
      k_spmm_finalize<<<dim3((unsigned)(((int64_t)v.n_long + 3) / 4)), dim3(256), 0, s>>>(
          v, partials, kWsFeatures, F, Y, ldY, nullptr, 0, nullptr);
      MRGCN_HIP_TRY(hipGetLastError());
    }
  }
  return MRGCN_OK;
}

// =====================================================================================================================
// ENTRY-SLICED product for views whose rows hold one or two entries (round 6; the general TRANSPOSED product: 1.7 entries
// per compact column at the AM shape).  With a slot of four lanes per ROW the wave's trip count is its longest row's
// and most gather slots idle: 436 us, 14.6 % of the roofline, bound by instruction issue.  Here a slot owns K = 4
// consecutive ENTRIES: every wave runs the same straight loop over super-rounds of 64 entries — stage (coalesced row-of-
// entry / operand-row / value), gather, accumulate with a flush whenever the row id changes — with the next super-
// round's staging in flight under this one's gathers.  Rows that span slots are joined by one segmented scan per
// super-round, rows that span super-rounds by a carry, rows that span WAVES by head / tail records that a second,
// tiny launch sums in wave order: no atomics, bitwise reproducible.  (Measured first as tools/lab/spmm_lab.hip::k_seg.)
// RESULT: 417 us against 436 — the product was not issue bound after all: 13.6 M random gathers of 40-byte rows out of a
// 67 MB dY are 1.7-2.2 GB of 128-byte line fetches from the Infinity Cache, whatever form issues them.  Rows that span
// slots are summed in a different order than by the slot-per-row kernels, so results differ in the last bit from the
// live / support forms of the same product: the form is OPT-IN (`spmm_t_seg`), the default stays the slot-per-row kernel.
// F <= 16 (four lanes x four floats); operand rows of >= 4 floats, read with the row's last piece clamped into the row.
// =====================================================================================================================
namespace mrgcn {
namespace {
using f32x4s = __attribute__((ext_vector_type(4))) float;
struct SegArgs {
  const int32_t *idx;   // [nnz] operand row of every entry
  const float *val;     // [nnz]
  const int32_t *row;   // [nnz] output row of every entry (non-decreasing)
  const float *D;
  int64_t ldD;
  int F;
  float *Y;
  int64_t ldY;
  int32_t *rec_row;     // [2 * nwaves]
  float *rec_val;       // [2 * nwaves][16]
  int64_t nwaves, nnz, xcd_per;
  int32_t last_row;     // what the padding past nnz pretends to belong to (value 0)
};

__device__ __forceinline__ void seg_store(const SegArgs &A, int32_t r, const float (&v)[4], int f0) {
  float *y = A.Y + (int64_t)r * A.ldY + f0;
  if (f0 + 4 <= A.F) {   // a whole piece of the row
    *reinterpret_cast<f32x4s *>(y) = f32x4s{v[0], v[1], v[2], v[3]};
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (f0 + i < A.F) y[i] = v[i];
}

template <int K>
__global__ __launch_bounds__(256) void k_seg(SegArgs A) {
  constexpr int G = 4, SLOTS = kWave / G, KR = K / G, SRE = SLOTS * K;
  constexpr int SRW = kSegWaveEntries / SRE;   // super-rounds per wave
  static_assert(K % G == 0 && kSegWaveEntries % SRE == 0, "a wave owns a whole number of super-rounds");
  const int lane = threadIdx.x & 63, slot = lane / G, q = lane % G;
  const int f0 = q * 4;
  const bool active = f0 < A.F;
  // the row's last piece starts inside the row (k_spmm3 `pack`): lo = min(f0, F - 4), shifted into place after the sum
  const int lo = active ? min(f0, A.F - 4) : 0, shift = active ? f0 - lo : 0;
  int64_t wb = blockIdx.x;
  if (A.xcd_per > 0) wb = (wb & 7) * A.xcd_per + (wb >> 3);
  const int64_t w = wb * 4 + (threadIdx.x >> 6);
  if (w >= A.nwaves) return;
  const int64_t e0 = w * (int64_t)kSegWaveEntries, e1 = e0 + kSegWaveEntries;
  const int32_t wave_prev_row = e0 > 0 ? A.row[e0 - 1] : -1;
  const int32_t wave_next_row = e1 < A.nnz ? A.row[e1] : -1;
  const int32_t wave_first_row = A.row[e0];
  const bool wave_open_left = wave_prev_row == wave_first_row;

  int32_t ci[KR], cr[KR], ci2[KR], cr2[KR];
  float ca[KR], ca2[KR];
  auto stage = [&](int64_t es, int32_t(&xi)[KR], float(&xa)[KR], int32_t(&xr)[KR]) {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int64_t m = es + slot * K + k * G + q;
      const int64_t mc = m < A.nnz ? m : A.nnz - 1;   // (unconditional loads at clamped addresses)
      const int32_t i_ = A.idx[mc], r_ = A.row[mc];
      const float a_ = A.val[mc];
      xi[k] = m < A.nnz ? i_ : 0;
      xa[k] = m < A.nnz ? a_ : 0.f;
      xr[k] = m < A.nnz ? r_ : A.last_row;
    }
  };
  stage(e0, ci, ca, cr);
  int32_t carry_row = -1;    // row of the chain that is open at the end of the previous super-round
  float carry[4] = {0.f, 0.f, 0.f, 0.f};
  int32_t prev_last_row = wave_prev_row;

  auto unshift = [&](const float(&v)[4], float(&o)[4]) {  // sums sit in the loaded layout: move them to f0 ..
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[i] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (shift == j && i + j < 4) o[i] = v[i + j];
    }
  };
  auto rec_write = [&](int which, int32_t r, const float(&v)[4]) {  // lanes of ONE slot call this
    if (active) {
      float o[4];
      unshift(v, o);
      float *p = A.rec_val + ((int64_t)2 * w + which) * 16 + f0;
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = o[i];
    }
    if (q == 0) A.rec_row[2 * w + which] = r;
  };
  // a finished row: to Y, or — when it began before this wave — to the wave's head record
  auto finish = [&](int32_t r, const float(&v)[4]) {
    if (wave_open_left && r == wave_first_row) rec_write(0, r, v);
    else if (active) {
      float o[4];
      unshift(v, o);
      seg_store(A, r, o, f0);
    }
  };

  for (int sr = 0; sr < SRW; ++sr) {
    const int64_t es = e0 + (int64_t)sr * SRE;
    // ---- gathers of this super-round (indices staged one super-round ago) ----------------------
    f32x4s x[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const int32_t c = __shfl(ci[t / G], slot * G + (t % G), kWave);
      x[t] = *reinterpret_cast<const f32x4s *>(A.D + (int64_t)c * A.ldD + lo);
    }
    const bool more = sr + 1 < SRW;  // compile-time per unrolled trip
    if (more) stage(es + SRE, ci2, ca2, cr2);
    // ---- rows before / after the slot's range ---------------------------------------------------
    int32_t prev_row = __shfl(cr[KR - 1], (slot > 0 ? slot - 1 : 0) * G + (G - 1), kWave);
    if (slot == 0) prev_row = prev_last_row;
    const int32_t sr_next_first = more ? __shfl(cr2[0], 0, kWave) : wave_next_row;
    int32_t next_row = __shfl(cr[0], (slot < SLOTS - 1 ? slot + 1 : 0) * G, kWave);
    if (slot == SLOTS - 1) next_row = sr_next_first;
    // ---- K entries of the slot ------------------------------------------------------------------
    int32_t cur_row = __shfl(cr[0], slot * G, kWave);
    const int32_t first_row = cur_row;
    const bool open_left = prev_row == first_row;
    bool is_first = true;
    float cur[4] = {0.f, 0.f, 0.f, 0.f}, first[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const int32_t r = __shfl(cr[t / G], slot * G + (t % G), kWave);
      const float a = __shfl(ca[t / G], slot * G + (t % G), kWave);
      if (r != cur_row) {
        if (is_first && open_left) {
#pragma unroll
          for (int i = 0; i < 4; ++i) first[i] = cur[i];
        } else {
          finish(cur_row, cur);
        }
        is_first = false;
        cur_row = r;
#pragma unroll
        for (int i = 0; i < 4; ++i) cur[i] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) cur[i] = fmaf(a, x[t][i], cur[i]);
    }
    const bool open_right = next_row == cur_row;
    const bool whole = is_first;  // one row fills the slot
    // last segment closed at the slot's end and not part of a chain from the left: done
    if (!open_right && !(whole && open_left)) finish(cur_row, cur);
    // ---- chains across slots: out = pass ? in + H : base -------------------------------------------
    const bool pass = whole && open_left;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = cur[i];  // whole: the slot's sum; else: its tail piece (chain start)
    bool reset = !pass;
#pragma unroll
    for (int d = 1; d < SLOTS; d <<= 1) {
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = __shfl_up(v[i], d * G, kWave);
      const int orst = __shfl_up((int)reset, d * G, kWave);
      if (slot >= d && !reset) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += o[i];
        reset = orst != 0;
      }
    }
    // chains that reach back beyond slot 0 take the carry of the previous super-round
    float cin[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cin[i] = __shfl(carry[i], q, kWave);  // carry lives in slot 0's lanes
    if (!reset) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += cin[i];
    }
    // what flows INTO each slot from the left
    float in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      in[i] = __shfl_up(v[i], G, kWave);
      if (slot == 0) in[i] = cin[i];
    }
    // a chain ends in this slot: first segment closed inside it, or the whole slot and nothing to the right
    if (open_left && (!whole || !open_right)) {
      float tot[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) tot[i] = whole ? v[i] : in[i] + first[i];
      finish(first_row, tot);
    }
    // ---- carry into the next super-round (kept in slot 0's lanes) -------------------------------------
    const int last = (SLOTS - 1) * G;
    const int32_t l_row = __shfl(cur_row, last, kWave);
    const int l_open = __shfl((int)open_right, last, kWave);
#pragma unroll
    for (int i = 0; i < 4; ++i) carry[i] = l_open ? __shfl(v[i], last + q, kWave) : 0.f;
    carry_row = l_open ? l_row : -1;
    prev_last_row = __shfl(cr[KR - 1], 63, kWave);
    if (more) {
#pragma unroll
      for (int k = 0; k < KR; ++k) { ci[k] = ci2[k]; ca[k] = ca2[k]; cr[k] = cr2[k]; }
    }
  }
  // ---- what is still open belongs to a row that continues in the next wave ---------------------------
  if (slot == 0) {
    const bool head_is_tail = carry_row >= 0 && wave_open_left && carry_row == wave_first_row;
    if (head_is_tail) {          // the wave lies inside one row: a single record
      rec_write(0, carry_row, carry);
      if (q == 0) A.rec_row[2 * w + 1] = -1;
    } else {
      if (carry_row >= 0) rec_write(1, carry_row, carry);
      else if (q == 0) A.rec_row[2 * w + 1] = -1;
      // head record: written by finish() if the first row closed in this wave and began before it
      if (!wave_open_left && q == 0) A.rec_row[2 * w] = -1;
    }
  }
}

// rows that span waves: records in wave order; the thread of a row's FIRST record sums them all
__global__ void k_seg_fix(SegArgs A) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = 2 * A.nwaves;
  if (i >= n) return;
  const int32_t r = A.rec_row[i];
  if (r < 0) return;
  for (int64_t k = i - 1; k >= 0; --k) {
    const int32_t pr = A.rec_row[k];
    if (pr == r) return;  // not the first record of this row
    if (pr >= 0) break;
  }
  float s[16];
#pragma unroll
  for (int f = 0; f < 16; ++f) s[f] = 0.f;
  for (int64_t k = i; k < n; ++k) {
    const int32_t kr = A.rec_row[k];
    if (kr < 0) continue;
    if (kr != r) break;
#pragma unroll
    for (int f = 0; f < 16; ++f) s[f] += A.rec_val[k * 16 + f];
  }
  for (int f = 0; f < A.F; ++f) A.Y[(int64_t)r * A.ldY + f] = s[f];
}

// Y[row, 0:F] = sum of val * D[idx] over the entries of `row` (entries sorted by row: CSC order for the TRANSPOSED view);
// `scratch`: seg_scratch_floats(nnz) floats
static int spmm_entry_sliced(const int32_t *idx, const float *val, const int32_t *row, int64_t nnz, int64_t rows,
                             const float *D, int64_t ldD, int F, float *Y, int64_t ldY, float *scratch, hipStream_t s) {
  SegArgs A{};
  A.idx = idx; A.val = val; A.row = row; A.D = D; A.ldD = ldD; A.F = F; A.Y = Y; A.ldY = ldY;
  A.nwaves = seg_waves(nnz);
  A.nnz = nnz;
  A.last_row = (int32_t)(rows - 1);
  A.rec_row = reinterpret_cast<int32_t *>(scratch);
  A.rec_val = scratch + 2 * A.nwaves;
  const int64_t blocks = (A.nwaves + 3) / 4;
  A.xcd_per = (blocks + 7) / 8;
  k_seg<4><<<dim3((unsigned)(A.xcd_per * 8)), dim3(256), 0, s>>>(A);
  MRGCN_HIP_TRY(hipGetLastError());
  k_seg_fix<<<dim3((unsigned)((2 * A.nwaves + 255) / 256)), dim3(256), 0, s>>>(A);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}
}  // namespace
}  // namespace mrgcn

extern "C" int mrgcn_spmm_bf16(const mrgcn_plan_t *plan, int32_t view, const uint16_t *D, int64_t ldD,
                               int32_t F, float *Y, int64_t ldY, const float *bias, int32_t relu,
                               const int32_t *out_index, void *stream) {
  using namespace mrgcn;
  MRGCN_REQUIRE(plan, "plan is NULL");
  MRGCN_REQUIRE(view >= MRGCN_VIEW_LITERAL && view <= MRGCN_VIEW_TRANSPOSED, "view");
  MRGCN_REQUIRE(F > 0 && ldD >= F && ldY >= F, "F / leading dimensions");
  MRGCN_REQUIRE(D && Y, "NULL operand");
  SparseView v = plan->view(view);
  if (view == MRGCN_VIEW_COMPACT) {  // its rows are class-major ranks: results go to row rowmap[rank]
    MRGCN_REQUIRE(out_index == nullptr, "out_index is not available on the COMPACT view");
    out_index = plan->rowmap;
  }
  MRGCN_REQUIRE((relu & ~(MRGCN_SPMM_RELU | MRGCN_SPMM_PAD_WRITABLE | MRGCN_SPMM_TWO_PASS)) == 0, "flags");
  const int pad_ok = (relu & MRGCN_SPMM_PAD_WRITABLE) != 0;
  const int fold = (relu & MRGCN_SPMM_TWO_PASS) == 0 && spmm3_fold_default();
  relu &= MRGCN_SPMM_RELU;
  float *partials;
  int32_t *ticket;
  {
    int rc = plan_scratch(plan, (hipStream_t)stream, &partials, &ticket);
    if (rc != MRGCN_OK) return rc;
  }
  v.ticket = (fold && plan->ticket_ints >= v.n_long) ? ticket : nullptr;
  int tile = 64;
  if (ldD % 8 == 0 && ((uintptr_t)D) % 16 == 0) tile = 256;  // kWsFeatures floats of partials per chunk
  else if (ldD % 4 == 0 && ((uintptr_t)D) % 8 == 0) tile = 256;
  else if (ldD % 2 == 0 && ((uintptr_t)D) % 4 == 0) tile = 128;
  for (int f = 0; f < F; f += tile) {
    const int w = (F - f < tile) ? (F - f) : tile;
    View3 w3 = view3_of(plan);
    w3.pad_ok = pad_ok;
    w3.fold = fold;
    w3.ticket = ticket;
    int rc = dispatch_bf16(v, D + f, ldD, ldD - f, w, Y + f, ldY, bias ? bias + f : nullptr, relu, out_index,
                           partials, (hipStream_t)stream, (view == MRGCN_VIEW_COMPACT && F <= 16 && !plan->lean) ? &w3 : nullptr);
    if (rc != MRGCN_OK) return rc;
  }
  return MRGCN_OK;
}

extern "C" int mrgcn_spmm_f32(const mrgcn_plan_t *plan, int32_t view, const float *D, int64_t ldD,
                              int32_t F, float *Y, int64_t ldY, const float *bias, int32_t relu,
                              const int32_t *out_index, void *stream) {
  using namespace mrgcn;
  MRGCN_REQUIRE(plan, "plan is NULL");
  MRGCN_REQUIRE(view >= MRGCN_VIEW_LITERAL && view <= MRGCN_VIEW_TRANSPOSED, "view");
  MRGCN_REQUIRE(F > 0 && ldD >= F && ldY >= F, "F / leading dimensions");
  MRGCN_REQUIRE(D && Y, "NULL operand");
  SparseView v = plan->view(view);
  hipStream_t s = (hipStream_t)stream;
  View3 lit3;
  const bool lit = view == MRGCN_VIEW_LITERAL && literal_on_compact(plan, F, out_index, s, &v, &lit3);
  if (view == MRGCN_VIEW_COMPACT || lit) {  // its rows are class-major ranks: results go to row rowmap[rank]
    MRGCN_REQUIRE(out_index == nullptr, "out_index is not available on the COMPACT view");
    out_index = plan->rowmap;
  }
  MRGCN_REQUIRE((relu & ~(MRGCN_SPMM_RELU | MRGCN_SPMM_PAD_WRITABLE | MRGCN_SPMM_TWO_PASS)) == 0, "flags");
  const int pad_ok = (relu & MRGCN_SPMM_PAD_WRITABLE) != 0;
  const int fold = (relu & MRGCN_SPMM_TWO_PASS) == 0 && spmm3_fold_default();
  relu &= MRGCN_SPMM_RELU;
  float *partials;
  int32_t *ticket;
  {
    int rc = plan_scratch(plan, s, &partials, &ticket);
    if (rc != MRGCN_OK) return rc;
  }
  v.ticket = (fold && plan->ticket_ints >= v.n_long) ? ticket : nullptr;  // split rows finished inside the product
  // the general TRANSPOSED product of a narrow layer: entry-sliced (rows of 1-2 entries; see k_seg)
  if (view == MRGCN_VIEW_TRANSPOSED && F >= 4 && F <= 16 && !out_index && !bias && !relu && !plan->lean &&
      plan->nnz > 0 && cfg(CFG_SPMM_T_SEG) != 0 && plan->partials_floats >= seg_scratch_floats(plan->nnz) &&
      plan_entry_cols(plan, s))
    return spmm_entry_sliced(plan->crow, plan->cval, plan->ecol, plan->nnz, plan->ncols, D, ldD, F, Y, ldY, partials, s);
  // feature tiles: one pass covers up to 64 lanes x VEC floats; the split-row workspace
  // holds kWsFeatures floats per chunk
  int tile = 64;  // scalar-load worst case
  if (ldD % 4 == 0 && ((uintptr_t)D) % 16 == 0) tile = 256;
  else if (ldD % 2 == 0 && ((uintptr_t)D) % 8 == 0) tile = 128;
  for (int f = 0; f < F; f += tile) {
    int w = (F - f < tile) ? (F - f) : tile;
    // optional tiny-row pre-pass for views whose rows are mostly 1-2 entries (the transposed view);
    // measured no faster than the general kernel on the AM shape, so opt-in (MRGCN_SPMM_TINY=1)
    const bool tiny_on = cfg(CFG_SPMM_TINY) != 0;
    const bool use_tiny = tiny_on && w <= 64 && v.rows > 0 && (plan->nnz < 3 * v.rows);
    const int64_t operand_rows = view == MRGCN_VIEW_LITERAL ? plan->num_relations * plan->num_nodes
                                 : view == MRGCN_VIEW_COMPACT ? plan->n_op : plan->num_rows;
    // (the compact operand is read front to back by the rows that own its single-use columns: a 16-byte load
    // that straddles two lines there fetches lines its neighbours need anyway)
    const bool operand_cached = operand_rows * ldD * 4 <= (int64_t)200 << 20 || view == MRGCN_VIEW_COMPACT || lit;
    // the COMPACT view of a narrow layer takes k_spmm3
    View3 w3 = lit ? lit3 : view3_of(plan);
    w3.pad_ok = pad_ok;
    w3.fold = fold;
    w3.ticket = ticket;
    int rc = dispatch(v, D + f, ldD, ldD - f, w, Y + f, ldY, bias ? bias + f : nullptr, relu,
                      out_index, partials, use_tiny, operand_cached, s,
                      ((view == MRGCN_VIEW_COMPACT || lit) && F <= 16 && !plan->lean) ? &w3 : nullptr);
    if (rc != MRGCN_OK) return rc;
  }
  return MRGCN_OK;
}
