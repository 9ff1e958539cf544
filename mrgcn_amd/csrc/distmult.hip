// DistMult decoder behind the R-GCN encoder (link prediction):
//   score  mrgcn/tasks/link_prediction.py:645-665   sum_h (E[s,h] * Rel[p,h]) * E[o,h]
//   loss   :57, :550-554                             BCEWithLogitsLoss (mean) and its gradient
//   ranks  :593-643                                  every fact against ALL nodes, tail then head,
//                                                    raw or filtered, tie-aware
// The reference materialises a [facts, nodes, 3] candidate tensor and a [facts, nodes] score
// matrix per batch; here the scores live only in registers: a block scores 256 candidate nodes
// against kFB facts at once from a transposed copy of E (coalesced), compares with the facts'
// true scores and adds (greater, ties) counts with integer atomics (deterministic).
//
// Rank arithmetic is float32, products associated as in the reference, accumulated
// sequentially over h with contraction off, so that ranks are bit-reproducible against
// oracle/lp_oracle.py.
#include <algorithm>

#include <hipcub/hipcub.hpp>

#include "common.hpp"
#include "config.hpp"

namespace mrgcn {
namespace {

constexpr int kTB = 256;
constexpr int kFB = 8;     // facts per block in the rank kernel
constexpr int kHT = 64;    // h tile of the per-fact query vectors in LDS

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
  return x;
}

// ---- training scores: one wave per triple -------------------------------------------------
__global__ void k_distmult_fwd(const float *__restrict__ E, int64_t ldE, const float *__restrict__ Rel,
                               int64_t ldR, int H, const int64_t *__restrict__ tr, int64_t n,
                               float *__restrict__ scores) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (kTB / kWave) + (threadIdx.x >> 6);
  if (t >= n) return;
  const float *s = E + tr[3 * t] * ldE, *p = Rel + tr[3 * t + 1] * ldR, *o = E + tr[3 * t + 2] * ldE;
  float acc = 0.f;
  for (int h = lane; h < H; h += kWave) acc += s[h] * p[h] * o[h];
  acc = wave_sum(acc);
  if (lane == 0) scores[t] = acc;
}

// dE[s] += g p*o ; dRel[p] += g s*o ; dE[o] += g s*p   (scatter with float atomics)
__global__ void k_distmult_bwd(const float *__restrict__ E, int64_t ldE, const float *__restrict__ Rel,
                               int64_t ldR, int H, const int64_t *__restrict__ tr, int64_t n,
                               const float *__restrict__ g, float *__restrict__ dE, int64_t lddE,
                               float *__restrict__ dRel, int64_t lddR) {
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (kTB / kWave) + (threadIdx.x >> 6);
  if (t >= n) return;
  const int64_t si = tr[3 * t], pi = tr[3 * t + 1], oi = tr[3 * t + 2];
  const float *s = E + si * ldE, *p = Rel + pi * ldR, *o = E + oi * ldE;
  const float gt = g[t];
  if (gt == 0.f) return;
  for (int h = lane; h < H; h += kWave) {
    const float sv = s[h], pv = p[h], ov = o[h];
    if (dE) {
      atomicAdd(dE + si * lddE + h, gt * pv * ov);
      atomicAdd(dE + oi * lddE + h, gt * sv * pv);
    }
    if (dRel) atomicAdd(dRel + pi * lddR + h, gt * sv * ov);
  }
}

// The same gradients with far fewer atomics: `order` lists the triples sorted by their WHICH-th
// component (0 = s, 1 = p, 2 = o); a wave walks kRun consecutive sorted triples, keeps the running
// sum for the current target row in registers and only touches memory when the row changes.  At
// the FB15k-237 shape every relation row is the target of ~1 400 triples and an average node row
// of ~40: the scatter kernel above spends its time in colliding L2 atomics (1.75 ms), these three
// passes re-read the (L2-resident) embedding rows instead.
constexpr int kRun = 32;   // sorted triples per wave
constexpr int kRunF = 256;   // features per pass of a wave (4 per lane)

template <int WHICH>
__global__ void k_distmult_bwd_sorted(const float *__restrict__ E, int64_t ldE, const float *__restrict__ Rel,
                                      int64_t ldR, int H, const int64_t *__restrict__ tr,
                                      const int64_t *__restrict__ order, int64_t n,
                                      const float *__restrict__ g, float *__restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * (kTB / kWave) + (threadIdx.x >> 6);
  const int64_t t0 = w * kRun;
  if (t0 >= n) return;
  const int h0 = blockIdx.y * kRunF;
  float acc[kRunF / kWave];
#pragma unroll
  for (int k = 0; k < kRunF / kWave; ++k) acc[k] = 0.f;
  int64_t cur = -1;
  auto flush = [&]() {
    if (cur < 0) return;
#pragma unroll
    for (int k = 0; k < kRunF / kWave; ++k) {
      const int h = h0 + lane + kWave * k;
      if (h < H && acc[k] != 0.f) atomicAdd(out + cur * ldo + h, acc[k]);
      acc[k] = 0.f;
    }
  };
  const int64_t t1 = (t0 + kRun < n) ? t0 + kRun : n;
  for (int64_t t = t0; t < t1; ++t) {
    const int64_t i = order[t];
    const int64_t si = tr[3 * i], pi = tr[3 * i + 1], oi = tr[3 * i + 2];
    const int64_t key = WHICH == 0 ? si : WHICH == 1 ? pi : oi;
    if (key != cur) {  // wave-uniform
      flush();
      cur = key;
    }
    const float gt = g[i];
    if (gt == 0.f) continue;
    const float *a = WHICH == 0 ? Rel + pi * ldR : E + si * ldE;            // the two rows that are multiplied
    const float *b = WHICH == 2 ? Rel + pi * ldR : E + oi * ldE;
#pragma unroll
    for (int k = 0; k < kRunF / kWave; ++k) {
      const int h = h0 + lane + kWave * k;
      if (h < H) acc[k] = fmaf(gt * a[h], b[h], acc[k]);
    }
  }
  flush();
}

// ---- the same two kernels for rows of whole 16-byte pieces (H % 4 == 0, H <= 256, 16-byte aligned rows) ----------
// lane = four features: a row is ONE load instruction, and the index words of a wave's triples are fetched with one
// coalesced load each up front — the kernels above chase order -> triple -> row through three dependent round trips
// per triple with one row in flight (145 us per sorted pass at the FB15k-237 shape, all of it latency).
using f32x4d = __attribute__((ext_vector_type(4))) float;
constexpr int kFwdTPW = 4;  // triples per wave of the forward

__global__ __launch_bounds__(kTB) void k_distmult_fwd4(const float *__restrict__ E, int64_t ldE,
                                                       const float *__restrict__ Rel, int64_t ldR, int H,
                                                       const int64_t *__restrict__ tr, int64_t n,
                                                       float *__restrict__ scores) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * (kTB / kWave) + (threadIdx.x >> 6);
  const int64_t t0 = w * kFwdTPW;
  if (t0 >= n) return;
  const int cnt = (int)((n - t0 < kFwdTPW) ? n - t0 : kFwdTPW);
  // lane l < 3 * cnt holds word l of the wave's triples
  const int32_t mine = (lane < 3 * cnt) ? (int32_t)tr[3 * t0 + lane] : 0;
  const bool on = 4 * lane < H;
  const int f0 = on ? 4 * lane : 0;
  f32x4d sv[kFwdTPW], pv[kFwdTPW], ov[kFwdTPW];
#pragma unroll
  for (int u = 0; u < kFwdTPW; ++u) {
    const int uu = u < cnt ? u : cnt - 1;
    const int64_t si = __builtin_amdgcn_readlane(mine, 3 * uu), pi = __builtin_amdgcn_readlane(mine, 3 * uu + 1),
                  oi = __builtin_amdgcn_readlane(mine, 3 * uu + 2);
    sv[u] = *reinterpret_cast<const f32x4d *>(E + si * ldE + f0);
    pv[u] = *reinterpret_cast<const f32x4d *>(Rel + pi * ldR + f0);
    ov[u] = *reinterpret_cast<const f32x4d *>(E + oi * ldE + f0);
  }
#pragma unroll
  for (int u = 0; u < kFwdTPW; ++u) {
    const f32x4d q = sv[u] * pv[u] * ov[u];
    float acc = on ? (q.x + q.y) + (q.z + q.w) : 0.f;
    acc = wave_sum(acc);
    if (lane == 0 && u < cnt) scores[t0 + u] = acc;
  }
}

// RUNS: runs of kRun sorted triples a wave walks one after the other, keeping its open sum across them.  The relation
// pass (WHICH = 1: a few hundred keys, segments of thousands of triples) takes four: every wave of a long segment adds
// its sum into the same 4 H-byte row, and a quarter of the waves means a quarter of those contended atomics
// (FB15k-237: 176 us with one run per wave — the entity passes, whose segments are short, take 111 / 123 us).
template <int WHICH, int RUNS = 1>
__global__ __launch_bounds__(kTB) void k_distmult_bwd_sorted4(const float *__restrict__ E, int64_t ldE,
                                                              const float *__restrict__ Rel, int64_t ldR, int H,
                                                              const int64_t *__restrict__ tr,
                                                              const int64_t *__restrict__ order, int64_t n,
                                                              const float *__restrict__ g, float *__restrict__ out,
                                                              int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * (kTB / kWave) + (threadIdx.x >> 6);
  if (w * RUNS * kRun >= n) return;
  const bool on = 4 * lane < H;
  const int f0 = on ? 4 * lane : 0;
  f32x4d acc = {0.f, 0.f, 0.f, 0.f};
  int32_t cur = -1;
  auto flush = [&]() {
    if (cur >= 0 && on) {
      float *o = out + (int64_t)cur * ldo + f0;
      if (acc.x != 0.f) atomicAdd(o + 0, acc.x);
      if (acc.y != 0.f) atomicAdd(o + 1, acc.y);
      if (acc.z != 0.f) atomicAdd(o + 2, acc.z);
      if (acc.w != 0.f) atomicAdd(o + 3, acc.w);
    }
    acc = f32x4d{0.f, 0.f, 0.f, 0.f};
  };
#pragma unroll 1
  for (int run = 0; run < RUNS; ++run) {
  const int64_t t0 = (w * RUNS + run) * kRun;
  if (t0 >= n) break;
  const int cnt = (int)((n - t0 < kRun) ? n - t0 : kRun);
  // lane l < cnt holds sorted triple t0 + l: its three ids and its score gradient (two dependent loads for the whole
  // wave instead of three per triple)
  const int64_t i = order[t0 + (lane < cnt ? lane : cnt - 1)];
  const int32_t si = (int32_t)tr[3 * i], pi = (int32_t)tr[3 * i + 1], oi = (int32_t)tr[3 * i + 2];
  const float gi = g[i];
  for (int tb = 0; tb < cnt; tb += 4) {
    f32x4d a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // four triples' rows in flight
      const int tt = (tb + u < cnt) ? tb + u : cnt - 1;
      const int64_t s_ = __builtin_amdgcn_readlane(si, tt), p_ = __builtin_amdgcn_readlane(pi, tt),
                    o_ = __builtin_amdgcn_readlane(oi, tt);
      const float *ap = WHICH == 0 ? Rel + p_ * ldR : E + s_ * ldE;  // the two rows that are multiplied
      const float *bp = WHICH == 2 ? Rel + p_ * ldR : E + o_ * ldE;
      a[u] = *reinterpret_cast<const f32x4d *>(ap + f0);
      b[u] = *reinterpret_cast<const f32x4d *>(bp + f0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (tb + u < cnt) {  // wave uniform
        const int32_t key = __builtin_amdgcn_readlane(WHICH == 0 ? si : WHICH == 1 ? pi : oi, tb + u);
        const float gt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gi), tb + u));
        if (key != cur) {
          flush();
          cur = key;
        }
        acc += (a[u] * gt) * b[u];
      }
    }
  }
  }
  flush();
}

// loss = mean(max(x,0) - x y + log1p(exp(-|x|))),  dx = (sigmoid(x) - y) / n
__global__ void k_bce_logits(const float *__restrict__ x, const float *__restrict__ y, int64_t n,
                             float *__restrict__ loss, float *__restrict__ dx) {
  __shared__ float s_part[kTB / kWave];
  float acc = 0.f;
  const float inv = 1.f / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * kTB + threadIdx.x; i < n; i += (int64_t)gridDim.x * kTB) {
    const float xi = x[i], yi = y[i];
    acc += fmaxf(xi, 0.f) - xi * yi + log1pf(expf(-fabsf(xi)));
    if (dx) dx[i] = (1.f / (1.f + expf(-xi)) - yi) * inv;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < kTB / kWave; ++i) t += s_part[i];
    atomicAdd(loss, t * inv);
  }
}

// ---- ranks -----------------------------------------------------------------------------------
// Et[h, c] = E[c, h]
__global__ void k_transpose(const float *__restrict__ E, int64_t ldE, int64_t N, int H,
                            float *__restrict__ Et) {
  __shared__ float tile[32][33];
  const int64_t c0 = (int64_t)blockIdx.x * 32;
  const int h0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int64_t c = c0 + r;
    const int h = h0 + tx;
    tile[r][tx] = (c < N && h < H) ? E[c * ldE + h] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int h = h0 + r;
    const int64_t c = c0 + tx;
    if (h < H && c < N) Et[(int64_t)h * N + c] = tile[tx][r];
  }
}

// true[f] = sum_h (E[s,h] Rel[p,h]) E[o,h], sequential; 0 for facts the reference never scores
__global__ void k_true_scores(const float *__restrict__ E, int64_t ldE, const float *__restrict__ Rel,
                              int64_t ldR, int H, const int64_t *__restrict__ tr, int64_t nf,
                              int64_t N, float *__restrict__ truth, int32_t *__restrict__ counts) {
#pragma clang fp contract(off)
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  counts[4 * f] = counts[4 * f + 1] = counts[4 * f + 2] = counts[4 * f + 3] = 0;
  float acc = 0.f;
  if (f < N) {
    const float *s = E + tr[3 * f] * ldE, *p = Rel + tr[3 * f + 1] * ldR, *o = E + tr[3 * f + 2] * ldE;
    for (int h = 0; h < H; ++h) {
      const float sp = s[h] * p[h];
      const float spo = sp * o[h];
      acc = acc + spo;
    }
  }
  truth[f] = acc;
}

__device__ __forceinline__ bool in_sorted(const int32_t *__restrict__ a, int64_t lo, int64_t hi, int32_t key) {
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int32_t v = a[mid];
    if (v == key) return true;
    if (v < key) lo = mid + 1; else hi = mid;
  }
  return false;
}

// grid (candidate tiles, fact tiles, 2 directions).  HEAD=false: candidate replaces o,
// score = (E[s] Rel[p]) * E[c];  HEAD=true: candidate replaces s, score = (E[c] Rel[p]) * E[o].
__global__ __launch_bounds__(kTB) void k_rank_counts(
    const float *__restrict__ Et, int64_t N, int H, const float *__restrict__ E, int64_t ldE,
    const float *__restrict__ Rel, int64_t ldR, const int64_t *__restrict__ tr, int64_t nf,
    const float *__restrict__ truth, const int64_t *__restrict__ tail_ptr,
    const int32_t *__restrict__ tail_idx, const int64_t *__restrict__ head_ptr,
    const int32_t *__restrict__ head_idx, int32_t *__restrict__ counts) {
#pragma clang fp contract(off)
  __shared__ float s_a[kFB][kHT];   // tail: E[s,h]*Rel[p,h]   head: Rel[p,h]
  __shared__ float s_b[kFB][kHT];   // head: E[o,h]
  __shared__ int s_cnt[kFB][2];
  const bool head = blockIdx.z == 1;
  const int64_t c = (int64_t)blockIdx.x * kTB + threadIdx.x;
  const int64_t f0 = (int64_t)blockIdx.y * kFB;
  const int nfb = (int)((nf - f0) < kFB ? (nf - f0) : kFB);
  const bool live = c < N;
  float acc[kFB];
#pragma unroll
  for (int i = 0; i < kFB; ++i) acc[i] = 0.f;
  if (threadIdx.x < kFB * 2) s_cnt[threadIdx.x >> 1][threadIdx.x & 1] = 0;

  for (int h0 = 0; h0 < H; h0 += kHT) {
    const int hn = (H - h0) < kHT ? (H - h0) : kHT;
    __syncthreads();
    for (int i = threadIdx.x; i < kFB * kHT; i += kTB) {
      const int fi = i / kHT, h = i % kHT;
      float a = 0.f, b = 0.f;
      if (fi < nfb && h < hn) {
        const int64_t f = f0 + fi;
        const float pv = Rel[tr[3 * f + 1] * ldR + h0 + h];
        if (head) {
          a = pv;
          b = E[tr[3 * f + 2] * ldE + h0 + h];
        } else {
          a = E[tr[3 * f] * ldE + h0 + h] * pv;
        }
      }
      s_a[fi][h] = a;
      s_b[fi][h] = b;
    }
    __syncthreads();
    if (live) {
      for (int h = 0; h < hn; ++h) {
        const float e = Et[(int64_t)(h0 + h) * N + c];
        if (head) {
#pragma unroll
          for (int i = 0; i < kFB; ++i) {
            const float ep = e * s_a[i][h];
            const float epo = ep * s_b[i][h];
            acc[i] = acc[i] + epo;
          }
        } else {
#pragma unroll
          for (int i = 0; i < kFB; ++i) {
            const float spe = s_a[i][h] * e;
            acc[i] = acc[i] + spe;
          }
        }
      }
    }
  }
  const int64_t *fptr = head ? head_ptr : tail_ptr;
  const int32_t *fidx = head ? head_idx : tail_idx;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < kFB; ++i) {
    bool gt = false, eq = false;
    if (i < nfb) {
      const int64_t f = f0 + i;
      const float sc = (f < N) ? acc[i] : 0.f;
      const float t = truth[f];
      bool masked = !live;
      if (!masked && fptr) masked = in_sorted(fidx, fptr[f], fptr[f + 1], (int32_t)c);
      gt = !masked && sc > t;
      eq = !masked && sc == t;
    }
    const int ngt = __popcll(__ballot(gt)), neq = __popcll(__ballot(eq));
    if (lane == 0) {
      if (ngt) atomicAdd(&s_cnt[i][0], ngt);
      if (neq) atomicAdd(&s_cnt[i][1], neq);
    }
  }
  __syncthreads();
  if (threadIdx.x < nfb * 2) {
    const int i = threadIdx.x >> 1, w = threadIdx.x & 1;
    const int v = s_cnt[i][w];
    if (v) atomicAdd(&counts[4 * (f0 + i) + (head ? 2 : 0) + w], v);
  }
}

// rank = greater + round_half_even((ties - 1) / 2) + 1
__global__ void k_rank_final(const int32_t *__restrict__ counts, int64_t nf, int64_t *__restrict__ ranks) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * nf) return;
  const bool head = i >= nf;
  const int64_t f = head ? i - nf : i;
  const int64_t gt = counts[4 * f + (head ? 2 : 0)], eq = counts[4 * f + (head ? 3 : 1)];
  const int64_t m = eq - 1;                       // >= 0: the fact's own answer always ties
  int64_t half = m >> 1;
  if ((m & 1) && (half & 1)) half += 1;           // x.5 rounds to the even neighbour
  ranks[i] = gt + half + 1;
}

}  // namespace
}  // namespace mrgcn

using namespace mrgcn;

namespace {
__global__ void k_triple_keys(const int64_t *__restrict__ triples, int64_t n, int col, int32_t *__restrict__ keys,
                              int32_t *__restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    keys[i] = (int32_t)triples[3 * i + col];
    idx[i] = (int32_t)i;
  }
}
// all three columns at once: element c * n + i has key (c << bits) | triples[i][c], so one sort leaves the three
// orders in the thirds of the index array
__global__ void k_triple_keys_all(const int64_t *__restrict__ triples, int64_t n, int bits, int32_t *__restrict__ keys,
                                  int32_t *__restrict__ idx) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < 3 * n) {
    const int c = (int)(e / n);
    const int64_t i = e - (int64_t)c * n;
    keys[e] = (int32_t)((c << bits) | (int32_t)triples[3 * i + c]);
    idx[e] = (int32_t)i;
  }
}
__global__ void k_widen3_i32(const int32_t *__restrict__ a, int64_t n, int64_t *__restrict__ o0, int64_t *__restrict__ o1,
                             int64_t *__restrict__ o2) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < 3 * n) {
    int64_t *o = e < n ? o0 : e < 2 * n ? o1 : o2;
    o[e - (e < n ? 0 : e < 2 * n ? n : 2 * n)] = a[e];
  }
}
__global__ void k_widen_i32(const int32_t *__restrict__ a, int64_t n, int64_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i];
}
// ---- the three orders of a SMALL triple set by counting sort (the freshly drawn corrupted facts of an epoch: 54 k at
// the FB15k-237 shape).  A radix sort of so few keys is ~25 launch-bound kernels; here: one histogram pass, one scan
// block per column, one fill pass.  Ties keep no particular order (none is needed).
// Histogram of one column (blockIdx.y) of the triples.  The bins of a block's slice are counted in LDS first and
// flushed with one global add per touched bin: a hub node is the subject of thousands of the corrupted facts, and as
// many same-address global atomics in a row were most of the first version's 120 us (FB15k-237 shape).  Columns with
// more bins than fit LDS go straight to global memory.
constexpr int kCountLdsBins = 16384;  // two arrays of them in k_fill3: 128 KB
constexpr int kCountTB = 1024, kCountPer = 4;
__global__ __launch_bounds__(kCountTB) void k_count3(const int64_t *__restrict__ tr, int64_t n, int32_t *__restrict__ cnt,
                                                     int64_t stride, int64_t bins_n, int64_t bins_r) {
  extern __shared__ int32_t s_bins[];
  const int c = blockIdx.y;
  const int64_t bins = c == 1 ? bins_r : bins_n;
  int32_t *g = cnt + c * stride;
  const bool lds = bins <= kCountLdsBins;
  if (lds) {
    for (int64_t t = threadIdx.x; t < bins; t += kCountTB) s_bins[t] = 0;
    __syncthreads();
  }
  const int64_t i0 = (int64_t)blockIdx.x * kCountTB * kCountPer;
  for (int u = 0; u < kCountPer; ++u) {
    const int64_t i = i0 + u * kCountTB + threadIdx.x;
    if (i < n) atomicAdd(lds ? &s_bins[tr[3 * i + c]] : &g[tr[3 * i + c]], 1);
  }
  if (lds) {
    __syncthreads();
    for (int64_t t = threadIdx.x; t < bins; t += kCountTB)
      if (s_bins[t]) atomicAdd(&g[t], s_bins[t]);
  }
}
// exclusive scan of cnt[c * stride .. + bins_c) in place, one 1024-thread block per column
__global__ __launch_bounds__(1024) void k_scan3(int32_t *__restrict__ cnt, int64_t stride, int64_t bins_n, int64_t bins_r) {
  __shared__ int32_t s_sum[1024];
  __shared__ int32_t s_wave[16];
  const int c = blockIdx.x;
  const int64_t bins = c == 1 ? bins_r : bins_n;
  int32_t *a = cnt + c * stride;
  const int64_t per = (bins + 1023) / 1024;
  const int64_t b0 = threadIdx.x * per, b1 = min(b0 + per, bins);
  constexpr int kPer = 16;
  if (per <= kPer && bins > 0) {
    // the slice in registers: its loads in flight together (one at a time, twice over, and twenty barriers of a
    // Hillis-Steele scan over 1 024 sums: 15 us for FB15k-237's 14 541 bins), a wave-level scan of the slice sums
    int32_t v[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) v[u] = a[min(b0 + u, bins - 1)];
    int32_t t = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      if (b0 + u >= b1) v[u] = 0;
      t += v[u];
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int32_t inc = t;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    int32_t base = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w)
      if (w < wv) base += s_wave[w];
    int32_t run = base + inc - t;  // exclusive prefix of this thread's slice
#pragma unroll
    for (int u = 0; u < kPer; ++u)
      if (b0 + u < b1) {
        a[b0 + u] = run;
        run += v[u];
      }
    return;
  }
  int32_t t = 0;
  for (int64_t k = b0; k < b1; ++k) t += a[k];
  s_sum[threadIdx.x] = t;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele over the 1024 partial sums
    const int32_t v = (int)threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0;
    __syncthreads();
    s_sum[threadIdx.x] += v;
    __syncthreads();
  }
  int32_t run = s_sum[threadIdx.x] - t;  // exclusive prefix of this thread's slice
  for (int64_t k = b0; k < b1; ++k) {
    const int32_t v = a[k];
    a[k] = run;
    run += v;
  }
}
// positions: a block counts its slice per bin (LDS), reserves one range per touched bin behind the bin's offset, and
// hands the range out to its own triples
__global__ __launch_bounds__(kCountTB) void k_fill3(const int64_t *__restrict__ tr, int64_t n, int32_t *__restrict__ off,
                                                    int64_t stride, int64_t bins_n, int64_t bins_r,
                                                    int64_t *__restrict__ o0, int64_t *__restrict__ o1,
                                                    int64_t *__restrict__ o2) {
  extern __shared__ int32_t s_bins[];  // counts, then cursors | bases
  const int c = blockIdx.y;
  const int64_t bins = c == 1 ? bins_r : bins_n;
  int32_t *g = off + c * stride;
  int64_t *out = c == 0 ? o0 : c == 1 ? o1 : o2;
  const bool lds = bins <= kCountLdsBins;
  const int64_t i0 = (int64_t)blockIdx.x * kCountTB * kCountPer;
  if (!lds) {
    for (int u = 0; u < kCountPer; ++u) {
      const int64_t i = i0 + u * kCountTB + threadIdx.x;
      if (i < n) out[atomicAdd(&g[tr[3 * i + c]], 1)] = i;
    }
    return;
  }
  int32_t *s_base = s_bins + bins;
  for (int64_t t = threadIdx.x; t < bins; t += kCountTB) s_bins[t] = 0;
  __syncthreads();
  int64_t key[kCountPer];
  for (int u = 0; u < kCountPer; ++u) {
    const int64_t i = i0 + u * kCountTB + threadIdx.x;
    key[u] = i < n ? tr[3 * i + c] : -1;
    if (key[u] >= 0) atomicAdd(&s_bins[key[u]], 1);
  }
  __syncthreads();
  for (int64_t t = threadIdx.x; t < bins; t += kCountTB) {
    const int32_t k = s_bins[t];
    if (k) s_base[t] = atomicAdd(&g[t], k);
    s_bins[t] = 0;
  }
  __syncthreads();
  for (int u = 0; u < kCountPer; ++u) {
    const int64_t i = i0 + u * kCountTB + threadIdx.x;
    if (key[u] >= 0) out[s_base[key[u]] + atomicAdd(&s_bins[key[u]], 1)] = i;
  }
}

inline bool rows_vec4_ok(const float *E, int64_t ldE, const float *Rel, int64_t ldR, int H) {
  const bool on = cfg(CFG_LP_VEC4) != 0;
  return on && H % 4 == 0 && H <= 256 && ldE % 4 == 0 && ldR % 4 == 0 && (((uintptr_t)E | (uintptr_t)Rel) & 15) == 0;
}
inline int key_bits(int64_t bound) {
  int b = 1;
  while (b < 31 && (bound >> b) != 0) ++b;
  return b;
}
inline size_t orders_sort_temp(int64_t n) {
  size_t tb = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                                           (int32_t *)nullptr, (int)n, 0, 31, (hipStream_t)0);
  return (tb + 255) / 256 * 256;
}
}  // namespace

namespace {
// out[i] = pi(i), i < k: pi a keyed bijection of [0, n) — a six-round Feistel network over the next even number of bits,
// walked along its cycle until it lands below n.  A random subset without replacement in ONE launch; torch.randperm(n)[:k]
// sorts n random keys (a merge sort of 19 launches at n = 272 k: 0.13 ms of the 1.9 ms FB15k-237 epoch).
__device__ __forceinline__ uint32_t feistel_f(uint32_t r, uint32_t key) {
  uint32_t h = r * 0x9E3779B1u + key;
  h ^= h >> 15;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__global__ void k_random_subset(int64_t n, int64_t k, const int64_t *__restrict__ seed, int half_bits,
                                int64_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const uint64_t sd = (uint64_t)*seed;
  uint32_t key[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) key[r] = feistel_f((uint32_t)(sd >> (r & 1 ? 32 : 0)) + 0x632BE5ABu * (r + 1), (uint32_t)(sd >> 17) ^ (r * 0x27D4EB2Fu));
  const uint64_t mask = (1ull << half_bits) - 1;
  uint64_t x = (uint64_t)i;
  do {
    uint32_t L = (uint32_t)(x >> half_bits), R = (uint32_t)(x & mask);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const uint32_t t = L ^ (uint32_t)(feistel_f(R, key[r]) & mask);
      L = R;
      R = t;
    }
    x = ((uint64_t)L << half_bits) | R;
  } while (x >= (uint64_t)n);
  out[i] = (int64_t)x;
}

// The whole in-batch corruption of train_model (tasks/link_prediction.py:239-263) in one launch: corrupted fact k
// (k < ncorrupt = n / 5) is a copy of fact pi(k) — pi the keyed bijection above, i.e. ncorrupt DISTINCT facts — whose
// head (k < nhead) or tail is replaced by a node of the batch drawn with replacement (`nodes`: the batch's node set).
__global__ void k_corrupt_triples(const int64_t *__restrict__ facts, int64_t n, const int64_t *__restrict__ nodes,
                                  int64_t n_nodes, const int64_t *__restrict__ seed, int half_bits, int64_t ncorrupt,
                                  int64_t nhead, int64_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncorrupt) return;
  const uint64_t sd = (uint64_t)*seed;
  uint32_t key[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) key[r] = feistel_f((uint32_t)(sd >> (r & 1 ? 32 : 0)) + 0x632BE5ABu * (r + 1), (uint32_t)(sd >> 17) ^ (r * 0x27D4EB2Fu));
  const uint64_t mask = (1ull << half_bits) - 1;
  uint64_t x = (uint64_t)i;
  do {
    uint32_t L = (uint32_t)(x >> half_bits), R = (uint32_t)(x & mask);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const uint32_t t = L ^ (uint32_t)(feistel_f(R, key[r]) & mask);
      L = R;
      R = t;
    }
    x = ((uint64_t)L << half_bits) | R;
  } while (x >= (uint64_t)n);
  int64_t s_ = facts[3 * x], p_ = facts[3 * x + 1], o_ = facts[3 * x + 2];
  // a 64-bit draw per corrupted fact, reduced to [0, n_nodes) by multiply-shift (no modulo bias worth the name)
  const uint64_t h = ((uint64_t)feistel_f((uint32_t)i, key[0] ^ 0x9E3779B9u) << 32) | feistel_f((uint32_t)i ^ 0x5bd1e995u, key[3]);
  const int64_t pick = nodes[(int64_t)(((unsigned __int128)h * (unsigned __int128)(uint64_t)n_nodes) >> 64)];
  if (i < nhead) s_ = pick; else o_ = pick;
  out[3 * i] = s_;
  out[3 * i + 1] = p_;
  out[3 * i + 2] = o_;
}
}  // namespace

extern "C" {

int mrgcn_corrupt_triples_i64(const int64_t *facts, int64_t n, const int64_t *nodes, int64_t n_nodes,
                              const int64_t *seed_dev, int64_t ncorrupt, int64_t nhead, int64_t *out, void *stream) {
  MRGCN_REQUIRE(n >= 0 && ncorrupt >= 0 && ncorrupt <= n && nhead >= 0 && nhead <= ncorrupt && n < ((int64_t)1 << 62),
                "0 <= nhead <= ncorrupt <= n");
  if (ncorrupt == 0) return MRGCN_OK;
  MRGCN_REQUIRE(facts && nodes && seed_dev && out && n_nodes > 0, "NULL / empty node set");
  int bits = 2;
  while (bits < 62 && ((int64_t)1 << bits) < n) ++bits;
  if (bits & 1) ++bits;
  MRGCN_REQUIRE(bits / 2 <= 32, "n too large");
  k_corrupt_triples<<<dim3((unsigned)((ncorrupt + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      facts, n, nodes, n_nodes, seed_dev, bits / 2, ncorrupt, nhead, out);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_random_subset_i64(int64_t n, int64_t k, const int64_t *seed_dev, int64_t *out, void *stream) {
  MRGCN_REQUIRE(n >= 0 && k >= 0 && k <= n && n < ((int64_t)1 << 62), "0 <= k <= n");
  if (k == 0) return MRGCN_OK;
  MRGCN_REQUIRE(seed_dev && out, "NULL");
  int bits = 2;
  while (bits < 62 && ((int64_t)1 << bits) < n) ++bits;
  if (bits & 1) ++bits;
  MRGCN_REQUIRE(bits / 2 <= 32, "n too large");
  k_random_subset<<<dim3((unsigned)((k + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(n, k, seed_dev, bits / 2, out);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}


int mrgcn_distmult_score_f32(const float *E, int64_t ldE, const float *Rel, int64_t ldR, int32_t H,
                             const int64_t *triples, int64_t n, float *scores, void *stream) {
  MRGCN_REQUIRE(E && Rel && triples && scores && H > 0 && n >= 0, "distmult_score: bad argument");
  if (n == 0) return MRGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int per = kTB / kWave;
  if (rows_vec4_ok(E, ldE, Rel, ldR, H)) {
    const int64_t waves = (n + kFwdTPW - 1) / kFwdTPW;
    k_distmult_fwd4<<<(unsigned)((waves + per - 1) / per), kTB, 0, st>>>(E, ldE, Rel, ldR, H, triples, n, scores);
  } else {
    k_distmult_fwd<<<(unsigned)((n + per - 1) / per), kTB, 0, st>>>(E, ldE, Rel, ldR, H, triples, n, scores);
  }
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_distmult_score_bwd_f32(const float *E, int64_t ldE, const float *Rel, int64_t ldR, int32_t H,
                                 const int64_t *triples, int64_t n, const float *dscores, float *dE,
                                 int64_t lddE, float *dRel, int64_t lddR, void *stream) {
  MRGCN_REQUIRE(E && Rel && triples && dscores && H > 0 && n >= 0, "distmult_score_bwd: bad argument");
  if (n == 0 || (!dE && !dRel)) return MRGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int per = kTB / kWave;
  k_distmult_bwd<<<(unsigned)((n + per - 1) / per), kTB, 0, st>>>(E, ldE, Rel, ldR, H, triples, n, dscores,
                                                                   dE, lddE, dRel, lddR);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

// The three permutations mrgcn_distmult_score_bwd_sorted_f32 takes, by radix sorts of 32-bit keys over the bits the
// keys can have (node ids: 14 bits at FB15k-237 -> two passes; torch.argsort of the int64 columns ran a merge sort of
// ~30 launches per column: 0.64 ms of a 2.6 ms epoch).  Stable: ties keep the order of the triples.
int64_t mrgcn_distmult_orders_workspace(int64_t n) {
  if (n <= 0) return 256;
  const int64_t arr = (3 * n * 4 + 255) / 256 * 256;   // (keys and indices of all three columns, in and out)
  return 4 * arr + (int64_t)orders_sort_temp(3 * n < ((int64_t)1 << 31) ? 3 * n : n);
}

int mrgcn_distmult_orders(const int64_t *triples, int64_t n, int64_t num_nodes, int64_t num_relations,
                          int64_t *order_s, int64_t *order_p, int64_t *order_o, void *workspace,
                          int64_t workspace_bytes, void *stream) {
  MRGCN_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && num_nodes > 0 && num_relations > 0, "sizes");
  MRGCN_REQUIRE(num_nodes < ((int64_t)1 << 31) && num_relations < ((int64_t)1 << 31), "ids must fit 31 bits");
  if (n == 0) return MRGCN_OK;
  MRGCN_REQUIRE(triples && workspace && workspace_bytes >= mrgcn_distmult_orders_workspace(n), "workspace");
  hipStream_t s = (hipStream_t)stream;
  const int64_t arr = (3 * n * 4 + 255) / 256 * 256;
  char *w = (char *)workspace;
  int32_t *k_in = (int32_t *)w, *k_out = (int32_t *)(w + arr), *i_in = (int32_t *)(w + 2 * arr),
          *i_out = (int32_t *)(w + 3 * arr);
  void *tmp = w + 4 * arr;
  size_t tb = orders_sort_temp(3 * n < ((int64_t)1 << 31) ? 3 * n : n);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  int64_t *outs[3] = {order_s, order_p, order_o};
  const int64_t bounds[3] = {num_nodes, num_relations, num_nodes};
  const int bits = key_bits(std::max(num_nodes, num_relations) - 1);
  if (order_s && order_p && order_o && bits <= 28 && 3 * n < ((int64_t)1 << 31)) {
    // one sort for the three columns (a third of the launches: the sorts are launch-bound at these sizes)
    const unsigned b3 = (unsigned)((3 * n + 255) / 256);
    k_triple_keys_all<<<dim3(b3), dim3(256), 0, s>>>(triples, n, bits, k_in, i_in);
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k_in, k_out, i_in, i_out, (int)(3 * n), 0, bits + 2, s));
    k_widen3_i32<<<dim3(b3), dim3(256), 0, s>>>(i_out, n, order_s, order_p, order_o);
    MRGCN_HIP_TRY(hipGetLastError());
    return MRGCN_OK;
  }
  for (int c = 0; c < 3; ++c) {
    if (!outs[c]) continue;
    k_triple_keys<<<dim3(blocks), dim3(256), 0, s>>>(triples, n, c, k_in, i_in);
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k_in, k_out, i_in, i_out, (int)n, 0, key_bits(bounds[c] - 1), s));
    k_widen_i32<<<dim3(blocks), dim3(256), 0, s>>>(i_out, n, outs[c]);
  }
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int64_t mrgcn_distmult_orders_counting_workspace(int64_t num_nodes, int64_t num_relations) {
  const int64_t stride = std::max(num_nodes, num_relations) + 1;
  return 3 * stride * (int64_t)sizeof(int32_t);
}

int mrgcn_distmult_orders_counting(const int64_t *triples, int64_t n, int64_t num_nodes, int64_t num_relations,
                                   int64_t *order_s, int64_t *order_p, int64_t *order_o, void *workspace,
                                   int64_t workspace_bytes, void *stream) {
  MRGCN_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) && num_nodes > 0 && num_relations > 0, "sizes");
  MRGCN_REQUIRE(num_nodes <= (1 << 22) && num_relations <= (1 << 22), "counting sort: at most 4 M bins per column");
  if (n == 0) return MRGCN_OK;
  MRGCN_REQUIRE(triples && order_s && order_p && order_o && workspace &&
                    workspace_bytes >= mrgcn_distmult_orders_counting_workspace(num_nodes, num_relations),
                "NULL / workspace");
  hipStream_t s = (hipStream_t)stream;
  const int64_t stride = std::max(num_nodes, num_relations) + 1;
  int32_t *cnt = (int32_t *)workspace;
  MRGCN_HIP_TRY(mrgcn::fill_async(cnt, 0, (size_t)(3 * stride) * sizeof(int32_t), s));
  const unsigned blocks = (unsigned)((n + kCountTB * kCountPer - 1) / (kCountTB * kCountPer));
  // (the columns whose bins fit LDS: the array is sized for the larger of them)
  const int64_t big = std::max(num_nodes <= kCountLdsBins ? num_nodes : 0, num_relations <= kCountLdsBins ? num_relations : 0);
  const size_t lds1 = (size_t)big * sizeof(int32_t);
  MRGCN_HIP_TRY(mrgcn::raise_lds_limit((const void *)k_count3, kCountLdsBins * sizeof(int32_t)));
  MRGCN_HIP_TRY(mrgcn::raise_lds_limit((const void *)k_fill3, 2 * kCountLdsBins * sizeof(int32_t)));
  k_count3<<<dim3(blocks, 3), dim3(kCountTB), lds1, s>>>(triples, n, cnt, stride, num_nodes, num_relations);
  k_scan3<<<dim3(3), dim3(1024), 0, s>>>(cnt, stride, num_nodes, num_relations);
  k_fill3<<<dim3(blocks, 3), dim3(kCountTB), 2 * lds1, s>>>(triples, n, cnt, stride, num_nodes, num_relations, order_s,
                                                          order_p, order_o);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_distmult_score_bwd_sorted_f32(const float *E, int64_t ldE, const float *Rel, int64_t ldR, int32_t H,
                                        const int64_t *triples, int64_t n, const float *dscores,
                                        const int64_t *order_s, const int64_t *order_p, const int64_t *order_o,
                                        float *dE, int64_t lddE, float *dRel, int64_t lddR, void *stream) {
  MRGCN_REQUIRE(E && Rel && triples && dscores && H > 0 && n >= 0, "distmult_score_bwd_sorted: bad argument");
  MRGCN_REQUIRE((!dE || (order_s && order_o)) && (!dRel || order_p), "distmult_score_bwd_sorted: missing order");
  if (n == 0 || (!dE && !dRel)) return MRGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t waves = (n + kRun - 1) / kRun;
  dim3 grid((unsigned)((waves + kTB / kWave - 1) / (kTB / kWave)), (unsigned)((H + kRunF - 1) / kRunF));
  if (rows_vec4_ok(E, ldE, Rel, ldR, H) && (!dE || (lddE % 4 == 0 && ((uintptr_t)dE & 15) == 0)) &&
      (!dRel || (lddR % 4 == 0 && ((uintptr_t)dRel & 15) == 0))) {
    dim3 g4((unsigned)((waves + kTB / kWave - 1) / (kTB / kWave)));
    if (dE) {
      k_distmult_bwd_sorted4<0><<<g4, kTB, 0, st>>>(E, ldE, Rel, ldR, H, triples, order_s, n, dscores, dE, lddE);
      k_distmult_bwd_sorted4<2><<<g4, kTB, 0, st>>>(E, ldE, Rel, ldR, H, triples, order_o, n, dscores, dE, lddE);
    }
    if (dRel) {
      constexpr int kRelRuns = 4;
      dim3 gr((unsigned)(((waves + kRelRuns - 1) / kRelRuns + kTB / kWave - 1) / (kTB / kWave)));
      k_distmult_bwd_sorted4<1, kRelRuns><<<gr, kTB, 0, st>>>(E, ldE, Rel, ldR, H, triples, order_p, n, dscores, dRel, lddR);
    }
    MRGCN_HIP_TRY(hipGetLastError());
    return MRGCN_OK;
  }
  if (dE) {
    k_distmult_bwd_sorted<0><<<grid, kTB, 0, st>>>(E, ldE, Rel, ldR, H, triples, order_s, n, dscores, dE, lddE);
    k_distmult_bwd_sorted<2><<<grid, kTB, 0, st>>>(E, ldE, Rel, ldR, H, triples, order_o, n, dscores, dE, lddE);
  }
  if (dRel)
    k_distmult_bwd_sorted<1><<<grid, kTB, 0, st>>>(E, ldE, Rel, ldR, H, triples, order_p, n, dscores, dRel, lddR);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_bce_logits_f32(const float *x, const float *y, int64_t n, float *loss, float *dx, void *stream) {
  MRGCN_REQUIRE(x && y && loss && n > 0, "bce_logits: bad argument");
  hipStream_t st = (hipStream_t)stream;
  MRGCN_HIP_TRY(mrgcn::fill_async(loss, 0, sizeof(float), st));
  int64_t b = (n + kTB - 1) / kTB;
  if (b > 256) b = 256;  // (one float atomic per block on ONE address, ~12 ns each: 1 024 of them were 12 of the kernel's 16 us)
  k_bce_logits<<<(unsigned)b, kTB, 0, st>>>(x, y, n, loss, dx);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int64_t mrgcn_distmult_ranks_workspace(int64_t num_nodes, int32_t H, int64_t num_facts) {
  if (num_nodes < 0 || H <= 0 || num_facts < 0) return -1;
  // Et [H, N] floats | truth [nf] floats | counts [4 nf] int32
  return (int64_t)sizeof(float) * ((int64_t)H * num_nodes + num_facts) + (int64_t)sizeof(int32_t) * 4 * num_facts;
}

int mrgcn_distmult_ranks(const float *E, int64_t ldE, int64_t num_nodes, const float *Rel, int64_t ldR,
                         int32_t H, const int64_t *triples, int64_t num_facts, const int64_t *tail_ptr,
                         const int32_t *tail_idx, const int64_t *head_ptr, const int32_t *head_idx,
                         void *workspace, int64_t workspace_bytes, int64_t *ranks, void *stream) {
  MRGCN_REQUIRE(E && Rel && triples && ranks && workspace && H > 0 && num_nodes > 0 && num_facts >= 0,
                "distmult_ranks: bad argument");
  MRGCN_REQUIRE((tail_ptr == nullptr) == (head_ptr == nullptr), "distmult_ranks: give both filter lists or none");
  MRGCN_REQUIRE(workspace_bytes >= mrgcn_distmult_ranks_workspace(num_nodes, H, num_facts),
                "distmult_ranks: workspace too small");
  if (num_facts == 0) return MRGCN_OK;
  hipStream_t st = (hipStream_t)stream;
  float *Et = (float *)workspace;
  float *truth = Et + (int64_t)H * num_nodes;
  int32_t *counts = (int32_t *)(truth + num_facts);
  dim3 tg((unsigned)((num_nodes + 31) / 32), (unsigned)((H + 31) / 32));
  k_transpose<<<tg, 256, 0, st>>>(E, ldE, num_nodes, H, Et);
  k_true_scores<<<(unsigned)((num_facts + 127) / 128), 128, 0, st>>>(E, ldE, Rel, ldR, H, triples, num_facts,
                                                                      num_nodes, truth, counts);
  const int64_t ftiles = (num_facts + kFB - 1) / kFB;
  MRGCN_REQUIRE(ftiles <= 65535, "distmult_ranks: more than 524280 facts per call");
  dim3 rg((unsigned)((num_nodes + kTB - 1) / kTB), (unsigned)ftiles, 2);
  k_rank_counts<<<rg, kTB, 0, st>>>(Et, num_nodes, H, E, ldE, Rel, ldR, triples, num_facts, truth, tail_ptr,
                                    tail_idx, head_ptr, head_idx, counts);
  k_rank_final<<<(unsigned)((2 * num_facts + 255) / 256), 256, 0, st>>>(counts, num_facts, ranks);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // extern "C"
