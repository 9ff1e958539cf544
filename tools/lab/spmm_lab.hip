// LAB (not part of libmrgcn_hip.so): stream + gather form of the stacked-CSR product, to measure
// operand layouts before they go into plan.hip / spmm.hip.
//
// A row's entries are split into  (a) gathered entries: explicit operand-row index into a small dense
// region Mg (rows of 16 floats, 64-B aligned: a read never straddles a line) and (b) a stream run:
// `ns` consecutive rows of Ms (F floats each, unpadded) starting at sptr[row], which needs no index.
// Consecutive output rows own consecutive runs, so the whole of Ms is read front to back.
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int kWave = 64;
constexpr int kLong = 32;
constexpr int kChunk = 512;

__device__ __forceinline__ void fma4(float (&acc)[4], float a, const float4 &x, bool on) {
  acc[0] = fmaf(a, on ? x.x : 0.f, acc[0]);
  acc[1] = fmaf(a, on ? x.y : 0.f, acc[1]);
  acc[2] = fmaf(a, on ? x.z : 0.f, acc[2]);
  acc[3] = fmaf(a, on ? x.w : 0.f, acc[3]);
}

template <int F, int SU>
__global__ __launch_bounds__(256) void k_sg(int64_t rows, const int32_t *__restrict__ gptr,
                                            const int32_t *__restrict__ gidx, const float *__restrict__ gval,
                                            const int32_t *__restrict__ sptr, const float *__restrict__ sval,
                                            const float *__restrict__ Mg, const float *__restrict__ Ms,
                                            float *__restrict__ Y, int n_chunks,
                                            const int32_t *__restrict__ c_row, const int32_t *__restrict__ c_gb,
                                            const int32_t *__restrict__ c_ge, const int32_t *__restrict__ c_sb,
                                            const int32_t *__restrict__ c_se, float *__restrict__ partials,
                                            int chunk_blocks, int64_t short_blocks, int64_t xcd_per) {
  constexpr int G = 4, SLOTS = kWave / G, LDG = 16;
  const int lane = threadIdx.x & 63, slot = lane >> 2, q = lane & 3;
  const int f0 = q * 4;
  const bool active = f0 < F;
  const int fq = active ? f0 : 0;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if ((int)blockIdx.x < chunk_blocks) {
    const int64_t chunk = ((int64_t)blockIdx.x * 256 + threadIdx.x) / kWave;
    if (chunk >= n_chunks) return;
    const int32_t gb = c_gb[chunk], n = c_ge[chunk] - gb;
    const int32_t sb = c_sb[chunk], ns = c_se[chunk] - sb;
    constexpr int T = kChunk / kWave;
    int32_t ci[T];
    float ca[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int32_t m = t * kWave + lane;
      ci[t] = (m < n) ? gidx[gb + m] : 0;
      ca[t] = (m < n) ? gval[gb + m] : 0.f;
    }
    // stream part: slot s takes rows s, s + 16, ... (one instruction = 16 consecutive rows)
    for (int32_t k0 = 0; k0 < ns; k0 += SLOTS * SU) {
      float4 x[SU];
      float a[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int32_t k = k0 + u * SLOTS + slot;
        const bool on = k < ns;
        const int32_t kk = on ? k : 0;
        a[u] = on ? sval[sb + kk] : 0.f;
        x[u] = *reinterpret_cast<const float4 *>(Ms + (int64_t)(sb + kk) * F + fq);
      }
#pragma unroll
      for (int u = 0; u < SU; ++u) fma4(acc, a[u], x[u], active && (k0 + u * SLOTS + slot < ns));
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t * kWave >= n) break;
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int src = u * SLOTS + slot;
        const int32_t c = __shfl(ci[t], src, kWave);
        const float a = __shfl(ca[t], src, kWave);
        const float4 x = *reinterpret_cast<const float4 *>(Mg + (int64_t)c * LDG + fq);
        fma4(acc, a, x, active && (t * kWave + src < n));
      }
    }
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
    }
    if (slot == 0 && active) {
      const int32_t row = c_row[chunk];
      float *p = row >= 0 ? Y + (int64_t)row * F + f0 : partials + chunk * 16 + f0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (f0 + i < F) p[i] = acc[i];
    }
    return;
  }
  int64_t sbk = (int64_t)blockIdx.x - chunk_blocks;
  if (xcd_per > 0) {
    sbk = (sbk & 7) * xcd_per + (sbk >> 3);
    if (sbk >= short_blocks) return;
  }
  const int64_t wave = (sbk * 256 + threadIdx.x) / kWave;
  const int64_t row = wave * SLOTS + slot;
  int32_t gb = 0, ng = 0, s0 = 0, ns = 0;
  if (row < rows) {
    gb = gptr[row];
    ng = gptr[row + 1] - gb;
    s0 = sptr[row];
    ns = sptr[row + 1] - s0;
  }
  const bool mine = row < rows && ng + ns <= kLong;
  if (!mine) ng = ns = 0;
  if (!__any(mine)) return;
  constexpr int T = kLong / G;
  int32_t ci[T];
  float ca[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const int32_t m = t * G + q;
    ci[t] = (m < ng) ? gidx[gb + m] : 0;
    ca[t] = (m < ng) ? gval[gb + m] : 0.f;
  }
  // stream run first: its addresses need no index
  for (int32_t t0 = 0; __any(t0 < ns); t0 += SU) {
    float4 x[SU];
    float a[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const bool on = t0 + u < ns;
      const int32_t kk = on ? t0 + u : 0;
      a[u] = on ? sval[s0 + kk] : 0.f;
      x[u] = *reinterpret_cast<const float4 *>(Ms + (int64_t)(s0 + kk) * F + fq);
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) fma4(acc, a[u], x[u], active && (t0 + u < ns));
  }
  const int sbase = slot * G;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (!__any(t * G < ng)) break;
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int32_t c = __shfl(ci[t], sbase + u, kWave);
      const float a = __shfl(ca[t], sbase + u, kWave);
      const float4 x = *reinterpret_cast<const float4 *>(Mg + (int64_t)c * LDG + fq);
      fma4(acc, a, x, active && (t * G + u < ng));
    }
  }
  if (mine && active) {
    float *y = Y + row * F + f0;
#pragma unroll
    for (int i = 0; i < 4; i += 2)
      if (f0 + i + 1 < F) *reinterpret_cast<float2 *>(y + i) = make_float2(acc[i], acc[i + 1]);
      else if (f0 + i < F) y[i] = acc[i];
  }
}

// multi-chunk rows: one wave per long row; lane = (chunk mod 4, feature), chunks cptr[lr]..cptr[lr+1]
__global__ void k_sg_finalize(int n_long, const int32_t *__restrict__ long_row, const int32_t *__restrict__ cptr,
                              const float *__restrict__ partials, int F, float *__restrict__ Y) {
  const int lr = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
  const int lane = threadIdx.x & 63, f = lane & 15, k = lane >> 4;
  if (lr >= n_long) return;
  const int c0 = cptr[lr], c1 = cptr[lr + 1];
  if (c1 - c0 <= 1) return;
  float s = 0.f;
  for (int c = c0 + k; c < c1; c += 4) s += partials[(int64_t)c * 16 + f];
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  if (k == 0 && f < F) Y[(int64_t)long_row[lr] * F + f] = s;
}

extern "C" int lab_sg(int F, int su, int64_t rows, const int32_t *gptr, const int32_t *gidx, const float *gval,
                      const int32_t *sptr, const float *sval, const float *Mg, const float *Ms, float *Y,
                      int n_chunks, const int32_t *c_row, const int32_t *c_gb, const int32_t *c_ge,
                      const int32_t *c_sb, const int32_t *c_se, float *partials, int n_long,
                      const int32_t *long_row, const int32_t *long_cptr, int xcd, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const int64_t short_waves = (rows + 15) / 16;
  const int64_t short_blocks = (short_waves + 3) / 4;
  const int64_t chunk_blocks = ((int64_t)n_chunks + 3) / 4;
  const int64_t xcd_per = xcd ? (short_blocks + 7) / 8 : 0;
  const int64_t launch_short = xcd ? xcd_per * 8 : short_blocks;
  dim3 grid((unsigned)(launch_short + chunk_blocks));
#define GO(F_, SU_)                                                                                          \
  k_sg<F_, SU_><<<grid, dim3(256), 0, s>>>(rows, gptr, gidx, gval, sptr, sval, Mg, Ms, Y, n_chunks, c_row, \
                                           c_gb, c_ge, c_sb, c_se, partials, (int)chunk_blocks,            \
                                           short_blocks, xcd_per)
  if (F == 10 && su == 4) GO(10, 4);
  else if (F == 10 && su == 8) GO(10, 8);
  else if (F == 10 && su == 2) GO(10, 2);
  else if (F == 16 && su == 4) GO(16, 4);
  else return 1;
#undef GO
  if (n_long > 0)
    k_sg_finalize<<<dim3((unsigned)((n_long + 3) / 4)), dim3(256), 0, s>>>(n_long, long_row, long_cptr,
                                                                                    partials, F, Y);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

__global__ void k_copy_i32(int32_t *dst, const int32_t *src, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
extern "C" int lab_copy_i32(int32_t *dst, const int32_t *src, int64_t n, void *stream) {
  k_copy_i32<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(dst, src, n);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// =====================================================================================================
// ENTRY-SLICED product: a slot of G lanes owns K consecutive ENTRIES (not a row).  No row pointer, no
// long/short split, no divergence in trip counts: every wave runs the same loop over super-rounds of
// (64/G)*K entries — stage (coalesced idx / val / row), gather all K rows of the slot, accumulate with a
// flush whenever the row id changes — so the next super-round's staging loads are in flight while this
// one's gathers are consumed.  Rows that span slots are joined by one segmented scan per super-round,
// rows that span waves by per-wave head / tail records that a tiny second kernel sums in wave order
// (bitwise reproducible: no atomics).  Arrays are padded to a whole number of waves with (idx 0, val 0,
// row = last row).
// =====================================================================================================
struct SegArgs {
  const int32_t *idx;
  const float *val;
  const int32_t *row;
  const float *D;
  int64_t ldD;
  int F;
  float *Y;
  int64_t ldY;
  const float *bias;
  int relu;
  int32_t *rec_row;  // [2 * nwaves]
  float *rec_val;    // [2 * nwaves][4 * G]
  int srw;           // super-rounds per wave
  int64_t nwaves, nnz_pad, xcd_per;
};

template <int G>
__device__ __forceinline__ void seg_store(const SegArgs &A, int32_t r, const float (&v)[4], int f0) {
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t = v[i];
    if (A.bias && f0 + i < A.F) t += A.bias[f0 + i];
    if (A.relu) t = fmaxf(t, 0.f);
    o[i] = t;
  }
  float *y = A.Y + (int64_t)r * A.ldY + f0;
  if ((A.ldY & 1) == 0) {
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
      if (f0 + i + 1 < A.F) *reinterpret_cast<float2 *>(y + i) = make_float2(o[i], o[i + 1]);
      else if (f0 + i < A.F) y[i] = o[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (f0 + i < A.F) y[i] = o[i];
  }
}

template <int G, int K>
__global__ __launch_bounds__(256) void k_seg(SegArgs A) {
  constexpr int SLOTS = kWave / G, KR = K / G, SRE = SLOTS * K;
  static_assert(K % G == 0, "K must be a multiple of G");
  const int lane = threadIdx.x & 63, slot = lane / G, q = lane % G;
  const int f0 = q * 4;
  const bool active = f0 < A.F;
  const int fq = active ? f0 : 0;
  int64_t wb = blockIdx.x;
  if (A.xcd_per > 0) wb = (wb & 7) * A.xcd_per + (wb >> 3);
  const int64_t w = wb * 4 + (threadIdx.x >> 6);
  if (w >= A.nwaves) return;
  const int64_t e0 = w * (int64_t)A.srw * SRE, e1 = e0 + (int64_t)A.srw * SRE;
  const int32_t wave_prev_row = e0 > 0 ? A.row[e0 - 1] : -1;
  const int32_t wave_next_row = e1 < A.nnz_pad ? A.row[e1] : -1;
  const int32_t wave_first_row = A.row[e0];
  const bool wave_open_left = wave_prev_row == wave_first_row;
  bool head_written = false;

  int32_t ci[KR], cr[KR], ci2[KR], cr2[KR];
  float ca[KR], ca2[KR];
  auto stage = [&](int64_t es, int32_t(&xi)[KR], float(&xa)[KR], int32_t(&xr)[KR]) {
#pragma unroll
    for (int k = 0; k < KR; ++k) {
      const int64_t m = es + slot * K + k * G + q;
      xi[k] = A.idx[m];
      xa[k] = A.val[m];
      xr[k] = A.row[m];
    }
  };
  stage(e0, ci, ca, cr);
  int32_t carry_row = -1;    // row of the chain that is open at the end of the previous super-round
  float carry[4] = {0.f, 0.f, 0.f, 0.f};
  int32_t prev_last_row = wave_prev_row;

  auto rec_write = [&](int which, int32_t r, const float(&v)[4]) {  // lanes of ONE slot call this
    if (active) {
      float *p = A.rec_val + ((int64_t)2 * w + which) * (4 * G) + f0;
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = v[i];
    }
    if (q == 0) A.rec_row[2 * w + which] = r;
  };
  // a finished row: to Y, or — when it began before this wave — to the wave's head record
  auto finish = [&](int32_t r, const float(&v)[4]) {
    if (wave_open_left && r == wave_first_row) rec_write(0, r, v);
    else if (active) seg_store<G>(A, r, v, f0);
  };

  for (int sr = 0; sr < A.srw; ++sr) {
    const int64_t es = e0 + (int64_t)sr * SRE;
    // ---- gathers of this super-round (indices staged one super-round ago) ----------------------
    float4 x[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const int32_t c = __shfl(ci[t / G], slot * G + (t % G), kWave);
      x[t] = *reinterpret_cast<const float4 *>(A.D + (int64_t)c * A.ldD + fq);
    }
    const bool more = sr + 1 < A.srw;  // wave uniform
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
      cur[0] = fmaf(a, x[t].x, cur[0]);
      cur[1] = fmaf(a, x[t].y, cur[1]);
      cur[2] = fmaf(a, x[t].z, cur[2]);
      cur[3] = fmaf(a, x[t].w, cur[3]);
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
  (void)head_written;
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
template <int G>
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
  float s[4 * G];
#pragma unroll
  for (int f = 0; f < 4 * G; ++f) s[f] = 0.f;
  for (int64_t k = i; k < n; ++k) {
    const int32_t kr = A.rec_row[k];
    if (kr < 0) continue;
    if (kr != r) break;
#pragma unroll
    for (int f = 0; f < 4 * G; ++f) s[f] += A.rec_val[k * (4 * G) + f];
  }
  for (int f = 0; f < A.F; ++f) {
    float t = s[f];
    if (A.bias) t += A.bias[f];
    if (A.relu) t = fmaxf(t, 0.f);
    A.Y[(int64_t)r * A.ldY + f] = t;
  }
}

extern "C" int lab_seg(int K, int srw, int64_t nnz_pad, const int32_t *idx, const float *val, const int32_t *row,
                       const float *D, int64_t ldD, int F, float *Y, int64_t ldY, int32_t *rec_row,
                       float *rec_val, int xcd, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  constexpr int G = 4;
  const int sre = (64 / G) * K;
  if (nnz_pad % ((int64_t)sre * srw) != 0) return 3;
  SegArgs A{idx, val, row, D, ldD, F, Y, ldY, nullptr, 0, rec_row, rec_val, srw, nnz_pad / ((int64_t)sre * srw),
            nnz_pad, 0};
  const int64_t blocks = (A.nwaves + 3) / 4;
  A.xcd_per = xcd ? (blocks + 7) / 8 : 0;
  const int64_t launch = xcd ? A.xcd_per * 8 : blocks;
  if (K == 8) k_seg<G, 8><<<dim3((unsigned)launch), dim3(256), 0, s>>>(A);
  else if (K == 4) k_seg<G, 4><<<dim3((unsigned)launch), dim3(256), 0, s>>>(A);
  else if (K == 16) k_seg<G, 16><<<dim3((unsigned)launch), dim3(256), 0, s>>>(A);
  else return 1;
  k_seg_fix<G><<<dim3((unsigned)((2 * A.nwaves + 255) / 256)), dim3(256), 0, s>>>(A);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// =====================================================================================================
// PERSISTENT, SOFTWARE-PIPELINED row-owner kernels.  Same row ownership as the production k_spmm (a slot
// of G lanes per row, 64/G rows per wave step; rows longer than T*G entries go to the chunk kernel), but a
// wave walks many row groups and keeps three stages in flight: the row pointers of group i+2, the
// index / value staging of group i+1 and the gathers of group i — one exposed round trip per group
// instead of pointer -> indices -> gathers one after the other.
// =====================================================================================================
template <int G, int T, int GB>
__global__ __launch_bounds__(256) void k_rows_p(int64_t rows, const int32_t *__restrict__ ptr,
                                                const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                const float *__restrict__ D, int64_t ldD, int F,
                                                float *__restrict__ Y, int64_t ldY, int64_t ngroups,
                                                int64_t groups_per_xcd, int waves_per_xcd) {
  constexpr int SLOTS = kWave / G, THR = T * G;
  const int lane = threadIdx.x & 63, slot = lane / G, q = lane % G;
  const int f0 = q * 4;
  const bool active = f0 < F;
  const int fq = active ? f0 : 0;
  const int xcd = blockIdx.x & 7;
  const int64_t widx = (int64_t)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
  const int64_t g_end = min(ngroups, (xcd + 1) * groups_per_xcd);
  int64_t g = xcd * groups_per_xcd + widx;
  if (g >= g_end) return;
  const int64_t stride = waves_per_xcd;

  auto load_ptr = [&](int64_t gg, int32_t &b, int32_t &n) {
    const int64_t row = gg * SLOTS + slot;
    b = 0; n = 0;
    if (gg < g_end && row < rows) {
      b = ptr[row];
      n = ptr[row + 1] - b;
      if (n > THR) n = 0;  // the chunk kernel owns this row
    }
  };
  auto stage = [&](int32_t b, int32_t n, int32_t(&xi)[T], float(&xa)[T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int32_t m = t * G + q;
      xi[t] = (m < n) ? idx[b + m] : 0;
      xa[t] = (m < n) ? val[b + m] : 0.f;
    }
  };
  int32_t b0, n0, b1, n1, b2, n2;
  int32_t ci[T], ci2[T];
  float ca[T], ca2[T];
  load_ptr(g, b0, n0);
  stage(b0, n0, ci, ca);
  load_ptr(g + stride, b1, n1);
  const int sbase = slot * G;
  for (;;) {
    stage(b1, n1, ci2, ca2);
    load_ptr(g + 2 * stride, b2, n2);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t0 = 0; t0 < T; t0 += GB) {
      if (!__any(t0 * G < n0)) break;
      float4 x[GB * G];
      float a[GB * G];
#pragma unroll
      for (int tt = 0; tt < GB; ++tt)
#pragma unroll
        for (int u = 0; u < G; ++u) {
          const int32_t c = __shfl(ci[t0 + tt], sbase + u, kWave);
          a[tt * G + u] = __shfl(ca[t0 + tt], sbase + u, kWave);
          x[tt * G + u] = *reinterpret_cast<const float4 *>(D + (int64_t)c * ldD + fq);
        }
#pragma unroll
      for (int k = 0; k < GB * G; ++k) fma4(acc, a[k], x[k], active && ((t0 + k / G) * G + (k % G) < n0));
    }
    {
      const int64_t row = g * SLOTS + slot;
      const int32_t nn = (row < rows) ? ptr[row + 1] - ptr[row] : THR + 1;  // cheap: the lines are cached
      if (row < rows && nn <= THR && active) {
        float *y = Y + row * ldY + f0;
#pragma unroll
        for (int i = 0; i < 4; i += 2)
          if (f0 + i + 1 < F) *reinterpret_cast<float2 *>(y + i) = make_float2(acc[i], acc[i + 1]);
          else if (f0 + i < F) y[i] = acc[i];
      }
    }
    g += stride;
    if (g >= g_end) break;
    b0 = b1; n0 = n1; b1 = b2; n1 = n2;
#pragma unroll
    for (int t = 0; t < T; ++t) { ci[t] = ci2[t]; ca[t] = ca2[t]; }
  }
}

// chunks of long rows (<= 512 entries, one wave each), persistent with the next chunk's staging in flight
template <int G, int GB>
__global__ __launch_bounds__(256) void k_chunks_p(int n_chunks, const int32_t *__restrict__ c_beg,
                                                  const int32_t *__restrict__ c_end,
                                                  const int32_t *__restrict__ c_row,
                                                  const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                  const float *__restrict__ D, int64_t ldD, int F,
                                                  float *__restrict__ Y, int64_t ldY, float *__restrict__ partials,
                                                  int total_waves) {
  constexpr int SLOTS = kWave / G, T = kChunk / kWave;
  const int lane = threadIdx.x & 63, slot = lane / G, q = lane % G;
  const int f0 = q * 4;
  const bool active = f0 < F;
  const int fq = active ? f0 : 0;
  int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  auto stage = [&](int cc, int32_t &n, int32_t(&xi)[T], float(&xa)[T]) {
    n = 0;
    int32_t b = 0;
    if (cc < n_chunks) { b = c_beg[cc]; n = c_end[cc] - b; }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int32_t m = t * kWave + lane;
      xi[t] = (m < n) ? idx[b + m] : 0;
      xa[t] = (m < n) ? val[b + m] : 0.f;
    }
  };
  int32_t n0, n1;
  int32_t ci[T], ci2[T];
  float ca[T], ca2[T];
  stage(c, n0, ci, ca);
  for (;;) {
    stage(c + total_waves, n1, ci2, ca2);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t * kWave >= n0) break;
#pragma unroll
      for (int u0 = 0; u0 < G; u0 += GB) {
        float4 x[GB];
        float a[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) {
          const int src = (u0 + u) * SLOTS + slot;
          const int32_t cc = __shfl(ci[t], src, kWave);
          a[u] = __shfl(ca[t], src, kWave);
          x[u] = *reinterpret_cast<const float4 *>(D + (int64_t)cc * ldD + fq);
        }
#pragma unroll
        for (int u = 0; u < GB; ++u) fma4(acc, a[u], x[u], active && (t * kWave + (u0 + u) * SLOTS + slot < n0));
      }
    }
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
    }
    if (slot == 0 && active) {
      const int32_t row = c_row[c];
      float *p = row >= 0 ? Y + (int64_t)row * ldY + f0 : partials + (int64_t)c * 16 + f0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (f0 + i < F) p[i] = acc[i];
    }
    c += total_waves;
    if (c >= n_chunks) break;
    n0 = n1;
#pragma unroll
    for (int t = 0; t < T; ++t) { ci[t] = ci2[t]; ca[t] = ca2[t]; }
  }
}

extern "C" int lab_rows_p(int T, int GB, int waves_per_cu, int chunk_waves_per_cu, int64_t rows,
                          const int32_t *ptr, const int32_t *idx, const float *val, const float *D, int64_t ldD,
                          int F, float *Y, int n_chunks, const int32_t *c_beg, const int32_t *c_end,
                          const int32_t *c_row, float *partials, int n_long, const int32_t *long_row,
                          const int32_t *long_cptr, int which, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  constexpr int G = 4;
  const int64_t ngroups = (rows + 15) / 16;
  const int blocks = 256 * waves_per_cu / 4;  // multiple of 8
  const int waves_per_xcd = blocks / 8 * 4;
  const int64_t gpx = (ngroups + 7) / 8;
  if (which & 1) {
#define RP(T_, GB_) k_rows_p<G, T_, GB_><<<dim3(blocks), dim3(256), 0, s>>>(rows, ptr, idx, val, D, ldD, F, Y, F, \
                                                                            ngroups, gpx, waves_per_xcd)
    if (T == 8 && GB == 1) RP(8, 1);
    else if (T == 8 && GB == 2) RP(8, 2);
    else if (T == 8 && GB == 4) RP(8, 4);
    else if (T == 4 && GB == 2) RP(4, 2);
    else if (T == 4 && GB == 4) RP(4, 4);
    else return 1;
#undef RP
  }
  if ((which & 2) && n_chunks > 0) {
    const int cblocks = 256 * chunk_waves_per_cu / 4;
    const int total = cblocks * 4;
    if (T == 8) k_chunks_p<G, 4><<<dim3(cblocks), dim3(256), 0, s>>>(n_chunks, c_beg, c_end, c_row, idx, val, D, ldD, F,
                                                                     Y, F, partials, total);
    else k_chunks_p<G, 4><<<dim3(cblocks), dim3(256), 0, s>>>(n_chunks, c_beg, c_end, c_row, idx, val, D, ldD, F, Y, F,
                                                              partials, total);
    if (n_long > 0)
      k_sg_finalize<<<dim3((unsigned)((n_long + 3) / 4)), dim3(256), 0, s>>>(n_long, long_row, long_cptr,
                                                                                      partials, F, Y);
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

// =====================================================================================================
// v3: every wave issues at most 8 gathers per lane in ONE batch (no dependent gather rounds).
//   S  rows of <= 8 entries: 16 consecutive rows per wave, 4 lanes each
//   M  rows of 9..32 entries (list): 4 rows per wave, 16 lanes each (4 sub-slots x <= 8 entries)
//   L  rows of > 32 entries: chunks of <= 128 entries, one wave each (16 slots x 8 entries)
// =====================================================================================================
__device__ __forceinline__ void store_f(float *y, const float (&acc)[4], int f0, int F) {
#pragma unroll
  for (int i = 0; i < 4; i += 2)
    if (f0 + i + 1 < F) *reinterpret_cast<float2 *>(y + i) = make_float2(acc[i], acc[i + 1]);
    else if (f0 + i < F) y[i] = acc[i];
}

// NT gathers issued back to back in one basic block, then consumed (SRC(t) -> source lane of the staged entry)
template <int NT, typename SrcFn, typename OnFn>
__device__ __forceinline__ void gather_batch(const int32_t (&ci)[2], const float (&ca)[2], int regdiv,
                                             const float *__restrict__ D, int64_t ldD, int fq, bool active,
                                             float (&acc)[4], SrcFn src, OnFn on) {
  float4 x[NT];
  float a[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int sl = src(t);
    const int32_t cc = __shfl(ci[t / regdiv], sl, kWave);
    a[t] = __shfl(ca[t / regdiv], sl, kWave);
    x[t] = *reinterpret_cast<const float4 *>(D + (int64_t)cc * ldD + fq);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) fma4(acc, a[t], x[t], active && on(t));
}
#define GATHER_SWITCH(NR, REGDIV, SRC, ON)                                                        \
  switch (NR) {                                                                                   \
    case 1: gather_batch<1>(ci, ca, REGDIV, D, ldD, fq, active, acc, SRC, ON); break;             \
    case 2: gather_batch<2>(ci, ca, REGDIV, D, ldD, fq, active, acc, SRC, ON); break;             \
    case 3: gather_batch<3>(ci, ca, REGDIV, D, ldD, fq, active, acc, SRC, ON); break;             \
    case 4: gather_batch<4>(ci, ca, REGDIV, D, ldD, fq, active, acc, SRC, ON); break;             \
    case 5: gather_batch<5>(ci, ca, REGDIV, D, ldD, fq, active, acc, SRC, ON); break;             \
    case 6: gather_batch<6>(ci, ca, REGDIV, D, ldD, fq, active, acc, SRC, ON); break;             \
    case 7: gather_batch<7>(ci, ca, REGDIV, D, ldD, fq, active, acc, SRC, ON); break;             \
    case 8: gather_batch<8>(ci, ca, REGDIV, D, ldD, fq, active, acc, SRC, ON); break;             \
    default: break;                                                                               \
  }

template <int G>
__global__ __launch_bounds__(256) void k_v3(int64_t rows, const int32_t *__restrict__ ptr,
                                            const int32_t *__restrict__ idx, const float *__restrict__ val,
                                            const float *__restrict__ D, int64_t ldD, int F,
                                            float *__restrict__ Y, int64_t ldY, int n_chunks,
                                            const int32_t *__restrict__ c_beg, const int32_t *__restrict__ c_end,
                                            const int32_t *__restrict__ c_row, float *__restrict__ partials,
                                            int n_mid, const int32_t *__restrict__ mid_rows, int chunk_blocks,
                                            int mid_blocks, int64_t short_blocks, int64_t xcd_per) {
  constexpr int SLOTS = kWave / G;
  const int lane = threadIdx.x & 63, slot = lane / G, q = lane % G;
  const int f0 = q * 4;
  const bool active = f0 < F;
  const int fq = active ? f0 : 0;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if ((int)blockIdx.x < chunk_blocks) {  // ---- L
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const int32_t b = c_beg[c], n = c_end[c] - b;
    int32_t ci[2];
    float ca[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int32_t m = t * kWave + lane;
      ci[t] = (m < n) ? idx[b + m] : 0;
      ca[t] = (m < n) ? val[b + m] : 0.f;
    }
    const int nr = __builtin_amdgcn_readfirstlane((n + SLOTS - 1) / SLOTS);  // gather rounds (wave uniform)
    GATHER_SWITCH(nr, 4, [&](int t) { return (t * SLOTS + slot) & 63; }, [&](int t) { return t * SLOTS + slot < n; })
#pragma unroll
    for (int off = G; off < kWave; off <<= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
    }
    if (slot == 0 && active) {
      const int32_t row = c_row[c];
      if (row >= 0) store_f(Y + (int64_t)row * ldY + f0, acc, f0, F);
      else {
        float *p = partials + (int64_t)c * 16 + f0;
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = acc[i];
      }
    }
    return;
  }
  if ((int)blockIdx.x < chunk_blocks + mid_blocks) {  // ---- M: 4 rows per wave
    const int w = (blockIdx.x - chunk_blocks) * 4 + (threadIdx.x >> 6);
    const int rsel = lane >> 4, l16 = lane & 15, ss = slot & 3;
    const int mi = w * 4 + rsel;
    int32_t b = 0, n = 0, row = -1;
    if (mi < n_mid) {
      row = mid_rows[mi];
      b = ptr[row];
      n = ptr[row + 1] - b;
    }
    if (!__any(row >= 0)) return;
    int32_t ci[2];
    float ca[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int32_t m = t * 16 + l16;
      ci[t] = (m < n) ? idx[b + m] : 0;
      ca[t] = (m < n) ? val[b + m] : 0.f;
    }
    const int base = rsel * 16;
    int nr = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (__any(t * 4 < n)) nr = t + 1;
    // sub-slot ss takes entry t*4 + ss of its row in round t
    GATHER_SWITCH(nr, 4, [&](int t) { return base + ((t * 4 + ss) & 15); }, [&](int t) { return t * 4 + ss < n; })
#pragma unroll
    for (int off = G; off < 16; off <<= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], off, kWave);
    }
    if (ss == 0 && row >= 0 && active) store_f(Y + (int64_t)row * ldY + f0, acc, f0, F);
    return;
  }
  // ---- S: 16 consecutive rows per wave, rows of > 8 entries are someone else's
  int64_t sb = (int64_t)blockIdx.x - chunk_blocks - mid_blocks;
  if (xcd_per > 0) {
    sb = (sb & 7) * xcd_per + (sb >> 3);
    if (sb >= short_blocks) return;
  }
  const int64_t row = (sb * 4 + (threadIdx.x >> 6)) * SLOTS + slot;
  int32_t b = 0, n = 0;
  if (row < rows) {
    b = ptr[row];
    n = ptr[row + 1] - b;
  }
  const bool mine = row < rows && n <= 8;
  if (!mine) n = 0;
  if (!__any(mine)) return;
  int32_t ci[2];
  float ca[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int32_t m = t * G + q;
    ci[t] = (m < n) ? idx[b + m] : 0;
    ca[t] = (m < n) ? val[b + m] : 0.f;
  }
  const int sbase = slot * G;
  int nr = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t)
    if (__any(t < n)) nr = t + 1;
  GATHER_SWITCH(nr, G, [&](int t) { return sbase + (t % G); }, [&](int t) { return t < n; })
  if (mine && active) store_f(Y + row * ldY + f0, acc, f0, F);
}

extern "C" int lab_v3(int64_t rows, const int32_t *ptr, const int32_t *idx, const float *val, const float *D,
                      int64_t ldD, int F, float *Y, int n_chunks, const int32_t *c_beg, const int32_t *c_end,
                      const int32_t *c_row, float *partials, int n_long, const int32_t *long_row,
                      const int32_t *long_cptr, int n_mid, const int32_t *mid_rows, int xcd, int which,
                      void *stream) {
  hipStream_t s = (hipStream_t)stream;
  constexpr int G = 4;
  const int64_t short_waves = (rows + 15) / 16;
  int64_t short_blocks = (short_waves + 3) / 4;
  int chunk_blocks = (n_chunks + 3) / 4;
  int mid_blocks = ((n_mid + 3) / 4 + 3) / 4;
  if (!(which & 1)) short_blocks = 0;
  if (!(which & 2)) chunk_blocks = 0;
  if (!(which & 4)) mid_blocks = 0;
  const int64_t xcd_per = xcd ? (short_blocks + 7) / 8 : 0;
  const int64_t launch_short = xcd ? xcd_per * 8 : short_blocks;
  const int64_t grid = launch_short + chunk_blocks + mid_blocks;
  if (grid == 0) return 0;
  k_v3<G><<<dim3((unsigned)grid), dim3(256), 0, s>>>(rows, ptr, idx, val, D, ldD, F, Y, F, n_chunks, c_beg, c_end,
                                                     c_row, partials, n_mid, mid_rows, chunk_blocks, mid_blocks,
                                                     short_blocks, xcd_per);
  if (n_long > 0 && (which & 2))
    k_sg_finalize<<<dim3((unsigned)((n_long + 3) / 4)), dim3(256), 0, s>>>(n_long, long_row, long_cptr,
                                                                                    partials, F, Y);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
