// Modality encoders that feed X — the dense work in front of the graph path.
//
//   k_mlp_gate_scatter_fwd / _bwd   the literal MLPs (mrgcn/models/perceptron.py:6-46: Linear -> ReLU blocks,
//       1 layer for numeric / boolean, 2 for the temporal datatypes, widths <= 16) FUSED with the gate multiply
//       and the masked scatter into the feature matrix (mrgcn/models/mrgcn.py:285-303):
//           XF[rows[i], off : off + d_out] = gate * MLP(enc[i, :])
//       one thread per literal, all weights in LDS; the backward recomputes the activations and reduces
//       dW = G^T A per block as a 16 x 16 product out of LDS (no per-lane atomics).
//   k_gemm_f32   C = act(alpha * op(A) . op(B) + bias) on the matrix cores (v_mfma_f32_16x16x4_f32: exact
//       fp32 products and sums, so the 1e-4 parity bar holds) with the epilogues the heads need: bias, ReLU,
//       ReLU mask of a saved activation.  The `pre_fc -> ReLU -> fc` heads of the string / image encoders
//       (mrgcn/models/imagecnn.py:31-41, transformer.py:29-38) and the TCNN's fully connected tail
//       (temporal_cnn.py:147-153) run forward and backward on it, and so does the TCNN's Conv1d in implicit
//       im2col form (A-loader modes below).
//   k_colsum_f32 bias gradients (column sums).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "common.hpp"
#include "config.hpp"

namespace mrgcn {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ---------------------------------------------------------------------------------------------------
// fused MLP + gate + scatter
// ---------------------------------------------------------------------------------------------------
constexpr int kMlpW = 16;  // widest layer the fused kernels take
constexpr int kMlpL = 4;   // deepest

struct MlpGrad {
  float *dW[4];               // [dims[l + 1]][dims[l]], accumulated into
  float *db[4];               // nullable
};

struct MlpDesc {
  int L;
  int dims[kMlpL + 1];        // dims[0] = input width, dims[l + 1] = output width of layer l
  const float *W[kMlpL];      // [dims[l + 1]][dims[l]] (nn.Linear layout)
  const float *b[kMlpL];      // nullable
};

__device__ __forceinline__ int mlp_w_off(const MlpDesc &d, int l) {  // offset of layer l inside the LDS copy
  int o = 0;
  for (int i = 0; i < l; ++i) o += d.dims[i + 1] * (d.dims[i] + 1);
  return o;
}

__device__ void mlp_stage_weights(const MlpDesc &d, float *s_w) {
  for (int l = 0; l < d.L; ++l) {
    const int din = d.dims[l], dout = d.dims[l + 1], base = mlp_w_off(d, l);
    for (int t = threadIdx.x; t < dout * (din + 1); t += blockDim.x) {
      const int j = t / (din + 1), k = t - j * (din + 1);
      s_w[base + t] = k < din ? d.W[l][j * din + k] : (d.b[l] ? d.b[l][j] : 0.f);  // bias rides as column din
    }
  }
}

// activations of every layer for one input row (registers); a[0] = input, a[l + 1] = relu(W_l a[l] + b_l)
__device__ __forceinline__ void mlp_forward_row(const MlpDesc &d, const float *s_w, float (&a)[kMlpL + 1][kMlpW]) {
#pragma unroll
  for (int l = 0; l < kMlpL; ++l) {
    if (l < d.L) {
      const int din = d.dims[l], dout = d.dims[l + 1];
      const float *w = s_w + mlp_w_off(d, l);
#pragma unroll
      for (int j = 0; j < kMlpW; ++j) {
        float y = 0.f;
        if (j < dout) {
          const float *wj = w + j * (din + 1);
          y = wj[din];
#pragma unroll
          for (int k = 0; k < kMlpW; ++k)
            if (k < din) y = fmaf(wj[k], a[l][k], y);
          y = fmaxf(y, 0.f);
        }
        a[l + 1][j] = y;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_mlp_gate_scatter_fwd(MlpDesc d, const float *__restrict__ X, int64_t ldx,
                                                              int64_t n, const float *__restrict__ gate,
                                                              const int64_t *__restrict__ rows,
                                                              float *__restrict__ XF, int64_t ldxf, int off) {
  extern __shared__ float s_w[];
  mlp_stage_weights(d, s_w);
  __syncthreads();
  const float g = *gate;
  const int dout = d.dims[d.L];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float a[kMlpL + 1][kMlpW];
#pragma unroll
    for (int k = 0; k < kMlpW; ++k) a[0][k] = k < d.dims[0] ? X[i * ldx + k] : 0.f;
    mlp_forward_row(d, s_w, a);
    float *o = XF + (rows ? rows[i] : i) * ldxf + off;
#pragma unroll
    for (int l = 1; l <= kMlpL; ++l)
      if (l == d.L) {
#pragma unroll
        for (int j = 0; j < kMlpW; ++j)
          if (j < dout) o[j] = a[l][j] * g;
      }
  }
}

// backward: dW_l, db_l, dgate (+=; zeroed by the caller).  Per layer the block's 256 rows leave their masked
// output gradient G[r][j] and input activation A[r][k] in LDS, thread (j, k) sums the 256 products.
__global__ __launch_bounds__(256) void k_mlp_gate_scatter_bwd(MlpDesc d, const float *__restrict__ X, int64_t ldx,
                                                              int64_t n, const float *__restrict__ gate,
                                                              const int64_t *__restrict__ rows,
                                                              const float *__restrict__ dXF, int64_t ldxf, int off,
                                                              MlpGrad gr, float *__restrict__ dgate) {
  extern __shared__ float s_mem[];
  int wtot = 0;
  for (int l = 0; l < d.L; ++l) wtot += d.dims[l + 1] * (d.dims[l] + 1);
  float *s_w = s_mem;
  float *s_g = s_mem + wtot;                 // [256][kMlpW + 1]
  float *s_a = s_g + 256 * (kMlpW + 1);      // [256][kMlpW + 1]
  mlp_stage_weights(d, s_w);
  __syncthreads();
  const float gt = *gate;
  const int tj = threadIdx.x >> 4, tk = threadIdx.x & 15;  // this thread's (j, k) of the 16 x 16 reduction
  float dg_local = 0.f;
  float accw[kMlpL], accb[kMlpL];
#pragma unroll
  for (int l = 0; l < kMlpL; ++l) accw[l] = accb[l] = 0.f;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < n; i0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i0 + threadIdx.x;
    const bool on = i < n;
    float a[kMlpL + 1][kMlpW];
#pragma unroll
    for (int k = 0; k < kMlpW; ++k) a[0][k] = (on && k < d.dims[0]) ? X[i * ldx + k] : 0.f;
    mlp_forward_row(d, s_w, a);
    float g[kMlpW];
    const float *go = dXF + (on ? (rows ? rows[i] : i) : 0) * ldxf + off;
#pragma unroll
    for (int l = 1; l <= kMlpL; ++l)
      if (l == d.L) {
#pragma unroll
        for (int j = 0; j < kMlpW; ++j) {
          const float gj = (on && j < d.dims[l]) ? go[j] : 0.f;
          dg_local = fmaf(a[l][j], gj, dg_local);   // d gate = <MLP(x), dXF>
          g[j] = gj * gt;
        }
      }
#pragma unroll
    for (int l = kMlpL - 1; l >= 0; --l) {
      if (l < d.L) {
        const int din = d.dims[l], dout = d.dims[l + 1];
#pragma unroll
        for (int j = 0; j < kMlpW; ++j) {
          if (!(j < dout && a[l + 1][j] > 0.f)) g[j] = 0.f;   // ReLU
          s_g[threadIdx.x * (kMlpW + 1) + j] = g[j];
        }
#pragma unroll
        for (int k = 0; k < kMlpW; ++k) s_a[threadIdx.x * (kMlpW + 1) + k] = a[l][k];
        __syncthreads();
        if (tj < dout && tk < din) {
          float sw = 0.f;
          for (int r = 0; r < 256; ++r) sw = fmaf(s_g[r * (kMlpW + 1) + tj], s_a[r * (kMlpW + 1) + tk], sw);
          accw[l] += sw;
        }
        if (tk == 0 && tj < dout) {
          float sb = 0.f;
          for (int r = 0; r < 256; ++r) sb += s_g[r * (kMlpW + 1) + tj];
          accb[l] += sb;
        }
        __syncthreads();
        // gradient w.r.t. this layer's input
        const float *w = s_w + mlp_w_off(d, l);
        float gp[kMlpW];
#pragma unroll
        for (int k = 0; k < kMlpW; ++k) {
          float s = 0.f;
          if (k < din) {
#pragma unroll
            for (int j = 0; j < kMlpW; ++j)
              if (j < dout) s = fmaf(w[j * (din + 1) + k], g[j], s);
          }
          gp[k] = s;
        }
#pragma unroll
        for (int k = 0; k < kMlpW; ++k) g[k] = gp[k];
      }
    }
  }
#pragma unroll
  for (int l = 0; l < kMlpL; ++l) {
    if (l < d.L) {
      const int din = d.dims[l], dout = d.dims[l + 1];
      if (tj < dout && tk < din && accw[l] != 0.f) atomicAdd(&gr.dW[l][tj * din + tk], accw[l]);
      if (tk == 0 && tj < dout && gr.db[l] && accb[l] != 0.f) atomicAdd(&gr.db[l][tj], accb[l]);
    }
  }
  // gate: block reduction
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dg_local += __shfl_xor(dg_local, o, kWave);
  __shared__ float s_dg[4];
  if ((threadIdx.x & 63) == 0) s_dg[threadIdx.x >> 6] = dg_local;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = s_dg[0] + s_dg[1] + s_dg[2] + s_dg[3];
    if (t != 0.f) atomicAdd(dgate, t);
  }
}

// ---------------------------------------------------------------------------------------------------
// C[M][N] = epilogue( alpha * sum_k Aop(m, k) * Bop(k, n) )     f32 MFMA 16x16x4, block tile 64 x 64, K step 16
//   A-loader modes: 0  A[m*lda + k]            (row major)
//                   1  A[k*lda + m]            (transposed storage)
//                   2  conv1d im2col of x[b][ci][t]:  m = b*Tout + t,  k = ci*KW + kw  ->
//                      x[(b*Cin + ci)*Tin + t + kw - pad]  (0 outside)                      (forward, dW)
//                   3  the same with m and k swapped (A^T of mode 2: dW = im2col(x)^T . dY)
//   B-loader modes: 0  B[k*ldb + n]   1  B[n*ldb + k]
//                   2  conv output gradient as [m = b*Tout + t][n = co]: dY[(b*Cout + co)*Tout + t]
//   C-store modes:  0  C[m*ldc + n]   2  conv layout out[(b*Cout + n)*Tout + t],  m = b*Tout + t
// ---------------------------------------------------------------------------------------------------
struct ConvGeom {
  int Cin = 0, Tin = 0, KW = 0, pad = 0, Tout = 0, Cout = 0;
};

struct GemmArgs {
  const float *A, *B;
  float *C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  int amode, bmode, cmode;
  const float *bias;        // [N] nullable
  int relu;
  const float *mask;        // nullable: C *= (mask > 0), same addressing as C
  float alpha;
  ConvGeom cg;
  int kchunk = 0;           // > 0: split K — block z takes k in [z kchunk, (z + 1) kchunk) and ADDS into a zeroed C
};

__device__ __forceinline__ float gemm_a(const GemmArgs &g, int m, int k) {
  if (m >= g.M || k >= g.K) return 0.f;
  if (g.amode == 0) return g.A[(int64_t)m * g.lda + k];
  if (g.amode == 1) return g.A[(int64_t)k * g.lda + m];
  int mm = m, kk = k;
  if (g.amode == 3) { mm = k; kk = m; }   // A^T of the im2col matrix
  const int b = mm / g.cg.Tout, t = mm - b * g.cg.Tout;
  const int ci = kk / g.cg.KW, kw = kk - ci * g.cg.KW;
  const int ti = t + kw - g.cg.pad;
  if (ti < 0 || ti >= g.cg.Tin) return 0.f;
  return g.A[((int64_t)b * g.cg.Cin + ci) * g.cg.Tin + ti];
}
__device__ __forceinline__ float gemm_b(const GemmArgs &g, int k, int n) {
  if (k >= g.K || n >= g.N) return 0.f;
  if (g.bmode == 0) return g.B[(int64_t)k * g.ldb + n];
  if (g.bmode == 1) return g.B[(int64_t)n * g.ldb + k];
  const int b = k / g.cg.Tout, t = k - b * g.cg.Tout;   // bmode 2: k runs over (b, t), n over channels
  return g.B[((int64_t)b * g.cg.Cout + n) * g.cg.Tout + t];
}
__device__ __forceinline__ int64_t gemm_c_index(const GemmArgs &g, int m, int n) {
  if (g.cmode == 0) return (int64_t)m * g.ldc + n;
  const int b = m / g.cg.Tout, t = m - b * g.cg.Tout;
  return ((int64_t)b * g.cg.Cout + n) * g.cg.Tout + t;
}

constexpr int kGT = 64, kGK = 16, kGP = kGK + 4;  // tile, K step, padded LDS row (rows on different banks)

__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs g) {
  __shared__ __align__(16) float As[kGT][kGP];   // [m][k]
  __shared__ __align__(16) float Bs[kGT][kGP];   // [n][k]
  const int m0 = blockIdx.y * kGT, n0 = blockIdx.x * kGT;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32;   // this wave's 32 x 32 quarter
  const int lm = lane & 15, kq = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // loader mapping: thread t fills As[t / 4][4 * (t % 4) + 0..3] and the same of Bs
  const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;
  for (int k0 = 0; k0 < g.K; k0 += kGK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      As[lr][lk + i] = gemm_a(g, m0 + lr, k0 + lk + i);
      Bs[lr][lk + i] = gemm_b(g, k0 + lk + i, n0 + lr);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < kGK; ks += 16) {
      f32x4 av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = *reinterpret_cast<const f32x4 *>(&As[wm + i * 16 + lm][ks + 4 * kq]);
        bv[i] = *reinterpret_cast<const f32x4 *>(&Bs[wn + i * 16 + lm][ks + 4 * kq]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // k slot kq of the s-th MFMA stands for k = k0 + 4 kq + s: any bijection of the K step works
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  // D: lane (n = lane & 15, q = lane >> 4) holds rows 4q + reg of its 16 x 16 tile
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = m0 + wm + i * 16 + 4 * kq + reg, n = n0 + wn + j * 16 + lm;
        if (m >= g.M || n >= g.N) continue;
        float v = acc[i][j][reg] * g.alpha;
        if (g.bias) v += g.bias[n];
        if (g.relu) v = fmaxf(v, 0.f);
        const int64_t ci = gemm_c_index(g, m, n);
        if (g.mask && !(g.mask[ci] > 0.f)) v = 0.f;
        g.C[ci] = v;
      }
}

// ---------------------------------------------------------------------------------------------------
// The tiled form of the product (128 x 128, 128 x 64 or 64 x 64 block tiles): the loader modes are template parameters, every element is fetched by a dword
// buffer load whose lanes run along the direction the operand is contiguous in memory, and there is no branch in
// the K loop.
//   * element (row i, column k) of an operand sits at byte offset  roff(i) + coff(k)  of its base; a row or a column
//     outside the matrix gets the offset kOOB, so that the sum falls behind the buffer descriptor's range and the
//     load returns 0 by itself (operands of at most 2^29 bytes: kOOB minus the few bytes by which the conv modes'
//     row offsets can be negative still lies behind them).  The conv modes carry the position inside the sequence
//     (rt + ct) and select kOOB where it leaves [0, Tin): the zero padding.
//   * index decompositions (m -> (b, t), k -> (ci, kw), ...) are a multiply-high by a magic number the launcher
//     computed (exact while x * d < 2^32, which it checks).
//   * ROWL operands (A modes 1, 2; B mode 0: memory runs along the tile rows): thread = (row, NE consecutive k);
//     the k of a wave are uniform, their decomposition is scalar work.  KL operands (A modes 0, 3; B modes 1, 2:
//     memory runs along k): thread = (k, NE rows).  Both leave Xs[row][k] (row stride BK + 4 floats) in LDS, from
//     which every MFMA fragment is one conflict-free ds_read_b128.
//   * two LDS stages, one barrier per K step; the next step's elements fly under this step's products.
//   * the conv output (cmode 2: y[b][n][t], contiguous along m = (b, t)) goes through LDS per wave so that a store
//     instruction writes 64 consecutive positions of one channel.
//   * split K (kchunk > 0): the partial tiles are added into a zeroed C with float atomics, block z = 0 adds the
//     bias; a ReLU / mask epilogue then runs as k_mm_finish.
// ---------------------------------------------------------------------------------------------------
using rsrc_t = __amdgpu_buffer_rsrc_t;
using u32x4m = __attribute__((ext_vector_type(4))) uint32_t;
using bf16x8m = __attribute__((ext_vector_type(8))) __bf16;
constexpr uint32_t kOOB = 1u << 30;
constexpr int64_t kMmMaxBytes = (int64_t)1 << 29;

struct MmArgs {
  const float *A, *B;
  float *C;
  uint32_t a_bytes, b_bytes;
  int M, N, K;
  int32_t lda, ldb;
  int64_t ldc;
  const float *bias;
  int relu;
  const float *mask;
  float alpha;
  ConvGeom cg;
  int kchunk;
  uint32_t mg_tout, mg_kw, mg_tout16;  // floor(2^32 / d) + 1 for d = Tout, KW, 16 Tout (0 stands for d == 1)
};

struct Coord {
  uint32_t off;  // bytes
  int32_t t;     // conv position contribution
};

__device__ __forceinline__ uint32_t mdiv(uint32_t x, uint32_t magic) { return magic ? __umulhi(x, magic) : x; }

template <bool IS_A, int MODE>
__device__ __forceinline__ Coord mm_row(const MmArgs &g, int i) {
  if (i >= (IS_A ? g.M : g.N)) return Coord{kOOB, 0};
  if constexpr (IS_A) {
    if constexpr (MODE == 0) return Coord{(uint32_t)(i * g.lda) * 4u, 0};
    else if constexpr (MODE == 1) return Coord{(uint32_t)i * 4u, 0};
    else if constexpr (MODE == 2) {
      const int b = (int)mdiv((uint32_t)i, g.mg_tout), t = i - b * g.cg.Tout;
      return Coord{(uint32_t)(b * g.cg.Cin * g.cg.Tin + t) * 4u, t};
    } else {
      const int ci = (int)mdiv((uint32_t)i, g.mg_kw), kw = i - ci * g.cg.KW;
      return Coord{(uint32_t)(ci * g.cg.Tin + kw - g.cg.pad) * 4u, kw - g.cg.pad};
    }
  } else {
    if constexpr (MODE == 0) return Coord{(uint32_t)i * 4u, 0};
    else if constexpr (MODE == 1) return Coord{(uint32_t)(i * g.ldb) * 4u, 0};
    else return Coord{(uint32_t)(i * g.cg.Tout) * 4u, 0};
  }
}
template <bool IS_A, int MODE>
__device__ __forceinline__ Coord mm_col(const MmArgs &g, int k, int kend) {
  if (k >= kend) return Coord{kOOB, 0};
  if constexpr (IS_A) {
    if constexpr (MODE == 0) return Coord{(uint32_t)k * 4u, 0};
    else if constexpr (MODE == 1) return Coord{(uint32_t)(k * g.lda) * 4u, 0};
    else if constexpr (MODE == 2) {
      const int ci = (int)mdiv((uint32_t)k, g.mg_kw), kw = k - ci * g.cg.KW;
      return Coord{(uint32_t)(ci * g.cg.Tin + kw - g.cg.pad) * 4u, kw - g.cg.pad};
    } else {
      const int b = (int)mdiv((uint32_t)k, g.mg_tout), t = k - b * g.cg.Tout;
      return Coord{(uint32_t)(b * g.cg.Cin * g.cg.Tin + t) * 4u, t};
    }
  } else {
    if constexpr (MODE == 0) return Coord{(uint32_t)(k * g.ldb) * 4u, 0};
    else if constexpr (MODE == 1) return Coord{(uint32_t)k * 4u, 0};
    else {
      const int b = (int)mdiv((uint32_t)k, g.mg_tout), t = k - b * g.cg.Tout;
      return Coord{(uint32_t)(b * g.cg.Cout * g.cg.Tout + t) * 4u, 0};
    }
  }
}

// ST: the element type the tile is STAGED in — float (v_mfma_f32_16x16x4_f32) or uint16_t = bf16 (the bf16 pipeline:
// operands stay fp32 in memory, are rounded as they enter LDS and multiplied by v_mfma_f32_16x16x32_bf16; BK = 32)
template <bool IS_A, int MODE, bool ROWL, int ROWS, int BK, typename ST = float>
struct MmLoader {
  static constexpr bool CONV = IS_A && MODE >= 2;
  static constexpr int NE = ROWS * BK / 256;  // elements per thread and K step
  static constexpr int LDK = sizeof(ST) == 4 ? BK + 4 : BK + 8;  // row stride in elements: rows stay 16-byte aligned
  static constexpr int RS = 256 / BK;         // KL: distance of a thread's rows
  static constexpr int NR = ROWL ? 1 : NE;
  uint32_t roff[NR];
  int32_t rt[CONV ? NR : 1];
  __device__ __forceinline__ void init(const MmArgs &g, int row0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int r = ROWL ? (int)(threadIdx.x % ROWS) : (int)(threadIdx.x / BK) + RS * i;
      const Coord c = mm_row<IS_A, MODE>(g, row0 + r);
      roff[i] = c.off;
      if constexpr (CONV) rt[i] = c.t;
    }
  }
  __device__ __forceinline__ uint32_t at(const MmArgs &g, int i, const Coord &c) const {
    uint32_t off = roff[i] + c.off;
    if constexpr (CONV) off = (uint32_t)(rt[i] + c.t) < (uint32_t)g.cg.Tin ? off : kOOB;
    return off;
  }
  __device__ __forceinline__ void load(const MmArgs &g, rsrc_t rs, int k0, int kend, float (&v)[NE]) const {
    if constexpr (ROWL) {
      // the wave's NE columns are the same for all its lanes (ROWS >= 64): lane j decomposes column j once, on the
      // vector unit, and the loads read its result by lane index (as scalar work the decompositions of the 20 waves
      // of a CU kept its one scalar unit as busy as the matrix cores)
      static_assert((NE & (NE - 1)) == 0, "NE is a power of two");
      const int ks = (int)(threadIdx.x / ROWS) * NE;
      const Coord cl = mm_col<IS_A, MODE>(g, k0 + ks + (int)(threadIdx.x & (NE - 1)), kend);
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        Coord c;
        c.off = (uint32_t)__builtin_amdgcn_readlane((int)cl.off, j);
        c.t = CONV ? __builtin_amdgcn_readlane(cl.t, j) : 0;
        v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, at(g, 0, c), 0, 0));
      }
    } else {
      const Coord c = mm_col<IS_A, MODE>(g, k0 + (int)(threadIdx.x % BK), kend);
#pragma unroll
      for (int i = 0; i < NE; ++i)
        v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, at(g, i, c), 0, 0));
    }
  }
  __device__ __forceinline__ void stage(ST *Xs, const float (&v)[NE]) const {
    if constexpr (ROWL) {
      ST *dst = Xs + (threadIdx.x % ROWS) * LDK + (threadIdx.x / ROWS) * NE;
      if constexpr (sizeof(ST) == 4) {
#pragma unroll
        for (int h = 0; h < NE / 4; ++h)
          *reinterpret_cast<f32x4 *>(dst + 4 * h) = f32x4{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
      } else {
        static_assert(sizeof(ST) == 4 || NE % 8 == 0, "bf16 staging stores whole 16-byte pieces");
#pragma unroll
        for (int h = 0; h < NE / 8; ++h) {
          u32x4m o;
          o.x = (uint32_t)f32_to_bf16(v[8 * h + 0]) | ((uint32_t)f32_to_bf16(v[8 * h + 1]) << 16);
          o.y = (uint32_t)f32_to_bf16(v[8 * h + 2]) | ((uint32_t)f32_to_bf16(v[8 * h + 3]) << 16);
          o.z = (uint32_t)f32_to_bf16(v[8 * h + 4]) | ((uint32_t)f32_to_bf16(v[8 * h + 5]) << 16);
          o.w = (uint32_t)f32_to_bf16(v[8 * h + 6]) | ((uint32_t)f32_to_bf16(v[8 * h + 7]) << 16);
          *reinterpret_cast<u32x4m *>(dst + 8 * h) = o;
        }
      }
    } else {
      ST *dst = Xs + (threadIdx.x / BK) * LDK + (threadIdx.x % BK);
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        if constexpr (sizeof(ST) == 4) dst[RS * i * LDK] = v[i];
        else dst[RS * i * LDK] = f32_to_bf16(v[i]);
      }
    }
  }
};

// MT / NT: 16 x 16 MFMA tiles per wave along m / n — block tile 32 MT x 32 NT (128 x 128, 128 x 64, 64 x 64)
// AR / BR: the lanes of the A / B loader run along the tile rows (else along k)
template <int AMODE, int BMODE, int CMODE, bool AR, bool BR, int MT, int NT, int BK, typename ST = float>
__global__ __launch_bounds__(256) void k_mm_tile(MmArgs g) {
  constexpr bool BF = sizeof(ST) == 2;
  static_assert(!BF || BK == 32, "the bf16 form multiplies one 32-wide k-step per stage");
  constexpr int BM = 32 * MT, BN = 32 * NT, LDK = BF ? BK + 8 : BK + 4, ASZ = BM * LDK, BSZ = BN * LDK;
  constexpr int WR = 16 * MT, CST = WR + 4;  // rows of a wave; row stride of its transposition buffer
  static_assert(2 * (ASZ + BSZ) * sizeof(ST) >= 4 * 16 * CST * sizeof(float),
                "the epilogue's transposition buffer fits the stages");
  __shared__ __align__(16) ST smem_st[2 * (ASZ + BSZ)];
  float *smem = reinterpret_cast<float *>(smem_st);  // (the epilogue's view)
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = (wv >> 1) * WR, wn = (wv & 1) * (16 * NT);
  const int lm = lane & 15, kq = lane >> 4;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)g.A, 0, g.a_bytes, 0x00020000);
  const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)g.B, 0, g.b_bytes, 0x00020000);
  MmLoader<true, AMODE, AR, BM, BK, ST> la;
  MmLoader<false, BMODE, BR, BN, BK, ST> lb;
  la.init(g, m0);
  lb.init(g, n0);
  const int kbeg = g.kchunk > 0 ? blockIdx.z * g.kchunk : 0;
  const int kend = g.kchunk > 0 ? min(g.K, kbeg + g.kchunk) : g.K;
  float va[decltype(la)::NE], vb[decltype(lb)::NE];
  la.load(g, ra, kbeg, kend, va);
  lb.load(g, rb, kbeg, kend, vb);
  la.stage(smem_st, va);
  lb.stage(smem_st + ASZ, vb);
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      la.load(g, ra, k0 + BK, kend, va);
      lb.load(g, rb, k0 + BK, kend, vb);
    }
    const ST *Ac = smem_st + cur * (ASZ + BSZ), *Bc = Ac + ASZ;
    if constexpr (BF) {
      // one k-step of 32: lane (m = lane & 15, g = lane >> 4) holds k = 8 g .. 8 g + 7 of its row (16 bytes)
      bf16x8m a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        a[i] = __builtin_bit_cast(bf16x8m, *reinterpret_cast<const u32x4m *>(Ac + (wm + 16 * i + lm) * LDK + 8 * kq));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        b[j] = __builtin_bit_cast(bf16x8m, *reinterpret_cast<const u32x4m *>(Bc + (wn + 16 * j + lm) * LDK + 8 * kq));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int kg = 0; kg < BK / 16; ++kg) {
        f32x4 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4 *>(Ac + (wm + 16 * i + lm) * LDK + 16 * kg + 4 * kq);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4 *>(Bc + (wn + 16 * j + lm) * LDK + 16 * kg + 4 * kq);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      ST *An = smem_st + (cur ^ 1) * (ASZ + BSZ);
      la.stage(An, va);
      lb.stage(An + ASZ, vb);
    }
    __syncthreads();
    cur ^= 1;
  }
  const bool split = g.kchunk > 0;
  const bool add_bias = g.bias && (!split || blockIdx.z == 0);
  auto emit = [&](float v, int n, int64_t ci) {   // one finished element (alpha applied) to its place in C
    if (add_bias) v += g.bias[n];
    if (split) {
      atomicAdd(g.C + ci, v);
      return;
    }
    if (g.relu) v = fmaxf(v, 0.f);
    if (g.mask && !(g.mask[ci] > 0.f)) v = 0.f;
    g.C[ci] = v;
  };
  if constexpr (CMODE == 0) {
    // D: lane (n = lane & 15, q = lane >> 4) holds rows 4q + reg of its 16 x 16 tile
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn + 16 * j + lm;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int m = m0 + wm + 16 * i + 4 * kq + reg;
          if (m < g.M && n < g.N) emit(acc[i][j][reg] * g.alpha, n, (int64_t)m * g.ldc + n);
        }
    }
  } else {
    // y[b][n][t]: the wave's rows (b, t) x 16 columns pass through LDS (Cs[column][row]) and leave in the order
    // memory has them
    float *Cs = smem + wv * (16 * CST);
    const int Tout = g.cg.Tout, mw = m0 + wm;
    if (Tout >= 64) {
      // long sequences: lane = row; a store instruction writes up to 64 consecutive positions of one channel
      const int m = mw + lane, mc = min(m, g.M - 1);
      const int b = (int)mdiv((uint32_t)mc, g.mg_tout), t = mc - b * Tout;
      const int64_t base = (int64_t)b * g.cg.Cout * Tout + t;
      const bool on = lane < WR && m < g.M;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = 0; i < MT; ++i) *reinterpret_cast<f32x4 *>(Cs + lm * CST + 16 * i + 4 * kq) = acc[i][j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const int n = n0 + wn + 16 * j + c;   // wave uniform
          if (on && n < g.N) emit(Cs[c * CST + min(lane, WR - 1)] * g.alpha, n, base + (int64_t)n * Tout);
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      // short sequences: the 16 channels x Tout positions of one b are contiguous in y; lanes enumerate
      // (b, channel, t) of the sequences the wave's rows touch
      const int mhi = min(g.M, mw + WR) - 1;   // last row of the wave inside the matrix (may be < mw)
      const int blo = (int)mdiv((uint32_t)min(mw, g.M - 1), g.mg_tout);
      const int bhi = (int)mdiv((uint32_t)max(mhi, 0), g.mg_tout);
      const int per_b = 16 * Tout, total = mhi >= mw ? (bhi - blo + 1) * per_b : 0;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int i = 0; i < MT; ++i) *reinterpret_cast<f32x4 *>(Cs + lm * CST + 16 * i + 4 * kq) = acc[i][j];
        __builtin_amdgcn_wave_barrier();
        for (int f = lane; f < total; f += 64) {
          const int bl = (int)mdiv((uint32_t)f, g.mg_tout16), rem = f - bl * per_b;
          const int c = (int)mdiv((uint32_t)rem, g.mg_tout), t = rem - c * Tout;
          const int b = blo + bl, row = b * Tout + t - mw, n = n0 + wn + 16 * j + c;
          if (row >= 0 && row <= mhi - mw && n < g.N)
            emit(Cs[c * CST + row] * g.alpha, n, ((int64_t)b * g.cg.Cout + n) * Tout + t);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
}

// ReLU / mask epilogue of a split product (the bias went in with block z = 0)
__global__ __launch_bounds__(256) void k_mm_finish(float *__restrict__ C, int64_t ldc, int M, int N, int relu,
                                                   const float *__restrict__ mask) {
  const int64_t total = (int64_t)M * N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = e / N, ci = m * ldc + (e - m * N);
    float v = C[ci];
    if (relu) v = fmaxf(v, 0.f);
    if (mask && !(mask[ci] > 0.f)) v = 0.f;
    C[ci] = v;
  }
}

// out[n] = sum_m X[m][n] (row major, ld) — bias gradients.  Block (x, y): 64 columns x the y-th slab of rows; a tall
// matrix (gridDim.y > 1) adds its slab sums into the zeroed out with float atomics
__global__ __launch_bounds__(256) void k_colsum_f32(const float *__restrict__ X, int64_t ld, int M, int N,
                                                    float *__restrict__ out) {
  __shared__ float s[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int per = (M + gridDim.y - 1) / gridDim.y, mb = blockIdx.y * per, me = min(M, mb + per);
  float t = 0.f;
  if (n < N)
    for (int m = mb + part; m < me; m += 4) t += X[(int64_t)m * ld + n];
  s[part][threadIdx.x & 63] = t;
  __syncthreads();
  if (part == 0 && n < N) {
    const float v = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
    if (gridDim.y > 1) atomicAdd(&out[n], v);
    else out[n] = v;
  }
}

}  // namespace
}  // namespace mrgcn

namespace mrgcn {
namespace {

// launch of k_mm_tile; MRGCN_ERR_UNSUPPORTED = not a shape / mode combination it takes (the caller goes on to the
// older forms)
int mm_tile_launch(const GemmArgs &o, hipStream_t stream, bool bf16 = false) {
  const int M = o.M, N = o.N, K = o.K;
  int cmode = o.cmode;
  int64_t ldc = o.ldc;
  if (cmode == 2 && o.cg.Tout == 1) {  // y[b][n][0] is a row-major [b][n]
    cmode = 0;
    ldc = o.cg.Cout;
  }
  // a convolution whose kernel covers the whole (unpadded) sequence is a plain product: x[b][(ci, kw)] and
  // y[b][n] are row-major matrices
  int amode = o.amode, bmode = o.bmode;
  int64_t lda = o.lda, ldb = o.ldb;
  if (amode >= 2 && o.cg.Tout == 1 && o.cg.pad == 0 && o.cg.Tin == o.cg.KW) {
    lda = (int64_t)o.cg.Cin * o.cg.KW;
    amode = amode == 2 ? 0 : 1;
  }
  if (bmode == 2 && o.cg.Tout == 1 && amode < 2) {
    ldb = o.cg.Cout;
    bmode = 0;
  }
  const bool conv = amode >= 2 || bmode == 2 || cmode == 2;
  // lanes along the tile rows where memory runs along them — and for the (b, t)-indexed reductions of short
  // sequences (dW: k = (b, t)), whose runs along k are Tout elements long while the rows are contiguous or evenly
  // strided
  const bool short_seq = conv && o.cg.Tout < 16;
  const bool ar = amode == 1 || amode == 2 || (amode == 3 && short_seq);
  const bool br = bmode == 0 || (bmode == 2 && short_seq);
  int combo = -1;
  if (amode < 2 && bmode < 2 && cmode == 0) combo = amode * 2 + bmode;
  else if (amode == 2 && bmode < 2) combo = (cmode == 2 ? 4 : 7) + (bmode == 0 ? 1 : 0);   // 4, 5 / 7, 8
  else if (amode == 3 && bmode == 2 && cmode == 0 && ar == br) combo = ar ? 9 : 6;
  if (combo < 0) return MRGCN_ERR_UNSUPPORTED;
  int64_t amax, bmax;
  if (amode < 2) amax = (amode == 0 ? (int64_t)(M - 1) * lda + K : (int64_t)(K - 1) * lda + M);
  else amax = (int64_t)((amode == 2 ? M : K) / o.cg.Tout + 1) * o.cg.Cin * o.cg.Tin;
  if (bmode < 2) bmax = (bmode == 0 ? (int64_t)(K - 1) * ldb + N : (int64_t)(N - 1) * ldb + K);
  else bmax = (int64_t)(K / o.cg.Tout + 1) * o.cg.Cout * o.cg.Tout;
  if (amax * 4 > kMmMaxBytes || bmax * 4 > kMmMaxBytes || K < 1) return MRGCN_ERR_UNSUPPORTED;
  if (conv && (int64_t)std::max(M, K) * std::max(16 * o.cg.Tout, o.cg.KW) >= ((int64_t)1 << 32)) return MRGCN_ERR_UNSUPPORTED;
  if (!((M >= 48 && N >= 48) || K >= 1024)) return MRGCN_ERR_UNSUPPORTED;
  MmArgs g{};
  g.A = o.A; g.B = o.B; g.C = o.C;
  g.a_bytes = (uint32_t)(amax * 4); g.b_bytes = (uint32_t)(bmax * 4);
  g.M = M; g.N = N; g.K = K;
  g.lda = (int32_t)lda; g.ldb = (int32_t)ldb; g.ldc = ldc;
  g.bias = o.bias; g.relu = o.relu; g.mask = o.mask; g.alpha = o.alpha;
  g.cg = o.cg;
  auto magic = [](int64_t d) { return d <= 1 ? 0u : (uint32_t)((((uint64_t)1) << 32) / (uint64_t)d + 1); };
  g.mg_tout = magic(o.cg.Tout);
  g.mg_kw = magic(o.cg.KW);
  g.mg_tout16 = magic((int64_t)16 * o.cg.Tout);
  // Tile shape and K split.  A candidate's cost = the padded tile area over the shape's efficiency, times what it
  // pays for its grid: a grid of fewer than two blocks per CU either splits K (the partial tiles meet in a zeroed C
  // through float atomics: about 145 splits / K of the product's own time) or leaves CUs idle or with one wave per
  // SIMD, whose load latencies nothing covers.
  const int target = (int)cfg(CFG_MM_BLOCKS);
  const int force_tile = (int)cfg(CFG_MM_TILE);
  const bool dense_c = cmode == 2 || ldc == N;
  const bool can_split = dense_c && K >= 512 && !(cmode == 2 && (o.relu || o.mask));
  struct Cand { int bm, bn; double eff; int occ; };  // occ: blocks per CU (registers / LDS)
  const Cand cands[3] = {{128, 128, 1.0, 3}, {128, 64, 0.95, 5}, {64, 64, 0.85, 4}};
  int best = -1, best_splits = 1;
  double best_cost = 0;
  for (int c = 0; c < 3; ++c) {
    if (force_tile >= 0 && c != force_tile) continue;
    if (cands[c].bn == 128 && N <= 64) continue;
    const int64_t tm = (M + cands[c].bm - 1) / cands[c].bm, tn = (N + cands[c].bn - 1) / cands[c].bn, t = tm * tn;
    double cost = (double)tm * cands[c].bm * tn * cands[c].bn / cands[c].eff;
    int splits = 1;
    if (t < 512) {
      if (can_split) {
        splits = (int)std::max<int64_t>(1, std::min<int64_t>((target + t / 2) / t, K / 256));
        cost *= 1.0 + 145.0 * splits / K;
      }
      if (t * splits < 512) cost *= std::max(1.0, 256.0 / (double)(t * splits)) * 1.4;
    }
    // the last round of blocks leaves part of the chip idle; a grid that is resident all at once is as slow as
    // its fullest CU (the blocks of a CU share its matrix cores)
    const double per_round = 256.0 * cands[c].occ, blocks = (double)t * splits;
    if (blocks > per_round) cost *= std::ceil(blocks / per_round) * per_round / blocks;
    else if (blocks > 256.0) cost *= std::ceil(blocks / 256.0) * 256.0 / blocks;
    if (best < 0 || cost < best_cost) best = c, best_cost = cost, best_splits = splits;
  }
  if (best < 0) return MRGCN_ERR_UNSUPPORTED;
  const int BM = cands[best].bm, BN = cands[best].bn, splits = best_splits;
  dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + BM - 1) / BM), 1);
  if (splits > 1) {
    g.kchunk = ((K + splits - 1) / splits + 31) / 32 * 32;
    grid.z = (unsigned)((K + g.kchunk - 1) / g.kchunk);
    const int64_t celems = cmode == 2 ? (int64_t)(M / o.cg.Tout) * o.cg.Cout * o.cg.Tout : (int64_t)M * N;
    MRGCN_HIP_TRY(mrgcn::fill_async(o.C, 0, (size_t)celems * sizeof(float), stream));
  }
#define MM_GO1(AM_, BM_, CM_, AR_, BR_, MT_, NT_, BK_) \
  k_mm_tile<AM_, BM_, CM_, AR_, BR_, MT_, NT_, BK_><<<grid, dim3(256), 0, stream>>>(g)
#define MM_GOB(AM_, BM_, CM_, AR_, BR_, MT_, NT_) \
  k_mm_tile<AM_, BM_, CM_, AR_, BR_, MT_, NT_, 32, uint16_t><<<grid, dim3(256), 0, stream>>>(g)
#define MM_GO(AM_, BM_, CM_, AR_, BR_)                                     \
  do {                                                                     \
    if (bf16) { /* operands rounded to bf16 as they are staged, v_mfma_f32_16x16x32_bf16, fp32 sums */ \
      if (BM == 64) MM_GOB(AM_, BM_, CM_, AR_, BR_, 2, 2);                 \
      else if (BN == 64) MM_GOB(AM_, BM_, CM_, AR_, BR_, 4, 2);            \
      else MM_GOB(AM_, BM_, CM_, AR_, BR_, 4, 4);                          \
    } else if (BM == 64) MM_GO1(AM_, BM_, CM_, AR_, BR_, 2, 2, 32);        \
    else if (BN == 64) MM_GO1(AM_, BM_, CM_, AR_, BR_, 4, 2, 16);          \
    else MM_GO1(AM_, BM_, CM_, AR_, BR_, 4, 4, 16);                        \
  } while (0)
  switch (combo) {
    case 0: MM_GO(0, 0, 0, false, true); break;
    case 1: MM_GO(0, 1, 0, false, false); break;
    case 2: MM_GO(1, 0, 0, true, true); break;
    case 3: MM_GO(1, 1, 0, true, false); break;
    case 4: MM_GO(2, 1, 2, true, false); break;
    case 5: MM_GO(2, 0, 2, true, true); break;
    case 6: MM_GO(3, 2, 0, false, false); break;
    case 7: MM_GO(2, 1, 0, true, false); break;
    case 8: MM_GO(2, 0, 0, true, true); break;
    default: MM_GO(3, 2, 0, true, true); break;
  }
#undef MM_GO
#undef MM_GOB
#undef MM_GO1
  MRGCN_HIP_TRY(hipGetLastError());
  if (splits > 1 && (o.relu || o.mask)) {
    const int64_t total = (int64_t)M * N;
    k_mm_finish<<<dim3((unsigned)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, stream>>>(o.C, ldc, M, N,
                                                                                                     o.relu, o.mask);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  return MRGCN_OK;
}

}  // namespace
}  // namespace mrgcn

using namespace mrgcn;

extern "C" {

static int mlp_desc(MlpDesc &d, int32_t L, const int32_t *dims, const float *const *W, const float *const *b) {
  MRGCN_REQUIRE(L >= 1 && L <= kMlpL && dims && W && b, "1 <= layers <= 4");
  d.L = L;
  for (int l = 0; l <= L; ++l) {
    MRGCN_REQUIRE(dims[l] >= 1 && dims[l] <= kMlpW, "layer widths must be 1..16");
    d.dims[l] = dims[l];
  }
  for (int l = 0; l < kMlpL; ++l) {
    d.W[l] = l < L ? W[l] : nullptr;
    d.b[l] = l < L ? b[l] : nullptr;
    MRGCN_REQUIRE(l >= L || d.W[l], "NULL weight");
  }
  return MRGCN_OK;
}

int32_t mrgcn_mlp_fused_supported(int32_t L, const int32_t *dims) {
  if (L < 1 || L > kMlpL || !dims) return 0;
  for (int l = 0; l <= L; ++l)
    if (dims[l] < 1 || dims[l] > kMlpW) return 0;
  return 1;
}

int mrgcn_mlp_gate_scatter_fwd_f32(int32_t L, const int32_t *dims, const float *const *W, const float *const *b,
                                   const float *X, int64_t ldx, int64_t n, const float *gate, const int64_t *rows,
                                   float *XF, int64_t ldxf, int32_t offset, void *stream) {
  MlpDesc d;
  int rc = mlp_desc(d, L, dims, W, b);
  if (rc != MRGCN_OK) return rc;
  MRGCN_REQUIRE(X && gate && XF && ldx >= dims[0] && ldxf >= offset + dims[L] && offset >= 0, "operands");
  if (n == 0) return MRGCN_OK;
  size_t lds = 0;
  for (int l = 0; l < L; ++l) lds += (size_t)dims[l + 1] * (dims[l] + 1) * sizeof(float);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  k_mlp_gate_scatter_fwd<<<dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream>>>(d, X, ldx, n, gate, rows, XF,
                                                                                          ldxf, offset);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_mlp_gate_scatter_bwd_f32(int32_t L, const int32_t *dims, const float *const *W, const float *const *b,
                                   const float *X, int64_t ldx, int64_t n, const float *gate, const int64_t *rows,
                                   const float *dXF, int64_t ldxf, int32_t offset, float *const *dW,
                                   float *const *db, float *dgate, void *stream) {
  MlpDesc d;
  int rc = mlp_desc(d, L, dims, W, b);
  if (rc != MRGCN_OK) return rc;
  MRGCN_REQUIRE(X && gate && dXF && dW && db && dgate, "NULL");
  if (n == 0) return MRGCN_OK;
  MlpGrad gr{};
  for (int l = 0; l < L; ++l) {
    MRGCN_REQUIRE(dW[l], "NULL weight gradient");
    gr.dW[l] = dW[l];
    gr.db[l] = db[l];
  }
  size_t lds = 2 * 256 * (kMlpW + 1) * sizeof(float);
  for (int l = 0; l < L; ++l) lds += (size_t)dims[l + 1] * (dims[l] + 1) * sizeof(float);
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  k_mlp_gate_scatter_bwd<<<dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream>>>(
      d, X, ldx, n, gate, rows, dXF, ldxf, offset, gr, dgate);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_gemm_f32(int32_t amode, int32_t bmode, int32_t cmode, int32_t M, int32_t N, int32_t K, const float *A,
                   int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, const float *bias,
                   int32_t relu, const float *mask, float alpha, const int32_t *conv_geom, void *stream) {
  MRGCN_REQUIRE(A && B && C && M >= 0 && N >= 0 && K >= 0, "operands");
  MRGCN_REQUIRE(amode >= 0 && amode <= 3 && bmode >= 0 && bmode <= 2 && (cmode == 0 || cmode == 2), "modes");
  MRGCN_REQUIRE((amode < 2 && bmode < 2 && cmode == 0) || conv_geom, "conv modes need the geometry");
  if (M == 0 || N == 0) return MRGCN_OK;
  GemmArgs g{A, B, C, lda, ldb, ldc, M, N, K, amode, bmode, cmode, bias, relu, mask, alpha, ConvGeom{}};
  if (conv_geom) g.cg = ConvGeom{conv_geom[0], conv_geom[1], conv_geom[2], conv_geom[3], conv_geom[4], conv_geom[5]};
  // the tiled form takes every product with at least 48 rows and columns or a long reduction (operands of at most
  // 2^29 bytes); the heads' few output columns over a short reduction, larger operands and MRGCN_GEMM_TILED=0 take
  // the 64 x 64 kernel with element loaders
  const bool tiled_on = cfg(CFG_GEMM_TILED) != 0;
  if (tiled_on) {
    const int rc = mm_tile_launch(g, (hipStream_t)stream);
    if (rc != MRGCN_ERR_UNSUPPORTED) return rc;
  }
  dim3 grid((unsigned)((N + kGT - 1) / kGT), (unsigned)((M + kGT - 1) / kGT));
  k_gemm_f32<<<grid, dim3(256), 0, (hipStream_t)stream>>>(g);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

// The same product with bf16 matrix-core arithmetic (the bf16 pipeline of BASELINE config 3): operands and result stay
// fp32 in memory, the operands are rounded to bf16 (nearest even) as tiles are staged, v_mfma_f32_16x16x32_bf16 with
// fp32 accumulation.  Shapes outside the tiled form's limits take the fp32 element-loader kernel (exact fp32).
int mrgcn_gemm_bf16mm_f32(int32_t amode, int32_t bmode, int32_t cmode, int32_t M, int32_t N, int32_t K, const float *A,
                          int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc, const float *bias,
                          int32_t relu, const float *mask, float alpha, const int32_t *conv_geom, void *stream) {
  MRGCN_REQUIRE(A && B && C && M >= 0 && N >= 0 && K >= 0, "operands");
  MRGCN_REQUIRE(amode >= 0 && amode <= 3 && bmode >= 0 && bmode <= 2 && (cmode == 0 || cmode == 2), "modes");
  MRGCN_REQUIRE((amode < 2 && bmode < 2 && cmode == 0) || conv_geom, "conv modes need the geometry");
  if (M == 0 || N == 0) return MRGCN_OK;
  GemmArgs g{A, B, C, lda, ldb, ldc, M, N, K, amode, bmode, cmode, bias, relu, mask, alpha, ConvGeom{}};
  if (conv_geom) g.cg = ConvGeom{conv_geom[0], conv_geom[1], conv_geom[2], conv_geom[3], conv_geom[4], conv_geom[5]};
  if (cfg(CFG_GEMM_TILED) != 0) {
    const int rc = mm_tile_launch(g, (hipStream_t)stream, true);
    if (rc != MRGCN_ERR_UNSUPPORTED) return rc;
  }
  dim3 grid((unsigned)((N + kGT - 1) / kGT), (unsigned)((M + kGT - 1) / kGT));
  k_gemm_f32<<<grid, dim3(256), 0, (hipStream_t)stream>>>(g);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int mrgcn_colsum_f32(const float *X, int64_t ld, int32_t M, int32_t N, float *out, void *stream) {
  MRGCN_REQUIRE(X && out && ld >= N, "operands");
  if (N == 0) return MRGCN_OK;
  const unsigned cols = (unsigned)((N + 63) / 64);
  unsigned slabs = M >= 512 ? std::min<unsigned>((unsigned)(M / 128), std::max(1u, 512u / cols)) : 1u;
  if (slabs > 1) MRGCN_HIP_TRY(mrgcn::fill_async(out, 0, (size_t)N * sizeof(float), (hipStream_t)stream));
  k_colsum_f32<<<dim3(cols, slabs), dim3(256), 0, (hipStream_t)stream>>>(X, ld, M, N, out);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // extern "C"
