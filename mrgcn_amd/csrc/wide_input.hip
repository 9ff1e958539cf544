// Backward of a WIDE featureless input layer with few bases (graph.py:66-81 and its autograd; the link-prediction
// encoder of configs/fb15k-237.toml: N x 200, 2 bases) without the compact gradient operand dM.
//
// The general path forms dM = A'^T dY (one 800-byte row per touched column: 260 MB at the FB15k-237 shape), then reads
// it twice more (dV, dcomp): 780 MB of HBM traffic and five launches for 23 MB of result.  Per ENTRY e = (row i, column
// (j, r), value a) of A the two gradients are
//     dV[j][b][:]  += comp[r][b] * a * dY[i][:]
//     dcomp[r][b]  += a * <dY[i][:], V[j][b][:]>
// and the entries of a source node j are contiguous in the plan's CSC order (columns are numbered by (j, r)).  So: a
// wave takes a node (or a 256-entry piece of a hub node), keeps the node's B rows of V and its dV accumulators in
// registers (lane = four features), gathers the dY rows of its entries eight at a time from the L2-resident dY
// (N x F: 11.6 MB), sums the dcomp products of a run of equal relations in registers and reduces a run once.
// dV leaves with one store per node (float atomics only for the pieces of hub nodes), dcomp through per-block LDS
// accumulators.  Nothing of size (columns x F) is ever written.
#include <cstdlib>

#include "common.hpp"
#include "config.hpp"

namespace mrgcn {
namespace {

using f32x4w = __attribute__((ext_vector_type(4))) float;

// erel[e] = relation of CSC entry e (entries of compact column c are cptr[c] .. cptr[c+1])
__global__ void k_entry_relations(const int32_t *__restrict__ cptr, const int32_t *__restrict__ urel, int64_t ncols,
                                  int32_t *__restrict__ erel) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  const int32_t r = urel[c];
  for (int32_t e = cptr[c]; e < cptr[c + 1]; ++e) erel[e] = r;
}

__device__ __forceinline__ float wave_sum4(f32x4w v) {
  float x = (v.x + v.y) + (v.z + v.w);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

template <int BT>
__global__ __launch_bounds__(256) void k_wide_input_bwd(const int32_t *__restrict__ unit_node,
                                                        const int32_t *__restrict__ unit_beg,
                                                        const int32_t *__restrict__ unit_end,
                                                        const uint8_t *__restrict__ unit_multi, int64_t n_units,
                                                        const int32_t *__restrict__ crow,
                                                        const float *__restrict__ cval,
                                                        const int32_t *__restrict__ erel,
                                                        const float *__restrict__ dY, int64_t ldY,
                                                        const float *__restrict__ V, const float *__restrict__ comp,
                                                        int R, int B, int F, float *__restrict__ dV,
                                                        float *__restrict__ dcomp) {
  extern __shared__ __align__(16) float s_mem[];  // comp [R][B] | dcomp accumulators [R][B]
  float *s_comp = s_mem, *s_dc = s_mem + R * B;
  for (int t = threadIdx.x; t < R * B; t += blockDim.x) {
    s_comp[t] = comp[t];
    s_dc[t] = 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int f0 = 4 * lane;
  const bool active = f0 < F;
  const int64_t ld4 = ldY >> 2;
  const f32x4w *dY4 = reinterpret_cast<const f32x4w *>(dY) + (active ? lane : 0);
  const f32x4w zero = {0.f, 0.f, 0.f, 0.f};
  for (int64_t u = (int64_t)blockIdx.x * 4 + wv; u < n_units; u += (int64_t)gridDim.x * 4) {
    const int64_t j = unit_node[u];
    const int32_t e0 = unit_beg[u], e1 = unit_end[u];
    f32x4w Vb[BT], accV[BT], part[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      Vb[b] = (active && b < B) ? *reinterpret_cast<const f32x4w *>(V + (j * B + b) * F + f0) : zero;
      accV[b] = zero;
      part[b] = zero;
    }
    int cur_r = -1;
    auto flush = [&]() {
      if (cur_r < 0) return;
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        if (b < B) {
          const float s = wave_sum4(part[b]);
          if (lane == 0 && s != 0.f) atomicAdd(&s_dc[cur_r * B + b], s);
        }
        part[b] = zero;
      }
    };
    for (int32_t eb = e0; eb < e1; eb += 64) {
      const int32_t me = (eb + lane < e1) ? eb + lane : e1 - 1;
      const int32_t mrow = crow[me], mrel = erel[me];
      const float mval = cval[me];
      const int cnt = (e1 - eb < 64) ? e1 - eb : 64;
      for (int t0 = 0; t0 < cnt; t0 += 8) {
        f32x4w x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {  // eight row gathers in flight (clamped: the tail repeats the last entry)
          const int tt = (t0 + k < cnt) ? t0 + k : cnt - 1;
          const int32_t row = __builtin_amdgcn_readlane(mrow, tt);
          x[k] = dY4[(int64_t)row * ld4];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (t0 + k < cnt) {  // wave uniform
            const int r = __builtin_amdgcn_readlane(mrel, t0 + k);
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mval), t0 + k));
            if (r != cur_r) {
              flush();
              cur_r = r;
            }
            const f32x4w d = active ? x[k] * a : zero;
#pragma unroll
            for (int b = 0; b < BT; ++b) {
              if (b < B) {
                accV[b] += d * s_comp[r * B + b];
                part[b] += d * Vb[b];
              }
            }
          }
        }
      }
    }
    flush();
    if (active) {
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        if (b < B) {
          float *o = dV + (j * B + b) * F + f0;
          if (unit_multi[u]) {
            atomicAdd(o + 0, accV[b].x);
            atomicAdd(o + 1, accV[b].y);
            atomicAdd(o + 2, accV[b].z);
            atomicAdd(o + 3, accV[b].w);
          } else {
            *reinterpret_cast<f32x4w *>(o) = accV[b];
          }
        }
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < R * B; t += blockDim.x) {
    const float x = s_dc[t];
    if (x != 0.f) atomicAdd(&dcomp[t], x);
  }
}

}  // namespace
}  // namespace mrgcn

extern "C" {

using namespace mrgcn;

int mrgcn_plan_entry_relations(const mrgcn_plan_t *p, int32_t *erel, void *stream) {
  MRGCN_REQUIRE(p && erel, "NULL");
  if (p->ncols == 0) return MRGCN_OK;
  k_entry_relations<<<dim3((unsigned)((p->ncols + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(p->cptr, p->urel,
                                                                                                   p->ncols, erel);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int32_t mrgcn_wide_input_bwd_supported(const mrgcn_plan_t *p, int32_t B, int32_t F) {
  const bool on = cfg(CFG_WIDE_BWD) != 0;
  return (on && p && B >= 1 && B <= 4 && F > 16 && F <= 256 && F % 4 == 0 &&
          (size_t)2 * p->num_relations * B * sizeof(float) <= 64 * 1024) ? 1 : 0;
}

int mrgcn_wide_input_bwd_f32(const mrgcn_plan_t *p, const int32_t *erel, const int32_t *unit_node,
                             const int32_t *unit_beg, const int32_t *unit_end, const uint8_t *unit_multi,
                             int64_t n_units, const float *dY, int64_t ldY, const float *V, const float *comp,
                             int32_t B, int32_t F, float *dV, float *dcomp, void *stream) {
  MRGCN_REQUIRE(p && erel && unit_node && unit_beg && unit_end && unit_multi && dY && V && comp && dV && dcomp, "NULL");
  MRGCN_REQUIRE(mrgcn_wide_input_bwd_supported(p, B, F), "shape outside mrgcn_wide_input_bwd_supported");
  MRGCN_REQUIRE(ldY % 4 == 0 && ldY >= F && (((uintptr_t)dY | (uintptr_t)V | (uintptr_t)dV) & 15) == 0,
                "dY / V / dV must be 16-byte aligned with rows of whole 16-byte pieces");
  hipStream_t s = (hipStream_t)stream;
  const int R = (int)p->num_relations;
  // every block of dV is written by exactly one unit (plain store) or by the pieces of a hub node (atomics): zero
  // first; dcomp accumulates per block
  MRGCN_HIP_TRY(mrgcn::fill_async(dV, 0, (size_t)p->num_nodes * B * F * sizeof(float), s));
  MRGCN_HIP_TRY(mrgcn::fill_async(dcomp, 0, (size_t)R * B * sizeof(float), s));
  if (n_units == 0) return MRGCN_OK;
  const size_t lds = (size_t)2 * R * B * sizeof(float);
  int64_t grid = (n_units + 3) / 4;
  if (grid > 256 * 8) grid = 256 * 8;
#define WIDE_GO(BT_)                                                                                            \
  k_wide_input_bwd<BT_><<<dim3((unsigned)grid), dim3(256), lds, s>>>(unit_node, unit_beg, unit_end, unit_multi, \
                                                                       n_units, p->crow, p->cval, erel, dY, ldY, \
                                                                       V, comp, R, B, F, dV, dcomp)
  if (B == 1) WIDE_GO(1);
  else if (B == 2) WIDE_GO(2);
  else WIDE_GO(4);
#undef WIDE_GO
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // extern "C"
