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

// DET (round 6): bitwise reproducible, and units of 64 entries (hub nodes stop being the tail of the launch: the
// FB15k-237 epoch 1.24 -> 1.18 ms; with float atomics those small pieces reordered cancelling hub sums enough to leave
// the step oracle's interval).  A unit of a node that has SEVERAL units writes its B x F partial sums to its own slot
// of `hub_part` and k_wide_hub_sum adds a node's slots in unit order; dcomp is summed per WAVE in LDS (one writer: no
// LDS atomics), per block in wave order, and k_wide_dcomp_final adds the blocks' sums in block order.  No float atomic
// anywhere: the same bits every run.
template <int BT, bool DET = false>
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
                                                        float *__restrict__ dcomp,
                                                        const int32_t *__restrict__ unit_slot = nullptr,
                                                        float *__restrict__ hub_part = nullptr,
                                                        float *__restrict__ dc_part = nullptr) {
  extern __shared__ __align__(16) float s_mem[];  // comp [R][B] | dcomp accumulators [R][B] (DET: one set per wave)
  float *s_comp = s_mem, *s_dc = s_mem + R * B;
  const int n_dc = DET ? 4 * R * B : R * B;
  for (int t = threadIdx.x; t < R * B; t += blockDim.x) s_comp[t] = comp[t];
  for (int t = threadIdx.x; t < n_dc; t += blockDim.x) s_dc[t] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float *my_dc = DET ? s_dc + wv * R * B : s_dc;
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
          if (DET) {
            if (lane == 0) my_dc[cur_r * B + b] += s;   // (this wave's own accumulators: one writer)
          } else if (lane == 0 && s != 0.f) {
            atomicAdd(&s_dc[cur_r * B + b], s);
          }
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
      const int32_t slot = DET ? unit_slot[u] : -1;
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        if (b < B) {
          float *o = dV + (j * B + b) * F + f0;
          if (DET) {
            if (slot >= 0) *reinterpret_cast<f32x4w *>(hub_part + ((int64_t)slot * B + b) * F + f0) = accV[b];
            else *reinterpret_cast<f32x4w *>(o) = accV[b];
          } else if (unit_multi[u]) {
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
  if (DET) {  // the block's sum, waves in order; the blocks' sums are added by k_wide_dcomp_final
    // (stored [R B][blocks]: the final pass reads a sum's terms as one coalesced run)
    for (int t = threadIdx.x; t < R * B; t += blockDim.x)
      dc_part[(int64_t)t * gridDim.x + blockIdx.x] =
          ((s_dc[t] + s_dc[R * B + t]) + s_dc[2 * R * B + t]) + s_dc[3 * R * B + t];
    return;
  }
  for (int t = threadIdx.x; t < R * B; t += blockDim.x) {
    const float x = s_dc[t];
    if (x != 0.f) atomicAdd(&dcomp[t], x);
  }
}

// dV[node] = sum of the node's unit slots in a fixed order (one wave per (hub node, basis), lane = four features):
// eight slots in flight — slot s0 + 8 i + k goes to accumulator k, the eight are added in order at the end
__global__ __launch_bounds__(256) void k_wide_hub_sum(const int32_t *__restrict__ hub_node, const int32_t *__restrict__ hub_ptr,
                                                      int64_t n_hubs, const float *__restrict__ hub_part, int B, int F,
                                                      float *__restrict__ dV) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t h = w / B;
  const int b = (int)(w - h * B);
  if (h >= n_hubs || 4 * lane >= F) return;
  const int64_t j = hub_node[h];
  const int32_t s0 = hub_ptr[h], s1 = hub_ptr[h + 1];
  const f32x4w zero = {0.f, 0.f, 0.f, 0.f};
  f32x4w acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = zero;
  for (int32_t sl = s0; sl < s1; sl += 8) {
    f32x4w x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int32_t q = min(sl + k, s1 - 1);
      x[k] = *reinterpret_cast<const f32x4w *>(hub_part + ((int64_t)q * B + b) * F + 4 * lane);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (sl + k < s1) acc[k] += x[k];
  }
  const f32x4w t = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  *reinterpret_cast<f32x4w *>(dV + (j * B + b) * F + 4 * lane) = t;
}

// dcomp[t] = sum over the blocks' partial sums in a FIXED order: one wave per element, lane l adds blocks l, l + 64, ...
// in rising order, then the lanes' sums meet in a fixed butterfly
__global__ __launch_bounds__(256) void k_wide_dcomp_final(const float *__restrict__ dc_part, int n_blocks, int RB,
                                                          float *__restrict__ dcomp) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= RB) return;
  const float *src = dc_part + (int64_t)t * n_blocks;
  float s = 0.f;
  for (int b = lane; b < n_blocks; b += 64) s += src[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) dcomp[t] = s;
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

// The same, bitwise reproducible (no float atomics): `unit_slot[u]` = the unit's slot in the hub workspace (-1: the
// only unit of its node, stored straight into dV), `hub_node / hub_ptr` = the nodes with several units and their slot
// ranges, `workspace`: mrgcn_wide_input_bwd_det_workspace(plan, n_slots, B, F) floats.
int64_t mrgcn_wide_input_bwd_det_workspace(const mrgcn_plan_t *p, int64_t n_slots, int32_t B, int32_t F) {
  if (!p) return -1;
  return n_slots * B * F + (int64_t)256 * 8 * p->num_relations * B;
}

int mrgcn_wide_input_bwd_det_f32(const mrgcn_plan_t *p, const int32_t *erel, const int32_t *unit_node,
                                 const int32_t *unit_beg, const int32_t *unit_end, const int32_t *unit_slot,
                                 int64_t n_units, const int32_t *hub_node, const int32_t *hub_ptr, int64_t n_hubs,
                                 int64_t n_slots, const float *dY, int64_t ldY, const float *V, const float *comp,
                                 int32_t B, int32_t F, float *dV, float *dcomp, float *workspace,
                                 int64_t workspace_floats, void *stream) {
  MRGCN_REQUIRE(p && erel && unit_node && unit_beg && unit_end && unit_slot && dY && V && comp && dV && dcomp && workspace,
                "NULL");
  MRGCN_REQUIRE(n_hubs == 0 || (hub_node && hub_ptr), "hub arrays");
  MRGCN_REQUIRE(mrgcn_wide_input_bwd_supported(p, B, F), "shape outside mrgcn_wide_input_bwd_supported");
  MRGCN_REQUIRE(ldY % 4 == 0 && ldY >= F && (((uintptr_t)dY | (uintptr_t)V | (uintptr_t)dV | (uintptr_t)workspace) & 15) == 0,
                "dY / V / dV / workspace must be 16-byte aligned with rows of whole 16-byte pieces");
  MRGCN_REQUIRE(workspace_floats >= mrgcn_wide_input_bwd_det_workspace(p, n_slots, B, F), "workspace");
  hipStream_t s = (hipStream_t)stream;
  const int R = (int)p->num_relations;
  // nodes without an entry keep zeros; every other block of dV is written once (by its unit or by k_wide_hub_sum)
  MRGCN_HIP_TRY(mrgcn::fill_async(dV, 0, (size_t)p->num_nodes * B * F * sizeof(float), s));
  if (n_units == 0) {
    MRGCN_HIP_TRY(mrgcn::fill_async(dcomp, 0, (size_t)R * B * sizeof(float), s));
    return MRGCN_OK;
  }
  float *hub_part = workspace, *dc_part = workspace + n_slots * B * F;
  const size_t lds = (size_t)5 * R * B * sizeof(float);
  int64_t grid = (n_units + 3) / 4;
  if (grid > 256 * 8) grid = 256 * 8;
#define WIDE_DET(BT_)                                                                                              \
  do {                                                                                                             \
    auto kfn = k_wide_input_bwd<BT_, true>;                                                                        \
    MRGCN_HIP_TRY(mrgcn::raise_lds_limit((const void *)kfn, lds));                                                 \
    kfn<<<dim3((unsigned)grid), dim3(256), lds, s>>>(unit_node, unit_beg, unit_end, nullptr, n_units, p->crow,     \
                                                     p->cval, erel, dY, ldY, V, comp, R, B, F, dV, dcomp,          \
                                                     unit_slot, hub_part, dc_part);                                \
  } while (0)
  if (B == 1) WIDE_DET(1);
  else if (B == 2) WIDE_DET(2);
  else WIDE_DET(4);
#undef WIDE_DET
  MRGCN_HIP_TRY(hipGetLastError());
  if (n_hubs > 0) {
    k_wide_hub_sum<<<dim3((unsigned)((n_hubs * B + 3) / 4)), dim3(256), 0, s>>>(hub_node, hub_ptr, n_hubs, hub_part, B, F, dV);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  k_wide_dcomp_final<<<dim3((unsigned)((R * B + 3) / 4)), dim3(256), 0, s>>>(dc_part, (int)grid, R * B, dcomp);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // extern "C"
