// EXPERIMENT RECORD (round 5) — NOT part of the library, not compiled by mrgcn_amd/build.py.
//
// A band-major, XCD-affine, persistent form of the layer-0 transform, built to cut the 3.9x gather redundancy of
// k_xform_mfma_fwd (X rows re-fetched through the fabric once per relation of their node).  It was correct (bit-equal to
// the relation-major kernel: tools/lab/xform_band_experiment_test.py) and did NOT pay.  Measured on the AM shape:
//
//   relation-major kernel (product)                         1 078 us   5.33 GB of 128-byte fabric reads (41.7 M)
//   old kernel, chunks of 4 096-node bands dealt to XCDs    1 578 us   3.0 GB   (a workgroup + two barriers per tiny chunk)
//   this file, fragment-shaped loads, 2 waves / SIMD        1 158 us   3.39 GB  TA busy 76 % (1 114 L1 accesses per tile)
//   this file as below (row-contiguous loads via LDS)       1 016 us   4.63 GB  at 12 waves / CU
//                                                           1 050 us   2.9 GB   at  8 waves / CU
//                                                           1 445 us   1.85 GB  at  4 waves / CU
//   ... without its stores 835 us, with one weight tile for all relations 937 us.
//
// What bounds it: an XCD's L2 is 4 MB and the bandwidth-delay product of its 32 CUs (72-96 KB in flight each) is
// 2.3-3 MB.  A band small enough to stay resident next to the weight tiles (1.7 MB for 267 relations) and the tiles in
// flight leaves so few tiles per (band, relation) group that the per-tile costs take over; with enough waves in flight to
// reach the fabric's rate the rows are evicted before the node's other relations come round (L2 hit rate 38 %).  Fewer
// waves re-use the rows (1.85 GB = 1.5x the compulsory bytes) and are latency bound.  Every arrangement lands at
// ~1.0 ms: the box moves ~4.7-4.9 TB/s through the fabric whatever is asked of it.
//
// (Needs the BandTiles builder that went with it: plan.hip `build_band_tiles` of the same commit, in git history.)
//
// Band-major per-relation transform of WIDE input rows (layer 0: X rows of 620 bytes at the AM shape) on the matrix
// cores:
//
//   Out[o(c), n] = sum_k In[node(c), k] * W[r_c][k][n]        (graph.py:93-94)      c = compact column (node, relation)
//
// The relation-major kernel (xform_mfma.hip) walks bands of 131 072 nodes: a band of X is 81 MB, every relation of the
// band gathers its rows again through the fabric (AM: 5.3 GB of 128-byte requests for a 1.03 GB X, 3.9x; the kernel
// sat at the rate the Infinity Cache delivers random rows).  Here the columns are cut into TILES of <= 16 columns of one
// (band, relation) group with bands of kTileBand = 4 096 nodes (common.hpp: BandTiles) — 2.5 MB of X, inside one XCD's
// 4 MB L2 — and the launch is persistent and XCD-affine: workgroup b runs on XCD b mod 8 (MI355X_MICROARCH.md, dispatch:
// for speed only), XCD x walks the bands x, x + 8, ... one after the other, its workgroups' waves take the band's
// tiles round robin.  An X row is fetched from HBM once, by the identity relation's tiles that sweep the band first,
// and re-read from that L2 by the node's other relations (measured with the old kernel's chunks dealt to XCDs this way:
// 5.33 -> 3.0 GB of fabric reads, but 1.58 ms instead of 1.08: a 256-thread workgroup per (band, relation) chunk
// that stages its weight tile in LDS behind two barriers for a handful of columns).
//
// So there is no workgroup barrier here: a wave multiplies one tile with `v_mfma_f32_16x16x4_f32` (exact fp32; the
// slot <-> k assignment of xform_mfma.hip: the same bits), its operands staged through its own LDS tile —
//   A: 16 rows x K of In, 16-byte pieces;
//   B: the relation's weight tile TRANSPOSED to [F][KP] (k contiguous, zero padded: a small launch in front writes
//      it into the stream's product scratch), L1 / L2 resident — every wave of the XCD reads the same few relations at
//      a time;
// the tile's indices (source node, output row) come one tile ahead, the tile descriptors two.
#include "common.hpp"

namespace mrgcn {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 band_load4_guarded(const float *row, int k, int K) {
  f32x4 v;
  v.x = (k + 0 < K) ? row[k + 0] : 0.f;
  v.y = (k + 1 < K) ? row[k + 1] : 0.f;
  v.z = (k + 2 < K) ? row[k + 2] : 0.f;
  v.w = (k + 3 < K) ? row[k + 3] : 0.f;
  return v;
}

// Wt[(r * F + n) * KP + k] = W[(r * K + k) * F + n], zeros for K <= k < KP
__global__ void k_band_wt(const float *__restrict__ W, int64_t R, int K, int F, int KP, float *__restrict__ Wt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * F * KP) return;
  const int k = (int)(i % KP);
  const int64_t rn = i / KP;
  const int n = (int)(rn % F);
  const int64_t r = rn / F;
  Wt[i] = k < K ? W[(r * K + k) * F + n] : 0.f;
}

// KS = ceil(K / 16) K steps; F <= 16; ldOut <= 16.  Operands go global -> registers -> the wave's own LDS tile -> MFMA
// layout.  The loads use a ROW-CONTIGUOUS lane map (lane = 4 * row + 16-byte segment): the matrix cores want lane
// l = (row l & 15, k slot l >> 4), and with that map on the global loads every 16 consecutive lanes touch 16 different
// rows — the texture path looks up one cache block per lane group and 16-byte piece: 1 114 L1 accesses per tile
// measured (TA busy 76-82 %, MFMA 26 %, the kernel 1.16-1.43 ms).  Row-contiguous, 16 lanes cover 4 rows x 64 bytes.
// The B tile (the relation's transposed weights) stays in LDS while the wave's tiles keep their relation (a wave takes
// RUN consecutive tiles: the identity block has 256 per band, the other hot relations dozens).
// Per tile: [next tile's A rows in flight under this tile's MFMA chain] -> A to LDS -> MFMAs fed by ds_read_b128.
constexpr int kBandRun = 4;
#ifndef MRGCN_BAND_SC1
#define MRGCN_BAND_SC1 1
#endif
constexpr bool kBandSc1 = MRGCN_BAND_SC1 != 0;
template <int KS, typename OT>
__global__ __launch_bounds__(256, 4) void k_xform_band_fwd(
    const int32_t *__restrict__ tile_beg, const int32_t *__restrict__ tile_rc, const int32_t *__restrict__ band_tptr,
    const int32_t *__restrict__ tnode, const int32_t *__restrict__ tout, int32_t n_bands, int32_t n_tiles,
    const float *__restrict__ In, int64_t ldIn, int K, const float *__restrict__ Wt, int F, OT *__restrict__ Out,
    int64_t ldOut, int dbg) {
  extern __shared__ __align__(16) float s_tiles[];  // per wave: [KS][16 rows][16 floats], 16-byte segments swizzled
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int m = lane & 15, kq = lane >> 4;      // MFMA map: row / column m, k slot kq
  const int lrow = lane >> 2, lseg = lane & 3;  // load map: row lrow, 16-byte segment lseg of a 64-byte K step
  constexpr int KP = KS * 16;
  const int64_t FKP = (int64_t)F * KP;
  f32x4 *sA = reinterpret_cast<f32x4 *>(s_tiles + wv * (KS * 256));
  // segment s of row r sits at slot s ^ g(r >> 2), g = {0, 2, 3, 1}: every 16-lane group of a ds_read_b128
  // ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) then covers the 64 banks once; the ds_write_b128 groups
  // (8 consecutive lanes = two whole rows) do anyway
  const int wr_at = lrow * 4 + (lseg ^ ((0x78 >> (2 * (lrow >> 2))) & 3));
  const int rd_at = m * 4 + (kq ^ ((0x78 >> (2 * (m >> 2))) & 3));
  // The wave's tile sequence: band xcd, xcd + 8, ...; the bands' runs of kBandRun tiles are dealt to the XCD's waves
  // round robin ACROSS bands (a band of 4 096 nodes has ~350 runs for 384 waves: dealt from wave 0 in every band, the
  // last waves never got a tile, and with smaller bands most of them never did).
  const int nwx = bpx * 4, wx = slot * 4 + wv;   // waves of this XCD, this wave's number
  int32_t band = xcd, t = 0, t_end = 0, run_left = 0, run_local = 0, nb_band = 0, a_band = 0, rbase = 0;
  auto next_tile = [&]() {  // advances to the wave's next tile; t = -1 when there is none
    if (run_left > 0 && t + 1 < t_end) { --run_left; ++t; return; }
    run_local += nwx;
    while (run_local >= nb_band) {
      if (band >= n_bands) { t = -1; return; }
      const int32_t a = band_tptr[band], b = band_tptr[band + 1];
      band += 8;
      nb_band = (b - a + kBandRun - 1) / kBandRun;
      run_local = wx - rbase;            // the first run of this band that falls to this wave
      if (run_local < 0) run_local += nwx;
      rbase = (rbase + nb_band) % nwx;
      a_band = a;
      t_end = b;
    }
    t = a_band + run_local * kBandRun;
    run_left = kBandRun - 1;
  };
  next_tile();
  if (t < 0) return;
  auto load_a = [&](int32_t node_l, f32x4 *a) {  // the rows of a tile, row-contiguous lane map
    const float *xrow = In + (int64_t)node_l * ldIn;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 16 + 4 * lseg;
      a[ks] = (k + 4 <= K) ? *reinterpret_cast<const f32x4 *>(xrow + k) : band_load4_guarded(xrow, k, K);
    }
  };
  // The index pipeline: tile descriptors three tiles ahead, a tile's indices (source node, output row; lanes of row
  // lrow) two ahead, its A rows one ahead — every load of an iteration has its address in registers when the iteration
  // starts (with the indices only one tile ahead the A rows waited for them: ~3.5 us per tile and wave).
  // Slots: 0 = the tile being multiplied, 1 = next (A rows in flight), 2 = indices in flight, 3 = descriptors in flight.
  const int32_t t0 = t;
  next_tile();
  const int32_t t1 = t;
  if (t1 >= 0) next_tile();
  const int32_t t2 = t1 >= 0 ? t : -1;
  if (t2 >= 0) next_tile();
  int32_t t3 = t2 >= 0 ? t : -1;
  auto desc = [&](int32_t tt, int32_t &beg, int32_t &rc) {
    beg = tile_beg[max(tt, 0)];
    rc = tile_rc[max(tt, 0)];
  };
  auto ids = [&](int32_t beg, int32_t rc, int32_t &nd, int32_t &orw) {
    const int32_t ee = beg + min(lrow, rc & 31);
    nd = tnode[ee];
    orw = tout[ee];
  };
  int32_t beg0, rc0, beg1, rc1, beg2, rc2, beg3, rc3;
  desc(t0, beg0, rc0);
  desc(t1, beg1, rc1);
  desc(t2, beg2, rc2);
  desc(t3, beg3, rc3);
  int32_t node0, orow0, node1, orow1, node2, orow2;
  ids(beg0, rc0, node0, orow0);
  ids(beg1, rc1, node1, orow1);
  ids(beg2, rc2, node2, orow2);
  bool v1 = t1 >= 0, v2 = t2 >= 0, v3 = t3 >= 0;
  f32x4 an[KS];
  load_a(node0, an);
  int32_t rel_lds = -1;
  f32x4 bv[KS];  // the relation's weight tile in MFMA layout, kept while the wave's tiles keep their relation
  for (;;) {
    const int32_t cnt = (dbg & 2) ? 0 : (rc0 & 31) + 1, rel = (dbg & 1) ? 0 : rc0 >> 5;
    if (rel != rel_lds) {  // wave uniform: a new relation's weight tile, through the LDS tile into MFMA layout
      const int n = lrow < F ? lrow : F - 1;
      const f32x4 *wrel = reinterpret_cast<const f32x4 *>(Wt + (int64_t)rel * FKP + n * KP + 4 * lseg);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bv[ks] = wrel[ks * 4];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) sA[ks * 64 + wr_at] = lrow < F ? bv[ks] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bv[ks] = sA[ks * 64 + rd_at];
      rel_lds = rel;
    }
    // ---- this tile's A rows: registers -> LDS (the previous reads of the tile have been consumed) ------------------
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) sA[ks * 64 + wr_at] = an[ks];
    // ---- in flight under the products: the next tile's A rows, the indices of the one after, descriptors beyond ---
    load_a(v1 ? node1 : node0, an);  // (the last tile re-reads its own rows: unconditional, straight-line loads)
    int32_t node3, orow3;
    ids(beg3, rc3, node3, orow3);
    if (v3) next_tile();
    const int32_t t4 = v3 ? t : -1;
    int32_t beg4, rc4;
    desc(t4, beg4, rc4);
    // ---- products ---------------------------------------------------------------------------------------------------
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 av = sA[ks * 64 + rd_at], bw = bv[ks];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bw.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bw.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bw.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bw.w, acc, 0, 0, 0);
    }
    // D: lane (n = m, g = kq) holds columns 4g + reg of the tile, feature n; column c's ids sit in lanes 4c .. 4c + 3
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int c = 4 * kq + reg;
      const int64_t row = __shfl(orow0, 4 * c, 64);
      if (c < cnt && m < ldOut) {  // zeros past F: the whole padded row
        if constexpr (sizeof(OT) == 4) {
          // written once, read by another kernel: `sc1` stores leave the XCD's L2 to the band's input rows
          if (kBandSc1) __hip_atomic_store(reinterpret_cast<float *>(Out) + row * ldOut + m, acc[reg], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
          else store_operand<OT>(Out + row * ldOut + m, acc[reg]);
        } else {
          store_operand<OT>(Out + row * ldOut + m, acc[reg]);
        }
      }
    }
    if (!v1) break;
    rc0 = rc1; node0 = node1; orow0 = orow1;
    rc1 = rc2; node1 = node2; orow1 = orow2; v1 = v2;
    rc2 = rc3; node2 = node3; orow2 = orow3; v2 = v3;
    beg3 = beg4; rc3 = rc4; v3 = t4 >= 0;
  }
}

}  // namespace

bool xform_band_fwd_supported(const mrgcn_plan *p, int K, int F, int64_t ldOut) {
  static const bool on = !(getenv("MRGCN_XFORM_BAND") && atoi(getenv("MRGCN_XFORM_BAND")) == 0);
  return on && p && p->tiles.n_tiles > 0 && K > kNarrowInput && K <= 256 && F <= 16 && ldOut <= 16 &&
         p->num_relations < (1 << 26);
}

int xform_band_fwd(const mrgcn_plan *p, bool operand_order, const float *In, int64_t ldIn, int K, const float *W, int F,
                   void *Out, int64_t ldOut, hipStream_t s, bool out_bf16) {
  const BandTiles &bt = p->tiles;
  if (bt.n_tiles == 0) return MRGCN_OK;
  const int32_t *tout = operand_order ? bt.tmpos : bt.tcol;
  static const int bpx_env = getenv("MRGCN_XFORM_BAND_BPX") ? atoi(getenv("MRGCN_XFORM_BAND_BPX")) : 0;
  const int bpx = bpx_env > 0 ? bpx_env : 32 * 3;   // workgroups per XCD: 32 CUs x 3 (LDS: 10 KB per wave at K = 155)
  const dim3 grid((unsigned)(8 * bpx)), block(256);
  static const int dbg = getenv("MRGCN_XFORM_BAND_DBG") ? atoi(getenv("MRGCN_XFORM_BAND_DBG")) : 0;
  const int ksteps = (K + 15) / 16;
  float *wt = nullptr;
  int32_t *ticket = nullptr;
  int rc = plan_scratch(p, s, &wt, &ticket);
  if (rc != MRGCN_OK) return rc;
#define XB_GO(KS_)                                                                                                   \
  do {                                                                                                               \
    const int64_t nwt = (int64_t)p->num_relations * F * (KS_ * 16);                                                  \
    k_band_wt<<<dim3((unsigned)((nwt + 255) / 256)), dim3(256), 0, s>>>(W, p->num_relations, K, F, KS_ * 16, wt);     \
    const size_t lds = (size_t)4 * KS_ * 256 * sizeof(float);                                                    \
    if (lds > 48 * 1024) {                                                                                           \
      static bool set_f = false, set_h = false;                                                                      \
      bool &done = out_bf16 ? set_h : set_f;                                                                         \
      if (!done) {                                                                                                   \
        if (out_bf16)                                                                                                \
          MRGCN_HIP_TRY(hipFuncSetAttribute((const void *)k_xform_band_fwd<KS_, uint16_t>,                           \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
        else                                                                                                         \
          MRGCN_HIP_TRY(hipFuncSetAttribute((const void *)k_xform_band_fwd<KS_, float>,                              \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
        done = true;                                                                                                 \
      }                                                                                                              \
    }                                                                                                                \
    if (out_bf16)                                                                                                    \
      k_xform_band_fwd<KS_, uint16_t><<<grid, block, lds, s>>>(bt.tile_beg, bt.tile_rc, bt.band_tptr, bt.tnode, tout,  \
                                                             (int32_t)bt.n_bands, bt.n_tiles, In, ldIn, K, wt, F,    \
                                                             (uint16_t *)Out, ldOut, dbg);                                \
    else                                                                                                             \
      k_xform_band_fwd<KS_, float><<<grid, block, lds, s>>>(bt.tile_beg, bt.tile_rc, bt.band_tptr, bt.tnode, tout,     \
                                                          (int32_t)bt.n_bands, bt.n_tiles, In, ldIn, K, wt, F,       \
                                                          (float *)Out, ldOut, dbg);                                      \
  } while (0)
  if (ksteps <= 4) XB_GO(4);
  else if (ksteps <= 8) XB_GO(8);
  else if (ksteps <= 10) XB_GO(10);
  else if (ksteps <= 12) XB_GO(12);
  else XB_GO(16);
#undef XB_GO
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

}  // namespace mrgcn
