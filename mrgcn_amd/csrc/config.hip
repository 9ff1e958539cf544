// The library's configuration table (config.hpp) and its C ABI: mrgcn_config_count / _name / _get / _set.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.hpp"
#include "config.hpp"

namespace mrgcn {
namespace {
struct Entry { const char *name, *env; int64_t def; const char *doc; };
const Entry kEntries[CFG_COUNT] = {
    {"lp_vec4", "MRGCN_LP_VEC4", 1, "decoder: float4 rows in the DistMult forward / sorted backward"},
    {"mm_blocks", "MRGCN_MM_BLOCKS", 768, "encoders' tiled product: workgroups to aim for when choosing tile and split-K"},
    {"mm_tile", "MRGCN_MM_TILE", -1, "encoders' tiled product: force a tile shape (-1: chosen per product)"},
    {"gemm_tiled", "MRGCN_GEMM_TILED", 1, "mrgcn_gemm_f32: the tiled matrix-core kernel (0: the scalar fallback)"},
    {"adam_nt", "MRGCN_ADAM_NT", 0, "dense Adam kernel: nontemporal loads / stores mode"},
    {"adam_grid", "MRGCN_ADAM_GRID", 8192, "dense Adam kernel: grid cap"},
    {"pool_keep_mb", "MRGCN_POOL_KEEP_MB", 4096, "device memory the block pool keeps before returning blocks to the driver (MB)"},
    {"replicate", "MRGCN_REPLICATE", 0, "plans replicate the lukewarm operand rows per reader by default"},
    {"node_band", "MRGCN_NODE_BAND", 131072, "source nodes per band of the relation-major column order (wide inputs)"},
    {"node_band_narrow", "MRGCN_NODE_BAND_NARROW", 32768, "the same for narrow inputs (<= 0: no second order)"},
    {"hot_min", "MRGCN_HOT_MIN", 16, "columns read by at least this many rows go to the hot region of the operand"},
    {"avoid_straddle", "MRGCN_AVOID_STRADDLE", 1, "operand order: re-read columns leave line-straddling slots to single-reader ones"},
    {"debug_captured_memset", "MRGCN_DEBUG_CAPTURED_MEMSET", 0, "fills inside a capture as hipMemsetAsync (reproduces the ROCm 7.2 memset-node fault)"},
    {"sup_timing", "MRGCN_SUP_TIMING", 0, "print the host timeline of a support chain's build"},
    {"xform_mfma", "MRGCN_XFORM_MFMA", 1, "per-relation transforms on the matrix cores (0: the LDS-FMA kernels)"},
    {"mix_pad", "MRGCN_MIX_PAD", 0, "basis mix forward writes zeros into the pad of an operand row"},
    {"mix_cols", "MRGCN_MIX_COLS", 0, "basis mix forward: the column-parallel form"},
    {"mix_wide", "MRGCN_MIX_WIDE", 1, "basis mix forward: the wide-row kernel for F > 16, B <= 4"},
    {"mix_mfma", "MRGCN_MIX_MFMA", 1, "basis mix forward on the matrix cores"},
    {"mix_fwd_tb", "MRGCN_MIX_FWD_TB", 1024, "block size of the scalar basis mix forward"},
    {"mix_node", "MRGCN_MIX_NODE", 1, "node-major mix backward / fused row Adam"},
    {"mix_bwd_tb", "MRGCN_MIX_BWD_TB", 512, "block size of the node-major mix backward"},
    {"mix_bwd_per_cu", "MRGCN_MIX_BWD_PER_CU", 0, "cap on blocks per CU of the node-major mix backward (0: by LDS)"},
    {"fused_adam", "MRGCN_FUSED_ADAM", 1, "row Adam with the gradient formed on the fly"},
    {"adam_list", "MRGCN_ADAM_LIST", 1, "fused row Adam on a support walks the live nodes as a list (pipelined, nontemporal)"},
    {"dcomp_wide", "MRGCN_DCOMP_WIDE", 1, "dcomp of wide layers by the wide kernel"},
    {"spmm_v3", "MRGCN_SPMM_V3", 1, "narrow forward product: k_spmm3"},
    {"spmm_wpe", "MRGCN_SPMM_WPE", 7, "k_spmm3: waves per SIMD the 32-bit-offset instantiation is built for (0: the plain form)"},
    {"spmm_xcd", "MRGCN_SPMM_XCD", 1, "general product: contiguous row runs per XCD"},
    {"spmm_tail", "MRGCN_SPMM_TAIL", 1, "16-byte gathers with a scalar tail on dword-aligned unpadded rows"},
    {"spmm_fold", "MRGCN_SPMM_FOLD", 1, "split rows finished in-kernel by the last arriver (0: a finalize launch)"},
    {"spmm_tiny", "MRGCN_SPMM_TINY", 0, "pre-pass for rows of very few entries"},
    {"sup_mix_tb", "MRGCN_SUP_MIX_TB", 512, "mix backward on a support: block size (512 or 1024)"},
    {"sup_mix_nb", "MRGCN_SUP_MIX_NB", 2, "mix backward on a support: nodes in flight per wave"},
    {"wide_bwd", "MRGCN_WIDE_BWD", 1, "wide featureless layer: backward straight from dY"},
    {"xform_cols_lds", "MRGCN_XFORM_COLS_LDS", 1, "narrow transform with every relation's weights in LDS, output order"},
    {"spmm_literal_v3", "MRGCN_SPMM_LITERAL_V3", 1, "LITERAL products of narrow layers on the compact view's row classes (k_spmm3 with literal columns)"},
    {"mix_add_vec", "MRGCN_MIX_ADD_VEC", 1, "basis mix forward: the feature term's rows come in as 16-byte pieces through LDS (0: four 4-byte loads per node)"},
    {"sup_rel_chunk", "MRGCN_SUP_REL_CHUNK", 512, "gradient supports: live columns of one (band, relation) group per transform block (64..1024; read when a support is built)"},
    {"adam_once", "MRGCN_ADAM_ONCE", 1, "fused row Adam on a support as a ONE-SHOT grid: list entries per wave (1, 2 or 4; 0: the persistent list kernel)"},
    {"mix_tickets", "MRGCN_MIX_TICKETS", 12, "basis mix forward: waves draw tiles of steps in order from this many ticket counters (1..64; 0: a stride through the node range)"},
    {"mix_ticket_tile", "MRGCN_MIX_TICKET_TILE", 4, "steps (two nodes each) a wave takes per ticket"},
    {"spmm_t_seg", "MRGCN_SPMM_T_SEG", 0, "general TRANSPOSED product of narrow layers: entry-sliced with a segmented sum (measured 417 vs 436 us at the AM shape — the product is bound by line fetches of the gathered rows, not by issue — and its sums are ordered differently from the live / support forms': opt-in)"},
    {"sup_mix_once", "MRGCN_SUP_MIX_ONCE", 0, "mix backward on a support as a one-shot grid (a wave per 64 list entries): 1 = with the LDS comp table (measured 739 vs 670 us in the AM epoch: its blocks re-stage 43 KB too often), 2 = comp rows read from the global table; 0: a resident grid striding through the list"}
};
std::atomic<int64_t> g_values[CFG_COUNT];
std::once_flag g_once;
void init_values() {
  for (int i = 0; i < CFG_COUNT; ++i) {
    const char *e = getenv(kEntries[i].env);
    g_values[i].store((e && *e) ? atoll(e) : kEntries[i].def, std::memory_order_relaxed);
  }
}
int find(const char *name) {
  if (!name) return -1;
  for (int i = 0; i < CFG_COUNT; ++i)
    if (!strcmp(name, kEntries[i].name) || !strcmp(name, kEntries[i].env)) return i;
  return -1;
}
}  // namespace

int64_t cfg(CfgKey k) {
  std::call_once(g_once, init_values);
  return g_values[k].load(std::memory_order_relaxed);
}
}  // namespace mrgcn

extern "C" {
int32_t mrgcn_config_count(void) { return mrgcn::CFG_COUNT; }
const char *mrgcn_config_name(int32_t i) { return (i >= 0 && i < mrgcn::CFG_COUNT) ? mrgcn::kEntries[i].name : nullptr; }
const char *mrgcn_config_doc(int32_t i) { return (i >= 0 && i < mrgcn::CFG_COUNT) ? mrgcn::kEntries[i].doc : nullptr; }
int mrgcn_config_get(const char *name, int64_t *value) {
  using namespace mrgcn;
  const int i = find(name);
  MRGCN_REQUIRE(i >= 0 && value, "unknown configuration key / NULL");
  *value = cfg((CfgKey)i);
  return MRGCN_OK;
}
int mrgcn_config_set(const char *name, int64_t value) {
  using namespace mrgcn;
  const int i = find(name);
  MRGCN_REQUIRE(i >= 0, "unknown configuration key");
  (void)cfg((CfgKey)i);  // (initialised before the first write)
  g_values[i].store(value, std::memory_order_relaxed);
  return MRGCN_OK;
}
}  // extern "C"
