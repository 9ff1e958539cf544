// Process-wide configuration of libmrgcn_hip.so: every switch the kernels' launchers look at, in ONE table.
// Initialised once, on the first read, from the MRGCN_* environment variables named in config.hip (so that a run can
// still be steered from outside); after that only mrgcn_config_set changes a value.  No compute entry point uses
// the C library's environment lookup, and none caches a switch in a function-local static: a value set through the ABI takes effect at the next
// call.  (Plans keep what they were built with.)
#pragma once
#include <stdint.h>

namespace mrgcn {
enum CfgKey : int {
  CFG_LP_VEC4,
  CFG_MM_BLOCKS,
  CFG_MM_TILE,
  CFG_GEMM_TILED,
  CFG_ADAM_NT,
  CFG_ADAM_GRID,
  CFG_POOL_KEEP_MB,
  CFG_REPLICATE,
  CFG_NODE_BAND,
  CFG_NODE_BAND_NARROW,
  CFG_HOT_MIN,
  CFG_AVOID_STRADDLE,
  CFG_DEBUG_CAPTURED_MEMSET,
  CFG_SUP_TIMING,
  CFG_XFORM_MFMA,
  CFG_MIX_PAD,
  CFG_MIX_COLS,
  CFG_MIX_WIDE,
  CFG_MIX_MFMA,
  CFG_MIX_FWD_TB,
  CFG_MIX_NODE,
  CFG_MIX_BWD_TB,
  CFG_MIX_BWD_PER_CU,
  CFG_FUSED_ADAM,
  CFG_ADAM_LIST,
  CFG_DCOMP_WIDE,
  CFG_SPMM_V3,
  CFG_SPMM_WPE,
  CFG_SPMM_XCD,
  CFG_SPMM_TAIL,
  CFG_SPMM_FOLD,
  CFG_SPMM_TINY,
  CFG_SUP_MIX_TB,
  CFG_SUP_MIX_NB,
  CFG_WIDE_BWD,
  CFG_XFORM_COLS_LDS,
  CFG_SPMM_LITERAL_V3,
  CFG_MIX_ADD_VEC,
  CFG_SUP_REL_CHUNK,
  CFG_ADAM_ONCE,
  CFG_MIX_TICKETS,
  CFG_MIX_TICKET_TILE,
  CFG_SPMM_T_SEG,
  CFG_SUP_MIX_ONCE,
  CFG_COUNT
};
int64_t cfg(CfgKey k);
}  // namespace mrgcn
