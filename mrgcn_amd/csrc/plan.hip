// Graph plan construction: the reference's uncoalesced COO (int64 indices, int8 or f32
// values; mrgcn/data/utils.py:165-170, mrgcn/data/batch.py:144-149) -> device-resident
// CSR over output rows + CSC over *touched* columns in (source node, relation) order.
// One-off work per adjacency (A is static across epochs), so device-wide hipcub
// sort/scan primitives are used; everything else is hand-written.
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include <chrono>
#include <cstdio>

#include "common.hpp"
#include "config.hpp"

namespace mrgcn {

thread_local std::string g_last_error;
void set_error(const std::string &msg) { g_last_error = msg; }

namespace {

constexpr int kTB = 256;
inline int nblocks(int64_t n) { return (int)((n + kTB - 1) / kTB); }

// key = row * RN + col; zero-valued entries get the sentinel when pruning so that they
// sort past the end.  Out-of-range indices raise *err.
__global__ void k_make_keys(const int64_t *__restrict__ rows, const int64_t *__restrict__ cols,
                            const void *__restrict__ vals, int val_dtype, int64_t nnz,
                            int64_t num_rows, int64_t RN, int prune, int64_t *__restrict__ keys,
                            float *__restrict__ fvals, unsigned int *__restrict__ kept,
                            int *__restrict__ err) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  int64_t r = rows[e], c = cols[e];
  float v = (val_dtype == MRGCN_VAL_I8) ? (float)((const int8_t *)vals)[e] : ((const float *)vals)[e];
  if (r < 0 || r >= num_rows || c < 0 || c >= RN) {
    *err = 1;
    keys[e] = INT64_MAX;
    fvals[e] = 0.f;
    return;
  }
  if (prune && v == 0.f) {
    keys[e] = INT64_MAX;
    fvals[e] = 0.f;
    return;
  }
  keys[e] = r * RN + c;
  fvals[e] = v;
  // the count of kept entries is only unknown when zeros are pruned (otherwise every entry of an error-free input is
  // kept: the launcher presets the counter): one same-address atomic per WAVE — what the compiler makes of a per-thread
  // add — cost 2.4 ms of the 15 ms AM plan build (213 k atomics on one word)
  if (prune) atomicAdd(kept, 1u);
}

// CSR -> the COO the plan builder consumes; `cast_i8`: the reference's boundary cast
// (scipy_sparse_to_pytorch_sparse(..., dtype=torch.int8), data/utils.py:165-170: C truncation)
__global__ void k_csr_to_coo(const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                             const float *__restrict__ data, int64_t num_rows, int64_t nnz, int cast_i8,
                             int64_t *__restrict__ rows, int64_t *__restrict__ cols,
                             float *__restrict__ vals) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  int64_t lo = 0, hi = num_rows;  // last row whose pointer is <= e
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if ((int64_t)indptr[mid] <= e) lo = mid; else hi = mid - 1;
  }
  rows[e] = lo;
  cols[e] = indices[e];
  const float v = data[e];
  vals[e] = cast_i8 ? (float)(int8_t)v : v;
}

__global__ void k_decode(const int64_t *__restrict__ keys, int64_t nnz, int64_t RN, int64_t N,
                         int64_t R, int32_t *__restrict__ rowidx, int32_t *__restrict__ lcol,
                         int64_t *__restrict__ key2, int32_t *__restrict__ eidx) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  int64_t k = keys[e];
  int64_t r = k / RN, c = k - r * RN;
  int64_t rel = c / N, j = c - rel * N;
  rowidx[e] = (int32_t)r;
  lcol[e] = (int32_t)c;
  key2[e] = j * R + rel;  // (source node, relation) order of the compact columns
  eidx[e] = (int32_t)e;
}

// ptr[i] = first position p in sorted `keys` with keys[p] >= i * stride, i in [0, n]
__global__ void k_lower_bound_ptr(const int64_t *__restrict__ keys, int64_t nnz, int64_t n,
                                  int64_t stride, int32_t *__restrict__ ptr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  int64_t target = i * stride;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  ptr[i] = (int32_t)lo;
}

__global__ void k_heads(const int64_t *__restrict__ key2s, int64_t nnz, int32_t *__restrict__ head) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  head[k] = (k == 0 || key2s[k] != key2s[k - 1]) ? 1 : 0;
}

__global__ void k_scatter_cols(const int64_t *__restrict__ key2s, const int32_t *__restrict__ eidxs,
                               const int32_t *__restrict__ cid1, const int32_t *__restrict__ head,
                               const int32_t *__restrict__ rowidx, const float *__restrict__ val,
                               int64_t nnz, int64_t R, int32_t *__restrict__ ccol,
                               int32_t *__restrict__ crow, float *__restrict__ cval,
                               int32_t *__restrict__ cptr, int32_t *__restrict__ urel,
                               int32_t *__restrict__ unode, int32_t *__restrict__ ulcol, int64_t N) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  int32_t e = eidxs[k];
  int32_t c = cid1[k] - 1;
  ccol[e] = c;
  crow[k] = rowidx[e];
  cval[k] = val[e];
  if (head[k]) {
    int64_t k2 = key2s[k];
    int64_t j = k2 / R;
    cptr[c] = (int32_t)k;
    unode[c] = (int32_t)j;
    int64_t rel = k2 - j * R;
    urel[c] = (int32_t)rel;
    ulcol[c] = (int32_t)(rel * N + j);
  }
}

// ptr[i] = first position in sorted int32 `keys` with keys[pos] >= i * stride, i in [0, n]
// (stride 1 over unode -> node pointers; stride N over sorted literal columns -> relation pointers)
__global__ void k_node_ptr(const int32_t *__restrict__ keys, int64_t count, int64_t n, int64_t stride,
                           int32_t *__restrict__ ptr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  int64_t target = i * stride;
  int64_t lo = 0, hi = count;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  ptr[i] = (int32_t)lo;
}

// storage order of the compact operand M (see common.hpp: mpos): hot columns (>= hot_min
// entries) first, most referenced first; then every other column in the order in which the FIRST
// output row that reads it is processed (class-major rank, k_class_keys).
// key = [hot: max_count - count | rest: max_count + 1 + rank(row)][compact id]
__global__ void k_mpos_keys(const int32_t *__restrict__ cptr, const int32_t *__restrict__ crow,
                            const int32_t *__restrict__ rank, int64_t ncols, int64_t max_count, int shift,
                            int hot_min, int64_t *__restrict__ keys, int32_t *__restrict__ ids) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  const int32_t b = cptr[c], e = cptr[c + 1];
  const int64_t cnt = e - b;
  int64_t hi;
  if (hot_min < 0) hi = 0;                                   // experiment: plain compact (node, relation) order
  else if (cnt >= hot_min) hi = max_count - cnt;            // in [0, max_count)
  else hi = max_count + 1 + (int64_t)rank[crow[b]];         // after every hot column, in processing order
  keys[c] = (hi << shift) | c;
  ids[c] = (int32_t)c;
}

__global__ void k_row_pos_keys(const int32_t *__restrict__ rowidx, const int32_t *__restrict__ rank,
                               const int32_t *__restrict__ pos, int64_t nnz, int64_t ncols,
                               int64_t *__restrict__ keys) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nnz) keys[e] = (int64_t)rank[rowidx[e]] * ncols + pos[e];
}

// ---- operand replicas ------------------------------------------------------------------------------------
// entries in class-major order carry the operand row of their column (`m`); with replicas every entry of a
// non-hot column (m >= n_hot) gets the next row of the stream instead
__global__ void k_is_stream(const int32_t *__restrict__ m, int64_t nnz, int32_t n_hot, int32_t *__restrict__ flag) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nnz) flag[e] = m[e] >= n_hot ? 1 : 0;
  if (e == nnz) flag[e] = 0;
}
__global__ void k_rep_assign(int32_t *__restrict__ m, const int32_t *__restrict__ spos, int64_t nnz, int32_t n_hot,
                             const int32_t *__restrict__ col_of_mpos, int32_t *__restrict__ ecol,
                             int32_t *__restrict__ mpos_new) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  const int32_t old = m[e];
  const int32_t c = col_of_mpos[old];
  ecol[e] = c;
  if (old >= n_hot) {
    const int32_t row = n_hot + spos[e];
    m[e] = row;
    atomicMin(&mpos_new[c], row);  // primary = the reader that is processed first
  } else {
    mpos_new[c] = old;  // hot: one shared row (every entry writes the same value)
  }
}
__global__ void k_rep_flag(const int32_t *__restrict__ m, const int32_t *__restrict__ ecol,
                           const int32_t *__restrict__ mpos_new, int64_t nnz, int32_t n_hot,
                           int32_t *__restrict__ flag) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nnz) flag[e] = (m[e] >= n_hot && m[e] != mpos_new[ecol[e]]) ? 1 : 0;
  if (e == nnz) flag[e] = 0;
}
__global__ void k_rep_fill(const int32_t *__restrict__ m, const int32_t *__restrict__ ecol,
                           const int32_t *__restrict__ mpos_new, const int32_t *__restrict__ flag,
                           const int32_t *__restrict__ pos, int64_t nnz, int32_t *__restrict__ src,
                           int32_t *__restrict__ dst) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nnz && flag[e]) {
    src[pos[e]] = mpos_new[ecol[e]];
    dst[pos[e]] = m[e];
  }
}
__global__ void k_iota_i32(int32_t *__restrict__ a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = (int32_t)i;
}
__global__ void k_fill_i32(int32_t *__restrict__ a, int64_t n, int32_t v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

// The COMPACT view processes the rows CLASS-MAJOR: first every row of <= kShort3Rows entries, then those
// of <= kMid3Rows, then the long ones, each class in row order.  Rows of different classes are served by
// different waves (k_spmm3), so with this order no two classes share a line of the index / value arrays or
// of the operand's first-touch region.  key = class * rows + row; sorting gives rank -> row.
__global__ void k_class_keys(const int32_t *__restrict__ ptr, int64_t rows, int s_max, int m_max, int64_t win_s,
                             int64_t win_m, int64_t cstride, int64_t *__restrict__ keys, int32_t *__restrict__ ids) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const int32_t n = ptr[i + 1] - ptr[i];
  const int64_t cls = n <= s_max ? 0 : (n <= m_max ? 1 : 2);
  // inside a class: windows of consecutive row ids, rows of a window by length (S and M: a wave's rows then need the
  // same number of gather rounds and none of its gathers is a masked filler), then by id
  const int64_t win = cls == 0 ? win_s : (cls == 1 ? win_m : 0);
  const int64_t sub = win > 0 ? (i / win) * 64 + n : 0;
  keys[i] = cls * cstride + sub * rows + i;
  ids[i] = (int32_t)i;
}
__global__ void k_rank_len(const int32_t *__restrict__ ptr, const int32_t *__restrict__ rowmap, int64_t rows,
                           int32_t *__restrict__ len) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < rows) len[k] = ptr[rowmap[k] + 1] - ptr[rowmap[k]];
  if (k == rows) len[k] = 0;
}

__global__ void k_keys_to_pos(const int64_t *__restrict__ keys, int64_t nnz, int64_t ncols,
                              int32_t *__restrict__ pos) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nnz) pos[e] = (int32_t)(keys[e] % ncols);
}

__global__ void k_invert_perm(const int32_t *__restrict__ order, int64_t n, int32_t *__restrict__ pos) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[order[i]] = (int32_t)i;
}

__global__ void k_gather_i32(const int32_t *__restrict__ table, const int32_t *__restrict__ idx,
                             int64_t n, int32_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = table[idx[i]];
}

// key of compact column c = (node j, relation r): (band(j) * R + r) * band_size + j % band_size
__global__ void k_band_keys(const int32_t *__restrict__ unode, const int32_t *__restrict__ urel,
                            int64_t ncols, int64_t R, int64_t band, int64_t *__restrict__ keys,
                            int32_t *__restrict__ ids) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  const int64_t j = unode[c], r = urel[c];
  const int64_t b = j / band;
  keys[c] = (b * R + r) * band + (j - b * band);
  ids[c] = (int32_t)c;
}

__global__ void k_iota(int32_t *__restrict__ a, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = (int32_t)i;
}

// per row: is it long, and how many chunks does it need
// chunk size of a row of `len` entries: `chunk`, or — with a cap on the chunks per row — the multiple of
// `chunk` that keeps the row within `cap` chunks (k_spmm3: a wave then walks several pieces of `chunk`
// entries, and a row never leaves more than `cap` partial sums)
__device__ __forceinline__ int32_t row_chunk(int32_t len, int chunk, int cap) {
  if (cap <= 0) return chunk;
  const int32_t n = (len + chunk - 1) / chunk;
  return n <= cap ? chunk : chunk * ((n + cap - 1) / cap);
}

// blockwise rows (k_spmm3's rows of more than kChunk3Entries entries): a row is cut into 4 * nb chunks — nb blocks
// of four waves, at most `cap` / 4 blocks — of equal size (a multiple of 16 entries, the last ones may be empty)
__device__ __forceinline__ int32_t row_blocks(int32_t len, int chunk, int cap) {
  const int32_t nb = (len + 4 * chunk - 1) / (4 * chunk);
  return nb < 1 ? 1 : (nb > cap / 4 ? cap / 4 : nb);
}
__device__ __forceinline__ int32_t row_chunk_blockwise(int32_t len, int chunk, int cap) {
  const int32_t nc = 4 * row_blocks(len, chunk, cap);
  return ((len + nc - 1) / nc + 15) / 16 * 16;
}

// rows of thresh < len (<= upper when upper > 0) are "long"
__global__ void k_long_count(const int32_t *__restrict__ ptr, int64_t rows, int thresh, int upper, int chunk, int cap,
                             int blockwise, int32_t *__restrict__ is_long, int32_t *__restrict__ nchunk,
                             int32_t *__restrict__ maxlen) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = i < rows;   // (every lane stays for the shuffles below)
  int32_t len = on ? ptr[i + 1] - ptr[i] : 0;
  int32_t lg = len > thresh && (upper <= 0 || len <= upper);
  if (i == rows) {  // the slot behind the last row: where the scans leave the totals
    is_long[i] = 0;
    nchunk[i] = 0;
  }
  if (on) {
    is_long[i] = lg;
    if (blockwise) {
      nchunk[i] = lg ? 4 * row_blocks(len, chunk, cap) : 0;
    } else {
      const int32_t rc = row_chunk(len, chunk, cap);
      nchunk[i] = lg ? (len + rc - 1) / rc : 0;
    }
  }
  // longest row: the wave's maximum by shuffles, and an atomic only if it beats what the word already holds (127 k
  // same-address atomics were 1.45 ms of the AM plan build)
  if (!maxlen) return;  // (wave uniform: a caller that does not ask)
  int32_t m = len;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, kWave));
  if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(maxlen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxlen, m);
}

__global__ void k_long_fill(const int32_t *__restrict__ ptr, int64_t rows, int chunk0, int cap, int blockwise,
                            const int32_t *__restrict__ is_long, const int32_t *__restrict__ long_pos,
                            const int32_t *__restrict__ chunk_pos, int32_t *__restrict__ long_row,
                            int32_t *__restrict__ long_cptr, int32_t *__restrict__ chunk_beg,
                            int32_t *__restrict__ chunk_end, int32_t *__restrict__ chunk_row, int32_t n_chunks) {
  // chunk_row holds 2 * n_chunks words: [c] the row (see below), [n_chunks + c] the position li of the chunk's row
  // among the long rows (the in-kernel finalize of the products finds its arrival counter there)
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows || !is_long[i]) return;
  int32_t li = long_pos[i], c0 = chunk_pos[i];
  long_row[li] = (int32_t)i;
  long_cptr[li] = c0;
  int32_t b = ptr[i], e = ptr[i + 1];
  if (blockwise) {  // every chunk names its row by position: chunk_row = -(li + 2)
    const int32_t nc = 4 * row_blocks(e - b, chunk0, cap), rc = row_chunk_blockwise(e - b, chunk0, cap);
    for (int32_t j = 0; j < nc; ++j) {
      const int32_t s = min(b + j * rc, e);
      chunk_beg[c0 + j] = s;
      chunk_end[c0 + j] = min(s + rc, e);
      chunk_row[c0 + j] = -(li + 2);
      chunk_row[n_chunks + c0 + j] = li;
    }
    return;
  }
  const int32_t chunk = row_chunk(e - b, chunk0, cap);
  for (int32_t s = b, c = c0; s < e; s += chunk, ++c) {
    chunk_beg[c] = s;
    chunk_end[c] = min(s + chunk, e);
    chunk_row[c] = (e - b <= chunk) ? (int32_t)i : -((int32_t)i + 2);  // < 0: one of several chunks of row -x - 2
    chunk_row[n_chunks + c] = li;
  }
}

// Device memory of a plan build: a small caching allocator in front of the device's stream-ordered pool
// (hipMallocAsync).  Building the plans of re-sampled mini-batch slices over and over asks for the same ~140 blocks
// (60 plan arrays, 80 scratch arrays) again and again; through the driver's pool every hipMallocAsync / hipFreeAsync
// costs tens of microseconds of host time once blocks have been freed (measured: 6-8 ms per slice plan against
// 1.5-3.4 ms while nothing had been freed yet).  Freed blocks therefore park here, tagged with the stream whose order
// protects them and the device's synchronisation epoch at the time: a request takes the smallest parked block of at
// least its size (and at most four times it) that was freed on the SAME stream (stream order: the block's last
// user runs before the new one) or before the last device-wide wait.  The cache keeps up to MRGCN_POOL_KEEP_MB
// (default 4096) per device; what does not fit goes back to the driver's pool.
constexpr uint64_t kPoolKeep = 4ull << 30;

struct ParkedBlock { void *ptr; size_t bytes; hipStream_t stream; uint64_t epoch; };
struct BlockCache {
  std::mutex mu;
  std::multimap<size_t, ParkedBlock> free_blocks;   // by size
  std::unordered_map<void *, size_t> live;            // blocks handed out: their real size
  uint64_t parked_bytes = 0, epoch = 1, safe_epoch = 0;  // blocks parked with epoch <= safe_epoch: free for any stream
  uint64_t keep = kPoolKeep;
  bool configured = false;
};
BlockCache &cache_of(int dev) {
  static BlockCache caches[64];
  return caches[(dev >= 0 && dev < 64) ? dev : 0];
}

hipError_t pool_alloc(void **p, size_t bytes, hipStream_t s) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  BlockCache &c = cache_of(dev);
  bytes = (bytes + 255) / 256 * 256;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    if (!c.configured) {
      hipMemPool_t pool;
      if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
        c.keep = (uint64_t)std::max<int64_t>(cfg(CFG_POOL_KEEP_MB), 0) << 20;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &c.keep);
      }
      c.configured = true;
    }
    for (auto it = c.free_blocks.lower_bound(bytes); it != c.free_blocks.end() && it->first <= 4 * bytes + 4096; ++it) {
      const ParkedBlock &b = it->second;
      if (b.stream == s || b.epoch <= c.safe_epoch) {
        *p = b.ptr;
        c.live[b.ptr] = b.bytes;
        c.parked_bytes -= b.bytes;
        c.free_blocks.erase(it);
        return hipSuccess;
      }
    }
  }
  hipError_t e = hipMallocAsync(p, bytes, s);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(c.mu);
    c.live[*p] = bytes;
  }
  return e;
}

// stream ordered free: the block may be reused by later work on `s`, or by anyone once a device-wide wait that began
// after this call has completed (`covered_epoch` != 0: such a wait — pool_sync_begin's return value — has completed
// already)
constexpr uint64_t kSafeNow = ~0ull;
void pool_free(void *p, hipStream_t s, uint64_t covered_epoch = 0) {
  if (!p) return;
  int dev = 0;
  (void)hipGetDevice(&dev);
  BlockCache &c = cache_of(dev);
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.live.find(p);
    if (it != c.live.end()) {
      bytes = it->second;
      c.live.erase(it);
      if (c.parked_bytes + bytes <= c.keep) {
        // (kSafeNow: every user has finished already — epoch 0 is below any safe_epoch)
        c.free_blocks.emplace(bytes, ParkedBlock{p, bytes, s, covered_epoch == kSafeNow ? 0 : covered_epoch ? covered_epoch : c.epoch});
        c.parked_bytes += bytes;
        return;
      }
    }
  }
  (void)hipFreeAsync(p, s);
}

// around a device-wide wait: blocks parked BEFORE pool_sync_begin are free for any stream once pool_sync_done ran
uint64_t pool_sync_begin() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  BlockCache &c = cache_of(dev);
  std::lock_guard<std::mutex> lock(c.mu);
  return c.epoch++;
}
void pool_sync_done(uint64_t e) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  BlockCache &c = cache_of(dev);
  std::lock_guard<std::mutex> lock(c.mu);
  if (e > c.safe_epoch) c.safe_epoch = e;
}

// An operand row that straddles a 128-byte line costs a re-reading row two line fetches instead of one.  A column read
// by ONE row sits in that row's sequential run, where the second line is fetched anyway; so inside each aligned group
// of 32 positions (4F-byte rows repeat their alignment every 32 rows at most) the columns with several readers trade
// places with single-reader ones until none of them sits on a straddling slot — for every row size in `row_bytes`
// (the operand layouts the plan's users announced: 48 = rows of 10..12 floats padded to 16 bytes, the default;
// 40 / 44 = packed rows of 10 / 11 floats).  In order: straddling multi-reader slots ascending, each takes the next
// free straddle-free single-reader slot.  A ragged last group stays as it is.
struct StraddleSizes { int32_t n; int32_t bytes[4]; };
__global__ void k_avoid_straddle(int32_t *__restrict__ order, const int32_t *__restrict__ cptr, int64_t ncols,
                                 StraddleSizes sz, const int32_t *__restrict__ n_hot_d) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t p0 = g * 32;
  if (p0 + 32 > ncols) return;
  // the group that holds the end of the hot region stays as it is: positions below n_hot are hot columns, the
  // replica path classifies by position
  const int64_t n_hot = n_hot_d[1];
  if (p0 < n_hot && p0 + 32 > n_hot) return;
  uint32_t bad = 0, multi = 0;
  for (int i = 0; i < 32; ++i) {
    for (int k = 0; k < sz.n; ++k) {
      const int64_t lo = (p0 + i) * sz.bytes[k], hi = lo + sz.bytes[k] - 1;
      if ((lo >> 7) != (hi >> 7)) bad |= 1u << i;
    }
    const int32_t c = order[p0 + i];
    if (cptr[c + 1] - cptr[c] > 1) multi |= 1u << i;
  }
  uint32_t todo = bad & multi, free_slots = ~bad & ~multi;
  while (todo && free_slots) {
    const int i = __ffs(todo) - 1, j = __ffs(free_slots) - 1;
    todo &= todo - 1;
    free_slots &= free_slots - 1;
    const int32_t t = order[p0 + i];
    order[p0 + i] = order[p0 + j];
    order[p0 + j] = t;
  }
}

struct Scratch {  // returns its allocations to the pool on scope exit, ordered after the build's work on `s`
  hipStream_t s = nullptr;
  std::vector<void *> ptrs;
  ~Scratch() { for (void *p : ptrs) pool_free(p, s); }
  template <typename T> hipError_t alloc(T **p, int64_t n) {
    hipError_t e = pool_alloc((void **)p, (size_t)std::max<int64_t>(n, 1) * sizeof(T), s);
    if (e == hipSuccess) ptrs.push_back(*p);
    return e;
  }
};

template <typename T> hipError_t plan_alloc(mrgcn_plan *p, T **dst, int64_t n) {
  size_t bytes = (size_t)std::max<int64_t>(n, 1) * sizeof(T);
  hipError_t e = pool_alloc((void **)dst, bytes, p->build_stream);
  if (e == hipSuccess) p->device_bytes += (int64_t)bytes;
  return e;
}

int exclusive_scan_i32(const int32_t *in, int32_t *out, int64_t n, hipStream_t s, Scratch &sc) {
  size_t tb = 0;
  MRGCN_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, (int)n, s));
  char *tmp = nullptr;
  MRGCN_HIP_TRY(sc.alloc(&tmp, (int64_t)tb));
  MRGCN_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tb, in, out, (int)n, s));
  return MRGCN_OK;
}

// builds the split-row descriptors of one orientation
int build_long(mrgcn_plan *p, const int32_t *ptr, int64_t rows, hipStream_t s, int32_t **long_row,
               int32_t **long_cptr, int32_t **chunk_beg, int32_t **chunk_end, int32_t **chunk_row,
               int32_t *n_long,
               int32_t *n_chunks, int64_t *max_len, int threshold = kLongThreshold, int chunk = kChunk,
               int cap = 0, int upper = 0, int blockwise = 0, const int32_t *known = nullptr) {
  if (known && known[0] == 0) {  // no long row (counted by the caller): the arrays keep one element each
    *n_long = 0;
    *n_chunks = 0;
    *max_len = 0;
    MRGCN_HIP_TRY(plan_alloc(p, long_row, 0));
    MRGCN_HIP_TRY(plan_alloc(p, long_cptr, 1));
    MRGCN_HIP_TRY(plan_alloc(p, chunk_beg, 0));
    MRGCN_HIP_TRY(plan_alloc(p, chunk_end, 0));
    MRGCN_HIP_TRY(plan_alloc(p, chunk_row, 0));
    k_fill_i32<<<1, 1, 0, s>>>(*long_cptr, 1, 0);
    MRGCN_HIP_TRY(hipGetLastError());
    return MRGCN_OK;
  }
  Scratch sc;
  sc.s = s;
  int32_t *is_long, *nchunk, *long_pos, *chunk_pos, *d_max;
  MRGCN_HIP_TRY(sc.alloc(&is_long, rows + 1));
  MRGCN_HIP_TRY(sc.alloc(&nchunk, rows + 1));
  MRGCN_HIP_TRY(sc.alloc(&long_pos, rows + 1));
  MRGCN_HIP_TRY(sc.alloc(&chunk_pos, rows + 1));
  MRGCN_HIP_TRY(sc.alloc(&d_max, 1));
  if (!known) MRGCN_HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(int32_t), s));
  k_long_count<<<nblocks(rows + 1), kTB, 0, s>>>(ptr, rows, threshold, upper, chunk, cap, blockwise, is_long, nchunk,
                                                 known ? nullptr : d_max);
  // scan over rows+1 elements so that position [rows] holds the totals
  int rc;
  if ((rc = exclusive_scan_i32(is_long, long_pos, rows + 1, s, sc))) return rc;
  if ((rc = exclusive_scan_i32(nchunk, chunk_pos, rows + 1, s, sc))) return rc;
  int32_t h[3] = {0, 0, 0};
  if (known) {  // the caller counted already (a gradient support's build): no wait, max_len stays unknown
    h[0] = known[0];
    h[1] = known[1];
  } else {
    MRGCN_HIP_TRY(hipMemcpyAsync(&h[0], long_pos + rows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MRGCN_HIP_TRY(hipMemcpyAsync(&h[1], chunk_pos + rows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MRGCN_HIP_TRY(hipMemcpyAsync(&h[2], d_max, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MRGCN_HIP_TRY(hipStreamSynchronize(s));
  }
  *n_long = h[0];
  *n_chunks = h[1];
  *max_len = h[2];
  MRGCN_HIP_TRY(plan_alloc(p, long_row, h[0]));
  MRGCN_HIP_TRY(plan_alloc(p, long_cptr, h[0] + 1));
  MRGCN_HIP_TRY(plan_alloc(p, chunk_beg, h[1]));
  MRGCN_HIP_TRY(plan_alloc(p, chunk_end, h[1]));
  MRGCN_HIP_TRY(plan_alloc(p, chunk_row, 2 * (int64_t)h[1]));  // (row | position among the long rows: k_long_fill)
  k_fill_i32<<<1, 1, 0, s>>>(*long_cptr + h[0], 1, h[1]);  // the total closes the chunk ranges (no host pointer: no wait)
  if (rows > 0 && h[0] > 0)
    k_long_fill<<<nblocks(rows), kTB, 0, s>>>(ptr, rows, chunk, cap, blockwise, is_long, long_pos, chunk_pos,
                                              *long_row, *long_cptr, *chunk_beg, *chunk_end, *chunk_row, h[1]);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;  // (stream ordered from here on: the scratch goes back tagged with the stream)
}

// long rows of several chunks: flag, (scan), positions
__global__ void k_multi_flag(const int32_t *__restrict__ long_cptr, int64_t n_long, int more_than,
                             int32_t *__restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_long) flag[i] = long_cptr[i + 1] - long_cptr[i] > more_than ? 1 : 0;
  if (i == n_long) flag[i] = 0;
}
__global__ void k_multi_fill(const int32_t *__restrict__ flag, const int32_t *__restrict__ pos, int64_t n_long,
                             int32_t *__restrict__ multi) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_long && flag[i]) multi[pos[i]] = (int32_t)i;
}

int bits_for(int64_t max_value) {
  int b = 1;
  while (b < 63 && (max_value >> b) != 0) ++b;
  return b;
}

// one relation-major order of the compact columns with bands of `band` source nodes (common.hpp: RelOrder)
int build_rel_order(mrgcn_plan *p, Scratch &sc, hipStream_t s, int64_t band, bool set_top_rel, int32_t **rperm,
                    int32_t **relptr, int32_t **rc_rel, int32_t **rc_beg, int32_t **rc_end, int32_t **rc_ptr,
                    int32_t **rc_ids, int32_t *n_chunks, int32_t *max_chunks, int64_t *band_out,
                    int64_t *nbands_out) {
  const int64_t N = p->num_nodes, R = p->num_relations, ncols = p->ncols;
  MRGCN_HIP_TRY(plan_alloc(p, rperm, ncols));
  const int64_t nbands = (N + band - 1) / band;
  *band_out = band;
  *nbands_out = nbands;
  const int64_t ngroups = nbands * R;
  MRGCN_HIP_TRY(plan_alloc(p, relptr, ngroups + 1));
  int32_t *ids;
  int64_t *k3, *k3_s;
  MRGCN_HIP_TRY(sc.alloc(&ids, ncols));
  MRGCN_HIP_TRY(sc.alloc(&k3, ncols));
  MRGCN_HIP_TRY(sc.alloc(&k3_s, ncols));
  if (ncols > 0) {
    k_band_keys<<<nblocks(ncols), kTB, 0, s>>>(p->unode, p->urel, ncols, R, band, k3, ids);
    MRGCN_HIP_TRY(hipGetLastError());
    size_t tb = 0;
    const int end_bit = bits_for(ngroups * band + band);
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k3, k3_s, ids, *rperm, (int)ncols, 0, end_bit, s));
    char *tmp;
    MRGCN_HIP_TRY(sc.alloc(&tmp, (int64_t)tb));
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k3, k3_s, ids, *rperm, (int)ncols, 0, end_bit, s));
  }
  k_lower_bound_ptr<<<nblocks(ngroups + 1), kTB, 0, s>>>(k3_s, ncols, ngroups, band, *relptr);
  MRGCN_HIP_TRY(hipGetLastError());
  // chunks (<= kRelChunk columns of one (band, relation) group each), built on the host
  std::vector<int32_t> h_gptr(ngroups + 1);
  MRGCN_HIP_TRY(hipMemcpyAsync(h_gptr.data(), *relptr, (ngroups + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MRGCN_HIP_TRY(hipStreamSynchronize(s));
  const int rel_chunk = kRelChunk;
  std::vector<int32_t> rel, beg, end;
  std::vector<std::vector<int32_t>> by_rel(R);
  for (int64_t g = 0; g < ngroups; ++g) {
    const int32_t r = (int32_t)(g % R);
    for (int32_t b0 = h_gptr[g]; b0 < h_gptr[g + 1]; b0 += rel_chunk) {
      rel.push_back(r);
      beg.push_back(b0);
      end.push_back(std::min(b0 + rel_chunk, h_gptr[g + 1]));
    }
  }
  for (size_t c = 0; c < rel.size(); ++c) by_rel[rel[c]].push_back((int32_t)c);
  if (set_top_rel) {  // the relation with the most compact columns: k_mix_bwd_nm keeps its dcomp row in registers
    std::vector<int64_t> per_rel(R, 0);
    for (int64_t g = 0; g < ngroups; ++g) per_rel[g % R] += h_gptr[g + 1] - h_gptr[g];
    int64_t best = 0;
    for (int64_t r = 0; r < R; ++r)
      if (per_rel[r] > best) { best = per_rel[r]; p->top_rel = (int32_t)r; }
  }
  std::vector<int32_t> cptr_rel(R + 1, 0), ids_by_rel;
  *max_chunks = 0;
  for (int64_t r = 0; r < R; ++r) {
    cptr_rel[r] = (int32_t)ids_by_rel.size();
    ids_by_rel.insert(ids_by_rel.end(), by_rel[r].begin(), by_rel[r].end());
    *max_chunks = std::max(*max_chunks, (int32_t)by_rel[r].size());
  }
  cptr_rel[R] = (int32_t)ids_by_rel.size();
  *n_chunks = (int32_t)rel.size();
  MRGCN_HIP_TRY(plan_alloc(p, rc_ptr, R + 1));
  MRGCN_HIP_TRY(hipMemcpy(*rc_ptr, cptr_rel.data(), (R + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
  MRGCN_HIP_TRY(plan_alloc(p, rc_ids, *n_chunks));
  MRGCN_HIP_TRY(plan_alloc(p, rc_rel, *n_chunks));
  MRGCN_HIP_TRY(plan_alloc(p, rc_beg, *n_chunks));
  MRGCN_HIP_TRY(plan_alloc(p, rc_end, *n_chunks));
  if (*n_chunks > 0) {
    size_t nb = rel.size() * sizeof(int32_t);
    MRGCN_HIP_TRY(hipMemcpy(*rc_ids, ids_by_rel.data(), nb, hipMemcpyHostToDevice));
    MRGCN_HIP_TRY(hipMemcpy(*rc_rel, rel.data(), nb, hipMemcpyHostToDevice));
    MRGCN_HIP_TRY(hipMemcpy(*rc_beg, beg.data(), nb, hipMemcpyHostToDevice));
    MRGCN_HIP_TRY(hipMemcpy(*rc_end, end.data(), nb, hipMemcpyHostToDevice));
  }
  return MRGCN_OK;
}

// op_node[mpos[c]] = unode[c], op_rel[mpos[c]] = urel[c] (rows without a primary column keep node -1)
__global__ void k_operand_ids(const int32_t *__restrict__ mpos, const int32_t *__restrict__ unode,
                              const int32_t *__restrict__ urel, int64_t ncols, int32_t *__restrict__ op_node,
                              int32_t *__restrict__ op_rel) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  const int32_t pos = mpos[c];
  op_node[pos] = unode[c];
  op_rel[pos] = urel[c];
}

int create_impl(mrgcn_plan *p, int64_t nnz_in, const int64_t *rows, const int64_t *cols,
                const void *vals, int val_dtype, uint32_t flags, hipStream_t s, const StraddleSizes &hint) {
  const int64_t N = p->num_nodes, R = p->num_relations, RN = R * N;
  const bool rep_default = cfg(CFG_REPLICATE) != 0;
  const bool replicate = !(flags & MRGCN_PLAN_NO_REPLICATE) && ((flags & MRGCN_PLAN_REPLICATE) || rep_default);
  Scratch sc;
  sc.s = s;
  int64_t *keys, *keys_s, *key2, *key2_s;
  float *fv;
  int32_t *eidx, *eidx_s, *head, *cid1;
  unsigned int *d_kept;
  int *d_err;
  MRGCN_HIP_TRY(sc.alloc(&keys, nnz_in));
  MRGCN_HIP_TRY(sc.alloc(&keys_s, nnz_in));
  MRGCN_HIP_TRY(sc.alloc(&fv, nnz_in));
  MRGCN_HIP_TRY(sc.alloc(&d_kept, 1));
  MRGCN_HIP_TRY(sc.alloc(&d_err, 1));
  MRGCN_HIP_TRY(hipMemsetAsync(d_kept, 0, sizeof(unsigned int), s));
  MRGCN_HIP_TRY(hipMemsetAsync(d_err, 0, sizeof(int), s));

  float *val_sorted = nullptr;  // becomes plan->val
  MRGCN_HIP_TRY(plan_alloc(p, &val_sorted, nnz_in));
  p->val = val_sorted;

  if (nnz_in > 0) {
    k_make_keys<<<nblocks(nnz_in), kTB, 0, s>>>(rows, cols, vals, val_dtype, nnz_in, p->num_rows, RN,
                                                (flags & MRGCN_PLAN_PRUNE_ZEROS) ? 1 : 0, keys, fv,
                                                d_kept, d_err);
    MRGCN_HIP_TRY(hipGetLastError());
    // sort by (row, col); only the bits that can be set take part (sentinel uses bit 62)
    size_t tb = 0;
    const int end_bit = 63;
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys, keys_s, fv, val_sorted,
                                                     (int)nnz_in, 0, end_bit, s));
    char *tmp;
    MRGCN_HIP_TRY(sc.alloc(&tmp, (int64_t)tb));
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys, keys_s, fv, val_sorted,
                                                     (int)nnz_in, 0, end_bit, s));
  }
  unsigned int h_kept = 0;
  int h_err = 0;
  MRGCN_HIP_TRY(hipMemcpyAsync(&h_kept, d_kept, sizeof(h_kept), hipMemcpyDeviceToHost, s));
  MRGCN_HIP_TRY(hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s));
  MRGCN_HIP_TRY(hipStreamSynchronize(s));
  if (h_err) {
    set_error("COO index out of range (row >= num_rows or col >= num_relations*num_nodes)");
    return MRGCN_ERR_RANGE;
  }
  const int64_t nnz = (flags & MRGCN_PLAN_PRUNE_ZEROS) ? (int64_t)h_kept : nnz_in;  // (counted only when pruning)
  p->nnz = nnz;

  MRGCN_HIP_TRY(plan_alloc(p, &p->rowptr, p->num_rows + 1));
  MRGCN_HIP_TRY(plan_alloc(p, &p->lcol, nnz));
  MRGCN_HIP_TRY(plan_alloc(p, &p->ccol, nnz));
  MRGCN_HIP_TRY(plan_alloc(p, &p->rowidx, nnz));
  MRGCN_HIP_TRY(plan_alloc(p, &p->crow, nnz));
  MRGCN_HIP_TRY(plan_alloc(p, &p->cval, nnz));
  MRGCN_HIP_TRY(plan_alloc(p, &p->nptr, N + 1));

  MRGCN_HIP_TRY(sc.alloc(&key2, nnz));
  MRGCN_HIP_TRY(sc.alloc(&key2_s, nnz));
  MRGCN_HIP_TRY(sc.alloc(&eidx, nnz));
  MRGCN_HIP_TRY(sc.alloc(&eidx_s, nnz));
  MRGCN_HIP_TRY(sc.alloc(&head, nnz));
  MRGCN_HIP_TRY(sc.alloc(&cid1, nnz));

  k_lower_bound_ptr<<<nblocks(p->num_rows + 1), kTB, 0, s>>>(keys_s, nnz, p->num_rows, RN, p->rowptr);
  MRGCN_HIP_TRY(hipGetLastError());
  int64_t ncols = 0;
  if (nnz > 0) {
    k_decode<<<nblocks(nnz), kTB, 0, s>>>(keys_s, nnz, RN, N, R, p->rowidx, p->lcol, key2, eidx);
    MRGCN_HIP_TRY(hipGetLastError());
    size_t tb = 0;
    const int end_bit = bits_for(RN);
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, key2, key2_s, eidx, eidx_s, (int)nnz,
                                                     0, end_bit, s));
    char *tmp;
    MRGCN_HIP_TRY(sc.alloc(&tmp, (int64_t)tb));
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, key2, key2_s, eidx, eidx_s, (int)nnz, 0,
                                                     end_bit, s));
    k_heads<<<nblocks(nnz), kTB, 0, s>>>(key2_s, nnz, head);
    MRGCN_HIP_TRY(hipGetLastError());
    size_t tb2 = 0;
    MRGCN_HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, tb2, head, cid1, (int)nnz, s));
    char *tmp2;
    MRGCN_HIP_TRY(sc.alloc(&tmp2, (int64_t)tb2));
    MRGCN_HIP_TRY(hipcub::DeviceScan::InclusiveSum(tmp2, tb2, head, cid1, (int)nnz, s));
    int32_t h_ncols = 0;
    MRGCN_HIP_TRY(hipMemcpyAsync(&h_ncols, cid1 + (nnz - 1), sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MRGCN_HIP_TRY(hipStreamSynchronize(s));
    ncols = h_ncols;
  }
  p->ncols = ncols;
  MRGCN_HIP_TRY(plan_alloc(p, &p->cptr, ncols + 1));
  MRGCN_HIP_TRY(plan_alloc(p, &p->urel, ncols));
  MRGCN_HIP_TRY(plan_alloc(p, &p->unode, ncols));
  MRGCN_HIP_TRY(plan_alloc(p, &p->ulcol, ncols));
  if (nnz > 0) {
    k_scatter_cols<<<nblocks(nnz), kTB, 0, s>>>(key2_s, eidx_s, cid1, head, p->rowidx, p->val, nnz, R,
                                                p->ccol, p->crow, p->cval, p->cptr, p->urel, p->unode,
                                                p->ulcol, N);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  int32_t h_nnz32 = (int32_t)nnz;
  MRGCN_HIP_TRY(hipMemcpyAsync(p->cptr + ncols, &h_nnz32, sizeof(int32_t), hipMemcpyHostToDevice, s));
  k_node_ptr<<<nblocks(N + 1), kTB, 0, s>>>(p->unode, ncols, N, 1, p->nptr);
  MRGCN_HIP_TRY(hipGetLastError());

  // order of the compact columns for the per-relation dense transforms: (node band, relation,
  // node).  Within a band of kNodeBand source nodes the columns are relation-major (one weight
  // tile per group); banding keeps the input rows a group gathers (X / H / dM) inside a window
  // that stays cache resident while every relation of the band is processed, instead of
  // re-streaming the whole input once per relation.  Two orders: wide bands for wide inputs, narrow bands
  // (kNodeBandNarrow) for narrow ones — a band of 40-byte rows then fits one XCD's L2 (common.hpp: RelOrder).
  int64_t band = kNodeBand;
  if (cfg(CFG_NODE_BAND) > 0) band = cfg(CFG_NODE_BAND);
  if (band > N) band = N;
  {
    int rc0 = build_rel_order(p, sc, s, band, true, &p->rperm, &p->relptr, &p->relchunk_rel, &p->relchunk_beg,
                              &p->relchunk_end, &p->relchunk_ptr, &p->relchunk_ids, &p->n_relchunks,
                              &p->max_relchunks, &p->node_band, &p->n_bands);
    if (rc0 != MRGCN_OK) return rc0;
    int64_t nband = kNodeBandNarrow;
    nband = cfg(CFG_NODE_BAND_NARROW);  // (<= 0: no second order)
    if (nband > 0 && nband < band && !(flags & MRGCN_PLAN_LEAN)) {
      rc0 = build_rel_order(p, sc, s, nband, false, &p->n_rperm, &p->n_relptr, &p->n_relchunk_rel, &p->n_relchunk_beg,
                            &p->n_relchunk_end, &p->n_relchunk_ptr, &p->n_relchunk_ids, &p->n_n_relchunks,
                            &p->n_max_relchunks, &p->n_node_band, &p->n_n_bands);
      if (rc0 != MRGCN_OK) return rc0;
    }
  }

  int rc;
  if ((rc = build_long(p, p->cptr, p->ncols, s, &p->c_long_row, &p->c_long_cptr, &p->c_chunk_beg,
                       &p->c_chunk_end, &p->c_chunk_row, &p->c_n_long, &p->c_n_chunks, &p->max_col_nnz)))
    return rc;
  if (flags & MRGCN_PLAN_LEAN) {
    // rows and operand keep their own orders: the COMPACT view is the row-major compact CSR itself
    p->lean = true;
    MRGCN_HIP_TRY(plan_alloc(p, &p->rowmap, p->num_rows));
    MRGCN_HIP_TRY(plan_alloc(p, &p->mpos, ncols));
    if (p->num_rows > 0) k_iota_i32<<<nblocks(p->num_rows), kTB, 0, s>>>(p->rowmap, p->num_rows);
    if (ncols > 0) k_iota_i32<<<nblocks(ncols), kTB, 0, s>>>(p->mpos, ncols);
    p->ptr3 = p->rowptr;
    p->mcol = p->ccol;
    p->mval = p->val;
    p->n_op = ncols;
    MRGCN_HIP_TRY(plan_alloc(p, &p->rnode, ncols));
    MRGCN_HIP_TRY(plan_alloc(p, &p->rmpos, ncols));
    if (ncols > 0) {
      k_gather_i32<<<nblocks(ncols), kTB, 0, s>>>(p->unode, p->rperm, ncols, p->rnode);
      k_gather_i32<<<nblocks(ncols), kTB, 0, s>>>(p->mpos, p->rperm, ncols, p->rmpos);
    }
    MRGCN_HIP_TRY(hipGetLastError());
    if ((rc = build_long(p, p->rowptr, p->num_rows, s, &p->r_long_row, &p->r_long_cptr, &p->r_chunk_beg,
                         &p->r_chunk_end, &p->r_chunk_row, &p->r_n_long, &p->r_n_chunks, &p->max_row_nnz)))
      return rc;
    p->q_long_row = p->r_long_row; p->q_long_cptr = p->r_long_cptr; p->q_chunk_beg = p->r_chunk_beg;
    p->q_chunk_end = p->r_chunk_end; p->q_chunk_row = p->r_chunk_row;
    p->q_n_long = p->r_n_long; p->q_n_chunks = p->r_n_chunks;
    const int64_t ws_l = (int64_t)std::max(p->r_n_chunks, p->c_n_chunks) * kWsFeatures;
    MRGCN_HIP_TRY(plan_alloc(p, &p->partials, ws_l));
    p->partials_floats = std::max<int64_t>(ws_l, 1);
    return MRGCN_OK;
  }
  // class-major processing order of the rows (COMPACT view)
  MRGCN_HIP_TRY(plan_alloc(p, &p->rowmap, p->num_rows));
  MRGCN_HIP_TRY(plan_alloc(p, &p->ptr3, p->num_rows + 1));
  int32_t *rank3 = nullptr;
  MRGCN_HIP_TRY(sc.alloc(&rank3, p->num_rows));
  {
    const int64_t rows = p->num_rows;
    int64_t *ck, *ck_s;
    int32_t *ids, *len3, *cls_ptr;
    MRGCN_HIP_TRY(sc.alloc(&ck, rows));
    MRGCN_HIP_TRY(sc.alloc(&ck_s, rows));
    MRGCN_HIP_TRY(sc.alloc(&ids, rows));
    MRGCN_HIP_TRY(sc.alloc(&len3, rows + 1));
    MRGCN_HIP_TRY(sc.alloc(&cls_ptr, 4));
    // length-sorted windows inside the S and M classes (kLenWindowS / kLenWindowM row ids; 0 = plain id order)
    const int64_t win_s = kLenWindowS, win_m = kLenWindowM;
    const int64_t wmin = std::min(win_s > 0 ? win_s : rows + 1, win_m > 0 ? win_m : rows + 1);
    const int64_t cstride = ((rows + wmin - 1) / wmin + 1) * 64 * std::max<int64_t>(rows, 1);
    if (rows > 0) {
      k_class_keys<<<nblocks(rows), kTB, 0, s>>>(p->rowptr, rows, kShort3Rows, kMid3Rows, win_s, win_m, cstride, ck, ids);
      MRGCN_HIP_TRY(hipGetLastError());
      size_t tb = 0;
      const int eb = bits_for(3 * cstride + cstride);
      MRGCN_REQUIRE(eb <= 62, "graph too large for the row-order key");
      MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, ck, ck_s, ids, p->rowmap, (int)rows, 0, eb, s));
      char *tmp;
      MRGCN_HIP_TRY(sc.alloc(&tmp, (int64_t)tb));
      MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, ck, ck_s, ids, p->rowmap, (int)rows, 0, eb, s));
      k_invert_perm<<<nblocks(rows), kTB, 0, s>>>(p->rowmap, rows, rank3);
      MRGCN_HIP_TRY(hipGetLastError());
    }
    k_lower_bound_ptr<<<1, kTB, 0, s>>>(ck_s, rows, 3, cstride, cls_ptr);  // first rank of each class
    k_rank_len<<<nblocks(rows + 1), kTB, 0, s>>>(p->rowptr, p->rowmap, rows, len3);
    MRGCN_HIP_TRY(hipGetLastError());
    int rc2;
    if ((rc2 = exclusive_scan_i32(len3, p->ptr3, rows + 1, s, sc))) return rc2;
    int32_t h[4] = {0, 0, 0, 0};
    MRGCN_HIP_TRY(hipMemcpyAsync(h, cls_ptr, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MRGCN_HIP_TRY(hipStreamSynchronize(s));
    p->n_short3 = h[1];
    p->n_mid3 = h[2] - h[1];
  }
  // storage order of M
  MRGCN_HIP_TRY(plan_alloc(p, &p->mpos, ncols));
  MRGCN_HIP_TRY(plan_alloc(p, &p->mcol, nnz));
  MRGCN_HIP_TRY(plan_alloc(p, &p->mval, nnz));
  if (ncols > 0) {
    int64_t *mk, *mk_s;
    int32_t *ids, *order, *n_hot_d = nullptr;
    MRGCN_HIP_TRY(sc.alloc(&mk, ncols));
    MRGCN_HIP_TRY(sc.alloc(&mk_s, ncols));
    MRGCN_HIP_TRY(sc.alloc(&ids, ncols));
    MRGCN_HIP_TRY(sc.alloc(&order, ncols));
    const int shift = bits_for(ncols);
    const int64_t max_count = p->max_col_nnz + 1;
    int hot_min = kHotMinRefs;
    if (cfg(CFG_HOT_MIN) > 1 || cfg(CFG_HOT_MIN) < 0) hot_min = (int)cfg(CFG_HOT_MIN);
    k_mpos_keys<<<nblocks(ncols), kTB, 0, s>>>(p->cptr, p->crow, rank3, ncols, max_count, shift, hot_min, mk, ids);
    MRGCN_HIP_TRY(hipGetLastError());
    const int end_bit = shift + bits_for(max_count + 1 + p->num_rows);
    MRGCN_REQUIRE(end_bit <= 62, "graph too large for the operand-order key");
    size_t tb = 0;
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, mk, mk_s, ids, order, (int)ncols, 0,
                                                     end_bit, s));
    char *tmp;
    MRGCN_HIP_TRY(sc.alloc(&tmp, (int64_t)tb));
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, mk, mk_s, ids, order, (int)ncols, 0,
                                                     end_bit, s));
    {
      const bool swap_on = cfg(CFG_AVOID_STRADDLE) != 0;
      const StraddleSizes sz = hint;
      // hot columns are the sorted keys below max_count << shift
      MRGCN_HIP_TRY(sc.alloc(&n_hot_d, 2));
      k_lower_bound_ptr<<<1, kTB, 0, s>>>(mk_s, ncols, 1, max_count << shift, n_hot_d);
      if (swap_on && ncols >= 32)
        k_avoid_straddle<<<nblocks(ncols / 32), kTB, 0, s>>>(order, p->cptr, ncols, sz, n_hot_d);
    }
    k_invert_perm<<<nblocks(ncols), kTB, 0, s>>>(order, ncols, p->mpos);
    MRGCN_HIP_TRY(hipGetLastError());
    // the COMPACT view walks the rows in class-major order (ptr3 / rowmap) and a row's entries in rising
    // operand position (its private, single-use operand rows are then one sequential run): own index +
    // value arrays in that order
    int32_t *mcol_u;
    int64_t *rk, *rk_s;
    MRGCN_HIP_TRY(sc.alloc(&mcol_u, nnz));
    MRGCN_HIP_TRY(sc.alloc(&rk, nnz));
    MRGCN_HIP_TRY(sc.alloc(&rk_s, nnz));
    k_gather_i32<<<nblocks(nnz), kTB, 0, s>>>(p->mpos, p->ccol, nnz, mcol_u);
    MRGCN_HIP_TRY(hipGetLastError());
    k_row_pos_keys<<<nblocks(nnz), kTB, 0, s>>>(p->rowidx, rank3, mcol_u, nnz, ncols, rk);
    MRGCN_HIP_TRY(hipGetLastError());
    {
      const int eb = bits_for(p->num_rows * ncols + ncols);
      size_t tb2 = 0;
      MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, rk, rk_s, p->val, p->mval, (int)nnz,
                                                       0, eb, s));
      char *tmp2;
      MRGCN_HIP_TRY(sc.alloc(&tmp2, (int64_t)tb2));
      MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp2, tb2, rk, rk_s, p->val, p->mval, (int)nnz, 0,
                                                       eb, s));
    }
    k_keys_to_pos<<<nblocks(nnz), kTB, 0, s>>>(rk_s, nnz, ncols, p->mcol);
    MRGCN_HIP_TRY(hipGetLastError());
    p->n_op = ncols;
    if (replicate && hot_min > 1) {
      // every entry of a non-hot column gets its own operand row, in processing order
      int32_t *flag, *spos, *ecol, *mpos_new, *rflag, *rpos;
      MRGCN_HIP_TRY(sc.alloc(&flag, nnz + 1));
      MRGCN_HIP_TRY(sc.alloc(&spos, nnz + 1));
      MRGCN_HIP_TRY(sc.alloc(&ecol, nnz));
      MRGCN_HIP_TRY(sc.alloc(&mpos_new, ncols));
      MRGCN_HIP_TRY(sc.alloc(&rflag, nnz + 1));
      MRGCN_HIP_TRY(sc.alloc(&rpos, nnz + 1));
      int32_t h2[2] = {0, 0};
      MRGCN_HIP_TRY(hipMemcpyAsync(h2, n_hot_d, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
      MRGCN_HIP_TRY(hipStreamSynchronize(s));
      const int32_t n_hot = h2[1];
      k_is_stream<<<nblocks(nnz + 1), kTB, 0, s>>>(p->mcol, nnz, n_hot, flag);
      int rc2;
      if ((rc2 = exclusive_scan_i32(flag, spos, nnz + 1, s, sc))) return rc2;
      k_fill_i32<<<nblocks(ncols), kTB, 0, s>>>(mpos_new, ncols, INT32_MAX);
      k_rep_assign<<<nblocks(nnz), kTB, 0, s>>>(p->mcol, spos, nnz, n_hot, order, ecol, mpos_new);
      k_rep_flag<<<nblocks(nnz + 1), kTB, 0, s>>>(p->mcol, ecol, mpos_new, nnz, n_hot, rflag);
      MRGCN_HIP_TRY(hipGetLastError());
      if ((rc2 = exclusive_scan_i32(rflag, rpos, nnz + 1, s, sc))) return rc2;
      int32_t h3[2] = {0, 0};
      MRGCN_HIP_TRY(hipMemcpyAsync(&h3[0], spos + nnz, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      MRGCN_HIP_TRY(hipMemcpyAsync(&h3[1], rpos + nnz, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      MRGCN_HIP_TRY(hipStreamSynchronize(s));
      p->n_op = (int64_t)n_hot + h3[0];
      p->n_rep = h3[1];
      MRGCN_HIP_TRY(plan_alloc(p, &p->rep_src, p->n_rep));
      MRGCN_HIP_TRY(plan_alloc(p, &p->rep_dst, p->n_rep));
      k_rep_fill<<<nblocks(nnz), kTB, 0, s>>>(p->mcol, ecol, mpos_new, rflag, rpos, nnz, p->rep_src, p->rep_dst);
      MRGCN_HIP_TRY(hipMemcpyAsync(p->mpos, mpos_new, ncols * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
      MRGCN_HIP_TRY(hipGetLastError());
      MRGCN_HIP_TRY(hipStreamSynchronize(s));
    }
  }
  // relation-major copies of the per-column indices (no dependent index chain in the transforms)
  MRGCN_HIP_TRY(plan_alloc(p, &p->rnode, ncols));
  MRGCN_HIP_TRY(plan_alloc(p, &p->rmpos, ncols));
  if (ncols > 0) {
    k_gather_i32<<<nblocks(ncols), kTB, 0, s>>>(p->unode, p->rperm, ncols, p->rnode);
    k_gather_i32<<<nblocks(ncols), kTB, 0, s>>>(p->mpos, p->rperm, ncols, p->rmpos);
    if (p->n_rperm) {
      MRGCN_HIP_TRY(plan_alloc(p, &p->n_rnode, ncols));
      MRGCN_HIP_TRY(plan_alloc(p, &p->n_rmpos, ncols));
      k_gather_i32<<<nblocks(ncols), kTB, 0, s>>>(p->unode, p->n_rperm, ncols, p->n_rnode);
      k_gather_i32<<<nblocks(ncols), kTB, 0, s>>>(p->mpos, p->n_rperm, ncols, p->n_rmpos);
    }
    MRGCN_HIP_TRY(hipGetLastError());
  }
  // the operand rows' (node, relation), for the narrow transform that writes the operand as one stream
  if (!(flags & MRGCN_PLAN_LEAN) && ncols > 0) {
    MRGCN_HIP_TRY(plan_alloc(p, &p->op_node, p->n_op));
    MRGCN_HIP_TRY(plan_alloc(p, &p->op_rel, p->n_op));
    MRGCN_HIP_TRY(hipMemsetAsync(p->op_node, 0xff, (size_t)p->n_op * sizeof(int32_t), s));
    MRGCN_HIP_TRY(hipMemsetAsync(p->op_rel, 0, (size_t)p->n_op * sizeof(int32_t), s));
    k_operand_ids<<<nblocks(ncols), kTB, 0, s>>>(p->mpos, p->unode, p->urel, ncols, p->op_node, p->op_rel);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  if ((rc = build_long(p, p->rowptr, p->num_rows, s, &p->r_long_row, &p->r_long_cptr, &p->r_chunk_beg,
                       &p->r_chunk_end, &p->r_chunk_row, &p->r_n_long, &p->r_n_chunks, &p->max_row_nnz)))
    return rc;
  // the COMPACT view's own split-row descriptors, over ranks: for k_spmm (chunks of kChunk) and for k_spmm3
  // (rows beyond kMid3Rows in chunks of kChunk3Entries, at most kChunk3Cap per row).  Descriptor "rows" are ranks.
  {
    int64_t dummy = 0;
    // chunk size of k_spmm's split rows on this view: 512 entries of 40-byte operand rows are 20 KB per wave, 512
    // entries of the 800-byte rows of a wide layer (FB15k-237: F = 200) are 410 KB walked by ONE wave while most of
    // the chip idles (634 k entries / 512 = 1 240 waves for 1 024 SIMDs).  A plan hinted with wide operand rows cuts
    // its long rows into chunks of 64 entries: ~10 k waves of 51 KB each.
    int wide_bytes = 0;
    for (int k = 0; k < hint.n; ++k) wide_bytes = std::max(wide_bytes, (int)hint.bytes[k]);
    const int q_chunk = wide_bytes >= 256 ? 64 : kChunk;
    if ((rc = build_long(p, p->ptr3, p->num_rows, s, &p->q_long_row, &p->q_long_cptr, &p->q_chunk_beg,
                         &p->q_chunk_end, &p->q_chunk_row, &p->q_n_long, &p->q_n_chunks, &dummy, kLongThreshold,
                         q_chunk)))
      return rc;
    // k_spmm3: rows of kMid3Rows < len <= kChunk3Entries are one chunk = one wave each (r3s_*); longer rows are cut
    // blockwise (r3_*: 4 * nb equal chunks, a block of four waves adds its four sums in LDS)
    int32_t *lr = nullptr, *lc = nullptr;
    int32_t nl = 0;
    if ((rc = build_long(p, p->ptr3, p->num_rows, s, &lr, &lc, &p->r3s_chunk_beg, &p->r3s_chunk_end,
                         &p->r3s_chunk_row, &nl, &p->r3s_n_chunks, &dummy, kMid3Rows, kChunk3Entries, 0,
                         kChunk3Entries, 0)))
      return rc;
    p->r3s_long_row = lr; p->r3s_long_cptr = lc;
    if ((rc = build_long(p, p->ptr3, p->num_rows, s, &p->r3_long_row, &p->r3_long_cptr, &p->r3_chunk_beg,
                         &p->r3_chunk_end, &p->r3_chunk_row, &p->r3_n_long, &p->r3_n_chunks, &dummy, kChunk3Entries,
                         kChunk3Entries, kChunk3Cap, 0, 1)))
      return rc;
    // one arrival counter per blockwise row (k_spmm3: the block that brings a row's last partial sum adds them up);
    // zero between launches — the last arriver puts its row's counter back
    // (and one per long row of the other views: the general product finishes its split rows the same way)
    p->ticket_ints = std::max<int64_t>(std::max<int64_t>(std::max(p->r3_n_long, p->r_n_long), std::max(p->q_n_long, p->c_n_long)), 1);
    MRGCN_HIP_TRY(plan_alloc(p, &p->r3_ticket, p->ticket_ints));
    MRGCN_HIP_TRY(hipMemsetAsync(p->r3_ticket, 0, (size_t)p->ticket_ints * sizeof(int32_t), s));
    MRGCN_HIP_TRY(plan_alloc(p, &p->work_tickets, (int64_t)kWorkTickets * kWorkTicketStride));
    MRGCN_HIP_TRY(hipMemsetAsync(p->work_tickets, 0, (size_t)kWorkTickets * kWorkTicketStride * sizeof(unsigned long long), s));
  }
  {  // the rows k_spmm3 leaves partial sums of (more than one block): the two-pass form's finalize launches one wave for each
    const int64_t nl = p->r3_n_long;
    int32_t *flag, *pos;
    MRGCN_HIP_TRY(sc.alloc(&flag, nl + 1));
    MRGCN_HIP_TRY(sc.alloc(&pos, nl + 1));
    k_multi_flag<<<nblocks(nl + 1), kTB, 0, s>>>(p->r3_long_cptr, nl, 4, flag);
    MRGCN_HIP_TRY(hipGetLastError());
    int rc2;
    if ((rc2 = exclusive_scan_i32(flag, pos, nl + 1, s, sc))) return rc2;
    int32_t h = 0;
    MRGCN_HIP_TRY(hipMemcpyAsync(&h, pos + nl, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MRGCN_HIP_TRY(hipStreamSynchronize(s));
    p->r3_n_multi = h;
    MRGCN_HIP_TRY(plan_alloc(p, &p->r3_multi, h));
    if (h > 0) k_multi_fill<<<nblocks(nl), kTB, 0, s>>>(flag, pos, nl, p->r3_multi);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  int64_t ws = (int64_t)std::max(std::max(p->r_n_chunks, p->q_n_chunks), p->c_n_chunks) * kWsFeatures;
  ws = std::max<int64_t>(ws, (int64_t)p->r3_n_chunks * 16);
  if (!p->lean) ws = std::max<int64_t>(ws, seg_scratch_floats(p->nnz));  // the entry-sliced transposed product's records
  MRGCN_HIP_TRY(plan_alloc(p, &p->partials, ws));
  p->partials_floats = std::max<int64_t>(ws, 1);
  return MRGCN_OK;
}

// hands every array of the plan back to its device's pool (`ep`: a wait that covers all users of the plan has completed:
// pool_free's covered_epoch) and deletes the descriptor
void release_plan(mrgcn_plan *q, uint64_t ep) {
  if (q->lean) {  // aliases of rowptr / ccol / val / r_*: one owner each
    q->ptr3 = nullptr; q->mcol = nullptr; q->mval = nullptr;
    q->q_long_row = q->q_long_cptr = q->q_chunk_beg = q->q_chunk_end = q->q_chunk_row = nullptr;
  }
  void *ptrs[] = {q->rowptr, q->lcol, q->ccol, q->rowidx, q->val, q->cptr, q->crow, q->urel, q->unode,
                  q->nptr, q->ulcol, q->mpos, q->mcol, q->mval, q->rperm, q->relptr, q->rnode, q->rmpos, q->relchunk_ptr, q->relchunk_ids, q->relchunk_rel, q->relchunk_beg, q->relchunk_end,
                  q->cval, q->r_long_row, q->r_long_cptr, q->r_chunk_beg, q->r_chunk_end,
                  q->c_long_row, q->c_long_cptr, q->c_chunk_beg, q->c_chunk_end, q->r_chunk_row, q->c_chunk_row,
                  q->r3_long_row, q->r3_long_cptr, q->r3_chunk_beg, q->r3_chunk_end, q->r3_chunk_row,
                  q->q_long_row, q->q_long_cptr, q->q_chunk_beg, q->q_chunk_end, q->q_chunk_row, q->rowmap, q->ptr3,
                  q->rep_src, q->rep_dst, q->partials, q->r3_multi, q->r3_ticket,
                  q->r3s_long_row, q->r3s_long_cptr, q->r3s_chunk_beg, q->r3s_chunk_end, q->r3s_chunk_row,
                  q->n_rperm, q->n_relptr, q->n_rnode, q->n_rmpos, q->n_relchunk_rel, q->n_relchunk_beg,
                  q->n_relchunk_end, q->n_relchunk_ptr, q->n_relchunk_ids, q->op_node, q->op_rel, q->mlcol,
                  q->work_tickets, q->ecol};
  // (after the wait any stream may take the blocks; the plan's own build stream is where the next build of a
  // similar slice will ask for them again: the pool hands them back without a driver call)
  for (void *a : ptrs) pool_free(a, q->build_stream, ep);
  for (size_t i = 1; i < q->stream_scratch.size(); ++i) {  // (set 0 is `partials` / `r3_ticket` above)
    pool_free(q->stream_scratch[i].partials, q->build_stream, ep);
    pool_free(q->stream_scratch[i].ticket, q->build_stream, ep);
  }
  delete q;
}

void free_plan(mrgcn_plan *p) {
  // the caller guarantees nothing that uses the plan is still to be SUBMITTED; work already in flight on any stream of
  // the plan's device is waited for (what hipFree did implicitly), then the blocks go back to that device's pool.
  // A plan dropped while a stream capture is under way (the wait is not allowed then) is parked and released by the
  // next plan that is created or destroyed outside a capture.
  static std::vector<mrgcn_plan *> parked;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int cur = 0;
  (void)hipGetDevice(&cur);
  parked.push_back(p);
  std::vector<mrgcn_plan *> keep;
  for (mrgcn_plan *q : parked) {
    const int qdev = q->device;
    if (qdev != cur) (void)hipSetDevice(qdev);
    const uint64_t ep = pool_sync_begin();
    if (hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      keep.push_back(q);
    } else {
      pool_sync_done(ep);
      release_plan(q, ep);
    }
    if (qdev != cur) (void)hipSetDevice(cur);
  }
  parked.swap(keep);
}

// the same without the device-wide wait: the caller names an event recorded behind the LAST work that uses the plan
// (on whichever stream); the host waits for that event and for the plan's own build stream only — other streams of the
// device keep running (a batch prefetcher drops the plans of finished steps while the next batch is being built)
int free_plan_after(mrgcn_plan *p, hipEvent_t ev) {
  int cur = 0;
  (void)hipGetDevice(&cur);
  const int qdev = p->device;
  if (qdev != cur) (void)hipSetDevice(qdev);
  hipError_t e = ev ? hipEventSynchronize(ev) : hipSuccess;
  if (e == hipSuccess) e = hipStreamSynchronize(p->build_stream);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (qdev != cur) (void)hipSetDevice(cur);
    free_plan(p);  // (e.g. a capture under way: the parking path)
    return MRGCN_OK;
  }
  // every user of THESE blocks has finished (nothing is said about other parked blocks: no epoch moves)
  release_plan(p, kSafeNow);
  if (qdev != cur) (void)hipSetDevice(cur);
  return MRGCN_OK;
}


// ---------------------------------------------------------------------------------------------
// Gradient support (common.hpp: mrgcn_support): one-off build per (plan, set of live output rows)
// ---------------------------------------------------------------------------------------------
// out[i] = flags[idx ? idx[i] : i] != 0 for i < n, out[n] = 0 (so that an exclusive scan over n + 1 ends in the total)
__global__ void k_sup_flags_i32(const uint8_t *__restrict__ flags, const int32_t *__restrict__ idx, int64_t n,
                                int32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = flags[idx ? idx[i] : i] != 0;
  else if (i == n) out[i] = 0;
}
// the compact columns of a node are contiguous, so are its live ones: nlptr[j] = lpos[nptr[j]]
__global__ void k_sup_nodes(const int32_t *__restrict__ nptr, const int32_t *__restrict__ lpos, int64_t N,
                            int32_t *__restrict__ nlptr, uint8_t *__restrict__ node_flags,
                            int32_t *__restrict__ nflag_i32) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > N) return;
  const int32_t a = lpos[nptr[j]];
  nlptr[j] = a;
  if (j < N) {
    const int32_t live = lpos[nptr[j + 1]] > a;
    node_flags[j] = (uint8_t)live;
    nflag_i32[j] = live;
  } else {
    nflag_i32[j] = 0;
  }
}
__global__ void k_sup_lnodes(const int32_t *__restrict__ nflag_i32, const int32_t *__restrict__ npos, int64_t N,
                             const int32_t *__restrict__ nlptr, int32_t *__restrict__ lnode,
                             int32_t *__restrict__ lnptr) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > N) return;
  if (j == N) { lnptr[npos[N]] = nlptr[N]; return; }
  if (!nflag_i32[j]) return;
  lnode[npos[j]] = (int32_t)j;
  lnptr[npos[j]] = nlptr[j];
}
// live columns in a relation-major order of the plan: position i of `rperm` -> live position rpos[i]
__global__ void k_sup_rfill(const int32_t *__restrict__ rperm, const int32_t *__restrict__ rnode,
                            const int32_t *__restrict__ rflag, const int32_t *__restrict__ rpos, int64_t ncols,
                            const int32_t *__restrict__ lpos, int32_t *__restrict__ lperm, int32_t *__restrict__ lrin) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncols || !rflag[i]) return;
  lperm[rpos[i]] = lpos[rperm[i]];
  lrin[rpos[i]] = rnode[i];
}

}  // namespace

namespace {
__global__ void k_fill_bytes(uint8_t *__restrict__ dst, uint32_t word, size_t head, size_t words, size_t tail) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < words) reinterpret_cast<uint32_t *>(dst + head)[t] = word;
  if (t < head) dst[t] = (uint8_t)word;
  if (t < tail) dst[head + 4 * words + t] = (uint8_t)word;
}
}  // namespace

hipError_t raise_lds_limit(const void *fn, size_t lds) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, size_t> allowed;
  if (lds <= 48 * 1024) return hipSuccess;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> g(mu);
  size_t &a = allowed[(uint64_t)(uintptr_t)fn * 64 + (uint64_t)(dev & 63)];
  if (a == 0) a = 48 * 1024;
  if (lds <= a) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) a = lds;
  return e;
}

hipError_t fill_async(void *dst, int byte_value, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  // (MRGCN_DEBUG_CAPTURED_MEMSET=1: the runtime's memset also inside a capture — tools/memset_node_repro.py shows the fault)
  const bool raw = cfg(CFG_DEBUG_CAPTURED_MEMSET) != 0;
  if (raw || hipStreamIsCapturing(s, &st) != hipSuccess || st == hipStreamCaptureStatusNone)
    return hipMemsetAsync(dst, byte_value, bytes, s);
  const uint32_t b = (uint32_t)(byte_value & 0xff);
  size_t head = (4 - ((uintptr_t)dst & 3)) & 3;
  if (head > bytes) head = bytes;
  const size_t words = (bytes - head) / 4, tail = bytes - head - 4 * words;
  const size_t n = words > 4 ? words : 4;
  k_fill_bytes<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>((uint8_t *)dst, b * 0x01010101u, head, words, tail);
  return hipGetLastError();
}

namespace {
// mlcol[e] = literal column of entry e of the COMPACT view: (relation, node) of the operand row it reads
__global__ void k_literal_cols(const int32_t *__restrict__ mcol, const int32_t *__restrict__ op_node,
                               const int32_t *__restrict__ op_rel, int64_t nnz, int64_t N,
                               int32_t *__restrict__ mlcol) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  const int32_t pos = mcol[e];
  mlcol[e] = (int32_t)((int64_t)op_rel[pos] * N + op_node[pos]);
}
}  // namespace

// p->mlcol (common.hpp), built by the first caller outside a capture; false: not available for this call
bool plan_literal_cols(const mrgcn_plan *p, hipStream_t s) {
  std::lock_guard<std::mutex> lock(p->scratch_mu);
  if (p->mlcol) return true;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
  int32_t *a = nullptr;
  if (pool_alloc((void **)&a, (size_t)p->nnz * sizeof(int32_t), s) != hipSuccess) return false;
  k_literal_cols<<<dim3((unsigned)((p->nnz + 255) / 256)), dim3(256), 0, s>>>(p->mcol, p->op_node, p->op_rel, p->nnz,
                                                                             p->num_nodes, a);
  // (products on other streams may read it next: one wait, once per plan)
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    pool_free(a, s);
    return false;
  }
  p->mlcol = a;
  return true;
}

__global__ void k_entry_cols(const int32_t *__restrict__ cptr, int64_t ncols, int32_t *__restrict__ ecol) {
  // one thread per column: its (few) entries get its id; long columns are written by their thread alone — a column of
  // 100 k entries is rare enough not to matter for a once-per-plan pass
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (int64_t)gridDim.x * blockDim.x)
    for (int32_t e = cptr[c]; e < cptr[c + 1]; ++e) ecol[e] = (int32_t)c;
}

bool plan_entry_cols(const mrgcn_plan *p, hipStream_t s) {
  std::lock_guard<std::mutex> lock(p->scratch_mu);
  if (p->ecol) return true;
  if (p->lean || p->nnz == 0 || !p->cptr) return false;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
  int32_t *a = nullptr;
  if (pool_alloc((void **)&a, (size_t)p->nnz * sizeof(int32_t), s) != hipSuccess) return false;
  k_entry_cols<<<dim3((unsigned)std::min<int64_t>((p->ncols + 255) / 256, 65535)), dim3(256), 0, s>>>(p->cptr, p->ncols, a);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    pool_free(a, s);
    return false;
  }
  p->ecol = a;
  return true;
}

int plan_scratch(const mrgcn_plan *p, hipStream_t s, float **partials, int32_t **ticket) {
  std::lock_guard<std::mutex> lock(p->scratch_mu);
  for (const auto &e : p->stream_scratch)
    if (e.stream == s) {
      *partials = e.partials;
      *ticket = e.ticket;
      return MRGCN_OK;
    }
  if (p->stream_scratch.empty()) {  // the set the plan was built with
    p->stream_scratch.push_back({s, p->partials, p->r3_ticket});
    *partials = p->partials;
    *ticket = p->r3_ticket;
    return MRGCN_OK;
  }
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
    set_error("the first product of a plan on a new stream allocates that stream's scratch: run one product on this "
              "stream before capturing it");
    return MRGCN_ERR_INVALID;
  }
  float *pa = nullptr;
  int32_t *ti = nullptr;
  MRGCN_HIP_TRY(pool_alloc((void **)&pa, (size_t)std::max<int64_t>(p->partials_floats, 1) * sizeof(float), s));
  const size_t tb = (size_t)std::max<int64_t>(p->ticket_ints, 1) * sizeof(int32_t);
  MRGCN_HIP_TRY(pool_alloc((void **)&ti, tb, s));
  MRGCN_HIP_TRY(hipMemsetAsync(ti, 0, tb, s));  // arrival counters start at zero (and return to zero after every launch)
  p->stream_scratch.push_back({s, pa, ti});
  *partials = pa;
  *ticket = ti;
  return MRGCN_OK;
}

namespace {

// out[k] = npos[node[k]] (rank of a node among the live nodes)
__global__ void k_sup_node_ord(const int32_t *__restrict__ node, const int32_t *__restrict__ idx,
                               const int32_t *__restrict__ npos, int64_t n, int32_t *__restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = npos[node[idx ? idx[k] : k]];
}

template <typename T> hipError_t sup_alloc(mrgcn_support *q, T **dst, int64_t n) {
  size_t bytes = (size_t)std::max<int64_t>(n, 1) * sizeof(T);
  hipError_t e = pool_alloc((void **)dst, bytes, q->build_stream);
  if (e == hipSuccess) {
    q->device_bytes += (int64_t)bytes;
    q->owned.push_back(*dst);
  }
  return e;
}

// ---- the build in two stages with ONE host wait between them -------------------------------------------------------
// Stage 1 needs nothing from the host: flags, scans and counts leave every size the arrays of stage 2 need — live
// columns / kept entries / live nodes / flagged rows, the split-row totals of both views, the live columns per (band,
// relation) group of both orders — in one small block that is copied to pinned host memory.  The row set of the next
// support of a chain (a mini-batch's next layer) is this one's NODE_FLAGS, on the device after stage 1: a chain queues
// every level's stage 1, waits once, and runs every stage 2.
// Work is proportional to the entries of the FLAGGED ROWS (E) plus a few passes over the plan's columns / nodes / rows:
// the entries are walked row-major (the flagged rows' CSR ranges), the kept entries per column come from a histogram,
// and the transposed arrays from one stable radix sort of the E (live column, entry) pairs — rows keep rising inside a
// column, the plan's own entry order.  (Going through the live COLUMNS instead walks every entry of a hub column —
// rdf:type with 10^5 entries is live for almost any sample: 0.2-0.4 ms per pass at the AM/4 shape; and the
// one-array-at-a-time build before that waited eight times per support: 3.2 ms for a mini-batch level.)
struct SupStage {
  mrgcn_support *q = nullptr;
  const uint8_t *row_flags = nullptr;
  bool forward = false;
  Scratch sc;
  int32_t *lpos = nullptr, *ckept = nullptr, *nflag = nullptr, *npos = nullptr;
  int32_t *rpos_rows = nullptr;
  struct Ord { int32_t *rflag = nullptr, *rpos = nullptr, *gptr = nullptr; int64_t ngroups = 0; } ow, on;
  int32_t *totals_d = nullptr;
  int64_t land_off = 0;  // this level's block in the pinned landing area (ints): totals[16] | gptr wide | gptr narrow
  int32_t *blk = nullptr;  // the orders' chunk lists on the device, at these offsets
  size_t offs_w[5] = {0, 0, 0, 0, 0}, offs_n[5] = {0, 0, 0, 0, 0};
};
enum { kTotL = 0, kTotE, kTotNL, kTotNR, kTotTLong, kTotTChunks, kTotFLong, kTotFChunks, kTotCount = 16 };

// per row: flag as an int, entries if flagged (position rows: 0), and the split-row totals of the forward view
__global__ void k_sup_rowcount(const uint8_t *__restrict__ row_flags, int64_t rows, const int32_t *__restrict__ rowptr,
                               int thresh, int chunk, int32_t *__restrict__ rflag, int32_t *__restrict__ rlen,
                               int32_t *__restrict__ totals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool on = i < rows && row_flags[i] != 0;
  const int32_t len = on ? rowptr[i + 1] - rowptr[i] : 0;
  if (i <= rows) {
    rflag[i] = on;
    rlen[i] = len;
  }
  const bool lg = len > thresh;
  const int32_t nlong = __popcll(__ballot(lg));
  int32_t nch = lg ? (len + chunk - 1) / chunk : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nch += __shfl_xor(nch, o, kWave);
  // one pair of global adds per BLOCK: the sample of a lower mini-batch layer is mostly hubs (thousands of long rows:
  // a pair of same-address atomics per wave was 28 of this kernel's 33 us)
  __shared__ int32_t s_tot[2];
  if (threadIdx.x < 2) s_tot[threadIdx.x] = 0;
  __syncthreads();
  if (lane == 0 && nlong > 0) {
    atomicAdd(&s_tot[0], nlong);
    atomicAdd(&s_tot[1], nch);
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_tot[0] > 0) {
    atomicAdd(&totals[kTotFLong], s_tot[0]);
    atomicAdd(&totals[kTotFChunks], s_tot[1]);
  }
}
// compact list of the flagged rows (arrays sized for every row): rank / id / entry range
__global__ void k_sup_rows2(const uint8_t *__restrict__ row_flags, const int32_t *__restrict__ rpos, int64_t rows,
                            const int32_t *__restrict__ funp, int32_t *__restrict__ rowrank,
                            int32_t *__restrict__ frow, int32_t *__restrict__ fptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > rows) return;
  if (i == rows) { fptr[rpos[i]] = funp[i]; return; }
  const bool on = row_flags[i] != 0;
  rowrank[i] = on ? rpos[i] : -1;
  if (on) {
    frow[rpos[i]] = (int32_t)i;
    fptr[rpos[i]] = funp[i];
  }
}
// the flagged row (by rank) of forward entry t: one search per wave for its first entry, then a short walk per lane
__device__ __forceinline__ int32_t sup_row_of(const int32_t *__restrict__ fptr, int32_t NR, int32_t t, int32_t t_wave0) {
  int32_t lo = 0, hi = NR;  // last q with fptr[q] <= t_wave0
  while (hi - lo > 1) {
    const int32_t mid = (lo + hi) >> 1;
    if (fptr[mid] <= t_wave0) lo = mid; else hi = mid;
  }
  int32_t q = lo;
  while (q + 1 < NR && fptr[q + 1] <= t) ++q;
  return q;
}
// stage 1, over the forward entries (count and rows read from device memory: grid-stride): marks the touched columns
// and counts the kept entries of each
__global__ void k_sup_fwd_mark(const int32_t *__restrict__ frow, const int32_t *__restrict__ fptr,
                               const int32_t *__restrict__ rpos_rows, int64_t rows, const int32_t *__restrict__ rowptr,
                               const int32_t *__restrict__ ccol, uint8_t *__restrict__ col_flags,
                               int32_t *__restrict__ ccnt) {
  const int32_t NR = rpos_rows[rows];
  if (NR == 0) return;
  const int32_t E = fptr[NR];
  const int lane = threadIdx.x & 63;
  for (int64_t t0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane; t0 < E; t0 += (int64_t)gridDim.x * blockDim.x) {
    const int32_t t = (int32_t)t0 + lane;
    if (t >= E) continue;
    const int32_t qr = sup_row_of(fptr, NR, t, (int32_t)t0);
    const int32_t c = ccol[rowptr[frow[qr]] + (t - fptr[qr])];
    col_flags[c] = 1;  // (same value from every writer)
    atomicAdd(&ccnt[c], 1);
  }
}
// per column: flag as an int (position ncols: 0) and the split-row totals of the transposed view from the kept counts
__global__ void k_sup_colflags(const uint8_t *__restrict__ col_flags, const int32_t *__restrict__ ccnt, int64_t ncols,
                               int thresh, int chunk, int32_t *__restrict__ cflag, int32_t *__restrict__ totals) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int32_t cnt = c < ncols ? ccnt[c] : 0;
  if (c <= ncols) cflag[c] = c < ncols && col_flags[c] != 0;
  const bool lg = cnt > thresh;
  const int32_t nlong = __popcll(__ballot(lg));
  int32_t nch = lg ? (cnt + chunk - 1) / chunk : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nch += __shfl_xor(nch, o, kWave);
  __shared__ int32_t s_tot[2];  // (one pair of global adds per block, as in k_sup_rowcount)
  if (threadIdx.x < 2) s_tot[threadIdx.x] = 0;
  __syncthreads();
  if (lane == 0 && nlong > 0) {
    atomicAdd(&s_tot[0], nlong);
    atomicAdd(&s_tot[1], nch);
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_tot[0] > 0) {
    atomicAdd(&totals[kTotTLong], s_tot[0]);
    atomicAdd(&totals[kTotTChunks], s_tot[1]);
  }
}
__global__ void k_sup_totals(const int32_t *__restrict__ lpos, int64_t ncols, const int32_t *__restrict__ ckept,
                             const int32_t *__restrict__ npos, int64_t N, const int32_t *__restrict__ rpos_rows,
                             int64_t rows, int32_t *__restrict__ totals) {
  totals[kTotL] = lpos[ncols];
  totals[kTotE] = ckept[ncols];
  totals[kTotNL] = npos[N];
  totals[kTotNR] = rpos_rows[rows];
}
// stage 2: number, relation and entry range of every live column (ckept: exclusive kept-entry offsets by column)
__global__ void k_sup_cols2(const uint8_t *__restrict__ col_flags, const int32_t *__restrict__ lpos, int64_t ncols,
                            const int32_t *__restrict__ urel, const int32_t *__restrict__ ckept,
                            int32_t *__restrict__ lcol, int32_t *__restrict__ lrel, int32_t *__restrict__ lptr) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > ncols) return;
  if (c == ncols) { lptr[lpos[c]] = ckept[c]; return; }
  if (!col_flags[c]) return;
  const int32_t k = lpos[c];
  lcol[k] = (int32_t)c;
  lrel[k] = urel[c];
  lptr[k] = ckept[c];
}
// stage 2, over the forward entries: live number of the column, value, rank of the row; the entry's own index as the
// payload of the transposing sort
__global__ void k_sup_fwd_fill(const int32_t *__restrict__ frow, const int32_t *__restrict__ fptr, int32_t NR, int32_t E,
                               const int32_t *__restrict__ rowptr, const int32_t *__restrict__ ccol,
                               const float *__restrict__ val, const int32_t *__restrict__ lpos,
                               int32_t *__restrict__ fcol, float *__restrict__ fval, int32_t *__restrict__ frank,
                               int32_t *__restrict__ ident, float *__restrict__ ones) {
  const int lane = threadIdx.x & 63;
  const int64_t t0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) - lane;
  const int32_t t = (int32_t)t0 + lane;
  if (t0 >= E || t >= E) return;
  const int32_t qr = sup_row_of(fptr, NR, t, (int32_t)t0);
  const int32_t src = rowptr[frow[qr]] + (t - fptr[qr]);
  fcol[t] = lpos[ccol[src]];
  fval[t] = val[src];
  frank[t] = qr;
  ident[t] = t;
  if (ones) ones[t] = 1.f;
}
// the transposed arrays from the sorted (live column, forward entry) pairs
__global__ void k_sup_t_fill(const int32_t *__restrict__ perm, int32_t E, const int32_t *__restrict__ frank,
                             const int32_t *__restrict__ frow, const float *__restrict__ fval,
                             int32_t *__restrict__ lrow, int32_t *__restrict__ lrow_rank, float *__restrict__ lval) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= E) return;
  const int32_t t = perm[o], qr = frank[t];
  lrow[o] = frow[qr];
  if (lrow_rank) lrow_rank[o] = qr;
  lval[o] = fval[t];
}

// pinned landing area of the builds' size blocks (grow-only; a build holds g_land_mu from its first copy to its wait)
int32_t *g_land = nullptr;
size_t g_land_ints = 0;
std::mutex g_land_mu;

int stage1_order(SupStage &b, hipStream_t s, const int32_t *rperm, const int32_t *relptr, int64_t nbands,
                 SupStage::Ord *o) {
  const mrgcn_plan *p = b.q->plan;
  const int64_t ncols = p->ncols;
  o->ngroups = nbands * p->num_relations;
  MRGCN_HIP_TRY(b.sc.alloc(&o->rflag, ncols + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&o->rpos, ncols + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&o->gptr, o->ngroups + 1));
  k_sup_flags_i32<<<nblocks(ncols + 1), kTB, 0, s>>>(b.q->col_flags, rperm, ncols, o->rflag);
  MRGCN_HIP_TRY(hipGetLastError());
  int rc;
  if ((rc = exclusive_scan_i32(o->rflag, o->rpos, ncols + 1, s, b.sc))) return rc;
  k_gather_i32<<<nblocks(o->ngroups + 1), kTB, 0, s>>>(o->rpos, relptr, o->ngroups + 1, o->gptr);
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int64_t stage_land_ints(const mrgcn_plan *p) {
  const int64_t R = p->num_relations;
  return kTotCount + (p->n_bands * R + 1) + (p->n_rperm ? p->n_n_bands * R + 1 : 0);
}

int support_stage1(SupStage &b, hipStream_t s) {
  mrgcn_support *q = b.q;
  const mrgcn_plan *p = q->plan;
  const int64_t N = p->num_nodes, ncols = p->ncols, rows = p->num_rows;
  b.sc.s = s;
  MRGCN_HIP_TRY(sup_alloc(q, &q->col_flags, ncols));
  MRGCN_HIP_TRY(sup_alloc(q, &q->node_flags, N));
  MRGCN_HIP_TRY(sup_alloc(q, &q->nlptr, N + 1));
  MRGCN_HIP_TRY(sup_alloc(q, &q->node_scratch, N));
  // (the flagged rows' list is at most every row: these three are sized for that, no host count needed)
  MRGCN_HIP_TRY(sup_alloc(q, &q->rowrank, rows));
  MRGCN_HIP_TRY(sup_alloc(q, &q->frow, rows));
  MRGCN_HIP_TRY(sup_alloc(q, &q->fptr, rows + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&b.totals_d, kTotCount));
  int32_t *rflag, *rlen, *funp, *cflag, *ccnt;
  MRGCN_HIP_TRY(b.sc.alloc(&rflag, rows + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&rlen, rows + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&b.rpos_rows, rows + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&funp, rows + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&cflag, ncols + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&b.lpos, ncols + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&ccnt, ncols + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&b.ckept, ncols + 1));
  MRGCN_HIP_TRY(hipMemsetAsync(b.totals_d, 0, kTotCount * sizeof(int32_t), s));
  MRGCN_HIP_TRY(hipMemsetAsync(q->col_flags, 0, (size_t)std::max<int64_t>(ncols, 1), s));
  MRGCN_HIP_TRY(hipMemsetAsync(ccnt, 0, (size_t)(ncols + 1) * sizeof(int32_t), s));
  int rc;
  // the flagged rows and their entry ranges
  k_sup_rowcount<<<nblocks(rows + 1), kTB, 0, s>>>(b.row_flags, rows, p->rowptr, kLongThreshold, kChunk, rflag, rlen,
                                                  b.totals_d);
  MRGCN_HIP_TRY(hipGetLastError());
  if ((rc = exclusive_scan_i32(rflag, b.rpos_rows, rows + 1, s, b.sc))) return rc;
  if ((rc = exclusive_scan_i32(rlen, funp, rows + 1, s, b.sc))) return rc;
  k_sup_rows2<<<nblocks(rows + 1), kTB, 0, s>>>(b.row_flags, b.rpos_rows, rows, funp, q->rowrank, q->frow, q->fptr);
  // touched columns and their kept entries
  if (ncols > 0 && p->nnz > 0) {
    int64_t grid = (p->nnz + kTB - 1) / kTB;
    if (grid > 2048) grid = 2048;
    k_sup_fwd_mark<<<dim3((unsigned)grid), kTB, 0, s>>>(q->frow, q->fptr, b.rpos_rows, rows, p->rowptr, p->ccol,
                                                       q->col_flags, ccnt);
  }
  k_sup_colflags<<<nblocks(ncols + 1), kTB, 0, s>>>(q->col_flags, ccnt, ncols, kLongThreshold, kChunk, cflag, b.totals_d);
  MRGCN_HIP_TRY(hipGetLastError());
  if ((rc = exclusive_scan_i32(cflag, b.lpos, ncols + 1, s, b.sc))) return rc;
  if ((rc = exclusive_scan_i32(ccnt, b.ckept, ncols + 1, s, b.sc))) return rc;
  // nodes (NODE_FLAGS: the row set of the next support of a chain)
  MRGCN_HIP_TRY(b.sc.alloc(&b.nflag, N + 1));
  MRGCN_HIP_TRY(b.sc.alloc(&b.npos, N + 1));
  k_sup_nodes<<<nblocks(N + 1), kTB, 0, s>>>(p->nptr, b.lpos, N, q->nlptr, q->node_flags, b.nflag);
  MRGCN_HIP_TRY(hipGetLastError());
  if ((rc = exclusive_scan_i32(b.nflag, b.npos, N + 1, s, b.sc))) return rc;
  k_sup_totals<<<1, 1, 0, s>>>(b.lpos, ncols, b.ckept, b.npos, N, b.rpos_rows, rows, b.totals_d);
  MRGCN_HIP_TRY(hipGetLastError());
  if ((rc = stage1_order(b, s, p->rperm, p->relptr, p->n_bands, &b.ow))) return rc;
  q->has_narrow = p->n_rperm != nullptr;
  if (q->has_narrow && (rc = stage1_order(b, s, p->n_rperm, p->n_relptr, p->n_n_bands, &b.on))) return rc;
  // the size block -> pinned host memory
  int32_t *land = g_land + b.land_off;
  MRGCN_HIP_TRY(hipMemcpyAsync(land, b.totals_d, kTotCount * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MRGCN_HIP_TRY(hipMemcpyAsync(land + kTotCount, b.ow.gptr, (b.ow.ngroups + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (q->has_narrow)
    MRGCN_HIP_TRY(hipMemcpyAsync(land + kTotCount + b.ow.ngroups + 1, b.on.gptr, (b.on.ngroups + 1) * sizeof(int32_t),
                                 hipMemcpyDeviceToHost, s));
  return MRGCN_OK;
}

// the chunk lists of one order from its group pointers (host): chunk_ptr[R+1] | ids | rel | beg | end, every array
// with at least one element.  Two passes, no allocation per call: `count` sizes, `write` fills `dst` (pinned memory).
// columns per chunk of a support's relation-major orders (<= kRelChunk: the transforms' LDS lists are sized for that)
// A small order (fewer live columns than ~1024 chunks' worth) is cut finer, down to 64 columns: its transforms are
// chains of dependent steps per block, and a few dozen blocks leave most of the chip idle.
int sup_rel_chunk(int64_t live_cols) {
  const int64_t v = cfg(CFG_SUP_REL_CHUNK);
  int64_t c = std::min<int64_t>(std::max<int64_t>(v, 64), kRelChunk);
  if (live_cols < c * 1024) c = std::min<int64_t>(c, std::max<int64_t>(64, (live_cols / 1024 + 63) / 64 * 64));
  return (int)c;
}
size_t order_chunks_count(const int32_t *h_gptr, int64_t ngroups, int64_t R, std::vector<int32_t> &cnt_rel,
                          int32_t *n_chunks, int32_t *max_chunks) {
  const int kChunk = sup_rel_chunk(h_gptr[ngroups] - h_gptr[0]);
  cnt_rel.assign((size_t)R, 0);
  int32_t n = 0;
  for (int64_t g = 0; g < ngroups; ++g) {
    const int32_t len = h_gptr[g + 1] - h_gptr[g];
    if (len <= 0) continue;
    const int32_t c = (len + kChunk - 1) / kChunk;
    cnt_rel[(size_t)(g % R)] += c;
    n += c;
  }
  int32_t mx = 0;
  for (int64_t r = 0; r < R; ++r) mx = std::max(mx, cnt_rel[(size_t)r]);
  *n_chunks = n;
  *max_chunks = mx;
  return (size_t)(R + 1) + 4 * (size_t)std::max(n, 1);
}
void order_chunks_write(const int32_t *h_gptr, int64_t ngroups, int64_t R, const std::vector<int32_t> &cnt_rel,
                        int32_t n_chunks, int32_t *dst, size_t base, size_t offs[5], std::vector<int32_t> &fill) {
  const int kChunk = sup_rel_chunk(h_gptr[ngroups] - h_gptr[0]);
  const size_t m = (size_t)std::max(n_chunks, 1);
  offs[0] = base;
  offs[1] = offs[0] + (size_t)(R + 1);
  offs[2] = offs[1] + m;
  offs[3] = offs[2] + m;
  offs[4] = offs[3] + m;
  int32_t *cptr = dst + offs[0], *ids = dst + offs[1], *rel = dst + offs[2], *beg = dst + offs[3], *end = dst + offs[4];
  fill.assign((size_t)R, 0);
  int32_t run = 0;
  for (int64_t r = 0; r < R; ++r) {
    cptr[r] = run;
    fill[(size_t)r] = run;
    run += cnt_rel[(size_t)r];
  }
  cptr[R] = run;
  if (n_chunks == 0) ids[0] = rel[0] = beg[0] = end[0] = 0;
  int32_t ci = 0;
  for (int64_t g = 0; g < ngroups; ++g) {
    const int32_t r = (int32_t)(g % R);
    for (int32_t b0 = h_gptr[g]; b0 < h_gptr[g + 1]; b0 += kChunk) {
      ids[fill[(size_t)r]++] = ci;
      rel[ci] = r;
      beg[ci] = b0;
      end[ci] = std::min(b0 + kChunk, h_gptr[g + 1]);
      ++ci;
    }
  }
}

int stage2_order(SupStage &b, hipStream_t s, const SupStage::Ord &o, const int32_t *rperm, const int32_t *rnode,
                 const int32_t *blk, const size_t *offs, mrgcn_support::Order *out) {
  mrgcn_support *q = b.q;
  const int64_t ncols = q->plan->ncols;
  out->chunk_ptr = const_cast<int32_t *>(blk) + offs[0];
  out->chunk_ids = const_cast<int32_t *>(blk) + offs[1];
  out->chunk_rel = const_cast<int32_t *>(blk) + offs[2];
  out->chunk_beg = const_cast<int32_t *>(blk) + offs[3];
  out->chunk_end = const_cast<int32_t *>(blk) + offs[4];
  MRGCN_HIP_TRY(sup_alloc(q, &out->lperm, q->L));
  MRGCN_HIP_TRY(sup_alloc(q, &out->lrin, q->L));
  if (ncols > 0)
    k_sup_rfill<<<nblocks(ncols), kTB, 0, s>>>(rperm, rnode, o.rflag, o.rpos, ncols, b.lpos, out->lperm, out->lrin);
  if (b.forward) {  // (forward arrays: the same list by rank among the live nodes)
    MRGCN_HIP_TRY(sup_alloc(q, &out->lrin_ord, q->L));
    if (q->L > 0) k_sup_node_ord<<<nblocks(q->L), kTB, 0, s>>>(out->lrin, nullptr, b.npos, q->L, out->lrin_ord);
  }
  MRGCN_HIP_TRY(hipGetLastError());
  return MRGCN_OK;
}

int long_rows_known(mrgcn_support *q, const int32_t *ptr, int64_t rows, hipStream_t s, int32_t n_long, int32_t n_chunks,
                    int32_t **long_row, int32_t **long_cptr, int32_t **chunk_beg, int32_t **chunk_end,
                    int32_t **chunk_row) {
  mrgcn_plan acct;  // build_long charges its arrays to a plan: this one only carries the stream and the byte count
  acct.build_stream = s;
  const int32_t known[2] = {n_long, n_chunks};
  int32_t nl = 0, nc = 0;
  int64_t max_len = 0;
  int rc = build_long(&acct, ptr, rows, s, long_row, long_cptr, chunk_beg, chunk_end, chunk_row, &nl, &nc, &max_len,
                      kLongThreshold, kChunk, 0, 0, 0, known);
  for (void *a : {(void *)*long_row, (void *)*long_cptr, (void *)*chunk_beg, (void *)*chunk_end, (void *)*chunk_row})
    if (a) q->owned.push_back(a);
  q->device_bytes += acct.device_bytes;
  return rc;
}

// pinned staging of the chunk lists on their way to the device (grow-only; the event says when the last upload has
// been read)
int32_t *g_up = nullptr;
size_t g_up_ints = 0;
hipEvent_t g_up_done = nullptr;
int g_up_dev = -1;  // the device g_up_done belongs to (an event records only on streams of its own device)
bool g_up_pending = false;

// the orders' chunk lists of every level: host work on the landed group pointers, then one asynchronous upload per
// support out of pinned memory, queued in front of the stage-2 kernels
int supports_upload(std::vector<SupStage> &st, hipStream_t s) {
  const int n = (int)st.size();
  static thread_local std::vector<int32_t> fill;
  struct Sz { size_t w = 0, n = 0; std::vector<int32_t> cw, cn; };
  std::vector<Sz> sz(n);
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    mrgcn_support *q = st[i].q;
    const int64_t R = q->plan->num_relations;
    const int32_t *land = g_land + st[i].land_off;
    sz[i].w = order_chunks_count(land + kTotCount, st[i].ow.ngroups, R, sz[i].cw, &q->wide.n_chunks, &q->wide.max_chunks);
    if (q->has_narrow)
      sz[i].n = order_chunks_count(land + kTotCount + st[i].ow.ngroups + 1, st[i].on.ngroups, R, sz[i].cn,
                                   &q->narrow.n_chunks, &q->narrow.max_chunks);
    total += sz[i].w + sz[i].n;
  }
  if (g_up_pending) {  // (the previous build's uploads: long done by now)
    MRGCN_HIP_TRY(hipEventSynchronize(g_up_done));
    g_up_pending = false;
  }
  if (total > g_up_ints) {
    if (g_up) (void)hipHostFree(g_up);
    g_up = nullptr;
    g_up_ints = 0;
    const size_t want = total + total / 2 + 4096;
    MRGCN_HIP_TRY(hipHostMalloc((void **)&g_up, want * sizeof(int32_t), hipHostMallocDefault));
    g_up_ints = want;
  }
  int dev_now = 0;
  MRGCN_HIP_TRY(hipGetDevice(&dev_now));
  if (g_up_done && g_up_dev != dev_now) {  // a build on another device of this process: the event follows it
    (void)hipEventDestroy(g_up_done);
    g_up_done = nullptr;
  }
  if (!g_up_done) {
    MRGCN_HIP_TRY(hipEventCreateWithFlags(&g_up_done, hipEventDisableTiming));
    g_up_dev = dev_now;
  }
  size_t at = 0;
  for (int i = 0; i < n; ++i) {
    mrgcn_support *q = st[i].q;
    const int64_t R = q->plan->num_relations;
    const int32_t *land = g_land + st[i].land_off;
    int32_t *dst = g_up + at;  // offsets are relative to this level's block
    order_chunks_write(land + kTotCount, st[i].ow.ngroups, R, sz[i].cw, q->wide.n_chunks, dst, 0, st[i].offs_w, fill);
    if (q->has_narrow)
      order_chunks_write(land + kTotCount + st[i].ow.ngroups + 1, st[i].on.ngroups, R, sz[i].cn, q->narrow.n_chunks, dst,
                         sz[i].w, st[i].offs_n, fill);
    const size_t ints = sz[i].w + sz[i].n;
    MRGCN_HIP_TRY(sup_alloc(q, &st[i].blk, (int64_t)ints));
    MRGCN_HIP_TRY(hipMemcpyAsync(st[i].blk, dst, ints * sizeof(int32_t), hipMemcpyHostToDevice, s));
    at += ints;
  }
  MRGCN_HIP_TRY(hipEventRecord(g_up_done, s));
  g_up_pending = true;
  return MRGCN_OK;
}

// a template for the forward-only arrays: owned by the support when it keeps them, scratch of the build otherwise
template <typename T> hipError_t stage_alloc(SupStage &b, bool keep, T **dst, int64_t n) {
  return keep ? sup_alloc(b.q, dst, n) : b.sc.alloc(dst, n);
}

int support_stage2(SupStage &b, hipStream_t s) {
  mrgcn_support *q = b.q;
  const mrgcn_plan *p = q->plan;
  const int64_t N = p->num_nodes, ncols = p->ncols;
  const int32_t *land = g_land + b.land_off;
  const int32_t *blk = b.blk;
  const size_t *offs_w = b.offs_w, *offs_n = b.offs_n;
  q->L = land[kTotL];
  q->E = land[kTotE];
  q->NL = land[kTotNL];
  q->NR = land[kTotNR];
  q->has_forward = b.forward;
  const bool fw = b.forward;
  const int32_t E = (int32_t)q->E;
  MRGCN_HIP_TRY(sup_alloc(q, &q->lcol, q->L));
  MRGCN_HIP_TRY(sup_alloc(q, &q->lrel, q->L));
  MRGCN_HIP_TRY(sup_alloc(q, &q->lptr, q->L + 1));
  MRGCN_HIP_TRY(sup_alloc(q, &q->lrow, q->E));
  MRGCN_HIP_TRY(sup_alloc(q, &q->lval, q->E));
  MRGCN_HIP_TRY(sup_alloc(q, &q->lnode, q->NL));
  MRGCN_HIP_TRY(sup_alloc(q, &q->lnptr, q->NL + 1));
  MRGCN_HIP_TRY(stage_alloc(b, fw, &q->fcol, q->E));
  MRGCN_HIP_TRY(stage_alloc(b, fw, &q->fval, q->E));
  if (fw) {
    MRGCN_HIP_TRY(sup_alloc(q, &q->ones, q->E));
    MRGCN_HIP_TRY(sup_alloc(q, &q->lrow_rank, q->E));
    MRGCN_HIP_TRY(sup_alloc(q, &q->lnode_ord, q->L));
  }
  int32_t *frank, *ident, *perm, *keys_out;
  MRGCN_HIP_TRY(b.sc.alloc(&frank, q->E));
  MRGCN_HIP_TRY(b.sc.alloc(&ident, q->E));
  MRGCN_HIP_TRY(b.sc.alloc(&perm, q->E));
  MRGCN_HIP_TRY(b.sc.alloc(&keys_out, q->E));
  k_sup_cols2<<<nblocks(ncols + 1), kTB, 0, s>>>(q->col_flags, b.lpos, ncols, p->urel, b.ckept, q->lcol, q->lrel, q->lptr);
  k_sup_lnodes<<<nblocks(N + 1), kTB, 0, s>>>(b.nflag, b.npos, N, q->nlptr, q->lnode, q->lnptr);
  MRGCN_HIP_TRY(hipGetLastError());
  if (E > 0) {
    // the flagged rows' entries (the forward view), then the same entries by (live column, row): one stable sort
    k_sup_fwd_fill<<<nblocks(E), kTB, 0, s>>>(q->frow, q->fptr, (int32_t)q->NR, E, p->rowptr, p->ccol, p->val, b.lpos,
                                             q->fcol, q->fval, frank, ident, fw ? q->ones : nullptr);
    MRGCN_HIP_TRY(hipGetLastError());
    size_t tb = 0;
    const int end_bit = bits_for(std::max<int64_t>(q->L - 1, 1));
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, q->fcol, keys_out, ident, perm, E, 0, end_bit, s));
    char *tmp = nullptr;
    MRGCN_HIP_TRY(b.sc.alloc(&tmp, (int64_t)tb));
    MRGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tb, q->fcol, keys_out, ident, perm, E, 0, end_bit, s));
    k_sup_t_fill<<<nblocks(E), kTB, 0, s>>>(perm, E, frank, q->frow, q->fval, q->lrow, fw ? q->lrow_rank : nullptr, q->lval);
    MRGCN_HIP_TRY(hipGetLastError());
  }
  int rc;
  // split rows of the filtered transposed view (live columns read by more than kLongThreshold live rows)
  q->t_n_long = land[kTotTLong];
  q->t_n_chunks = land[kTotTChunks];
  if ((rc = long_rows_known(q, q->lptr, q->L, s, q->t_n_long, q->t_n_chunks, &q->t_long_row, &q->t_long_cptr,
                            &q->t_chunk_beg, &q->t_chunk_end, &q->t_chunk_row)))
    return rc;
  MRGCN_HIP_TRY(sup_alloc(q, &q->partials, (int64_t)std::max(q->t_n_chunks, 1) * kWsFeatures));
  if (fw) {
    if (q->L > 0) k_sup_node_ord<<<nblocks(q->L), kTB, 0, s>>>(p->unode, q->lcol, b.npos, q->L, q->lnode_ord);
    MRGCN_HIP_TRY(hipGetLastError());
    q->f_n_long = land[kTotFLong];
    q->f_n_chunks = land[kTotFChunks];
    if ((rc = long_rows_known(q, q->fptr, q->NR, s, q->f_n_long, q->f_n_chunks, &q->f_long_row, &q->f_long_cptr,
                              &q->f_chunk_beg, &q->f_chunk_end, &q->f_chunk_row)))
      return rc;
    MRGCN_HIP_TRY(sup_alloc(q, &q->f_partials, (int64_t)std::max(q->f_n_chunks, 1) * kWsFeatures));
  }
  {
    const int64_t nt = std::max<int64_t>(std::max(q->t_n_long, q->f_n_long), 1);
    MRGCN_HIP_TRY(sup_alloc(q, &q->ticket, nt));
    MRGCN_HIP_TRY(hipMemsetAsync(q->ticket, 0, (size_t)nt * sizeof(int32_t), s));
  }
  if ((rc = stage2_order(b, s, b.ow, p->rperm, p->rnode, blk, offs_w, &q->wide))) return rc;
  if (q->has_narrow && (rc = stage2_order(b, s, b.on, p->n_rperm, p->n_rnode, blk, offs_n, &q->narrow))) return rc;
  return MRGCN_OK;
}

// builds the supports qs[0..n): qs[0] on `row_flags`, qs[i+1] on NODE_FLAGS of qs[i]
int build_support_chain(mrgcn_support **qs, int n, const uint8_t *row_flags, hipStream_t s, bool forward) {
  std::vector<SupStage> st(n);
  std::lock_guard<std::mutex> lock(g_land_mu);
  const int64_t per = stage_land_ints(qs[0]->plan);
  if ((size_t)(per * n) > g_land_ints) {
    if (g_land) (void)hipHostFree(g_land);
    g_land = nullptr;
    g_land_ints = 0;
    MRGCN_HIP_TRY(hipHostMalloc((void **)&g_land, (size_t)(per * n) * sizeof(int32_t), hipHostMallocDefault));
    g_land_ints = (size_t)(per * n);
  }
  const bool timing = cfg(CFG_SUP_TIMING) != 0;
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  int rc;
  for (int i = 0; i < n; ++i) {
    st[i].q = qs[i];
    st[i].forward = forward;
    st[i].row_flags = i == 0 ? row_flags : qs[i - 1]->node_flags;
    st[i].land_off = per * i;
    if ((rc = support_stage1(st[i], s))) return rc;
  }
  const auto t1 = clk::now();
  MRGCN_HIP_TRY(hipStreamSynchronize(s));  // the one wait of the build
  const auto t2 = clk::now();
  if ((rc = supports_upload(st, s))) return rc;
  const auto t3 = clk::now();
  for (int i = 0; i < n; ++i)
    if ((rc = support_stage2(st[i], s))) return rc;
  if (timing) {
    const auto us = [](clk::time_point a, clk::time_point b) {
      return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count();
    };
    fprintf(stderr, "mrgcn support chain (%d): stage 1 queued %ld us, wait %ld us, chunk lists + uploads %ld us, stage 2 queued %ld us\n",
            n, us(t0, t1), us(t1, t2), us(t2, t3), us(t3, clk::now()));
  }
  return MRGCN_OK;  // (stream ordered from here on: the scratch goes back tagged with the stream)
}

void release_support(mrgcn_support *q) {
  // like free_plan: wait for the work in flight on the device, then hand the blocks back; a support dropped while a
  // capture is under way is parked until the next one is released outside a capture
  static std::vector<mrgcn_support *> parked;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int cur = 0;
  (void)hipGetDevice(&cur);
  parked.push_back(q);
  std::vector<mrgcn_support *> keep;
  for (mrgcn_support *x : parked) {
    const int xdev = x->device;  // (x may be deleted below)
    if (xdev != cur) (void)hipSetDevice(xdev);
    const uint64_t ep = pool_sync_begin();
    if (hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      keep.push_back(x);
    } else {
      pool_sync_done(ep);
      for (void *a : x->owned) pool_free(a, x->build_stream, ep);
      delete x;
    }
    if (xdev != cur) (void)hipSetDevice(cur);
  }
  parked.swap(keep);
}

}  // namespace
}  // namespace mrgcn

extern "C" {

int mrgcn_support_create(mrgcn_support_t **out, const mrgcn_plan_t *plan, const uint8_t *row_flags, void *stream) {
  return mrgcn_support_create_ex(out, plan, row_flags, 0, stream);
}

int mrgcn_support_create_ex(mrgcn_support_t **out, const mrgcn_plan_t *plan, const uint8_t *row_flags, uint32_t flags,
                            void *stream) {
  using namespace mrgcn;
  MRGCN_REQUIRE(out && plan && row_flags, "NULL");
  MRGCN_REQUIRE((flags & ~MRGCN_SUPPORT_FORWARD) == 0, "flags");
  MRGCN_REQUIRE(!plan->lean, "a lean plan (mini-batch slice) keeps no transposed view to build a support on");
  return mrgcn_support_create_chain(out, 1, plan, row_flags, flags, stream);
}

int mrgcn_support_create_chain(mrgcn_support_t **out, int32_t n, const mrgcn_plan_t *plan, const uint8_t *row_flags,
                               uint32_t flags, void *stream) {
  using namespace mrgcn;
  MRGCN_REQUIRE(out && plan && row_flags && n >= 1 && n <= 64, "NULL / n");
  MRGCN_REQUIRE((flags & ~MRGCN_SUPPORT_FORWARD) == 0, "flags");
  MRGCN_REQUIRE(!plan->lean, "a lean plan (mini-batch slice) keeps no transposed view to build a support on");
  MRGCN_REQUIRE(n == 1 || plan->num_rows == plan->num_nodes, "a chain needs rows = nodes (the stacked adjacency)");
  std::vector<mrgcn_support *> qs(n);
  for (int i = 0; i < n; ++i) {
    qs[i] = new mrgcn_support();
    qs[i]->plan = plan;
    (void)hipGetDevice(&qs[i]->device);
    qs[i]->build_stream = (hipStream_t)stream;
  }
  int rc = build_support_chain(qs.data(), n, row_flags, (hipStream_t)stream, (flags & MRGCN_SUPPORT_FORWARD) != 0);
  if (rc != MRGCN_OK) {
    for (mrgcn_support *q : qs) release_support(q);
    return rc;
  }
  for (int i = 0; i < n; ++i) out[i] = qs[i];
  return MRGCN_OK;
}

int mrgcn_support_destroy(mrgcn_support_t *sup) {
  if (sup) mrgcn::release_support(sup);
  return MRGCN_OK;
}

int mrgcn_support_destroy_ordered(mrgcn_support_t *sup) {
  using namespace mrgcn;
  if (!sup) return MRGCN_OK;
  // no wait: the blocks go back to the pool under the order of the stream the support was built on — later work on
  // that stream may take them at once, anyone else after the next device-wide wait (pool_free)
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (sup->device != cur) (void)hipSetDevice(sup->device);
  for (void *a : sup->owned) pool_free(a, sup->build_stream);
  if (sup->device != cur) (void)hipSetDevice(cur);
  delete sup;
  return MRGCN_OK;
}

int mrgcn_support_info(const mrgcn_support_t *q, mrgcn_support_info_t *h) {
  MRGCN_REQUIRE(q && h, "NULL");
  h->live_cols = q->L;
  h->live_entries = q->E;
  h->live_nodes = q->NL;
  h->device_bytes = q->device_bytes;
  h->chunks_wide = q->wide.n_chunks;
  h->chunks_narrow = q->has_narrow ? q->narrow.n_chunks : 0;
  h->flagged_rows = q->has_forward ? q->NR : -1;
  return MRGCN_OK;
}

int mrgcn_support_array(const mrgcn_support_t *q, int32_t which, const void **d_ptr, int64_t *h_count) {
  MRGCN_REQUIRE(q && d_ptr && h_count, "NULL");
  const mrgcn_plan *p = q->plan;
  switch (which) {
    case MRGCN_SUP_COL_FLAGS: *d_ptr = q->col_flags; *h_count = p->ncols; break;
    case MRGCN_SUP_NODE_FLAGS: *d_ptr = q->node_flags; *h_count = p->num_nodes; break;
    case MRGCN_SUP_LCOL: *d_ptr = q->lcol; *h_count = q->L; break;
    case MRGCN_SUP_LREL: *d_ptr = q->lrel; *h_count = q->L; break;
    case MRGCN_SUP_NLPTR: *d_ptr = q->nlptr; *h_count = p->num_nodes + 1; break;
    case MRGCN_SUP_LPTR: *d_ptr = q->lptr; *h_count = q->L + 1; break;
    case MRGCN_SUP_LROW: *d_ptr = q->lrow; *h_count = q->E; break;
    case MRGCN_SUP_LVAL: *d_ptr = q->lval; *h_count = q->E; break;
    case MRGCN_SUP_LNODE: *d_ptr = q->lnode; *h_count = q->NL; break;
    case MRGCN_SUP_LPERM: *d_ptr = q->wide.lperm; *h_count = q->L; break;
    case MRGCN_SUP_FROW: *d_ptr = q->frow; *h_count = q->has_forward ? q->NR : 0; break;
    case MRGCN_SUP_FPTR: *d_ptr = q->fptr; *h_count = q->has_forward ? q->NR + 1 : 0; break;
    case MRGCN_SUP_FCOL: *d_ptr = q->fcol; *h_count = q->has_forward ? q->E : 0; break;
    case MRGCN_SUP_FVAL: *d_ptr = q->fval; *h_count = q->has_forward ? q->E : 0; break;
    case MRGCN_SUP_LNODE_ORD: *d_ptr = q->lnode_ord; *h_count = q->has_forward ? q->L : 0; break;
    case MRGCN_SUP_ROWRANK: *d_ptr = q->rowrank; *h_count = q->has_forward ? p->num_rows : 0; break;
    default: MRGCN_REQUIRE(false, "unknown support array");
  }
  return MRGCN_OK;
}

int mrgcn_abi_version(void) { return MRGCN_ABI_VERSION; }
const char *mrgcn_arch(void) { return "gfx950"; }
const char *mrgcn_last_error(void) { return mrgcn::g_last_error.c_str(); }

static int straddle_hint(const int32_t *row_bytes, int32_t n, mrgcn::StraddleSizes *out) {
  // no hint: rows of 10..12 floats padded to 48 bytes, the layout of the F = 10 / 11 layers before packed rows
  mrgcn::StraddleSizes sz{1, {48, 0, 0, 0}};
  if (n > 0) {
    MRGCN_REQUIRE(row_bytes != nullptr, "operand_row_bytes is NULL");
    sz.n = 0;
    for (int32_t i = 0; i < n && sz.n < 4; ++i) {
      MRGCN_REQUIRE(row_bytes[i] > 0 && row_bytes[i] % 2 == 0, "operand_row_bytes: positive, even");
      // rows that never straddle a line (divisors of 128) or always do (>= 128 bytes) have nothing to choose
      if (row_bytes[i] >= 128 || 128 % row_bytes[i] == 0) continue;
      sz.bytes[sz.n++] = row_bytes[i];
    }
  }
  *out = sz;
  return MRGCN_OK;
}

int mrgcn_plan_create_hinted(mrgcn_plan_t **plan, int64_t num_rows, int64_t num_nodes, int32_t num_relations,
                             int64_t nnz, const int64_t *coo_rows, const int64_t *coo_cols, const void *coo_vals,
                             int32_t val_dtype, uint32_t flags, const int32_t *operand_row_bytes,
                             int32_t n_row_bytes, void *stream) {
  MRGCN_REQUIRE(plan != nullptr, "plan is NULL");
  *plan = nullptr;
  MRGCN_REQUIRE(num_rows >= 0 && num_nodes > 0 && num_relations > 0, "bad shape");
  MRGCN_REQUIRE(nnz >= 0 && nnz < (int64_t)INT32_MAX, "nnz must be < 2^31");
  MRGCN_REQUIRE((int64_t)num_relations * num_nodes < (int64_t)INT32_MAX,
                "num_relations*num_nodes must be < 2^31");
  MRGCN_REQUIRE(num_rows < (int64_t)INT32_MAX, "num_rows must be < 2^31");
  MRGCN_REQUIRE(val_dtype == MRGCN_VAL_I8 || val_dtype == MRGCN_VAL_F32, "val_dtype");
  MRGCN_REQUIRE(nnz == 0 || (coo_rows && coo_cols && coo_vals), "NULL COO array");
  mrgcn::StraddleSizes hint;
  int rc = straddle_hint(operand_row_bytes, n_row_bytes, &hint);
  if (rc != MRGCN_OK) return rc;
  mrgcn_plan *p = new mrgcn_plan();
  p->num_rows = num_rows;
  p->num_nodes = num_nodes;
  p->num_relations = num_relations;
  (void)hipGetDevice(&p->device);
  p->build_stream = (hipStream_t)stream;
  rc = mrgcn::create_impl(p, nnz, coo_rows, coo_cols, coo_vals, val_dtype, flags, (hipStream_t)stream, hint);
  if (rc != MRGCN_OK) {
    mrgcn::free_plan(p);
    return rc;
  }
  *plan = p;
  return MRGCN_OK;
}

int mrgcn_plan_create(mrgcn_plan_t **plan, int64_t num_rows, int64_t num_nodes, int32_t num_relations,
                      int64_t nnz, const int64_t *coo_rows, const int64_t *coo_cols, const void *coo_vals,
                      int32_t val_dtype, uint32_t flags, void *stream) {
  return mrgcn_plan_create_hinted(plan, num_rows, num_nodes, num_relations, nnz, coo_rows, coo_cols, coo_vals,
                                  val_dtype, flags, nullptr, 0, stream);
}

int mrgcn_plan_create_csr_hinted(mrgcn_plan_t **plan, int64_t num_rows, int64_t num_nodes, int32_t num_relations,
                                 int64_t nnz, const int32_t *indptr, const int32_t *indices, const float *data,
                                 int32_t boundary_cast_i8, uint32_t flags, const int32_t *operand_row_bytes,
                                 int32_t n_row_bytes, void *stream) {
  MRGCN_REQUIRE(plan != nullptr, "plan is NULL");
  *plan = nullptr;
  MRGCN_REQUIRE(nnz >= 0 && nnz < (int64_t)INT32_MAX, "nnz must be < 2^31");
  MRGCN_REQUIRE(num_rows >= 0 && indptr, "indptr");
  MRGCN_REQUIRE(nnz == 0 || (indices && data), "NULL CSR array");
  hipStream_t s = (hipStream_t)stream;
  int64_t *rows = nullptr, *cols = nullptr;
  float *vals = nullptr;
  const size_t n = (size_t)(nnz > 0 ? nnz : 1);
  hipError_t e1 = mrgcn::pool_alloc((void **)&rows, n * sizeof(int64_t), s);
  hipError_t e2 = mrgcn::pool_alloc((void **)&cols, n * sizeof(int64_t), s);
  hipError_t e3 = mrgcn::pool_alloc((void **)&vals, n * sizeof(float), s);
  int rc = MRGCN_OK;
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
    mrgcn::set_error("plan_create_csr: out of device memory for the COO expansion");
    rc = MRGCN_ERR_HIP;
  } else {
    if (nnz > 0)
      mrgcn::k_csr_to_coo<<<mrgcn::nblocks(nnz), mrgcn::kTB, 0, s>>>(indptr, indices, data, num_rows, nnz,
                                                                     boundary_cast_i8 ? 1 : 0, rows, cols, vals);
    rc = mrgcn_plan_create_hinted(plan, num_rows, num_nodes, num_relations, nnz, rows, cols, vals, MRGCN_VAL_F32,
                                  flags, operand_row_bytes, n_row_bytes, stream);
  }
  mrgcn::pool_free(rows, s);  // stream ordered: after the build's last read
  mrgcn::pool_free(cols, s);
  mrgcn::pool_free(vals, s);
  return rc;
}

int mrgcn_plan_create_csr(mrgcn_plan_t **plan, int64_t num_rows, int64_t num_nodes, int32_t num_relations,
                          int64_t nnz, const int32_t *indptr, const int32_t *indices, const float *data,
                          int32_t boundary_cast_i8, uint32_t flags, void *stream) {
  return mrgcn_plan_create_csr_hinted(plan, num_rows, num_nodes, num_relations, nnz, indptr, indices, data,
                                      boundary_cast_i8, flags, nullptr, 0, stream);
}

int mrgcn_plan_destroy(mrgcn_plan_t *plan) {
  if (plan) mrgcn::free_plan(plan);
  return MRGCN_OK;
}

int mrgcn_plan_destroy_after(mrgcn_plan_t *plan, void *event) {
  if (!plan) return MRGCN_OK;
  return mrgcn::free_plan_after(plan, (hipEvent_t)event);
}

int mrgcn_plan_info(const mrgcn_plan_t *p, mrgcn_plan_info_t *h) {
  MRGCN_REQUIRE(p && h, "NULL");
  h->num_rows = p->num_rows;
  h->num_nodes = p->num_nodes;
  h->num_relations = p->num_relations;
  h->nnz = p->nnz;
  h->ncols = p->ncols;
  h->max_row_nnz = p->max_row_nnz;
  h->max_col_nnz = p->max_col_nnz;
  h->long_rows = p->r_n_long;
  h->long_cols = p->c_n_long;
  h->device_bytes = p->device_bytes;
  h->operand_rows = p->n_op;
  h->replicas = p->n_rep;
  return MRGCN_OK;
}

static int plan_lookup(const mrgcn_plan_t *p, int32_t which, const void **out, int64_t *count) {
  const void *src = nullptr;
  int64_t n = 0;
  switch (which) {
    case MRGCN_ARR_ROWPTR: src = p->rowptr; n = p->num_rows + 1; break;
    case MRGCN_ARR_LCOL: src = p->lcol; n = p->nnz; break;
    case MRGCN_ARR_CCOL: src = p->ccol; n = p->nnz; break;
    case MRGCN_ARR_VAL: src = p->val; n = p->nnz; break;
    case MRGCN_ARR_CPTR: src = p->cptr; n = p->ncols + 1; break;
    case MRGCN_ARR_CROW: src = p->crow; n = p->nnz; break;
    case MRGCN_ARR_CVAL: src = p->cval; n = p->nnz; break;
    case MRGCN_ARR_UREL: src = p->urel; n = p->ncols; break;
    case MRGCN_ARR_UNODE: src = p->unode; n = p->ncols; break;
    case MRGCN_ARR_NPTR: src = p->nptr; n = p->num_nodes + 1; break;
    case MRGCN_ARR_ROWIDX: src = p->rowidx; n = p->nnz; break;
    case MRGCN_ARR_ULCOL: src = p->ulcol; n = p->ncols; break;
    case MRGCN_ARR_RPERM: src = p->rperm; n = p->ncols; break;
    case MRGCN_ARR_RELPTR: src = p->relptr; n = p->n_bands * p->num_relations + 1; break;
    case MRGCN_ARR_MPOS: src = p->mpos; n = p->ncols; break;
    case MRGCN_ARR_MCOL: src = p->mcol; n = p->nnz; break;
    case MRGCN_ARR_MVAL: src = p->mval; n = p->nnz; break;
    case MRGCN_ARR_ROWMAP: src = p->rowmap; n = p->num_rows; break;
    case MRGCN_ARR_PTR3: src = p->ptr3; n = p->num_rows + 1; break;
    default: MRGCN_REQUIRE(false, "unknown plan array");
  }
  *out = src;
  *count = n;
  return MRGCN_OK;
}

int mrgcn_plan_array(const mrgcn_plan_t *p, int32_t which, const void **d_ptr, int64_t *h_count) {
  MRGCN_REQUIRE(p && d_ptr && h_count, "NULL");
  return plan_lookup(p, which, d_ptr, h_count);
}

int mrgcn_plan_export(const mrgcn_plan_t *p, int32_t which, void *h_dst, int64_t capacity_bytes) {
  MRGCN_REQUIRE(p && h_dst, "NULL");
  const void *src = nullptr;
  int64_t n = 0;
  int rc = plan_lookup(p, which, &src, &n);
  if (rc != MRGCN_OK) return rc;
  MRGCN_REQUIRE(capacity_bytes >= n * 4, "destination too small");
  // (the build ends stream ordered, not synchronised: wait for its stream before the blocking copy)
  MRGCN_HIP_TRY(hipStreamSynchronize(p->build_stream));
  if (n > 0) MRGCN_HIP_TRY(hipMemcpy(h_dst, src, (size_t)n * 4, hipMemcpyDeviceToHost));
  return MRGCN_OK;
}

int mrgcn_event_create(void **event) {
  MRGCN_REQUIRE(event, "NULL");
  hipEvent_t e;
  MRGCN_HIP_TRY(hipEventCreate(&e));
  *event = (void *)e;
  return MRGCN_OK;
}
int mrgcn_event_destroy(void *event) {
  if (event) MRGCN_HIP_TRY(hipEventDestroy((hipEvent_t)event));
  return MRGCN_OK;
}
int mrgcn_event_record(void *event, void *stream) {
  MRGCN_HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return MRGCN_OK;
}
int mrgcn_event_elapsed_ms(void *start, void *stop, float *h_ms) {
  MRGCN_REQUIRE(h_ms, "NULL");
  MRGCN_HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
  MRGCN_HIP_TRY(hipEventElapsedTime(h_ms, (hipEvent_t)start, (hipEvent_t)stop));
  return MRGCN_OK;
}

}  // extern "C"
