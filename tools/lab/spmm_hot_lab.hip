// LAB (not part of libmrgcn_hip.so): the stacked-CSR product with the HOTTEST operand rows staged in LDS — the one
// form north_star names ("LDS-staged feature tiles") that rounds 1-5 never measured.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC tools/lab/spmm_hot_lab.hip -o tools/lab/libspmm_hot_lab.so
// One persistent workgroup of 1 024 threads per CU (16 waves: what one 140 KB LDS image leaves room for) keeps the
// first `H` operand rows — the driver orders the operand by reader count — in LDS and walks groups of 16 consecutive
// SHORT rows per wave (<= 8 entries: k_spmm3's S class, 4 lanes per row, every entry's gather issued in one batch).
// An entry whose operand row is < H is a ds_read_b128; the others are global loads, issued under the complementary
// exec mask so that a hot entry makes NO L1 / L2 request at all.  H = 0 is the same kernel without the staging: the
// A/B.  F <= 12 (three 16-byte pieces per row), packed operand rows of `ld` floats (ld = F = 10: 40 bytes).
#include <hip/hip_runtime.h>
#include <stdint.h>

using f4 = __attribute__((ext_vector_type(4))) float;

template <bool HOT>
__global__ __launch_bounds__(1024) void k_hot(int64_t n_short, const int32_t *__restrict__ ptr,
                                              const int32_t *__restrict__ idx, const float *__restrict__ val,
                                              const int32_t *__restrict__ rowmap, const float *__restrict__ M,
                                              int ld, int F, int H, float *__restrict__ Y, int ldY) {
  extern __shared__ __align__(16) float s_hot[];  // [H][12]: rows padded to three whole pieces
  if (HOT) {
    for (int t = threadIdx.x; t < H * 12; t += blockDim.x) {
      const int r = t / 12, f = t - r * 12;
      s_hot[t] = f < F ? M[(int64_t)r * ld + f] : 0.f;
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, slot = lane >> 2, q = lane & 3;
  const bool active = 4 * q < F;
  // the row's last piece overlaps its neighbour's instead of leaving the packed row (k_spmm3 `pack`)
  const int lo = active ? min(4 * q, F - 4) : 0, shift = active ? 4 * q - lo : 0;
  const int64_t groups = (n_short + 15) / 16;
  const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  // blocks b and b + 8 share an XCD: every XCD walks one contiguous run of groups
  const int64_t per_xcd = (groups + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int64_t w_in_xcd = (int64_t)(blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t waves_per_xcd = nw / 8;
  for (int64_t gi = w_in_xcd; gi < per_xcd; gi += waves_per_xcd) {
    const int64_t g = xcd * per_xcd + gi;
    if (g >= groups) break;
    const int64_t rk = g * 16 + slot;
    int32_t b = 0, n = 0;
    if (rk < n_short) {
      b = ptr[rk];
      n = ptr[rk + 1] - b;
    }
    // lane q stages entries q and q + 4 of its row
    int32_t ci[2];
    float ca[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int e = q + 4 * t;
      ci[t] = e < n ? idx[b + e] : 0;
      ca[t] = e < n ? val[b + e] : 0.f;
    }
    f4 x[8];
    float a[8];
    int32_t c[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int src = (lane & ~3) | (t & 3);
      c[t] = __shfl(ci[t >> 2], src, 64);
      a[t] = __shfl(ca[t >> 2], src, 64);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const bool on = t < n;
      const bool hot = HOT && c[t] < H;
      x[t] = f4{0.f, 0.f, 0.f, 0.f};
      if (on && !hot) x[t] = *reinterpret_cast<const f4 *>(M + (int64_t)c[t] * ld + lo);
    }
    if (HOT) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const bool on = t < n;
        if (on && c[t] < H) x[t] = *reinterpret_cast<const f4 *>(s_hot + c[t] * 12 + lo);
      }
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fmaf(a[t], x[t][i], acc[i]);
    }
    if (rk < n_short && active) {
      float *y = Y + (int64_t)rowmap[rk] * ldY + 4 * q;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i + shift < 4 && 4 * q + i < F) y[i] = acc[i + shift];
    }
  }
}

extern "C" int lab_hot(int64_t n_short, const int32_t *ptr, const int32_t *idx, const float *val, const int32_t *rowmap,
                       const float *M, int ld, int F, int H, float *Y, int ldY, int grid, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)H * 12 * sizeof(float);
  if (H > 0) {
    static size_t allowed = 0;
    if (lds > allowed) {
      hipError_t e = hipFuncSetAttribute((const void *)k_hot<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      allowed = lds;
    }
    k_hot<true><<<dim3(grid), dim3(1024), lds, s>>>(n_short, ptr, idx, val, rowmap, M, ld, F, H, Y, ldY);
  } else {
    k_hot<false><<<dim3(grid), dim3(1024), 0, s>>>(n_short, ptr, idx, val, rowmap, M, ld, F, 0, Y, ldY);
  }
  return (int)hipGetLastError();
}
