// Microbenchmark for DESIGN §7 item 0: Adam on the node table with gradient and moments kept NODE-MAJOR
// ([N][B][F]: the B basis rows of a node are one contiguous 1600-byte block, so nodes without gradient
// are skipped in any node numbering) while the parameter stays in the reference's [B][N][F] layout.
// A block transposes a tile of T consecutive nodes through LDS: p in and out as B runs of T*F floats,
// g / m / v as one contiguous run per live node.  Compared with the plain streaming Adam (7 streams).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/adam_nodemajor.hip -o /tmp/an && /tmp/an [live_fraction]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int B = 40, F = 10;

__device__ __forceinline__ void upd(float &p, float g, float &m, float &v) {
  m = fmaf(0.9f, m, 0.1f * g);
  v = fmaf(0.999f, v, 0.001f * g * g);
  p -= 0.01f * (m / (sqrtf(v) / 0.0316f + 1e-8f));
}

__global__ __launch_bounds__(256) void k_adam_plain(float4 *p, const float4 *g, float4 *m, float4 *v, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 P = p[i], G = g[i], M = m[i], V = v[i];
    upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
    p[i] = P; m[i] = M; v[i] = V;
  }
}

// p: [B][N][F]; g, m, v: [N][B][F]; live: one byte per node
template <int T, int TB>
__global__ __launch_bounds__(TB) void k_adam_nodemajor(float *p, const float *g, float *m, float *v,
                                                        const unsigned char *live, int64_t N) {
  __shared__ __align__(16) float s_p[B][T * F + 4];
  __shared__ int s_any;
  const int64_t ntiles = (N + T - 1) / T;
  const int64_t slab = N * F;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t j0 = tile * T;
    const int nt = (int)((N - j0 < T) ? N - j0 : T);
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    if ((int)threadIdx.x < nt && live[j0 + threadIdx.x]) s_any = 1;
    __syncthreads();
    if (!s_any) continue;  // block uniform
    const int run4 = nt * F / 4;  // nt*F is a multiple of 4 for full tiles (T*F = 320)
    for (int q = threadIdx.x; q < B * run4; q += TB) {
      const int b = q / run4, x = q - b * run4;
      const float4 t = *reinterpret_cast<const float4 *>(p + (int64_t)b * slab + j0 * F + 4 * x);
      *reinterpret_cast<float4 *>(&s_p[b][4 * x]) = t;
    }
    __syncthreads();
    // node-major side: B*F = 400 floats = 100 float4 per node
    for (int q = threadIdx.x; q < nt * 100; q += TB) {
      const int t = q / 100, w = q - t * 100;
      if (!live[j0 + t]) continue;
      const int64_t i4 = (j0 + t) * 100 + w;
      const float4 G = reinterpret_cast<const float4 *>(g)[i4];
      float4 M = reinterpret_cast<const float4 *>(m)[i4], V = reinterpret_cast<const float4 *>(v)[i4];
      float *gm[4] = {&M.x, &M.y, &M.z, &M.w};
      float *gv[4] = {&V.x, &V.y, &V.z, &V.w};
      const float gg[4] = {G.x, G.y, G.z, G.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = 4 * w + k, b = e / F, o = e - b * F;
        upd(s_p[b][t * F + o], gg[k], *gm[k], *gv[k]);
      }
      reinterpret_cast<float4 *>(m)[i4] = M;
      reinterpret_cast<float4 *>(v)[i4] = V;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < B * run4; q += TB) {
      const int b = q / run4, x = q - b * run4;
      *reinterpret_cast<float4 *>(p + (int64_t)b * slab + j0 * F + 4 * x) = *reinterpret_cast<const float4 *>(&s_p[b][4 * x]);
    }
    __syncthreads();
  }
}

int main(int argc, char **argv) {
  const double frac = argc > 1 ? atof(argv[1]) : 0.5;
  const int64_t N = 1666764 / 64 * 64;  // whole tiles
  const int64_t n = (int64_t)B * N * F;
  float *p, *g, *m, *v;
  unsigned char *live;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4));
  CK(hipMalloc(&live, N));
  CK(hipMemset(p, 0, n * 4)); CK(hipMemset(g, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
  std::vector<unsigned char> h(N);
  srand(1);
  int64_t nl = 0;
  for (int64_t i = 0; i < N; ++i) { h[i] = (rand() / (double)RAND_MAX) < frac; nl += h[i]; }
  CK(hipMemcpy(live, h.data(), N, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) k_adam_plain<<<8192, 256>>>((float4 *)p, (const float4 *)g, (float4 *)m, (float4 *)v, n / 4);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  printf("plain streaming Adam: %.3f ms  (%.0f GB/s over 7 streams)\n", ms / 5, 7.0 * n * 4 / (ms / 5 * 1e-3) / 1e9);
  auto run = [&](auto kern, const char *name, int tb, int grid) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) kern<<<grid, tb>>>(p, g, m, v, live, N);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double bytes = 2.0 * n * 4 + 5.0 * nl * B * F * 4;
    printf("node-major g/m/v, %.0f %% live nodes, %s, grid %d: %.3f ms  (%.0f GB/s over %.1f GB)\n", 100.0 * nl / N, name, grid,
           ms / 5, bytes / (ms / 5 * 1e-3) / 1e9, bytes / 1e9);
  };
  run(k_adam_nodemajor<32, 256>, "T=32 256 thr", 256, 3072);
  run(k_adam_nodemajor<32, 512>, "T=32 512 thr", 512, 3072);
  run(k_adam_nodemajor<16, 256>, "T=16 256 thr", 256, 6144);
  run(k_adam_nodemajor<16, 128>, "T=16 128 thr", 128, 6144);
  run(k_adam_nodemajor<8, 128>, "T=8  128 thr", 128, 12288);
  return 0;
}
