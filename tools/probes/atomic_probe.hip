// How fast are fp32 atomic adds into a [2012 x 2048] fp32 matrix when 256 blocks each add a 256x256 tile (split-K epilogue pattern)?
// agent scope (coherent across XCDs) vs workgroup scope (L2-local) vs plain read-modify-write (baseline, racy — timing only).
// hipcc --offload-arch=gfx950 -O3 tools/probes/atomic_probe.hip -o gpurun_out/atomic_probe && gpurun_out/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* x, int M, int N, int splits) {
  // block b -> tile (b / splits), split (b % splits): all splits of a tile add to the same 256x256 region
  const int tile = blockIdx.x / splits, tm = tile / (N / 256), tn = tile % (N / 256);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = w; r < 256; r += 4) {
    const int m = tm * 256 + r;
    if (m >= M) break;
    for (int c = lane; c < 256; c += 64) {
      float* p = x + (long long)m * N + tn * 256 + c;
      const float v = 1.0f + (float)(blockIdx.x % splits);
      if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else *p += v;
    }
  }
}
template <int MODE>
void run(const char* name, float* x, int M, int N, int splits) {
  const int tiles = ((M + 255) / 256) * (N / 256);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(tiles * splits), dim3(256), 0, 0, x, M, N, splits);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(tiles * splits), dim3(256), 0, 0, x, M, N, splits);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-28s splits %d: %7.1f us per launch (%.1f MB of adds -> %.2f TB/s of adds)\n", name, splits, ms / 20 * 1e3, tiles * splits * 0.262144, tiles * splits * 262144.0 / (ms / 20 * 1e-3) / 1e12);
}
int main() {
  const int M = 2012, N = 2048;
  float* x; hipMalloc(&x, (size_t)M * N * 4); hipMemset(x, 0, (size_t)M * N * 4);
  for (int s : {1, 2, 4}) { run<0>("atomic agent scope", x, M, N, s); run<1>("atomic workgroup scope", x, M, N, s); run<2>("plain rmw (racy)", x, M, N, s); }
  // correctness of agent scope: zero, 1 launch with 4 splits -> every element == 1+2+3+4 = 10
  hipMemset(x, 0, (size_t)M * N * 4);
  hipLaunchKernelGGL(k<0>, dim3(64 * 4), dim3(256), 0, 0, x, M, N, 4);
  float* h = (float*)malloc((size_t)M * N * 4); hipMemcpy(h, x, (size_t)M * N * 4, hipMemcpyDeviceToHost);
  long bad = 0; for (long i = 0; i < (long)M * N; ++i) bad += h[i] != 10.0f;
  printf("agent-scope result check: %ld wrong of %ld\n", bad, (long)M * N);
  hipMemset(x, 0, (size_t)M * N * 4);
  hipLaunchKernelGGL(k<1>, dim3(64 * 4), dim3(256), 0, 0, x, M, N, 4);
  hipMemcpy(h, x, (size_t)M * N * 4, hipMemcpyDeviceToHost);
  bad = 0; for (long i = 0; i < (long)M * N; ++i) bad += h[i] != 10.0f;
  printf("workgroup-scope result check (splits of a tile on DIFFERENT XCDs: b %% 8 differs): %ld wrong of %ld\n", bad, (long)M * N);
  return 0;
}
