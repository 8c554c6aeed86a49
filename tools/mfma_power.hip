// Pure-MFMA loop (no memory traffic): throughput of the two bf16 MFMA shapes with one or two waves per SIMD, for power readings taken
// next to it with rocm-smi (tools/pwr_probe.sh).   usage: mfma_power <shape 32|16> <waves_per_simd> <seconds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void mfma_loop(float* out, int iters) {
  extern __shared__ char sm[];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x % 7); b[i] = (short)(0x3f00 + threadIdx.x % 5); }
  float r = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  } else if (SHAPE == 33) {  // 16 accumulators of 32x32: 256 registers (AGPRs), the footprint of a 128x128 wave tile
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j];
  }
  if (r == 12345.f) out[0] = r;
}

int main(int argc, char** argv) {
  const int shape = atoi(argv[1]), wps = atoi(argv[2]);
  const double secs = atof(argv[3]);
  float* out;
  hipMalloc(&out, 4);
  const int iters = 20000;
  const int lds = wps == 1 ? 100 * 1024 : (wps == 2 ? 70 * 1024 : 36 * 1024);  // blocks per CU through the LDS footprint
  auto k = shape == 32 ? mfma_loop<32> : shape == 33 ? mfma_loop<33> : mfma_loop<16>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = 256 * wps;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double total_ms = 0; long launches = 0;
  while (total_ms < secs * 1e3) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    total_ms += ms; launches += 20;
  }
  // flops per launch: grid * 4 waves * iters * (8 x 32x32x16 | 16 x 16x16x32) * 2*M*N*K
  const double per_it = shape == 32 ? 8.0 * 2 * 32 * 32 * 16 : shape == 33 ? 16.0 * 2 * 32 * 32 * 16 : 16.0 * 2 * 16 * 16 * 32;
  const double fl = (double)grid * 4 * iters * per_it * launches;
  printf("shape %d  waves/SIMD %d : %.0f TF/s\n", shape, wps, fl / (total_ms * 1e-3) / 1e12);
  return 0;
}
