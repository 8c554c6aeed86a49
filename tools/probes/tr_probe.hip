// ds_read_b64_tr_b16 semantics probe: LDS image = row-major [64 keys][96 d] shorts with value key * 100 + d.  Lane (l31, hi) should get
// keys 8 hi + 4 half + {0,1,2,3} of column d = l31 when each lane of a 16-lane group supplies the address of a 4-short chunk:
//   row = key_base + ((lane & 15) >> 2), cols = 16 * ((lane >> 4) & 1) + 4 * (lane & 3) .. + 3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  for (int i = threadIdx.x; i < 64 * 96; i += 64) reinterpret_cast<short*>(sm)[i] = (short)((i / 96) * 100 + (i % 96));
  __syncthreads();
  typedef __attribute__((address_space(3))) v4s* lp;
  const int lane = threadIdx.x, hi = lane >> 5;
  for (int half = 0; half < 2; ++half) {
    const int row = 8 * hi + 4 * half + ((lane & 15) >> 2);
    const int addr = row * 192 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(sm + addr));
    for (int j = 0; j < 4; ++j) out[(lane * 2 + half) * 4 + j] = r[j];
  }
}
int main() {
  short* d; short h[64 * 8];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 64 * 192, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int half = 0; half < 2; ++half)
      for (int j = 0; j < 4; ++j) {
        const int want = (8 * (lane >> 5) + 4 * half + j) * 100 + (lane & 31);
        const int got = h[(lane * 2 + half) * 4 + j];
        if (want != got) { if (bad < 12) printf("lane %d half %d j %d: got %d want %d\n", lane, half, j, got, want); ++bad; }
      }
  printf("tr16_b64 probe: %d mismatches of 512\n", bad);
  return 0;
}
