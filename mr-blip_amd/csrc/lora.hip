// LoRA rank-8 "thin" products of the T5 path (peft Linear.forward / backward; configured at blip2_mr.py:182-200: r = 8, alpha = 8,
// lora_dropout = 0.05 on every q/k/v/o/wi_0/wi_1/wo/lm_head):
//     forward   u[m, r]  = sum_k dropout(x)[m, k] * (s A)[r, k]        (lora_A(lora_dropout(x)); the "up" product rides in the main GEMM)
//     backward  g[m, r]  = sum_n dy[m, n]         * (s B^T)[r, n]
// Both are [M x K] x [K x R] with R = 8 * adapters-of-the-group <= 32: 0.1-0.6 GFLOP per launch, i.e. nothing — what they cost is the
// launch and the operand traffic.  Shape of the kernels here: the R thin vectors live in LDS (<= 128 KB per 2048-wide K chunk), one
// wave owns a row, reads it ONCE in full 16-B pieces (coalesced: a wave instruction covers 1 KB of one row), applies the lora_dropout
// mask from the counter hash in registers, and accumulates R dot products with v_dot2c_f32_bf16 against ds_read_b128 fragments; one
// wave reduction per (row, r) at the very end.  (The MFMA skinny kernel they replace gathers 32 different rows per load instruction and
// was bound by the address unit: 13-22 us at M = 2012, 16 us for the decoder's single-block launches.)
//
// rmsnorm_lora_fwd additionally fuses T5LayerNorm (modeling_t5.py:254-277): the normalised row is still in registers when its LoRA
// projection is taken, so `xn` and `u` of the norm-fed projections (q/k/v, wi_0/wi_1, EncDecAttention.q) cost one launch.
#include "common.h"
#include "lora_thin.h"
#include <stdlib.h>

extern "C" int mrblip_rmsnorm_fwd(const float* x, long long ldx, const float* weight, int M, int D, float eps, void* out_bf16,
                                  long long ldob, float* out_f32, long long ldof, hipStream_t stream);  // norm.hip

typedef __bf16 mrb_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2bf(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(mrb_bf2, a), __builtin_bit_cast(mrb_bf2, b), c, false);
}

struct LoraRowsArgs {
  const bf16_t* X; long long ldx;   // [M, K] rows (bf16)
  const bf16_t* A; long long lda;   // [R, K] thin vectors (bf16, already scaled)
  bf16_t* U; long long ldu;         // [M, >= R] out (bf16): columns 0..R-1 are written
  int M, K, R;
  // rows [8j, 8j+8) of A are non-zero only in columns [seg_k0[j], seg_k1[j]) — the block-diagonal s*B^T of a fused q/k/v or wi_0/wi_1
  // group in the backward's g = dy B; chunks outside a group's range skip its staging and its dot products.  Dense A: [0, K) for all.
  int seg_k0[4], seg_k1[4];
  DropoutArg drop;                  // mask of X (element index m * K + k, pair-hashed); the kept values' 1/(1-p) is applied to the result
  // optional side job for the K-split GEMM that follows on the stream (gemm.hip, k_splits): init_dst[m, 0:init_n] = init_src[m, :] (or 0)
  // for the same rows — the fp32 output its blocks then add their partial products to; costs no launch of its own
  float* init_dst; const float* init_src; long long ld_idst, ld_isrc; int init_n;
  // batched form (thin kernel only, gridDim.y groups; round 4): group g reads X + g * x_gstride, A + g * a_gstride, writes U + g * u_gstride
  // (elements) and draws its mask with call-site id drop.site + g * site_stride — the LoRA "down" products of ALL decoder layers'
  // cross-attention K / V adapters on one encoder output (x_gstride = 0), or their backward's g = dy B on the layers' dy column blocks
  long long x_gstride, a_gstride, u_gstride; uint32_t site_stride;
};

#define LORA_KC 2048  // K chunk held in LDS: R x (up to) 2048 bf16

typedef __attribute__((address_space(3))) void* lora_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* lora_glb_ptr_t;

// stage A[:, kc0 : kc0 + kcw] into LDS as [R][kcw] by LDS-DMA (global_load ... lds, 16 B per lane, every piece in flight at once: a
// register-staged copy loop was the whole run time of the first version of these kernels).  The LDS image is linear in the piece
// index (piece i = row i / (kcw/8), 16-B column i % (kcw/8)), which is what the DMA's "wave base + lane * 16" addressing needs.
__device__ __forceinline__ void lora_fill(char* smem, const bf16_t* A, long long lda, int R, int kcw, int kc0, uint32_t active = 0xfu) {
  const int per_row = kcw >> 3, n16 = R * per_row;
  const int wv = threadIdx.x >> 6;
  for (int base = 0; base < n16; base += 256) {
    const int idx = base + (int)threadIdx.x;
    const int r = idx / per_row;
    if (idx < n16 && ((active >> (r >> 3)) & 1u)) {
      const int c = (idx - r * per_row) << 3;
      const bf16_t* src = A + (long long)r * lda + kc0 + c;
      __builtin_amdgcn_global_load_lds((lora_glb_ptr_t)src, (lora_lds_ptr_t)(smem + (long long)(base + wv * 64) * 16), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Blocks walk row groups of 4 * RPW rows (one wave = RPW rows).  K <= 2048 (one chunk): the grid is capped at the CU count and the thin
// vectors are staged ONCE per block; longer K: one row group per block, the chunk loop re-stages.
template <int NR, int RPW>  // NR = R / 8, RPW = rows per wave
__global__ __launch_bounds__(256) void lora_rows_kernel(const LoraRowsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int R = NR * 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t seed = p.drop.seed_ptr ? *p.drop.seed_ptr : 0u;
  const float post = p.drop.seed_ptr ? p.drop.inv_keep : 1.0f;
  const int nrg = (p.M + 4 * RPW - 1) / (4 * RPW);
  const bool one_chunk = p.K <= LORA_KC;
  bool staged = false;
  for (int rg = blockIdx.x; rg < nrg; rg += gridDim.x) {
    const int row0 = (rg * 4 + wv) * RPW;
    float acc[RPW][R];
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
      for (int r = 0; r < R; ++r) acc[i][r] = 0.f;
    for (int kc0 = 0; kc0 < p.K; kc0 += LORA_KC) {
      const int kcw = min(LORA_KC, p.K - kc0);
      uint32_t active = 0u;  // row groups of A that are non-zero somewhere in this chunk (block-uniform)
#pragma unroll
      for (int j = 0; j < NR; ++j) active |= (kc0 < p.seg_k1[j] && kc0 + kcw > p.seg_k0[j]) ? (1u << j) : 0u;
      if (!(one_chunk && staged)) {
        if (staged) __syncthreads();  // every wave is done with the previous chunk
        lora_fill(smem, p.A, p.lda, R, kcw, kc0, active);
        __syncthreads();
        staged = true;
      }
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int row = row0 + i;
        if (row >= p.M) break;  // wave-uniform
        if (p.init_dst && kc0 == 0) {
          for (int c = lane * 4; c < p.init_n; c += 256) {
            const float4 v = p.init_src ? *reinterpret_cast<const float4*>(p.init_src + (long long)row * p.ld_isrc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(p.init_dst + (long long)row * p.ld_idst + c) = v;
          }
        }
        const bf16_t* xr = p.X + (long long)row * p.ldx + kc0;
        uint4 xv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int c = (it * 64 + lane) * 8;
          xv[it] = (c < kcw) ? *reinterpret_cast<const uint4*>(xr + c) : make_uint4(0u, 0u, 0u, 0u);
        }
        if (p.drop.seed_ptr) {
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const uint32_t e = (uint32_t)row * (uint32_t)p.K + (uint32_t)(kc0 + (it * 64 + lane) * 8);
            uint32_t* d = reinterpret_cast<uint32_t*>(&xv[it]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              bool k0, k1;
              mrb_keep2(e + 2 * q, seed, p.drop.site, p.drop.thresh24, k0, k1);
              d[q] = (k0 ? d[q] & 0xffffu : 0u) | (k1 ? d[q] & 0xffff0000u : 0u);
            }
          }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int c = (it * 64 + lane) * 8;
          if (c < kcw) {
#pragma unroll
            for (int j = 0; j < NR; ++j) {
              if (!((active >> j) & 1u)) continue;  // uniform
#pragma unroll
              for (int r8 = 0; r8 < 8; ++r8) {
                const int r = j * 8 + r8;
                const uint4 av = *reinterpret_cast<const uint4*>(smem + ((long long)r * kcw + c) * 2);
                float a = acc[i][r];
                a = dot2bf(xv[it].x, av.x, a); a = dot2bf(xv[it].y, av.y, a); a = dot2bf(xv[it].z, av.z, a); a = dot2bf(xv[it].w, av.w, a);
                acc[i][r] = a;
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int row = row0 + i;
      if (row >= p.M) break;
      float mine = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float s = wave_sum_uniform(acc[i][r]);
        if (lane == r) mine = s;
      }
      if (lane < R) p.U[(long long)row * p.ldu + lane] = f2bf(mine * post);
    }
  }
}

static int lora_num_cu() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return n;
}

// ---- the same product for TALL inputs (M >= LORA_THIN_MIN_M rows: the T5 encoder's M = 2012) on the matrix cores ------------------
// Round 3: at M = 2012 the row kernel above is a chain of exposed latencies (stage A -> barrier -> a row's loads -> 32..96 KB of LDS
// fragment reads per 4-KB row -> next row; 8-30 us per launch where the operand streams in 2-10 us, ~4.5 ms of main-stream time per step).
// Here a block owns 16 rows, its 8 waves split K; every wave issues a whole batch of k-steps' operands as 16-B loads straight from
// global memory into v_mfma_f32_16x16x32_bf16 fragments (x: lane (m = l & 15, kg = l >> 4) holds 8 consecutive k of row m, 16 rows x 64
// contiguous bytes per instruction; A: the same lane map over r — A is 32..96 KB and L2 resident) — nothing is staged in LDS, every x
// byte is fetched once, the structural zeros of a block-diagonal A cost nothing extra, and 8 x 8 KB are in flight per CU.  The 8
// partial accumulators meet in LDS at the end (fixed order: deterministic).  Same dropout mask, same result up to fp32 summation order.
#define LORA_THIN_MIN_M 512

template <int NT, int UB, int ROWS>  // NT = 16-wide r tiles (R <= 16 * NT), UB = k-steps per batch, ROWS = rows per block: 16, or 8 (the MFMA's
                                     // other 8 rows idle, their lanes load nothing) when 16-row blocks would leave half of the CUs without one
__global__ __launch_bounds__(512) void lora_thin_kernel(const LoraRowsArgs p) {   // body: lora_thin.h (shared with the tile GEMM's thin role)
  __shared__ f32x4 red[8][NT][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.x * ROWS;
  const int grp = blockIdx.y;                              // batched form: this block's group (0 otherwise)
  ThinArgs t;
  t.X = p.X + grp * p.x_gstride; t.ldx = p.ldx; t.A = p.A + grp * p.a_gstride; t.lda = p.lda; t.U = p.U + grp * p.u_gstride; t.ldu = p.ldu;
  t.M = p.M; t.K = p.K; t.R = p.R;
  t.seed_ptr = p.drop.seed_ptr; t.site = p.drop.site + (uint32_t)grp * p.site_stride; t.thresh16 = p.drop.thresh24; t.inv_keep = p.drop.inv_keep;
  lora_thin_body<NT, UB, ROWS, 8, false>(t, m0, red, w, lane, [&]() __attribute__((always_inline)) {
    // side job (see LoraRowsArgs): the block's fp32 init rows, ROWS / 8 rows per wave
    if (p.init_dst) {
#pragma unroll
      for (int i = 0; i < ROWS / 8; ++i) {
        const int r = m0 + (ROWS / 8) * w + i;
        if (r < p.M)
          for (int c = lane * 4; c < p.init_n; c += 256) {
            const float4 v = p.init_src ? *reinterpret_cast<const float4*>(p.init_src + (long long)r * p.ld_isrc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(p.init_dst + (long long)r * p.ld_idst + c) = v;
          }
      }
    }
  });
}

static bool lora_thin_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("MRB_LORA_THIN");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on == 1;
}

static int lora_thin_min_m() {   // rows from which the matrix-core thin kernel replaces the LDS-resident row kernel (MRB_LORA_THIN_MIN_M)
  static int m = -1;
  if (m < 0) {
    const char* e = getenv("MRB_LORA_THIN_MIN_M");
    m = e ? atoi(e) : LORA_THIN_MIN_M;
  }
  return m;
}

static int launch_thin(const LoraRowsArgs& a, hipStream_t st, int groups = 1) {
  // 8-row blocks (MRB_LORA_THIN_ROWS=8; M = 2012: 252 blocks instead of 126) are FASTER stand-alone (enc g wi 17.8 vs 21.1 us, g qkv 14.4 vs
  // 16.3) and SLOWER in the train step (71.95 vs 71.36 ms; row kernel: 72.55): the 126-block form leaves half of the CUs to the
  // gradient side stream and the look-ahead ViT that run beside it.  16 rows is the default.
  static int rows8 = -1;
  if (rows8 < 0) { const char* e = getenv("MRB_LORA_THIN_ROWS"); rows8 = (e && atoi(e) == 8) ? 1 : 0; }
  const bool half = rows8 && groups == 1 && (a.M + 15) / 16 < lora_num_cu();
  const dim3 grid(half ? (a.M + 7) / 8 : (a.M + 15) / 16, groups);
  if (a.R <= 16) {
    if (half) hipLaunchKernelGGL((lora_thin_kernel<1, 8, 8>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((lora_thin_kernel<1, 8, 16>), grid, dim3(512), 0, st, a);
  } else {
    if (half) hipLaunchKernelGGL((lora_thin_kernel<2, 4, 8>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((lora_thin_kernel<2, 4, 16>), grid, dim3(512), 0, st, a);
  }
  return mrblip_check_launch("lora_thin");
}

template <int NR>
static int launch_rows(const LoraRowsArgs& a, hipStream_t st) {
  const int kcw = a.K < LORA_KC ? a.K : LORA_KC;
  const int LDS = NR * 8 * kcw * 2;
  constexpr int LDS_MAX = NR * 8 * LORA_KC * 2;
  const int ncu = lora_num_cu();
  // rows per wave.  One chunk: 1 (the blocks loop over their row groups, nothing is re-staged).  Several chunks: every block re-stages
  // all of A, so fewer, taller blocks — about one per CU.
  const int rpw = a.K <= LORA_KC ? 1 : (a.M <= 4 * ncu ? 1 : a.M <= 8 * ncu ? 2 : 4);
  const int nrg = (a.M + 4 * rpw - 1) / (4 * rpw);
  const int grid = a.K <= LORA_KC ? (nrg < ncu ? nrg : ncu) : nrg;
  static bool attr[3] = {};
#define MRB_LR_LAUNCH(I, RPW_)                                                                                                     \
  {                                                                                                                                \
    auto k = lora_rows_kernel<NR, RPW_>;                                                                                           \
    if (!attr[I]) {                                                                                                                \
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess) {                \
        mrblip_set_error("lora_rows: cannot raise dynamic LDS to %d", LDS_MAX);                                                    \
        return MRBLIP_ELAUNCH;                                                                                                     \
      }                                                                                                                            \
      attr[I] = true;                                                                                                              \
    }                                                                                                                              \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, st, a);                                                                      \
  }
  if (rpw == 1) MRB_LR_LAUNCH(0, 1)
  else if (rpw == 2) MRB_LR_LAUNCH(1, 2)
  else MRB_LR_LAUNCH(2, 4)
#undef MRB_LR_LAUNCH
  return mrblip_check_launch("lora_rows");
}

// U[M, 0:R] = dropout(X)[M, K] A[R, K]^T (bf16 in / out, fp32 accumulate).  p_drop = 0 (or seed_ptr NULL): no mask (the backward's g = dy B).
// seg (optional, 2 * R/8 ints: [k0_0, k1_0, k0_1, k1_1, ...]): the column range in which rows [8j, 8j+8) of A are non-zero.
static int lora_rows_impl(const void* X, long long ldx, const void* A, long long lda, int M, int R, int K, void* U, long long ldu, const int* seg,
                          const uint32_t* seed_ptr, uint32_t site, float p_drop, float* init_dst, long long ld_idst, const float* init_src,
                          long long ld_isrc, int init_n, hipStream_t stream) {
  MRB_REQUIRE(M > 0 && K > 0 && (K % 8) == 0 && R > 0 && R <= 32 && (R % 8) == 0, "lora_rows: bad shape (M=%d R=%d K=%d)", M, R, K);
  MRB_REQUIRE((ldx % 8) == 0 && (lda % 8) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)A % 16) == 0 && ldu >= R, "lora_rows: 16-B alignment");
  MRB_REQUIRE(!(p_drop > 0.f) || seed_ptr, "lora_rows: dropout needs a device seed pointer");
  LoraRowsArgs a;
  a.X = (const bf16_t*)X; a.ldx = ldx; a.A = (const bf16_t*)A; a.lda = lda; a.U = (bf16_t*)U; a.ldu = ldu; a.M = M; a.K = K; a.R = R;
  for (int j = 0; j < 4; ++j) {
    a.seg_k0[j] = (seg && j < R / 8) ? seg[2 * j] : 0;
    a.seg_k1[j] = (seg && j < R / 8) ? seg[2 * j + 1] : K;
  }
  a.drop.seed_ptr = (p_drop > 0.f) ? seed_ptr : nullptr;
  a.drop.site = site;
  a.drop.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f);
  a.drop.inv_keep = 1.0f / (1.0f - p_drop);
  MRB_REQUIRE(!init_dst || ((init_n % 4) == 0 && (ld_idst % 4) == 0 && (!init_src || (ld_isrc % 4) == 0)), "lora_rows: init job needs 16-B rows");
  a.init_dst = init_dst; a.init_src = init_src; a.ld_idst = ld_idst; a.ld_isrc = ld_isrc; a.init_n = init_dst ? init_n : 0;
  a.x_gstride = a.a_gstride = a.u_gstride = 0; a.site_stride = 0;
  if (M >= lora_thin_min_m() && (K % 32) == 0 && (ldu % 4) == 0 && ((uintptr_t)U % 8) == 0 && (long long)M * ldx * 2 < (1ll << 31) && lora_thin_enabled())
    return launch_thin(a, stream);
  switch (R / 8) {
    case 1: return launch_rows<1>(a, stream);
    case 2: return launch_rows<2>(a, stream);
    case 3: return launch_rows<3>(a, stream);
    default: return launch_rows<4>(a, stream);
  }
}

extern "C" int mrblip_lora_rows(const void* X, long long ldx, const void* A, long long lda, int M, int R, int K, void* U, long long ldu,
                                const int* seg, const uint32_t* seed_ptr, uint32_t site, float p_drop, hipStream_t stream) {
  return lora_rows_impl(X, ldx, A, lda, M, R, K, U, ldu, seg, seed_ptr, site, p_drop, nullptr, 0, nullptr, 0, 0, stream);
}

// The same product for `groups` problems in ONE launch (round 4): group g reads X + g * x_gstride (0: every group reads the same rows),
// A + g * a_gstride, writes U + g * u_gstride (elements) and masks with call-site id site0 + g * site_stride.  Matrix-core thin kernel only:
// K % 32 == 0, ldu % 4 == 0, R <= 32 (any M: short inputs just leave CUs idle).
extern "C" int mrblip_lora_rows_batched(const void* X, long long ldx, long long x_gstride, const void* A, long long lda, long long a_gstride, int M, int R,
                                        int K, void* U, long long ldu, long long u_gstride, int groups, const uint32_t* seed_ptr, uint32_t site0,
                                        uint32_t site_stride, float p_drop, hipStream_t stream) {
  MRB_REQUIRE(M > 0 && K > 0 && (K % 32) == 0 && R > 0 && R <= 32 && (R % 8) == 0 && groups > 0 && groups <= 65535, "lora_rows_batched: bad shape (M=%d R=%d K=%d groups=%d)", M, R, K, groups);
  MRB_REQUIRE((ldx % 8) == 0 && (lda % 8) == 0 && (x_gstride % 8) == 0 && (a_gstride % 8) == 0 && (u_gstride % 4) == 0 && (ldu % 4) == 0 && ((uintptr_t)X % 16) == 0 &&
                  ((uintptr_t)A % 16) == 0 && ((uintptr_t)U % 8) == 0 && ldu >= R, "lora_rows_batched: alignment");
  MRB_REQUIRE((long long)M * ldx * 2 < (1ll << 31), "lora_rows_batched: X exceeds the 2 GiB buffer range");
  MRB_REQUIRE(!(p_drop > 0.f) || seed_ptr, "lora_rows_batched: dropout needs a device seed pointer");
  LoraRowsArgs a;
  a.X = (const bf16_t*)X; a.ldx = ldx; a.A = (const bf16_t*)A; a.lda = lda; a.U = (bf16_t*)U; a.ldu = ldu; a.M = M; a.K = K; a.R = R;
  for (int j = 0; j < 4; ++j) { a.seg_k0[j] = 0; a.seg_k1[j] = K; }
  a.drop.seed_ptr = (p_drop > 0.f) ? seed_ptr : nullptr;
  a.drop.site = site0;
  a.drop.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f);
  a.drop.inv_keep = 1.0f / (1.0f - p_drop);
  a.init_dst = nullptr; a.init_src = nullptr; a.ld_idst = a.ld_isrc = 0; a.init_n = 0;
  a.x_gstride = x_gstride; a.a_gstride = a_gstride; a.u_gstride = u_gstride; a.site_stride = site_stride;
  return launch_thin(a, stream, groups);
}

// mrblip_lora_rows + the side job  init_dst[m, 0:init_n] = init_src ? init_src[m, :] : 0  over the same M rows (fp32): prepares the
// output that a following K-split mrblip_gemm_bf16 (tile_cfg bits 17..20) accumulates into.
extern "C" int mrblip_lora_rows_init(const void* X, long long ldx, const void* A, long long lda, int M, int R, int K, void* U, long long ldu,
                                     const int* seg, const uint32_t* seed_ptr, uint32_t site, float p_drop, float* init_dst, long long ld_idst,
                                     const float* init_src, long long ld_isrc, int init_n, hipStream_t stream) {
  MRB_REQUIRE(init_dst && init_n > 0, "lora_rows_init: no destination");
  return lora_rows_impl(X, ldx, A, lda, M, R, K, U, ldu, seg, seed_ptr, site, p_drop, init_dst, ld_idst, init_src, ld_isrc, init_n, stream);
}

// ---- T5 RMSNorm + LoRA "down" of the normalised row in one launch -------------------------------------------------------------
struct NormLoraArgs {
  const float* x; long long ldx;     // fp32 residual stream [M, D]
  const float* gamma;
  bf16_t* xn; long long ldn;          // bf16 normalised rows (the GEMM operand)
  const bf16_t* A; long long lda;     // [R, D]
  bf16_t* U; long long ldu;
  int M, D, R;
  float eps;
  DropoutArg drop;
};

template <int NR>
__global__ __launch_bounds__(256) void rmsnorm_lora_kernel(const NormLoraArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [R][D] bf16
  constexpr int R = NR * 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nv = p.D >> 2;
  lora_fill(smem, p.A, p.lda, R, p.D, 0);
  __syncthreads();
  const uint32_t seed = p.drop.seed_ptr ? *p.drop.seed_ptr : 0u;
  const float post = p.drop.seed_ptr ? p.drop.inv_keep : 1.0f;
  for (int row = blockIdx.x * 4 + wv; row < p.M; row += gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(p.x + (long long)row * p.ldx);
    float4 v[8];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = lane + 64 * j;
      v[j] = (i < nv) ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      q += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)p.D + p.eps);
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = lane + 64 * j;
      if (i < nv) {
        const float4 g = reinterpret_cast<const float4*>(p.gamma)[i];
        uint2 o = make_uint2(pack2bf(v[j].x * rstd * g.x, v[j].y * rstd * g.y), pack2bf(v[j].z * rstd * g.z, v[j].w * rstd * g.w));
        reinterpret_cast<uint2*>(p.xn + (long long)row * p.ldn)[i] = o;
        if (p.drop.seed_ptr) {
          const uint32_t e = (uint32_t)row * (uint32_t)p.D + (uint32_t)(i * 4);
          bool k0, k1, k2, k3;
          mrb_keep2(e, seed, p.drop.site, p.drop.thresh24, k0, k1);
          mrb_keep2(e + 2, seed, p.drop.site, p.drop.thresh24, k2, k3);
          o.x = (k0 ? o.x & 0xffffu : 0u) | (k1 ? o.x & 0xffff0000u : 0u);
          o.y = (k2 ? o.y & 0xffffu : 0u) | (k3 ? o.y & 0xffff0000u : 0u);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const uint2 av = *reinterpret_cast<const uint2*>(smem + ((long long)r * p.D + i * 4) * 2);
          acc[r] = dot2bf(o.y, av.y, dot2bf(o.x, av.x, acc[r]));
        }
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float s = wave_sum_uniform(acc[r]);
      if (lane == r) mine = s;
    }
    if (lane < R) p.U[(long long)row * p.ldu + lane] = f2bf(mine * post);
  }
}

// xn = bf16(RMSNorm(x) * gamma);  U[:, 0:R] = dropout(xn) A^T.   D <= 2048, D % 8 == 0.
extern "C" int mrblip_rmsnorm_lora_fwd(const float* x, long long ldx, const float* weight, int M, int D, float eps, void* out_bf16, long long ldob,
                                       const void* A, long long lda, int R, void* U, long long ldu, const uint32_t* seed_ptr, uint32_t site,
                                       float p_drop, hipStream_t stream) {
  MRB_REQUIRE(M > 0 && D > 0 && D <= 2048 && (D % 8) == 0, "rmsnorm_lora: need 0 < D <= 2048, D %% 8 == 0 (M=%d D=%d)", M, D);
  MRB_REQUIRE(R > 0 && R <= 32 && (R % 8) == 0 && ldu >= R && (lda % 8) == 0 && (ldx % 4) == 0 && (ldob % 4) == 0, "rmsnorm_lora: bad shape (R=%d)", R);
  MRB_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)out_bf16 % 8) == 0, "rmsnorm_lora: alignment");
  MRB_REQUIRE(!(p_drop > 0.f) || seed_ptr, "rmsnorm_lora: dropout needs a device seed pointer");
  if (M >= lora_thin_min_m() && (D % 32) == 0 && (ldob % 8) == 0 && ((uintptr_t)out_bf16 % 16) == 0 && lora_thin_enabled()) {
    // tall inputs: the norm at HBM speed, then the matrix-core thin product on the rows it just wrote (they are still in the L2 / MALL)
    if (int e = mrblip_rmsnorm_fwd(x, ldx, weight, M, D, eps, out_bf16, ldob, nullptr, 0, stream)) return e;
    return lora_rows_impl(out_bf16, ldob, A, lda, M, R, D, U, ldu, nullptr, seed_ptr, site, p_drop, nullptr, 0, nullptr, 0, 0, stream);
  }
  NormLoraArgs a;
  a.x = x; a.ldx = ldx; a.gamma = weight; a.xn = (bf16_t*)out_bf16; a.ldn = ldob; a.A = (const bf16_t*)A; a.lda = lda; a.U = (bf16_t*)U; a.ldu = ldu;
  a.M = M; a.D = D; a.R = R; a.eps = eps;
  a.drop.seed_ptr = (p_drop > 0.f) ? seed_ptr : nullptr;
  a.drop.site = site;
  a.drop.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f);
  a.drop.inv_keep = 1.0f / (1.0f - p_drop);
  const int LDS = R * D * 2;
  const int grid = (M + 3) / 4 < 256 ? (M + 3) / 4 : 256;  // every block stages the thin vectors once and walks its rows
  static bool attr[5] = {};
#define MRB_NL_LAUNCH(NR_)                                                                                                         \
  case NR_: {                                                                                                                      \
    auto k = rmsnorm_lora_kernel<NR_>;                                                                                             \
    if (!attr[NR_]) {                                                                                                              \
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, NR_ * 8 * 2048 * 2) != hipSuccess) {      \
        mrblip_set_error("rmsnorm_lora: cannot raise dynamic LDS");                                                               \
        return MRBLIP_ELAUNCH;                                                                                                     \
      }                                                                                                                            \
      attr[NR_] = true;                                                                                                            \
    }                                                                                                                              \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, stream, a);                                                                  \
    break;                                                                                                                         \
  }
  switch (R / 8) {
    MRB_NL_LAUNCH(1)
    MRB_NL_LAUNCH(2)
    MRB_NL_LAUNCH(3)
    MRB_NL_LAUNCH(4)
  }
#undef MRB_NL_LAUNCH
  return mrblip_check_launch("rmsnorm_lora");
}
