// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] * W[N,K]^T (+ optional K-extension segment for LoRA)
// with fused epilogues (bias / exact-erf GELU / gated-GELU / dropout / fp32 residual add).
//
// Replaces the torch call sites  F.linear / nn.Linear / Conv2d-as-GEMM  of the reference hot path:
//   eva_vit.py:120-126 (qkv), :146 (proj), :54-61 (fc1/GELU/fc2), :196-203 (patch embed)
//   Qformer.py:141-147, 285-289, 349-375 (query/key/value/dense/intermediate/output)
//   modeling_t5.py:323-329 (wi_0/wi_1/wo gated-GELU), :536-560 (q/k/v/o), :1870 (lm_head), blip2_mr.py:491 (t5_proj)
//
// Structure (tile kernel): BMxBNx64 block tile, 32x32x16 bf16 MFMA, operands staged HBM->LDS with
// buffer_load ... lds (16 B/lane, bounds-checked SRD so ragged M/N edges need no branches), LDS rows are
// 128 B with a 16-B-chunk XOR swizzle applied on the SOURCE address (LDS-DMA writes lane-linear) and on the
// ds_read_b128 address, two LDS stages, one barrier per K tile.  The MFMA is issued "swapped"
// (A-operand = W tile, B-operand = X tile) so each lane owns one output row and 4 consecutive columns per
// accumulator group -> 16-B (fp32) / 8-B (bf16) row-major stores.
#include "common.h"

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* W;
  const bf16_t* Aext;  // [M,64] K-extension (LoRA "u") or nullptr
  const bf16_t* Wext;  // [N,64]
  void* out;
  void* out2;           // optional bf16 pre-activation output (GELU / gated), or nullptr
  const float* bias;    // [N] or nullptr
  const float* residual;  // fp32 [M,ldr] or nullptr
  long long lda, ldw, ldaext, ldwext, ldo, ldo2, ldr;
  int M, N, K;  // N = rows of W (for GATED: 2*Nh, output has Nh columns)
  int act;      // 0 none, 1 gelu(erf)
  int tiles_m, tiles_n;
  DropoutArg drop;
};

__device__ __forceinline__ void epilogue_store4(const GemmArgs& p, bool out_f32, int m, int n0, float v[4], int ncols) {
  // v: 4 consecutive columns n0..n0+3 of row m (accumulator + nothing else yet)
  if (p.bias) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n0);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (p.out2) {
    uint2 pk = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out2) + (long long)m * p.ldo2 + n0) = pk;
  }
  if (p.act == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
  }
  if (p.drop.seed_ptr) {
    const uint32_t seed = *p.drop.seed_ptr;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = mrb_keep((uint32_t)m * (uint32_t)ncols + (uint32_t)(n0 + i), seed, p.drop.site, p.drop.thresh24) ? v[i] * p.drop.inv_keep : 0.f;
  }
  if (p.residual) {
    const float4 r = *reinterpret_cast<const float4*>(p.residual + (long long)m * p.ldr + n0);
    v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
  }
  if (out_f32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (long long)m * p.ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    uint2 pk = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long long)m * p.ldo + n0) = pk;
  }
}

__device__ __forceinline__ void epilogue_gated4(const GemmArgs& p, int m, int n0, const float h0[4], const float h1[4], int nh) {
  // y = dropout(gelu(h0) * h1) -> bf16 out[m, n0..]; optional out2 = [h0 | h1] stacked ([M, 2*nh])
  if (p.out2) {
    bf16_t* o2 = reinterpret_cast<bf16_t*>(p.out2) + (long long)m * p.ldo2;
    *reinterpret_cast<uint2*>(o2 + n0) = make_uint2(pack2bf(h0[0], h0[1]), pack2bf(h0[2], h0[3]));
    *reinterpret_cast<uint2*>(o2 + nh + n0) = make_uint2(pack2bf(h1[0], h1[1]), pack2bf(h1[2], h1[3]));
  }
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = gelu_erf(h0[i]) * h1[i];
  if (p.drop.seed_ptr) {
    const uint32_t seed = *p.drop.seed_ptr;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v[i] = mrb_keep((uint32_t)m * (uint32_t)nh + (uint32_t)(n0 + i), seed, p.drop.site, p.drop.thresh24) ? v[i] * p.drop.inv_keep : 0.f;
  }
  *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long long)m * p.ldo + n0) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int BM, int BN, int WGM, int WGN, bool OUT_F32, bool GATED>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_tile_kernel(const GemmArgs p) {
  constexpr int NW = WGM * WGN;
  constexpr int TM = BM / WGM / 32;  // 32x32 tiles per wave along M
  constexpr int TN = BN / WGN / 32;
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
  constexpr int JA = BM / 8 / NW, JW = BN / 8 / NW;  // LDS-DMA instructions per wave per operand tile
  static_assert(!GATED || TN == 2, "gated epilogue pairs the wave's two n-tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w / WGN, wn = w % WGN;
  const int hi = lane >> 5, l31 = lane & 31;

  // ---- block -> tile: XCD-contiguous remap (bijective), then grouped ordering for L2 reuse of the W panel
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int gid = bid / per_group;
  const int first_m = gid * GROUP_M;
  const int gsize = min(p.tiles_m - first_m, GROUP_M);
  const int bm = first_m + (bid % per_group) % gsize;
  const int bn = (bid % per_group) / gsize;

  const int Nh = p.N >> 1;                    // GATED only
  constexpr int BNO = GATED ? BN / 2 : BN;  // output columns per block

  // ---- buffer resources (bounds-checked: rows >= M / >= N read as zero, never fault)
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(uint32_t)((long long)p.M * p.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)(uint32_t)((long long)p.N * p.ldw * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rAe = __builtin_amdgcn_make_buffer_rsrc((void*)p.Aext, 0, (int)(uint32_t)((long long)p.M * p.ldaext * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rWe = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wext, 0, (int)(uint32_t)((long long)p.N * p.ldwext * 2), 0x00020000);

  // per-lane source offset inside one 8-row LDS-DMA piece: row (lane>>3), swizzled 16-B chunk
  const int prow = lane >> 3;
  const int chunk = (lane & 7) ^ ((w * 4 + (lane >> 4)) & 7);  // == (lane&7) ^ ((tile_row>>1)&7)
  const uint32_t vA = (uint32_t)((long long)prow * p.lda * 2) + chunk * 16;
  const uint32_t vW = (uint32_t)((long long)prow * p.ldw * 2) + chunk * 16;
  const uint32_t vAe = (uint32_t)((long long)prow * p.ldaext * 2) + chunk * 16;
  const uint32_t vWe = (uint32_t)((long long)prow * p.ldwext * 2) + chunk * 16;

  const int nk_main = p.K >> 6;
  const int nk = nk_main + (p.Aext ? 1 : 0);

  auto w_row_base = [&](int j) -> int {  // global W row of tile row (j*NW + w)*8
    const int tr = (j * NW + w) * 8;
    if (GATED) return (tr < BN / 2) ? bn * BNO + tr : Nh + bn * BNO + (tr - BN / 2);
    return bn * BN + tr;
  };

  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE;
    const bool ext = kt >= nk_main;
    if (!ext) {
      const uint32_t koff = (uint32_t)kt * 128u;
#pragma unroll
      for (int j = 0; j < JA; ++j) {
        const int tr = (j * NW + w) * 8;
        const uint32_t so = (uint32_t)((long long)(bm * BM + tr) * p.lda * 2) + koff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(base + tr * 128), 16, vA, so, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < JW; ++j) {
        const int tr = (j * NW + w) * 8;
        const uint32_t so = (uint32_t)((long long)w_row_base(j) * p.ldw * 2) + koff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(base + A_BYTES + tr * 128), 16, vW, so, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < JA; ++j) {
        const int tr = (j * NW + w) * 8;
        const uint32_t so = (uint32_t)((long long)(bm * BM + tr) * p.ldaext * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rAe, (lds_ptr_t)(base + tr * 128), 16, vAe, so, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < JW; ++j) {
        const int tr = (j * NW + w) * 8;
        const uint32_t so = (uint32_t)((long long)w_row_base(j) * p.ldwext * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rWe, (lds_ptr_t)(base + A_BYTES + tr * 128), 16, vWe, so, 0, 0);
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (bytes) inside a stage
  const int swz = (lane >> 1) & 7;
  const int a_row_off = (wm * (BM / WGM) + l31) * 128;
  int w_row_off[TN];
#pragma unroll
  for (int nt = 0; nt < TN; ++nt)
    w_row_off[nt] = A_BYTES + (GATED ? (nt * (BN / 2) + wn * 32 + l31) : (wn * (BN / WGN) + nt * 32 + l31)) * 128;

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* base = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int coff = (((kk * 2 + hi) ^ swz) << 4);
      bf16x8 xf[TM], wf[TN];
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) xf[mt] = *reinterpret_cast<const bf16x8*>(base + a_row_off + mt * 32 * 128 + coff);
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) wf[nt] = *reinterpret_cast<const bf16x8*>(base + w_row_off[nt] + coff);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
    }
  }

  // ---- epilogue: lane owns output row m = ... + l31; per accumulator group g: 4 consecutive columns
  const int ncols = GATED ? Nh : p.N;
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) {
    const int m = bm * BM + wm * (BM / WGM) + mt * 32 + l31;
    if (m >= p.M) continue;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nloc = 8 * g + 4 * hi;
      if (GATED) {
        const int n0 = bn * BNO + wn * 32 + nloc;
        if (n0 < Nh) {
          float h0[4], h1[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { h0[i] = acc[mt][0][4 * g + i]; h1[i] = acc[mt][TN - 1][4 * g + i]; }
          epilogue_gated4(p, m, n0, h0, h1, Nh);
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          const int n0 = bn * BN + wn * (BN / WGN) + nt * 32 + nloc;
          if (n0 < p.N) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[mt][nt][4 * g + i];
            epilogue_store4(p, OUT_F32, m, n0, v, ncols);
          }
        }
      }
    }
  }
}

// ---- skinny-M kernel (decoder rows, M <= a few 32-row tiles): weight-streaming bound.  One block = 32 output
// columns x 32 rows; its 4 waves split K, each lane streams 64 contiguous bytes of one W row per 64-wide k block
// (the contraction order inside a k block is permuted identically for W and X so both load 16-B vectors from full
// 128-B lines), partial sums are reduced through LDS.
template <bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const GemmArgs p) {
  __shared__ float red[3][16][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int n_row = blockIdx.x * 32 + l31;  // W row (MFMA A-operand row)
  const int m_row = blockIdx.y * 32 + l31;  // X row (MFMA B-operand column)
  const bool n_ok = n_row < p.N, m_ok = m_row < p.M;
  const int nkb = p.K >> 6;  // 64-wide k blocks of the main segment
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const bf16_t* wp = p.W + (long long)n_row * p.ldw + hi * 32;
  const bf16_t* xp = p.A + (long long)m_row * p.lda + hi * 32;
  // each wave owns a contiguous quarter of the k blocks; 4 blocks (32 independent 16-B loads per lane) are in flight at a time
  const int per = (nkb + 3) >> 2;
  const int kb_beg = w * per, kb_end = min(nkb, kb_beg + per);
#pragma unroll 1
  for (int kb0 = kb_beg; kb0 < kb_end; kb0 += 4) {
    bf16x8 wf[4][4], xf[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kb = min(kb0 + u, kb_end - 1);
      const bool live = kb0 + u < kb_end;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        wf[u][s] = (n_ok && live) ? *reinterpret_cast<const bf16x8*>(wp + kb * 64 + s * 8) : zero;
        xf[u][s] = (m_ok && live) ? *reinterpret_cast<const bf16x8*>(xp + kb * 64 + s * 8) : zero;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u][s], xf[u][s], acc, 0, 0, 0);
  }
  if (p.Aext && w == 3) {  // K-extension segment (one 64-wide block), taken by the last wave
    const bf16_t* wpe = p.Wext + (long long)n_row * p.ldwext + hi * 32;
    const bf16_t* xpe = p.Aext + (long long)m_row * p.ldaext + hi * 32;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 wf = n_ok ? *reinterpret_cast<const bf16x8*>(wpe + s * 8) : zero;
      const bf16x8 xf = m_ok ? *reinterpret_cast<const bf16x8*>(xpe + s * 8) : zero;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
    }
  }
  if (w > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
    if (m_ok) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = blockIdx.x * 32 + 8 * g + 4 * hi;
        if (n0 < p.N) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[4 * g + i];
          epilogue_store4(p, OUT_F32, m_row, n0, v, p.N);
        }
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN, bool OUT_F32, bool GATED>
static int launch_tile(GemmArgs& a, hipStream_t st) {
  constexpr int BNO = GATED ? BN / 2 : BN;
  const int ncols = GATED ? a.N / 2 : a.N;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (ncols + BNO - 1) / BNO;
  constexpr int LDS = 2 * (BM + BN) * 128;
  auto kern = gemm_tile_kernel<BM, BN, WGM, WGN, OUT_F32, GATED>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      mrblip_set_error("gemm: cannot raise dynamic LDS to %d", LDS);
      return MRBLIP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(WGM * WGN * 64), LDS, st, a);
  return mrblip_check_launch("gemm_tile");
}

extern "C" int mrblip_gemm_bf16(const void* A, long long lda, const void* W, long long ldw, const void* Aext, long long ldaext,
                                const void* Wext, long long ldwext, int M, int N, int K, void* out, long long ldo, int out_f32,
                                void* out2, long long ldo2, const float* bias, const float* residual, long long ldr, int act,
                                int gated, const uint32_t* seed_ptr, uint32_t site, float p_drop, int tile_cfg,
                                hipStream_t stream) {
  MRB_REQUIRE(M > 0 && N > 0 && K >= 0 && (K % 64) == 0, "gemm: need M,N>0 and K%%64==0 (M=%d N=%d K=%d)", M, N, K);
  MRB_REQUIRE(K > 0 || Aext, "gemm: empty contraction");
  MRB_REQUIRE((N % 8) == 0, "gemm: N %% 8 != 0 (N=%d)", N);
  MRB_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && (ldo % 4) == 0, "gemm: leading dims must keep 16-B alignment");
  MRB_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm: pointers must be 16-B aligned");
  MRB_REQUIRE((Aext == nullptr) == (Wext == nullptr), "gemm: Aext/Wext must come together");
  MRB_REQUIRE(((long long)(M + 256) * lda * 2 + 256) < (1ll << 32) && ((long long)(N + 256) * ldw * 2 + 256) < (1ll << 32),
              "gemm: operand exceeds the 4 GiB buffer-descriptor range");
  MRB_REQUIRE(!gated || (!out_f32 && !bias && !residual && act == 0 && (N % 16) == 0), "gemm: gated mode takes no bias/residual/act");
  GemmArgs a;
  a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.Aext = (const bf16_t*)Aext; a.Wext = (const bf16_t*)Wext;
  a.out = out; a.out2 = out2; a.bias = bias; a.residual = residual;
  a.lda = lda; a.ldw = ldw; a.ldaext = Aext ? ldaext : 0; a.ldwext = Wext ? ldwext : 0; a.ldo = ldo; a.ldo2 = ldo2; a.ldr = ldr;
  a.M = M; a.N = N; a.K = K; a.act = act; a.tiles_m = a.tiles_n = 0;
  a.drop.seed_ptr = (p_drop > 0.f) ? seed_ptr : nullptr;
  a.drop.site = site;
  a.drop.thresh24 = (uint32_t)(p_drop * 16777216.0f + 0.5f);
  a.drop.inv_keep = 1.0f / (1.0f - p_drop);
  MRB_REQUIRE(!(p_drop > 0.f) || seed_ptr, "gemm: dropout needs a device seed pointer");
  // tile_cfg: 0 auto, 1 = 256x256, 2 = 128x128, 3 = skinny
  int cfg = tile_cfg;
  if (cfg == 0) {
    if (M <= 64 && !gated) cfg = 3;
    else {
      // measured on MI355X (tools/gemm_bench.py): the 256x256 one-barrier-per-K-tile kernel (1 block/CU) only wins for long-K,
      // many-tile problems; the 128x128 variant (2 blocks/CU hide each other's staging latency) wins at the hot-path shapes.
      const long long t256 = (long long)((M + 255) / 256) * (((gated ? N / 2 : N) + (gated ? 127 : 255)) / (gated ? 128 : 256));
      cfg = (t256 >= 512 && K >= 4096) ? 1 : 2;
    }
  }
  if (cfg == 3) {
    MRB_REQUIRE(!gated, "gemm: skinny kernel has no gated epilogue");
    dim3 grid((N + 31) / 32, (M + 31) / 32);
    if (out_f32) hipLaunchKernelGGL(gemm_skinny_kernel<true>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gemm_skinny_kernel<false>, grid, dim3(256), 0, stream, a);
    return mrblip_check_launch("gemm_skinny");
  }
  if (cfg == 1) {
    if (gated) return launch_tile<256, 256, 2, 4, false, true>(a, stream);
    return out_f32 ? launch_tile<256, 256, 2, 4, true, false>(a, stream) : launch_tile<256, 256, 2, 4, false, false>(a, stream);
  }
  if (gated) return launch_tile<128, 128, 2, 2, false, true>(a, stream);
  return out_f32 ? launch_tile<128, 128, 2, 2, true, false>(a, stream) : launch_tile<128, 128, 2, 2, false, false>(a, stream);
}
