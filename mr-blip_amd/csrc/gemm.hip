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
#include "lora_thin.h"
#include <type_traits>
// TILE_PIPE (experiment, off): 1 = fragments of k-step kk+1 requested before the MFMAs of kk AND the next stage's LDS-DMA pieces spread over
// the k-steps (64x128 tile, K = 10240: 129.7 us against 107.3 us — pieces issued later land later, the 2-stage ring waits for them);
// 2 = the fragment prefetch alone (107.5 us: equal).  The compiler's own schedule stays.
#ifndef TILE_PIPE
#define TILE_PIPE 0
#endif
// TILE_REGSTAGE (experiment): operand tiles through registers (buffer_load_b128 -> VGPRs -> ds_write_b128, same LDS image as the LDS-DMA
// pieces) instead of LDS-DMA: an LDS-DMA piece occupies a SIMD's issue for ~100-200 clocks, which caps a CU's fill rate at ~45 GB/s with
// one block per CU (75 GB/s with two) — the per-K-tile time of the small tiles tracks their DMA bytes, not their MFMAs.
#ifndef TILE_REGSTAGE
#define TILE_REGSTAGE 0
#endif
#include <stdlib.h>

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
  int act;      // 0 none, 1 gelu(erf), 2 (round 6, generic tile kernel, bf16 out): GELU BACKWARD — out = bf16(bf16(acc) * gelu'(out2[m, n])), out2 is READ
                // (the forward's saved pre-activation): the same bits as a bf16 GEMM followed by mrblip_gelu_bwd
  int tiles_m, tiles_n;
  DropoutArg drop;
  // LoRA backward form  out = A W^T + mask(ext_drop) * (Aext Wext^T):  the K-extension tile is taken FIRST and the accumulators
  // (then holding only its product) are masked in registers before the main K loop adds to them
  int ext_first;
  DropoutArg ext_drop;
  int m_rows_per_block;  // skinny kernel only: 32, or 16 / 8 when few output columns leave most CUs without a block (more blocks stream
                         // the tall operand in parallel: one CU sustains only ~10 B/clk from HBM)
  DropoutArg a_drop;  // skinny kernel only: dropout of the A operand as it is loaded (zeroing; the 1/(1-p) is applied to the result)
  // skinny kernel only: gridDim.z blocks share one output tile along K and ADD their partial products into the fp32 output with atomics
  // (the caller pre-initialised it: residual or zero).  With M <= 32 rows a [32 x N] output has only N / 32 tiles — 64 for the T5
  // d_model — and one CU streams ~20 GB/s of weights: the K split puts every CU on the weight stream.
  int k_splits;
  long long part_stride;   // 4-wave kernel in SPLIT form: elements between the partial outputs (gemm_w4_kernel)
  // round 4 (tile kernels, bf16 output, plain / bias epilogue only; set through mrblip_gemm_set_extra for the NEXT launch of the thread):
  // head-transposed copies of up to three consecutive column ranges of width t_inner (heads of 64) — what mrblip_head_transpose would
  // write from the output, pad columns [t_rows, t_spad) included: tout[j][b][h][d][s] = out[b * t_rows + s][j * t_inner + h * 64 + d];
  // needs B == 1 or t_rows % 32 == 0 (a wave's 32-row slab then lies inside one clip, 32-aligned).
  // t_count > 0: range j < t_count goes to tout[0] + j * t_stride (elements) instead — any number of ranges in one buffer.
  bf16_t* tout[3]; int t_inner, t_rows, t_spad; long long t_bs, t_hs, t_stride; int t_count;
  // the K extension of output-column group g = n / ext_group_n reads columns [64 g, 64 g + 64) of Aext (0: one group): one GEMM for the
  // cross-attention K / V projections of ALL decoder layers, each layer with its own LoRA "down" activations
  int ext_group_n;
  int group_m;      // generic tile kernel: row tiles per group of the tile walk (launch_tile)
  // round 4, generic tile kernel: the FIRST pf_blocks workgroups of the grid (a multiple of 8: the tiles keep their XCD) do not compute;
  // they read [pf_ptr, pf_ptr + 16 pf_n16) and drop it — the weights of a LATER launch, pulled into the memory-side cache while this
  // one computes (mrblip_gemm_set_prefetch)
  // thin role (round 4, generic tile kernel; mrblip_gemm_set_thin): the th_blocks workgroups behind the prefetch ones compute the K
  // extension's A operand themselves — Aext[m, 0:th_R] = dropout(A)[m, 0:th_K] th_A^T, the LoRA "down" product (body: lora_thin.h), 16
  // rows each — write it with write-through stores and set th_flags[row block] = th_epoch; a tile waits for the flags of its rows before
  // it stages the K extension (its LAST K-tile) and reads Aext past its XCD's L2
  const bf16_t* th_A; long long th_lda; int th_R, th_K; DropoutArg th_drop; uint32_t* th_flags; uint32_t* th_err; uint32_t* th_tick; uint32_t th_epoch; int th_blocks; int th_stall;   // th_stall: TEST HOOK (mrblip_gemm_debug_stall_thin) — the role exits without publishing
  int pf_base, th_base, tile_base;   // first block id of the prefetch / thin / tile workgroups (launch_tile lays the three groups out)
  const void* pf_ptr; long long pf_n16; const void* pf_ptr2; long long pf_n16_2; int pf_blocks;   // (a second, usually small range: the LoRA K-extension operand)
};

// v0..v3: 4 consecutive columns n0..n0+3 of row m (raw accumulator). Applies bias -> (pre-activation copy) -> GELU -> dropout -> residual.
// (scalars by reference and `pre` always produced: a float[4] handed over by pointer, or `flag ? &local : nullptr`, makes the
// compiler keep the values in scratch memory)
__device__ __forceinline__ void epilogue_apply4(const GemmArgs& p, int m, int n0, float& v0, float& v1, float& v2, float& v3, int ncols, uint2& pre) {
  if (p.bias) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n0);
    v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w;
  }
  pre = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
  if (p.act == 1) {
    gelu_erf2(v0, v1);
    gelu_erf2(v2, v3);
  }
  if (p.drop.seed_ptr) {
    const uint32_t seed = mrb_seed_load(p.drop.seed_ptr);
    const uint32_t e = (uint32_t)m * (uint32_t)ncols + (uint32_t)n0;
    bool k0, k1, k2, k3;   // (n0 % 4 == 0 and an even row length: e is even)
    mrb_keep4(e, seed, p.drop.site, p.drop.thresh24, k0, k1, k2, k3);
    v0 = k0 ? v0 * p.drop.inv_keep : 0.f;
    v1 = k1 ? v1 * p.drop.inv_keep : 0.f;
    v2 = k2 ? v2 * p.drop.inv_keep : 0.f;
    v3 = k3 ? v3 * p.drop.inv_keep : 0.f;
  }
  if (p.residual) {
    const float4 r = *reinterpret_cast<const float4*>(p.residual + (long long)m * p.ldr + n0);
    v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
  }
}

__device__ __forceinline__ void epilogue_store4(const GemmArgs& p, bool out_f32, int m, int n0, float v0, float v1, float v2, float v3, int ncols) {
  uint2 pre;
  epilogue_apply4(p, m, n0, v0, v1, v2, v3, ncols, pre);
  if (p.out2) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out2) + (long long)m * p.ldo2 + n0) = pre;
  if (out_f32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (long long)m * p.ldo + n0) = make_float4(v0, v1, v2, v3);
  } else {
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long long)m * p.ldo + n0) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
  }
}

// 8 consecutive columns of one row: 16-B bf16 stores (half the store instructions of the 4-wide form), 2 x 16 B for fp32
__device__ __forceinline__ void epilogue_store8(const GemmArgs& p, bool out_f32, int m, int n0, float4 a, float4 b, int ncols) {
  uint2 pre0, pre1;
  epilogue_apply4(p, m, n0, a.x, a.y, a.z, a.w, ncols, pre0);
  epilogue_apply4(p, m, n0 + 4, b.x, b.y, b.z, b.w, ncols, pre1);
  if (p.out2) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out2) + (long long)m * p.ldo2 + n0) = make_uint4(pre0.x, pre0.y, pre1.x, pre1.y);
  if (out_f32) {
    float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (long long)m * p.ldo + n0);
    o[0] = a;
    o[1] = b;
  } else {
#ifdef EXP_NOSTORE
    if (a.x == 12345.678f)  // EXPERIMENT (wrong results): epilogue math without the global stores
#endif
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (long long)m * p.ldo + n0) =
        make_uint4(pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w));
  }
}

// ---- the same epilogue with the operands ALREADY LOADED (tile kernel).  Round 3, ISA reading of the run-time-flag helpers above inside
// the tile kernel: every `if (p.bias) load`, `*p.drop.seed_ptr` (a VECTOR load of the seed) and `if (p.residual) load` sat in its own
// branch, each followed by s_waitcnt vmcnt(0) — per 8-column item 2 seed + 2 residual round trips in series, 4 items per wave and
// tile: ~10 us of exposed memory latency per [2012 x 2048] tile whose K loop takes ~25 us.  Now the kernel fetches the seed once with
// a scalar load, and reads bias / residual of all its items up front through bounds-checked buffer resources (absent operand: zero
// records -> zeros, no branch), so all loads of a slab are in flight together.
__device__ __forceinline__ void epilogue_apply4v(const GemmArgs& p, uint32_t seed, int m, int n0, float& v0, float& v1, float& v2, float& v3, int ncols,
                                                 uint2& pre, const float4 b, const float4 r) {
  if (p.bias) { v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w; }
  pre = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
  if (p.act == 1) {
    gelu_erf2(v0, v1);
    gelu_erf2(v2, v3);
  }
  if (p.drop.seed_ptr) {
    const uint32_t e = (uint32_t)m * (uint32_t)ncols + (uint32_t)n0;
    bool k0, k1, k2, k3;   // (n0 % 4 == 0 and an even row length: e is even)
    mrb_keep4(e, seed, p.drop.site, p.drop.thresh24, k0, k1, k2, k3);
    v0 = k0 ? v0 * p.drop.inv_keep : 0.f;
    v1 = k1 ? v1 * p.drop.inv_keep : 0.f;
    v2 = k2 ? v2 * p.drop.inv_keep : 0.f;
    v3 = k3 ? v3 * p.drop.inv_keep : 0.f;
  }
  if (p.residual) { v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w; }
}

// act == 2: v <- bf16(v) * gelu'(h) for the 8 saved pre-activations h = out2[m, n0 .. n0 + 7] (rounded through bf16 first: what the bf16 output
// of a plain GEMM would have handed mrblip_gelu_bwd)
__device__ __forceinline__ void epilogue_gelu_bwd8(const GemmArgs& p, int m, int n0, float4& a, float4& b) {
  const uint4 h = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.out2) + (long long)m * p.ldo2 + n0);
  a.x = bf2f(f2bf(a.x)) * gelu_erf_grad(bf2f((bf16_t)(h.x & 0xffffu))); a.y = bf2f(f2bf(a.y)) * gelu_erf_grad(bf2f((bf16_t)(h.x >> 16)));
  a.z = bf2f(f2bf(a.z)) * gelu_erf_grad(bf2f((bf16_t)(h.y & 0xffffu))); a.w = bf2f(f2bf(a.w)) * gelu_erf_grad(bf2f((bf16_t)(h.y >> 16)));
  b.x = bf2f(f2bf(b.x)) * gelu_erf_grad(bf2f((bf16_t)(h.z & 0xffffu))); b.y = bf2f(f2bf(b.y)) * gelu_erf_grad(bf2f((bf16_t)(h.z >> 16)));
  b.z = bf2f(f2bf(b.z)) * gelu_erf_grad(bf2f((bf16_t)(h.w & 0xffffu))); b.w = bf2f(f2bf(b.w)) * gelu_erf_grad(bf2f((bf16_t)(h.w >> 16)));
}

__device__ __forceinline__ void epilogue_store8v(const GemmArgs& p, uint32_t seed, bool out_f32, int m, int n0, float4 a, float4 b, int ncols,
                                                 const float4 b0, const float4 b1, const float4 r0, const float4 r1) {
  uint2 pre0, pre1;
  if (p.act == 2) {   // (launch-uniform)
    epilogue_gelu_bwd8(p, m, n0, a, b);
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (long long)m * p.ldo + n0) =
        make_uint4(pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w));
    return;
  }
  epilogue_apply4v(p, seed, m, n0, a.x, a.y, a.z, a.w, ncols, pre0, b0, r0);
  epilogue_apply4v(p, seed, m, n0 + 4, b.x, b.y, b.z, b.w, ncols, pre1, b1, r1);
  if (p.out2) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out2) + (long long)m * p.ldo2 + n0) = make_uint4(pre0.x, pre0.y, pre1.x, pre1.y);
  if (out_f32) {
    float4* o = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (long long)m * p.ldo + n0);
    o[0] = a;
    o[1] = b;
  } else {
    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (long long)m * p.ldo + n0) =
        make_uint4(pack2bf(a.x, a.y), pack2bf(a.z, a.w), pack2bf(b.x, b.y), pack2bf(b.z, b.w));
  }
}

__device__ __forceinline__ void epilogue_store4v(const GemmArgs& p, uint32_t seed, bool out_f32, int m, int n0, float v0, float v1, float v2, float v3,
                                                 int ncols, const float4 b, const float4 r) {
  uint2 pre;
  epilogue_apply4v(p, seed, m, n0, v0, v1, v2, v3, ncols, pre, b, r);
  if (p.out2) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out2) + (long long)m * p.ldo2 + n0) = pre;
  if (out_f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (long long)m * p.ldo + n0) = make_float4(v0, v1, v2, v3);
  else *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long long)m * p.ldo + n0) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
}

// gated: y = dropout(gelu(h0) * h1) for 8 consecutive columns; optional out2 = [h0 | h1] stacked ([M, 2*nh])
__device__ __forceinline__ void epilogue_gated8(const GemmArgs& p, uint32_t seed, int m, int n0, const float h0[8], const float h1[8], int nh) {
  if (p.out2) {
    bf16_t* o2 = reinterpret_cast<bf16_t*>(p.out2) + (long long)m * p.ldo2;
    *reinterpret_cast<uint4*>(o2 + n0) = make_uint4(pack2bf(h0[0], h0[1]), pack2bf(h0[2], h0[3]), pack2bf(h0[4], h0[5]), pack2bf(h0[6], h0[7]));
    *reinterpret_cast<uint4*>(o2 + nh + n0) = make_uint4(pack2bf(h1[0], h1[1]), pack2bf(h1[2], h1[3]), pack2bf(h1[4], h1[5]), pack2bf(h1[6], h1[7]));
  }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    float g0 = h0[i], g1 = h0[i + 1];
    gelu_erf2(g0, g1);
    v[i] = g0 * h1[i];
    v[i + 1] = g1 * h1[i + 1];
  }
  if (p.drop.seed_ptr) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {   // (n0 % 8 == 0, nh even: pairs share a hash)
      bool k0, k1;
      mrb_keep2((uint32_t)m * (uint32_t)nh + (uint32_t)(n0 + i), seed, p.drop.site, p.drop.thresh24, k0, k1);
      v[i] = k0 ? v[i] * p.drop.inv_keep : 0.f;
      v[i + 1] = k1 ? v[i + 1] * p.drop.inv_keep : 0.f;
    }
  }
  *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + (long long)m * p.ldo + n0) =
      make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef uint32_t mrb_u32x4 __attribute__((ext_vector_type(4)));
typedef float mrb_f32x4 __attribute__((ext_vector_type(4)));

// One K-tile of operands: CA + CW LDS-DMA pieces per wave (1 KiB each, lane-linear in LDS); per-piece source offsets come in VGPRs,
// the only per-K-tile scalar is the byte offset along k.  (A device function, not a lambda: the buffer-resource type does not
// exist in the host pass, and a kernel lambda holding one silently loses its host stub.)
// PARTS / PART: issue only the pieces whose running index (A pieces first, then W) is PART modulo PARTS — the pipelined K loop spreads
// a stage's pieces over the k-steps of the K-tile that is being multiplied (PARTS = 1: all of them)
template <int CA, int CW, int NW_, int PIECE_BYTES, int PARTS = 1, int PART = 0>
__device__ __forceinline__ void gemm_stage_dma(char* dstA, char* dstW, const void* pa, uint32_t bytes_a, const void* pw, uint32_t bytes_w,
                                               const uint32_t* va, const uint32_t* vw, int w, uint32_t koff) {
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pa), 0, (int)bytes_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pw), 0, (int)bytes_w, 0x00020000);
#pragma unroll
  for (int j = 0; j < CA; ++j)
    if (j % PARTS == PART) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(dstA + (j * NW_ + w) * PIECE_BYTES), 16, va[j], koff, 0, 0);
#pragma unroll
  for (int j = 0; j < CW; ++j)
    if ((CA + j) % PARTS == PART) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dstW + (j * NW_ + w) * PIECE_BYTES), 16, vw[j], koff, 0, 0);
}
// (same body under a second name: two call sites of ONE function get merged into a call with selected array pointers, which sends
// the offset arrays through scratch memory)
// AUX_A: cache-policy bits of the A pieces (16 = sc1: agent-scope read past the XCD's L2 — the thin role of another XCD wrote them)
template <int CA, int CW, int NW_, int PIECE_BYTES, int PARTS = 1, int PART = 0, int AUX_A = 0>
__device__ __forceinline__ void gemm_stage_dma_ext(char* dstA, char* dstW, const void* pa, uint32_t bytes_a, const void* pw, uint32_t bytes_w,
                                               const uint32_t* va, const uint32_t* vw, int w, uint32_t koff) {
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pa), 0, (int)bytes_a, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pw), 0, (int)bytes_w, 0x00020000);
#pragma unroll
  for (int j = 0; j < CA; ++j)
    if (j % PARTS == PART) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(dstA + (j * NW_ + w) * PIECE_BYTES), 16, va[j], koff, 0, AUX_A);
#pragma unroll
  for (int j = 0; j < CW; ++j)
    if ((CA + j) % PARTS == PART) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(dstW + (j * NW_ + w) * PIECE_BYTES), 16, vw[j], koff, 0, 0);
}

// blocks per CU the register allocation must allow: what the LDS footprint permits (160 KB per CU), at most 16 waves per CU
constexpr int gemm_min_blocks(int lds_bytes, int nwaves) {
  int b = lds_bytes <= 40 * 1024 ? 4 : lds_bytes <= 53 * 1024 ? 3 : lds_bytes <= 80 * 1024 ? 2 : 1;
  while (b > 1 && b * nwaves > 16) --b;
  return nwaves > 4 && lds_bytes > 80 * 1024 ? 1 : b;
}

// BK = k extent of one LDS stage (64: 128-B rows, 8 chunks; 32: 64-B rows, 4 chunks), NS = stages in the ring (2, or 3 for the
// non-persistent BK = 32 form: prefetch distance two K-tiles with a counted vmcnt wait — the main loop holds only LDS-DMA loads,
// which retire in order).
template <int BM, int BN, int WGM, int WGN, bool OUT_F32, bool GATED, int BK = 64, int NS = 2>
__global__ __launch_bounds__(WGM* WGN * 64, gemm_min_blocks(NS * (BM + BN) * BK * 2, WGM * WGN)) void gemm_tile_kernel(const GemmArgs p) {
  constexpr int NW = WGM * WGN;
  constexpr int TM = BM / WGM / 32;  // 32x32 tiles per wave along M
  constexpr int TN = BN / WGN / 32;
  constexpr int RB = BK * 2, CPR = BK / 8, RPI = 64 / CPR;  // LDS row bytes, 16-B chunks per row, rows per LDS-DMA instruction
  constexpr int KK = BK / 16, NEXT = 64 / BK;                // MFMA k-steps per stage, stages of the 64-wide K extension
  constexpr int A_BYTES = BM * RB, W_BYTES = BN * RB, STAGE = A_BYTES + W_BYTES;
  constexpr int JA = BM / RPI / NW, JW = BN / RPI / NW;  // LDS-DMA instructions per wave per operand tile
  static_assert(BK == 64 || BK == 32, "BK");
  static_assert(!GATED || TN == 2, "gated epilogue pairs the wave's two n-tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // the THIN role workgroups sit in FRONT of the tiles (handed out before any tile that waits for them; round 4 also put them BEHIND the
  // tiles where those left slots idle — starved under contention, see launch_tile); the prefetch ones, which nothing waits for, sit in
  // front, or behind the tiles where the tiles leave workgroup slots idle / in the last, partly empty round of a multi-round grid
  // (launch_tile decides: pf_base / th_base / tile_base)
  // Round 5: a launch with a THIN role hands out its roles by TICKET, not by block id.  Every workgroup draws a ticket when it starts to
  // run (one relaxed agent-scope atomic on th_tick[0]); tickets [0, thb) compute the thin product, the next pfb prefetch, the rest walk
  // the tiles.  A consumer tile therefore only ever waits for producers that STARTED BEFORE IT — resident and running, whatever order the
  // eight XCD dispatchers hand the grid out in.  By block id the producers merely sat at the front of the grid: XCDs advance through
  // their shares independently, and with two processes on one GPU a tile on one XCD span on a producer that another XCD — its CUs held by
  // the OTHER process's spinning tiles — had not started: a cross-process deadlock until the bounded waits ran out (found by the error
  // word's first reader, tests/test_train_entry_gpu.py::test_bench_two_ranks_share_one_gpu; every run, both ranks).  The last workgroup to
  // finish (second counter, th_tick[1]) clears both counters: the next launch of the stream, and every replay of a captured graph, starts
  // from zero without any host-side state.
  int vbid = (int)blockIdx.x;
  const bool ticketed = p.th_tick != nullptr;   // uniform
  if (ticketed) {
    if (threadIdx.x == 0) *reinterpret_cast<volatile uint32_t*>(smem) = __hip_atomic_fetch_add(p.th_tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    vbid = __builtin_amdgcn_readfirstlane((int)*reinterpret_cast<volatile uint32_t*>(smem));   // (block-uniform: a VECTOR register here turned every tile index of the kernel into vector arithmetic)
    __syncthreads();
  }
  auto wg_done = [&]() __attribute__((always_inline)) {   // every exit of a ticketed launch counts; the last one resets the counters
    if (ticketed && threadIdx.x == 0) {
      const uint32_t old = __hip_atomic_fetch_add(p.th_tick + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == gridDim.x - 1) {
        __hip_atomic_store(p.th_tick + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p.th_tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  const int pfb = p.pf_blocks;   // uniform
  const int rid = vbid - p.pf_base;   // prefetch index if in [0, pfb)
  if (rid >= 0 && rid < pfb) {   // prefetch role: stream a later launch's weights through the memory-side cache, keep nothing
    const mrb_u32x4* __restrict__ q = reinterpret_cast<const mrb_u32x4*>(p.pf_ptr);
    const long long step = (long long)pfb * (NW * 64);
    long long i = (long long)rid * (NW * 64) + threadIdx.x;
    uint32_t keep = 0;
    // (plain loads: with the non-temporal hint the lines are not kept by the memory-side cache — measured, the consumer ran at its cold speed)
    for (; i + 3 * step < p.pf_n16; i += 4 * step) {
      const mrb_u32x4 a = q[i], b = q[i + step], c = q[i + 2 * step], d = q[i + 3 * step];
      keep |= a[0] ^ b[1] ^ c[2] ^ d[3];
    }
    for (; i < p.pf_n16; i += step) keep |= q[i][0];
    const mrb_u32x4* __restrict__ q2 = reinterpret_cast<const mrb_u32x4*>(p.pf_ptr2);
    for (i = (long long)rid * (NW * 64) + threadIdx.x; i < p.pf_n16_2; i += step) keep |= q2[i][1];
    asm volatile("" ::"v"(keep));   // the loads stay, no store
    wg_done();
    return;
  }

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int thb = p.th_blocks;   // uniform
  const int rb = vbid - p.th_base;
  if (rb >= 0 && rb < thb) {   // thin role: 16 rows of the K extension's A operand (see GemmArgs)
    if (rb * 16 < p.M && !p.th_stall) {
      ThinArgs t;
      t.X = p.A; t.ldx = p.lda; t.A = p.th_A; t.lda = p.th_lda; t.U = const_cast<bf16_t*>(p.Aext); t.ldu = p.ldaext;
      t.M = p.M; t.K = p.th_K; t.R = p.th_R;
      t.seed_ptr = p.th_drop.seed_ptr; t.site = p.th_drop.site; t.thresh16 = p.th_drop.thresh24; t.inv_keep = p.th_drop.inv_keep;
      // (operand batches sized to the kernel's register budget — 128 VGPRs with 16 waves: the batch depth does not change the summation order)
      constexpr int UB1 = NW >= 16 ? 4 : 8, UB2 = NW >= 16 ? 2 : 4;
      if (p.th_R <= 16) lora_thin_body<1, UB1, 16, NW, true>(t, rb * 16, reinterpret_cast<f32x4(*)[1][64]>(smem), w, lane, []() {});
      else lora_thin_body<2, UB2, 16, NW, true>(t, rb * 16, reinterpret_cast<f32x4(*)[2][64]>(smem), w, lane, []() {});
      __syncthreads();   // the storing waves have drained their write-through stores
      if (threadIdx.x == 0) __hip_atomic_store(p.th_flags + rb, p.th_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wg_done();
    return;
  }
  const int wm = w / WGN, wn = w % WGN;
  const int hi = lane >> 5, l31 = lane & 31;

  int bm = 0, bn = 0;  // current tile (persistent loop below)

  const int Nh = p.N >> 1;                    // GATED only
  const uint32_t drop_seed = p.drop.seed_ptr ? mrb_seed_load(p.drop.seed_ptr) : 0u;  // one scalar load per block
  constexpr int BNO = GATED ? BN / 2 : BN;  // output columns per block

  // ---- operands are read through bounds-checked buffer resources (built in stage()): rows >= M / >= N read as zero, never fault

  // per-lane source offset inside one LDS-DMA piece (RPI rows): row lane / CPR, swizzled 16-B chunk
  //   BK = 64: chunk ^= (tile_row >> 1) & 7  == (w*4 + (lane>>4)) & 7        BK = 32: chunk ^= (tile_row >> 2) & 3 == (lane >> 4) & 3
  const int prow = lane / CPR;
  const int chunk = BK == 64 ? ((lane & 7) ^ ((w * 4 + (lane >> 4)) & 7)) : ((lane & 3) ^ ((lane >> 4) & 3));
  const uint32_t vA = (uint32_t)((long long)prow * p.lda * 2) + chunk * 16;
  const uint32_t vW = (uint32_t)((long long)prow * p.ldw * 2) + chunk * 16;
  const uint32_t vAe = (uint32_t)((long long)prow * p.ldaext * 2) + chunk * 16;
  const uint32_t vWe = (uint32_t)((long long)prow * p.ldwext * 2) + chunk * 16;

  const int nk_main = p.K / BK;
  const int nk = nk_main + (p.Aext ? NEXT : 0);

  auto w_row_base = [&](int j) __attribute__((always_inline)) -> int {  // global W row of tile row (j*NW + w)*RPI
    const int tr = (j * NW + w) * RPI;
    if (GATED) return (tr < BN / 2) ? bn * BNO + tr : Nh + bn * BNO + (tr - BN / 2);
    return bn * BN + tr;
  };

  // Per-piece source offsets live in VGPRs (computed once per tile: row base of the piece + the lane's row / swizzled chunk) and the only
  // scalar that changes per K-tile is the byte offset along k (the buffer instruction's soffset).  The kernel is far beyond the 106
  // SGPRs (the compiler parks ~100 scalars in VGPR lanes): recomputing row_base * ld per piece and per K-tile from spilled scalars put
  // a dozen v_readlane / s_mul / s_nop in front of every LDS-DMA instruction of the main loop.  Having the row offset in voffset also
  // puts it under the resource's range check (rows >= M / N read as zero whatever the hardware does with soffset).
  uint32_t vpa[JA], vpw[JW];      // main segment (the K extension computes its own on the fly: one or two tiles of the loop)
  auto tile_offsets = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int tr = (j * NW + w) * RPI;
      vpa[j] = vA + (uint32_t)((long long)(bm * BM + tr) * p.lda * 2);
    }
#pragma unroll
    for (int j = 0; j < JW; ++j) vpw[j] = vW + (uint32_t)((long long)w_row_base(j) * p.ldw * 2);
  };
  // parts / part (compile-time, handed over as integral constants): see gemm_stage_dma
  auto stage_part = [&](int kt, int buf, auto parts_c, auto part_c) __attribute__((always_inline)) {
    constexpr int PARTS = decltype(parts_c)::value, PART = decltype(part_c)::value;
    char* base = smem + buf * STAGE;
    const bool ext = kt >= nk_main;  // uniform; the K extension is one or two tiles of the whole loop
    if (!ext) {
#ifdef EXP_SAMEK
      const uint32_t koff = 0;  // EXPERIMENT (wrong results): every K-tile re-loads the first one (cache-resident fill)
#else
      const uint32_t koff = (uint32_t)kt * (uint32_t)RB;
#endif
      gemm_stage_dma<JA, JW, NW, RPI * RB, PARTS, PART>(base, base + A_BYTES, p.A, (uint32_t)((long long)p.M * p.lda * 2), p.W,
                                                        (uint32_t)((long long)p.N * p.ldw * 2), vpa, vpw, w, koff);
    } else {
      if (p.th_flags && kt == nk_main) {   // uniform: the tile's rows of Aext come from this launch's thin role — wait for their flags
        const int rbk = ((bm * BM) >> 4) + lane;
        if (lane < BM / 16 && rbk * 16 < p.M) {
          uint32_t tries = 0;   // (bounded: a protocol error must show as a wrong result in the tests, not as a hung GPU)
          while (__hip_atomic_load(p.th_flags + rbk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.th_epoch && ++tries < (1u << 19)) __builtin_amdgcn_s_sleep(4);
          // loud: the host checks this word (non-zero = failure; it carries the first-come diagnostics 0x80000000 | row block << 16 | block id)
          if (tries >= (1u << 19)) __hip_atomic_store(p.th_err, 0x80000000u | ((uint32_t)(rbk & 0x7fff) << 16) | ((uint32_t)blockIdx.x & 0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("" ::: "memory");
      }
      uint32_t ea[JA], ew[JW];
      const uint32_t egrp = p.ext_group_n ? (uint32_t)((bn * BNO) / p.ext_group_n) * 128u : 0u;   // this tile's 64-column slot of Aext
#pragma unroll
      for (int j = 0; j < JA; ++j) ea[j] = vAe + egrp + (uint32_t)((long long)(bm * BM + (j * NW + w) * RPI) * p.ldaext * 2);
#pragma unroll
      for (int j = 0; j < JW; ++j) ew[j] = vWe + (uint32_t)((long long)w_row_base(j) * p.ldwext * 2);
      if (p.th_flags)
        gemm_stage_dma_ext<JA, JW, NW, RPI * RB, PARTS, PART, 16>(base, base + A_BYTES, p.Aext, (uint32_t)((long long)p.M * p.ldaext * 2), p.Wext,
                                                                  (uint32_t)((long long)p.N * p.ldwext * 2), ea, ew, w, (uint32_t)(kt - nk_main) * (uint32_t)RB);
      else
        gemm_stage_dma_ext<JA, JW, NW, RPI * RB, PARTS, PART>(base, base + A_BYTES, p.Aext, (uint32_t)((long long)p.M * p.ldaext * 2), p.Wext,
                                                              (uint32_t)((long long)p.N * p.ldwext * 2), ea, ew, w, (uint32_t)(kt - nk_main) * (uint32_t)RB);
    }
  };
  auto stage = [&](int kt, int buf) __attribute__((always_inline)) { stage_part(kt, buf, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); };

  constexpr bool RSTG = TILE_REGSTAGE != 0 && NS == 2;
  mrb_u32x4 rs_a[JA], rs_w[JW];
  auto rs_issue = [&](int kt) __attribute__((always_inline)) {
    const bool ext = kt >= nk_main;  // uniform
    if (!ext) {
      const uint32_t koff = (uint32_t)kt * (uint32_t)RB;
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)(uint32_t)((long long)p.M * p.lda * 2), 0x00020000);
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W), 0, (int)(uint32_t)((long long)p.N * p.ldw * 2), 0x00020000);
#pragma unroll
      for (int j = 0; j < JA; ++j) rs_a[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, vpa[j], koff, 0);
#pragma unroll
      for (int j = 0; j < JW; ++j) rs_w[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, vpw[j], koff, 0);
    } else {
      const uint32_t koff = (uint32_t)(kt - nk_main) * (uint32_t)RB;
      const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Aext), 0, (int)(uint32_t)((long long)p.M * p.ldaext * 2), 0x00020000);
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Wext), 0, (int)(uint32_t)((long long)p.N * p.ldwext * 2), 0x00020000);
#pragma unroll
      for (int j = 0; j < JA; ++j) rs_a[j] = __builtin_amdgcn_raw_buffer_load_b128(ra, vAe + (p.ext_group_n ? (uint32_t)((bn * BNO) / p.ext_group_n) * 128u : 0u) + (uint32_t)((long long)(bm * BM + (j * NW + w) * RPI) * p.ldaext * 2), koff, 0);
#pragma unroll
      for (int j = 0; j < JW; ++j) rs_w[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, vWe + (uint32_t)((long long)w_row_base(j) * p.ldwext * 2), koff, 0);
    }
  };
  auto rs_commit = [&](int buf) __attribute__((always_inline)) {   // the LDS image of the LDS-DMA pieces: piece (j * NW + w), lane-linear
    char* base = smem + buf * STAGE + lane * 16;
#pragma unroll
    for (int j = 0; j < JA; ++j) *reinterpret_cast<mrb_u32x4*>(base + (j * NW + w) * (RPI * RB)) = rs_a[j];
#pragma unroll
    for (int j = 0; j < JW; ++j) *reinterpret_cast<mrb_u32x4*>(base + A_BYTES + (j * NW + w) * (RPI * RB)) = rs_w[j];
  };
  f32x16 acc[TM][TN];

  // fragment read offsets (bytes) inside a stage
  const int swz = BK == 64 ? ((lane >> 1) & 7) : ((lane >> 2) & 3);
  const int a_row_off = (wm * (BM / WGM) + l31) * RB;
  int w_row_off[TN];
#pragma unroll
  for (int nt = 0; nt < TN; ++nt)
    w_row_off[nt] = A_BYTES + (GATED ? (nt * (BN / 2) + wn * 32 + l31) : (wn * (BN / WGN) + nt * 32 + l31)) * RB;

#ifdef EXP_STAGGER
  if (blockIdx.x >= 256 && blockIdx.x < 512) {  // EXPERIMENT: the second resident block of every CU starts EXP_STAGGER x 10 ns late
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)(EXP_STAGGER)) __builtin_amdgcn_s_sleep(32);
  }
#endif
  // ---- persistent tile loop: grid = resident blocks; a block's epilogue stores drain while it already stages the next tile
  const int ntiles = p.tiles_m * p.tiles_n;
  for (int tile = vbid - p.tile_base; tile < ntiles; tile += (int)gridDim.x - pfb - thb) {
  {  // tile id -> (bm, bn): XCD-contiguous remap (bijective), then grouped ordering for L2 reuse of the W panel
    int bid = tile;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int GROUP_M = p.group_m;
    const int per_group = GROUP_M * p.tiles_n;
    const int gid = bid / per_group;
    const int first_m = gid * GROUP_M;
    const int gsize = min(p.tiles_m - first_m, GROUP_M);
    bm = first_m + (bid % per_group) % gsize;
    bn = (bid % per_group) / gsize;
  }
  tile_offsets();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool ext_first = p.ext_first != 0;  // uniform
  auto kmap = [&](int i) __attribute__((always_inline)) { return ext_first ? (i < NEXT ? nk_main + i : i - NEXT) : i; };
  if constexpr (RSTG) {
    rs_issue(kmap(0));
    rs_commit(0);
    if (nk > 1) rs_issue(kmap(1));
  } else {
    stage(kmap(0), 0);
#pragma unroll
    for (int i = 1; i < NS - 1; ++i)
      if (i < nk) stage(kmap(i), i);
  }
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (RSTG) {
      // this wave's ds_writes of stage kt are done (NOT its global loads of K-tile kt + 1: no vmcnt wait here); behind the barrier
      // every wave has written stage kt and finished reading stage kt - 1
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      // stage kt must have landed; with a deeper ring the NS - 2 newest stages may still be in flight (LDS-DMA loads retire in order)
#ifndef EXP_NOSYNC
      if (NS >= 4 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (JA + JW)) : "memory");
      else if (NS >= 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(JA + JW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#endif
#if !defined(EXP_NODMA) && TILE_PIPE != 1
      if (kt + NS - 1 < nk) stage(kmap(kt + NS - 1), (kt + NS - 1) % NS);
#endif
    }
    const char* base = smem + (kt % NS) * STAGE;
#if TILE_PIPE
    // Round 3: software-pipelined K-tile.  The compiler's own schedule of the plain loop below issues the whole next stage (6-8 LDS-DMA
    // pieces, each an issue stall) right behind the barrier and then reads every fragment just in time (ds_read -> s_waitcnt lgkmcnt(0)
    // -> MFMA): with one or two waves per SIMD nothing covers either.  Here the fragments of k-step kk+1 are requested before the MFMAs
    // of k-step kk, and the stage's pieces are spread over the k-steps; sched_barrier pins that order.
    {
      const bool do_stage = kt + NS - 1 < nk;   // uniform
      const int skt = kmap(kt + NS - 1), sbuf = (kt + NS - 1) % NS;
      bf16x8 xf[2][TM], wf[2][TN];
      auto loadf = [&](int b, int kk) __attribute__((always_inline)) {
        const int coff = (((kk * 2 + hi) ^ swz) << 4);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) xf[b][mt] = *reinterpret_cast<const bf16x8*>(base + a_row_off + mt * 32 * RB + coff);
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) wf[b][nt] = *reinterpret_cast<const bf16x8*>(base + w_row_off[nt] + coff);
      };
      loadf(0, 0);
#define TILE_PIPE_STEP(KKI)                                                                                                \
      if (KKI < KK) {                                                                                                      \
        if (KKI + 1 < KK) loadf((KKI + 1) & 1, KKI + 1);                                                                    \
        if (TILE_PIPE == 1 && do_stage) stage_part(skt, sbuf, std::integral_constant<int, KK>{}, std::integral_constant<int, (KKI < KK ? KKI : 0)>{});  \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
        _Pragma("unroll") for (int mt = 0; mt < TM; ++mt)                                                                  \
          _Pragma("unroll") for (int nt = 0; nt < TN; ++nt)                                                                \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[KKI & 1][nt], xf[KKI & 1][mt], acc[mt][nt], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
      }
      TILE_PIPE_STEP(0) TILE_PIPE_STEP(1) TILE_PIPE_STEP(2) TILE_PIPE_STEP(3)
#undef TILE_PIPE_STEP
    }
#else
#ifdef EXP_NOLDS
    {  // EXPERIMENT (wrong results): fragments read once per K-tile
      const int coff = (((0 * 2 + hi) ^ swz) << 4);
      bf16x8 xf[TM], wf[TN];
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) xf[mt] = *reinterpret_cast<const bf16x8*>(base + a_row_off + mt * 32 * RB + coff);
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) wf[nt] = *reinterpret_cast<const bf16x8*>(base + w_row_off[nt] + coff);
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
    }
#else
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int coff = (((kk * 2 + hi) ^ swz) << 4);
      bf16x8 xf[TM], wf[TN];
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) xf[mt] = *reinterpret_cast<const bf16x8*>(base + a_row_off + mt * 32 * RB + coff);
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) wf[nt] = *reinterpret_cast<const bf16x8*>(base + w_row_off[nt] + coff);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
    }
#endif
#endif  // TILE_PIPE
    if constexpr (RSTG) {
      if (kt + 1 < nk) {             // K-tile kt + 1 (requested one K-tile ago) goes into the stage that K-tile kt - 1 used
        rs_commit((kt + 1) & 1);
        if (kt + 2 < nk) rs_issue(kmap(kt + 2));
      }
    }
    if (!GATED && ext_first && kt == NEXT - 1 && p.ext_drop.seed_ptr) {  // acc == Aext Wext^T: apply the LoRA input-dropout mask to it
      const uint32_t seed = mrb_seed_load(p.ext_drop.seed_ptr);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt) {
        const uint32_t m = (uint32_t)(bm * BM + wm * (BM / WGM) + mt * 32 + l31);
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const uint32_t n0 = (uint32_t)(bn * BN + wn * (BN / WGN) + nt * 32 + 8 * g + 4 * hi);
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
              bool k0, k1;
              mrb_keep2(m * (uint32_t)p.N + n0 + i, seed, p.ext_drop.site, p.ext_drop.thresh24, k0, k1);
              acc[mt][nt][4 * g + i] = k0 ? acc[mt][nt][4 * g + i] * p.ext_drop.inv_keep : 0.f;
              acc[mt][nt][4 * g + i + 1] = k1 ? acc[mt][nt][4 * g + i + 1] * p.ext_drop.inv_keep : 0.f;
            }
          }
      }
    }
  }

  // ---- epilogue, staged through LDS so that global stores are full contiguous row segments (a lane holds 4 consecutive columns
  // of ONE row per accumulator group: stored directly, every store instruction would touch 32 different 128-B lines).
  // Each wave transposes 32 x (TN*32) fp32 slabs through its private LDS region (row stride padded by 16 B: conflict-free
  // ds_write_b128 / ds_read_b128), then 16 lanes cover one row: 256-B (fp32) / 128-B (bf16) contiguous stores.
  constexpr int SLAB_COLS = TN * 32;
  constexpr int RS = SLAB_COLS * 4 + 16;          // bytes per slab row
  constexpr int CPR = SLAB_COLS / 4;               // 16-B chunks per row
  // (the launcher allocates max(NS * STAGE, 32 * RS * NW) bytes of dynamic LDS)
  __syncthreads();                                 // every wave is done reading the staging buffers
  char* slab = smem + w * (32 * RS);
  const int ncols = GATED ? Nh : p.N;
  // bias [N] and residual [M, ldr] as buffer resources: an absent operand gets zero records (every load returns 0), rows >= M lie
  // beyond the residual's last byte
  const __amdgpu_buffer_rsrc_t bias_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t res_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual), 0, p.residual ? (int)((((long long)p.M - 1) * p.ldr + p.N) * 4) : 0, 0x00020000);
#ifdef EXP_NOEPI
  if (p.M == -12345)  // EXPERIMENT (wrong results): no epilogue at all
#endif
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(slab + l31 * RS + (nt * 32 + 8 * g + 4 * hi) * 4) =
            make_float4(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
    const int m_base = bm * BM + wm * (BM / WGM) + mt * 32;
    if (GATED) {
      constexpr int ITEMS = 32 * 4;                // (row, 8-column group) items of the 32 x 32 gated output slab
#pragma unroll
      for (int i = 0; i < ITEMS / 64; ++i) {
        const int idx = i * 64 + lane, r = idx >> 2, c = (idx & 3) * 2;
        const float4 a0 = *reinterpret_cast<const float4*>(slab + r * RS + c * 16), a1 = *reinterpret_cast<const float4*>(slab + r * RS + (c + 1) * 16);
        const float4 b0 = *reinterpret_cast<const float4*>(slab + r * RS + (c + 8) * 16), b1 = *reinterpret_cast<const float4*>(slab + r * RS + (c + 9) * 16);
        const int m = m_base + r, n0 = bn * BNO + wn * 32 + c * 4;
        if (m < p.M && n0 < Nh) {
          const float h0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, h1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          epilogue_gated8(p, drop_seed, m, n0, h0, h1, Nh);
        }
      }
    } else {
      constexpr int GPR = CPR / 2;                 // 8-column groups per slab row
      constexpr int NIT = 32 * GPR / 64;
      // bias / residual of every item of this slab first (bounds-checked: rows >= M and absent operands read as zero), all in flight
      mrb_f32x4 bq[NIT][2], rq[NIT][2];
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int idx = i * 64 + lane, r = idx / GPR, c = (idx % GPR) * 2;
        const int m = m_base + r, n0 = bn * BN + wn * (BN / WGN) + c * 4;
        const uint32_t ob = (uint32_t)n0 * 4u, orr = (uint32_t)(((long long)m * p.ldr + n0) * 4);
        bq[i][0] = __builtin_bit_cast(mrb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(bias_rsrc, ob, 0, 0));
        bq[i][1] = __builtin_bit_cast(mrb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(bias_rsrc, ob + 16u, 0, 0));
        rq[i][0] = __builtin_bit_cast(mrb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, orr, 0, 0));
        rq[i][1] = __builtin_bit_cast(mrb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, orr + 16u, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int idx = i * 64 + lane, r = idx / GPR, c = (idx % GPR) * 2;
        const float4 a0 = *reinterpret_cast<const float4*>(slab + r * RS + c * 16), a1 = *reinterpret_cast<const float4*>(slab + r * RS + (c + 1) * 16);
        const int m = m_base + r, n0 = bn * BN + wn * (BN / WGN) + c * 4;
        if (m < p.M && n0 < p.N)
          epilogue_store8v(p, drop_seed, OUT_F32, m, n0, a0, a1, ncols, make_float4(bq[i][0][0], bq[i][0][1], bq[i][0][2], bq[i][0][3]),
                           make_float4(bq[i][1][0], bq[i][1][1], bq[i][1][2], bq[i][1][3]), make_float4(rq[i][0][0], rq[i][0][1], rq[i][0][2], rq[i][0][3]),
                           make_float4(rq[i][1][0], rq[i][1][1], rq[i][1][2], rq[i][1][3]));
      }
      if constexpr (!OUT_F32 && SLAB_COLS == 64) {
        if (p.t_inner > 0) {   // block-uniform: the head-transposed copies.  Lane = one output column: it reads the slab column-wise (row stride 68
                               // words: 64 distinct banks), and writes its 32 positions as 64 contiguous bytes of the [.., d, s] row
          const int n = bn * BN + wn * (BN / WGN) + lane;
          const int bclip = m_base / p.t_rows, s0 = m_base - bclip * p.t_rows;
          const int which = n / p.t_inner;
          bf16_t* td = nullptr;
          if (n < p.N && s0 < p.t_spad && m_base < p.M) {   // (a slab wholly past the last row belongs to no clip; the slab holding a clip's last
                                                            // row reaches the end of its 32-padded tile: t_spad == roundup32(t_rows))
            if (p.t_count > 0) td = which < p.t_count ? p.tout[0] + (long long)which * p.t_stride : nullptr;
            else if (which < 3) td = p.tout[which];
          }
          if (td) {
            const int cc = n - which * p.t_inner;
            const float bv = p.bias ? p.bias[n] : 0.f;
            uint32_t pk[16];
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
              float a = *reinterpret_cast<const float*>(slab + (2 * r2) * RS + lane * 4) + bv;
              float b = *reinterpret_cast<const float*>(slab + (2 * r2 + 1) * RS + lane * 4) + bv;
              a = (m_base + 2 * r2 < p.M) ? a : 0.f;          // rows past the problem = the pad columns of the transposed tile: zeros
              b = (m_base + 2 * r2 + 1 < p.M) ? b : 0.f;
              pk[r2] = pack2bf(a, b);
            }
            uint4* q = reinterpret_cast<uint4*>(td + bclip * p.t_bs + (long long)(cc >> 6) * p.t_hs + (long long)(cc & 63) * p.t_spad + s0);
            q[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            q[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            q[2] = make_uint4(pk[8], pk[9], pk[10], pk[11]);
            q[3] = make_uint4(pk[12], pk[13], pk[14], pk[15]);
          }
        }
      }
    }
  }
  // the next tile's LDS-DMA overwrites the slabs: wait for this wave's LDS reads only (NOT for the global stores) and meet
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  }  // persistent tile loop
  wg_done();
}

// one LDS-DMA piece (1 KiB, lane-linear in LDS) of an operand tile
__device__ __forceinline__ void gemm_dma_piece(char* dst, const void* ptr, uint32_t bytes, uint32_t voff, uint32_t koff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, voff, koff, 0, 0);
}

typedef uint32_t mrb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mrb_u32x4 gemm_load_piece(const void* ptr, uint32_t bytes, uint32_t voff, uint32_t koff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, (int)bytes, 0x00020000);
  return __builtin_amdgcn_raw_buffer_load_b128(r, voff, koff, 0);
}

#ifndef W4_STORE_AUX
#define W4_STORE_AUX 0   // cache-policy bits of the 4-wave kernel's output stores (experiment: 2 = nt; measured)
#endif
#ifndef W4_GROUP_M
#define W4_GROUP_M 8
#endif
// ---- 256 x (64 TN) x 64 tile, FOUR waves of 128 x (32 TN) (one wave per SIMD, the accumulators in AGPRs), one persistent block per CU.
// TN = 4: 256x256, half the LDS fragment traffic per MFMA of the 16-wave form (a wave re-uses each 16-B fragment against four tiles
// of the other operand).  TN = 3: 256x192 for outputs whose 256-wide tiling leaves the last round mostly empty (ViT fc2 / proj,
// N = 1408: 366 tiles = 1.43 rounds of 256 CUs, against 488 = 1.9 rounds).
// With one wave per SIMD nothing hides a wave's own latencies, so the K loop is software-pipelined by hand: the fragments of k-slice
// kk+1 are read from LDS while the MFMAs of slice kk run, and the stage hand-over (wait for the LDS-DMA of K-tile kt+1, barrier,
// LDS-DMA of K-tile kt+2, first fragments of kt+1) sits in front of the LAST slice of K-tile kt, whose MFMAs cover it.
// Plain epilogues only (bias / GELU / fp32 residual): the frozen-ViT GEMMs.  ACT (0 | 1 = GELU) and RES (fp32 residual) are
// compile-time: with one wave per SIMD the run-time flag tests of the shared epilogue helpers cost more than the epilogue's real work.
#ifdef EXP_W4_STAMPS
// EXPERIMENT: wall-clock stamps (s_memrealtime, 100 MHz) of every tile a block works on: [block][tile][tile start, K loop start, K loop
// end, epilogue end]; read back with mrblip_debug_w4_stamps (tools/w4_stamps.py)
__device__ unsigned long long w4_stamps[256 * 8 * 4];
extern "C" int mrblip_debug_w4_stamps(unsigned long long* host_dst) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(w4_stamps), sizeof(w4_stamps)) == hipSuccess ? 0 : -1;
}
#define W4_STAMP(I) if (threadIdx.x == 0 && stamp_tile < 8) w4_stamps[(blockIdx.x * 8 + stamp_tile) * 4 + (I)] = __builtin_amdgcn_s_memrealtime();
#else
#define W4_STAMP(I)
#endif
// Where the 1-KB LDS-DMA pieces of a stage are issued.  A stage (K-tile s) may be written from the hand-over slice of K-tile s - 2
// (phase 3: its buffer was just released by the barrier) through slice 2 of K-tile s - 1 (phases 0, 1, 2), and must have landed at the
// barrier that ends that slice.  W4_DIST = pieces per phase in the order 3, 0, 1, 2 (the wave's JA A pieces first, then its JW W
// pieces); inside a slice they are spread evenly over the MFMA slots.  The four waves of the block pass the same slots together and the
// CU's address unit takes ~16 clocks per piece: 8 pieces per wave in one 16-MFMA slice keep it 100 % busy and every piece queues behind
// the other waves' (60-185 clocks of issue stall per piece, MI355X_MICROARCH.md), which is why the load is spread.
#ifndef W4_DIST
#define W4_DIST 8, 8, 0, 0
#endif
__device__ constexpr int w4_dist(int ph_idx) { constexpr int d[4] = {W4_DIST}; return d[ph_idx]; }              // ph_idx: 0 -> phase 3, 1..3 -> phases 0..2
__device__ constexpr int w4_piece_phidx(int i, int total) {                                                       // scaled to the kernel's piece count
  const int sum = w4_dist(0) + w4_dist(1) + w4_dist(2) + w4_dist(3);
  int acc = 0;
  for (int q = 0; q < 4; ++q) {
    acc += w4_dist(q);
    if (i * sum < acc * total) return q;
  }
  return 3;
}
__device__ constexpr int w4_piece_first(int phidx, int total) { int i = 0; while (i < total && w4_piece_phidx(i, total) != phidx) ++i; return i; }
__device__ constexpr int w4_piece_count(int phidx, int total) { int n = 0; for (int i = 0; i < total; ++i) n += (w4_piece_phidx(i, total) == phidx); return n; }
__device__ constexpr int w4_piece_slot(int i, int total, int nslot) {
  const int q = w4_piece_phidx(i, total), n = w4_piece_count(q, total), j = i - w4_piece_first(q, total);
  const int off = (q != 0 && 2 * n <= nslot) ? 1 : 0;   // phases 0..2 start one slot late (slot 0 carries the first fragment read)
  return j * nslot / n + off;
}
// F16 (round 4): the operands (and a 16-bit output) are IEEE fp16 instead of bf16 — v_mfma_f32_32x32x16_f16, same rate, same image.
// W3 (round 4, experiment -> cfg 17): a THIRD stage buffer for the W operand (A0 W0 A1 W1 W2 = 160 KB, all of the CU's LDS).  With two
// stages a stage's pieces can only be issued once the hand-over has freed its buffer — A one K-tile, W three k-slices (~1.1 us) before the
// barrier that needs them; a piece that misses the L2 lands later than that and all four waves (one per SIMD: nothing else to run) wait.
// With W2 the W pieces of K-tile kt + 2 go out in the first slice of K-tile kt (seven slices ahead); the hand-over waits with
// vmcnt(JW): only those youngest pieces may still be in flight (LDS-DMA loads retire in order).
// SPLIT (round 5): K-SPLIT units for outputs with too few tiles to fill the chip — the T5 encoder's [2012 x 2048] input gradients are 64
// tiles of 256x256 with K = 6144 / 10240.  A unit = (part, tile): part s < k_splits multiplies the K range [s, s + 1) K / k_splits and
// writes its 256 x BN block of out + s * part_stride (fp32 or bf16 PARTIAL products; the consumer adds the parts in part order — a fixed
// order, so the result does not depend on which unit finishes first); with a K extension (Aext [M, 64], Wext [N, 64]: the LoRA term,
// which the consumer MASKS before adding it) its product is the LAST part, from one-K-tile units that every XCD's queue holds behind
// its main units (they fill the tail of the last round).  No bias / residual / activation.
// GATED (round 6): the T5 gated-GELU projection [wi_0; wi_1] (modeling_t5.py:323-329) on this tile.  A tile covers 128 OUTPUT columns: each
// wave's 128 tile columns are 64 rows of wi_0 (gate, nt 0 / 1) and the SAME 64 rows of wi_1 (linear, nt 2 / 3), fetched from the two halves
// of the stacked weight by the per-piece source rows below, so gate and linear value of an output element meet in one wave's slab and the
// epilogue is y = dropout(gelu(h0) * h1) with the pre-activations [h0 | h1] as a second bf16 output — bit for bit gemm_tile_kernel's gated
// epilogue (same K order, same GELU, same dropout hash over m * Nh + n).  The LoRA term rides as 64 more K columns ([xn | u] x [W | B]^T).
template <bool OUT_F32, int ACT, bool RES, int TN, bool F16 = false, bool W3 = false, bool SPLIT = false, bool GATED = false>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const GemmArgs p) {
  static_assert(!GATED || (TN == 4 && !OUT_F32 && !RES && !SPLIT && !F16 && ACT == 0), "gated: bf16 out, 256 x 256 tile");
  constexpr int BM = 256, BN = 64 * TN, WN = BN / 2, RB = 128, NW = 4, RPI = 8;
  constexpr int A_BYTES = BM * RB, W_BYTES = BN * RB, STAGE = A_BYTES + W_BYTES;
  constexpr int JA = BM / RPI / NW, JW = BN / RPI / NW;  // LDS-DMA pieces per wave: 8 of A, 8 / 6 of W
  constexpr int NSLOT = 4 * TN, NFRAG = 4 + TN;          // MFMAs and fragment reads per k-slice
  static_assert(!W3 || (w4_piece_count(0, JA + JW) == JA && w4_piece_count(1, JA + JW) == JW), "W3: A pieces in the hand-over slice, W pieces in the slice behind it");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int chunk = (lane & 7) ^ ((w * 4 + (lane >> 4)) & 7);
  const uint32_t vA = (uint32_t)((long long)(lane >> 3) * p.lda * 2) + chunk * 16;
  const uint32_t vW = (uint32_t)((long long)(lane >> 3) * p.ldw * 2) + chunk * 16;
  const int nk = p.K / 64;
  const int swz = (lane >> 1) & 7;
  const int a_off = (wm * 128 + l31) * RB;
  const int w_off = A_BYTES + (wn * WN + l31) * RB;
  const uint32_t bytes_a = (uint32_t)((long long)p.M * p.lda * 2), bytes_w = (uint32_t)((long long)p.N * p.ldw * 2);
  int bm = 0, bn = 0;
  uint32_t vpa[JA], vpw[JW];
  f32x16 acc[4][TN];
  // the unit's operands (SPLIT: set by W4_OPEN_TILE; otherwise the launch's)
  const bf16_t* tA = p.A;
  const bf16_t* tW = p.W;
  uint32_t tba = bytes_a, tbw = bytes_w;
  int nk_t = nk, part = 0;
  (void)part;

#ifdef EXP_W4_NOSYNC
#define W4_SYNC
#else
#define W4_SYNC asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
#define W4_SYNC_KEEP(N) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); __builtin_amdgcn_s_barrier();
#endif
#ifdef EXP_W4_NODMA
#define W4_DO_DMA false
#else
#define W4_DO_DMA true
#endif
  // fragment I of a k-slice, in the order the MFMAs below first need them: A0, W0 .. W(TN-1), A1, A2, A3
#define W4_FRAG(FA, FB, BASE, BASEW, KK, I)                                                                              \
  {                                                                                                                      \
    const int coff_ = ((((KK) * 2 + hi) ^ swz) << 4);                                                                    \
    if ((I) == 0) FA[0] = *reinterpret_cast<const bf16x8*>((BASE) + a_off + coff_);                                      \
    else if ((I) <= TN) FB[(I) - 1] = *reinterpret_cast<const bf16x8*>((BASEW) + w_off + ((I) - 1) * 32 * RB + coff_);   \
    else FA[(I) - TN] = *reinterpret_cast<const bf16x8*>((BASE) + a_off + ((I) - TN) * 32 * RB + coff_);                 \
  }
  // W3: byte offset of W stage buffer i (0, 1, 2) = behind A0, behind A1, behind everything
#define W4_WOFS(I3) ((I3) == 0 ? A_BYTES : (I3) == 1 ? STAGE + A_BYTES : 2 * STAGE)
  // one k-slice: MFMA j = (mt, nt) = (j / TN, j % TN) on (FA, FB); the NFRAG fragment reads of the NEXT slice (into GA, GB) are spread
  // evenly over its slots.  The LDS-DMA of a stage is spread over two slices (all four waves pass the hand-over together: all pieces
  // at once would queue up in front of the address unit and stall the MFMA issue behind them): DMA_A = the A pieces of K-tile kt + 2
  // (hand-over slice, into the buffer the barrier just freed), DMA_W = the W pieces of K-tile kt + 1 (first slice of the following
  // K-tile).  sched_barrier pins the order.
#define W4_SLICE(FA, FB, GA, GB, NBASE, NBASEW, NKK, NEXT, PHIDX, PACTIVE)                                                \
  _Pragma("unroll") for (int j_ = 0; j_ < NSLOT; ++j_) {                                                                 \
    if ((PACTIVE) && W4_DO_DMA) {                                                                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < JA + JW; ++i_)                                                             \
        if (w4_piece_phidx(i_, JA + JW) == (PHIDX) && j_ == w4_piece_slot(i_, JA + JW, NSLOT)) {                         \
          const int st_ = ((PHIDX) == 0 || W3) ? kt + 2 : kt + 1;                                                        \
          if (i_ < JA)                                                                                                   \
            gemm_dma_piece(smem + (st_ & 1) * STAGE + (i_ * NW + w) * (RPI * RB), tA, tba, vpa[i_ < JA ? i_ : 0], (uint32_t)st_ * (uint32_t)RB); \
          else                                                                                                           \
            gemm_dma_piece(smem + (W3 ? w3ofs2 : (st_ & 1) * STAGE + A_BYTES) + ((i_ - JA) * NW + w) * (RPI * RB), tW, tbw, \
                           vpw[i_ >= JA ? i_ - JA : 0], (uint32_t)st_ * (uint32_t)RB);                                   \
        }                                                                                                                \
    }                                                                                                                    \
    if (NEXT) {                                                                                                          \
      _Pragma("unroll") for (int f_ = 0; f_ < NFRAG; ++f_)                                                               \
        if (j_ == f_ * NSLOT / NFRAG) W4_FRAG(GA, GB, NBASE, NBASEW, NKK, f_)                                            \
    }                                                                                                                    \
    acc[j_ / TN][j_ % TN] = mfma32x16<F16>(FB[j_ % TN], FA[j_ / TN], acc[j_ / TN][j_ % TN]);                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
  // (w3 = kt % 3 under W3: the W stage of this K-tile; its fragment base is taken so that + w_off lands in the buffer)
#define W4_KTILE(DMA, NEXT)                                                                                              \
  {                                                                                                                      \
    const char* base = smem + (kt & 1) * STAGE;                                                                          \
    const char* nbase = smem + ((kt + 1) & 1) * STAGE;                                                                   \
    const int w3n = w3 == 2 ? 0 : w3 + 1, w3nn = w3n == 2 ? 0 : w3n + 1;                                                 \
    const char* basew = W3 ? smem + W4_WOFS(w3) - A_BYTES : base;                                                        \
    const char* nbasew = W3 ? smem + W4_WOFS(w3n) - A_BYTES : nbase;                                                     \
    const int w3ofs2 = W4_WOFS(w3nn);                                                                                    \
    (void)w3ofs2;                                                                                                        \
    W4_SLICE(fa0, fb0, fa1, fb1, base, basew, 1, true, 1, (W3 ? (DMA) : (NEXT)))                                         \
    W4_SLICE(fa1, fb1, fa0, fb0, base, basew, 2, true, 2, NEXT)                                                          \
    W4_SLICE(fa0, fb0, fa1, fb1, base, basew, 3, true, 3, NEXT)                                                          \
    if (W3 && (DMA)) { W4_SYNC_KEEP(JW) } else { W4_SYNC }                                                               \
    W4_SLICE(fa1, fb1, fa0, fb0, nbase, nbasew, 0, NEXT, 0, DMA)                                                         \
    w3 = w3n;                                                                                                            \
  }

  // Persistent walk: block b works on the tiles b, b + grid, ... ; tile ids congruent mod 8 stay on one XCD (blocks are dealt to the
  // XCDs round-robin) and the XCD-contiguous remap below lands neighbouring tiles in one L2.  (Drawing tiles at run time from per-XCD
  // atomic counters instead measured the same, alone and in the train step, and needed device-side state: dropped.)
  const int ntiles = p.tiles_m * p.tiles_n;
  const int my_xcd = blockIdx.x & 7;
  // SPLIT: the XCD's queue = its share of the k_splits * ntiles main units, then its share of the ntiles K-extension units
  const int n_main = SPLIT ? ntiles * p.k_splits : ntiles;
  const int my_main = (n_main - my_xcd + 7) >> 3;
  const int my_count = my_main + ((SPLIT && p.Aext) ? (ntiles - my_xcd + 7) >> 3 : 0);       // units in this XCD's queue
  const int nk_main = SPLIT ? nk / p.k_splits : nk;
  int cur = blockIdx.x >> 3;                             // position in this XCD's list
  // tile id -> (bm, bn) and the per-piece source offsets, then the LDS-DMA of its first K-tile into buffer 0
#define W4_OPEN_TILE(T)                                                                                                   \
  {                                                                                                                      \
    const bool ext_ = SPLIT && (T) >= my_main;                                                                           \
    const int nq_ = ext_ ? ntiles : n_main;                                                                              \
    int bid = ((T) - (ext_ ? my_main : 0)) * 8 + my_xcd;                                                                 \
    const int q = nq_ >> 3, r = nq_ & 7, xcd = bid & 7, idx = bid >> 3;                                                  \
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;                                                 \
    long long lda_ = p.lda, ldw_ = p.ldw;                                                                                \
    uint32_t k0_ = 0;                                                                                                    \
    if (SPLIT) {                                                                                                         \
      part = ext_ ? p.k_splits : bid / ntiles;                                                                           \
      bid -= ext_ ? 0 : part * ntiles;                                                                                   \
      tA = ext_ ? p.Aext : p.A;                                                                                          \
      tW = ext_ ? p.Wext : p.W;                                                                                          \
      lda_ = ext_ ? p.ldaext : p.lda;                                                                                    \
      ldw_ = ext_ ? p.ldwext : p.ldw;                                                                                    \
      tba = (uint32_t)((long long)p.M * lda_ * 2);                                                                       \
      tbw = (uint32_t)((long long)p.N * ldw_ * 2);                                                                       \
      nk_t = ext_ ? 1 : nk_main;                                                                                         \
      k0_ = ext_ ? 0u : (uint32_t)(part * nk_main * RB);                                                                 \
    }                                                                                                                    \
    constexpr int GROUP_M = W4_GROUP_M;                                                                                  \
    const int per_group = GROUP_M * p.tiles_n;                                                                           \
    const int gid = bid / per_group;                                                                                     \
    const int first_m = gid * GROUP_M;                                                                                   \
    const int gsize = min(p.tiles_m - first_m, GROUP_M);                                                                 \
    bm = first_m + (bid % per_group) % gsize;                                                                            \
    bn = (bid % per_group) / gsize;                                                                                      \
    const uint32_t vA_ = SPLIT ? (uint32_t)((long long)(lane >> 3) * lda_ * 2) + chunk * 16 + k0_ : vA;                  \
    const uint32_t vW_ = SPLIT ? (uint32_t)((long long)(lane >> 3) * ldw_ * 2) + chunk * 16 + k0_ : vW;                  \
    _Pragma("unroll") for (int j = 0; j < JA; ++j) vpa[j] = vA_ + (uint32_t)((long long)(bm * BM + (j * NW + w) * RPI) * lda_ * 2); \
    _Pragma("unroll") for (int j = 0; j < JW; ++j) {                                                                      \
      const int tr_ = (j * NW + w) * RPI;                                     /* tile row of the piece's first W row */  \
      const long long srow_ = GATED ? (long long)((tr_ >> 6) & 1) * (p.N >> 1) + bn * 128 + (tr_ >> 7) * 64 + (tr_ & 63)  \
                                    : (long long)bn * BN + tr_;                                                         \
      vpw[j] = vW_ + (uint32_t)(srow_ * ldw_ * 2);                                                                        \
    }                                                                                                                    \
    gemm_stage_dma<JA, JW, NW, RPI * RB>(smem, smem + A_BYTES, tA, tba, tW, tbw, vpa, vpw, w, 0u);                       \
  }
#ifdef EXP_W4_STAGGER
  {  // EXPERIMENT: 8 phase groups of 32 CUs (4 per XCD) start EXP_W4_STAGGER x 10 ns apart, so the epilogue store bursts of a round interleave
    const uint64_t wait = (uint64_t)((blockIdx.x >> 3) & 7) * (uint64_t)(EXP_W4_STAGGER);
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }
#endif
  if (cur < my_count) W4_OPEN_TILE(cur)
#ifdef EXP_W4_STAMPS
  int stamp_tile = 0;
#endif
  while (cur < my_count) {
    W4_STAMP(0)
    // (bm, bn), the piece offsets and the LDS-DMA of K-tile 0 were set up by W4_OPEN_TILE: before the loop, or under the previous
    // tile's epilogue, whose slabs lie behind buffer 0
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int w3 = 0;    // (W3) kt % 3
    (void)w3;
    const int nk = nk_t;   // (SPLIT: this unit's K-tiles)
    if (nk > 1) {  // the pieces of K-tile 1 a hand-over slice would have issued (the rest go out in the slices of K-tile 0; W3: all of K-tile 1)
      constexpr int N3 = W3 ? JA + JW : w4_piece_count(0, JA + JW);
#pragma unroll
      for (int i = 0; i < JA + JW; ++i)
        if (W3 || w4_piece_phidx(i, JA + JW) == 0) {
          if (i < JA) gemm_dma_piece(smem + STAGE + (i * NW + w) * (RPI * RB), tA, tba, vpa[i < JA ? i : 0], (uint32_t)RB);
          else gemm_dma_piece(smem + STAGE + A_BYTES + ((i - JA) * NW + w) * (RPI * RB), tW, tbw, vpw[i >= JA ? i - JA : 0], (uint32_t)RB);
        }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N3) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    bf16x8 fa0[4], fb0[TN], fa1[4], fb1[TN];
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) W4_FRAG(fa0, fb0, smem, smem, 0, i)
    W4_STAMP(1)
    int kt = 0;
    for (; kt < nk - 2; ++kt) W4_KTILE(true, true)
    if (kt < nk - 1) {
      W4_KTILE(false, true)
      ++kt;
    }
    W4_KTILE(false, false)
#undef W4_KTILE
#undef W4_SLICE
#undef W4_FRAG
#undef W4_WOFS

    // ---- epilogue: 32-row x WN-column fp32 slabs per wave through LDS (the transposition of gemm_tile_kernel), then LPR lanes cover
    // one row and every lane owns the SAME 8 columns in all passes: its bias values are loaded once per tile.
    constexpr int RS = WN * 4 + 16;
    constexpr int LPR = WN / 8, ROWS = 64 / LPR, PASSES = (32 + ROWS - 1) / ROWS;  // 16 lanes x 4 rows x 8 | 12 lanes x 5 rows x 7
    W4_STAMP(2)
    __syncthreads();  // every wave is done with both stage buffers
    const int nxt = cur + (int)(gridDim.x >> 3);
    const int bm_e = bm, bn_e = bn;
    const uint32_t part_off = SPLIT ? (uint32_t)((long long)part * p.part_stride * (OUT_F32 ? 4 : 2)) : 0u;
    if (nxt < my_count) W4_OPEN_TILE(nxt)  // the next tile's first K-tile flies into buffer 0 under this epilogue
    char* slab = smem + STAGE + w * (32 * RS);
    const bool lane_on = lane < LPR * ROWS;
    const int lrow = lane / LPR;
    const int c8 = (lane % LPR) * 8;                      // the lane's 8 columns inside the wave's WN
    const int n0 = bn_e * BN + wn * WN + c8;
    const bool n_ok = lane_on && n0 < p.N;
    // Branch-free global side (round 3).  ISA reading of the first form (residual loads and stores inside `if (ok)`): the compiler put a
    // wait in front of every conditional block, so each of the 4 row groups waited for its own residual loads AND for the previous
    // group's stores to be acknowledged — the fp32-residual epilogue of fc2 / proj stamped 27 us per tile against 6 us for bias only.
    // Now bias, residual and output go through bounds-checked buffer resources (rows >= M / columns >= N get an out-of-range offset:
    // loads return 0, stores are dropped; every resource is < 2 GiB, the sentinel offset is 2^31), the residual rows of group mt + 1 are requested before group mt is finished, and nothing
    // waits for a store.
    if constexpr (GATED) {
      // 8 lanes cover the 64 output columns of a slab row (8 each), 8 rows per pass, 4 passes per 32-row slab; global side branch-free
      // through bounds-checked resources (rows >= M / columns >= Nh get the sentinel offset: the store is dropped)
      const int Nh = p.N >> 1;
      const int grow = lane >> 3, gc8 = (lane & 7) * 8;
      const int gn0 = bn_e * 128 + wn * 64 + gc8;
      const bool gn_ok = gn0 < Nh;
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((((long long)p.M - 1) * p.ldo + Nh) * 2), 0x00020000);
      const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(p.out2, 0, p.out2 ? (int)((((long long)p.M - 1) * p.ldo2 + 2 * Nh) * 2) : 0, 0x00020000);
      const bool has_drop = p.drop.seed_ptr != nullptr;
      const uint32_t seed = has_drop ? mrb_seed_load(p.drop.seed_ptr) : 0u;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(slab + l31 * RS + (nt * 32 + 8 * g + 4 * hi) * 4) =
                make_float4(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
        const int m_base = bm_e * BM + wm * 128 + mt * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = i * 8 + grow, m = m_base + r;
          const bool ok = gn_ok && m < p.M;
          const char* sp = slab + r * RS + gc8 * 4;
          const float4 a0 = *reinterpret_cast<const float4*>(sp), a1 = *reinterpret_cast<const float4*>(sp + 16);
          const float4 b0 = *reinterpret_cast<const float4*>(sp + 256), b1 = *reinterpret_cast<const float4*>(sp + 272);
          const float h0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const float h1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          if (p.out2) {
            const uint32_t oh = ok ? (uint32_t)(((long long)m * p.ldo2 + gn0) * 2) : 0x80000000u;
            __builtin_amdgcn_raw_buffer_store_b128(mrb_u32x4{pack2bf(h0[0], h0[1]), pack2bf(h0[2], h0[3]), pack2bf(h0[4], h0[5]), pack2bf(h0[6], h0[7])}, rh, oh, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(mrb_u32x4{pack2bf(h1[0], h1[1]), pack2bf(h1[2], h1[3]), pack2bf(h1[4], h1[5]), pack2bf(h1[6], h1[7])}, rh,
                                                   ok ? oh + (uint32_t)Nh * 2u : 0x80000000u, 0, 0);
          }
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            float g0 = h0[e], g1 = h0[e + 1];
            gelu_erf2(g0, g1);
            v[e] = g0 * h1[e];
            v[e + 1] = g1 * h1[e + 1];
          }
          if (has_drop) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {   // (gn0 % 8 == 0, Nh even: pairs share a hash — as epilogue_gated8)
              bool k0, k1;
              mrb_keep2((uint32_t)m * (uint32_t)Nh + (uint32_t)(gn0 + e), seed, p.drop.site, p.drop.thresh24, k0, k1);
              v[e] = k0 ? v[e] * p.drop.inv_keep : 0.f;
              v[e + 1] = k1 ? v[e + 1] * p.drop.inv_keep : 0.f;
            }
          }
          __builtin_amdgcn_raw_buffer_store_b128(mrb_u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])}, ry,
                                                 ok ? (uint32_t)(((long long)m * p.ldo + gn0) * 2) : 0x80000000u, 0, 0);
        }
      }
      __syncthreads();
      cur = nxt;
      continue;
    }
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual), 0, RES ? (int)((((long long)p.M - 1) * p.ldr + p.N) * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout =
        __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(((SPLIT ? (long long)(p.k_splits + (p.Aext ? 1 : 0) - 1) * p.part_stride : 0ll) + ((long long)p.M - 1) * p.ldo + p.N) * (OUT_F32 ? 4 : 2)), 0x00020000);
    float bias8[8];
    {
      const uint32_t ob = n_ok ? (uint32_t)n0 * 4u : 0x80000000u;
      const mrb_f32x4 b0 = __builtin_bit_cast(mrb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, ob, 0, 0));
      const mrb_f32x4 b1 = __builtin_bit_cast(mrb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, ob + 16u, 0, 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) { bias8[i] = b0[i]; bias8[4 + i] = b1[i]; }
    }
    mrb_f32x4 rq[2][PASSES][2];
    auto res_fetch = [&](int mt, int buf) __attribute__((always_inline)) {
      const int m_base = bm_e * BM + wm * 128 + mt * 32;
#pragma unroll
      for (int i = 0; i < PASSES; ++i) {
        const int r = i * ROWS + lrow;
        const bool ok = n_ok && r < 32 && m_base + r < p.M;
        const uint32_t off = ok ? (uint32_t)(((long long)(m_base + r) * p.ldr + n0) * 4) : 0x80000000u;
        rq[buf][i][0] = __builtin_bit_cast(mrb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, off, 0, 0));
        rq[buf][i][1] = __builtin_bit_cast(mrb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, off + 16u, 0, 0));
      }
    };
    if (RES) res_fetch(0, 0);
#ifdef EXP_NOEPI
    if (p.M == -12345)
#endif
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (RES && mt + 1 < 4) res_fetch(mt + 1, (mt + 1) & 1);
#pragma unroll
      for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(slab + l31 * RS + (nt * 32 + 8 * g + 4 * hi) * 4) =
              make_float4(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1], acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
      const int m_base = bm_e * BM + wm * 128 + mt * 32;
#pragma unroll
      for (int i = 0; i < PASSES; ++i) {  // row ROWS i + lane / LPR of the slab
        const int r = i * ROWS + lrow, m = m_base + r;
        const bool ok = n_ok && r < 32 && m < p.M;
        const char* sp = slab + (r < 32 ? r : 0) * RS + c8 * 4;
        const float4 x0 = *reinterpret_cast<const float4*>(sp), x1 = *reinterpret_cast<const float4*>(sp + 16);
        float v[8] = {x0.x + bias8[0], x0.y + bias8[1], x0.z + bias8[2], x0.w + bias8[3],
                      x1.x + bias8[4], x1.y + bias8[5], x1.z + bias8[6], x1.w + bias8[7]};
        if (ACT == 1) {
          gelu_erf2(v[0], v[1]); gelu_erf2(v[2], v[3]); gelu_erf2(v[4], v[5]); gelu_erf2(v[6], v[7]);
        }
        if (RES) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += rq[mt & 1][i][0][e]; v[4 + e] += rq[mt & 1][i][1][e]; }
        }
        if (OUT_F32) {
          const uint32_t off = ok ? (uint32_t)(((long long)m * p.ldo + n0) * 4) + part_off : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mrb_u32x4, mrb_f32x4{v[0], v[1], v[2], v[3]}), rout, off, 0, W4_STORE_AUX);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(mrb_u32x4, mrb_f32x4{v[4], v[5], v[6], v[7]}), rout, off + 16u, 0, W4_STORE_AUX);
        } else {
          const uint32_t off = ok ? (uint32_t)(((long long)m * p.ldo + n0) * 2) + part_off : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(mrb_u32x4{pack2x<F16>(v[0], v[1]), pack2x<F16>(v[2], v[3]), pack2x<F16>(v[4], v[5]), pack2x<F16>(v[6], v[7])}, rout, off, 0, W4_STORE_AUX);
        }
      }
    }
    __syncthreads();  // every wave is done with the slabs: the next tile's K-tile 1 goes into buffer 1, which they overlap
    W4_STAMP(3)
#ifdef EXP_W4_STAMPS
    ++stamp_tile;
#endif
    cur = nxt;
  }
#undef W4_OPEN_TILE
}

// ---- the same 256x256x64 tile on v_mfma_f32_16x16x32_bf16 (cfg 16): four waves of 128x128 = 8 x 8 accumulator tiles of 16x16, two k-slices
// of 32 per K-tile (64 MFMAs of 16 clocks each), 16 fragment reads per slice spread one per 4 MFMAs.  Same LDS image, LDS-DMA fill and
// persistent tile walk as gemm_w4_kernel; hipBLASLt's asm kernel for this tile is built on this instruction (its main loop: 128 MFMAs,
// 32 ds_read_b128, 16 buffer_load ... lds per K-tile).  Swapped operands: lane (m = l & 15, g = l >> 4) owns output row m and the four
// consecutive columns 4 g .. 4 g + 3 of every 16x16 tile.
template <bool OUT_F32, int ACT, bool RES>
__global__ __launch_bounds__(256, 1) void gemm_w16_kernel(const GemmArgs p) {
  constexpr int BM = 256, BN = 256, RB = 128, NW = 4, RPI = 8;
  constexpr int A_BYTES = BM * RB, W_BYTES = BN * RB, STAGE = A_BYTES + W_BYTES;
  constexpr int JA = BM / RPI / NW, JW = BN / RPI / NW;  // 8 + 8 LDS-DMA pieces per wave and K-tile
  constexpr int NSLOT = 64, NFRAG = 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, kg = lane >> 4;
  const int chunk = (lane & 7) ^ ((w * 4 + (lane >> 4)) & 7);
  const uint32_t vA = (uint32_t)((long long)(lane >> 3) * p.lda * 2) + chunk * 16;
  const uint32_t vW = (uint32_t)((long long)(lane >> 3) * p.ldw * 2) + chunk * 16;
  const int nk = p.K / 64;
  const int swz = l15 >> 1;                                // ((row >> 1) & 7) of row = 16 i + l15
  const int a_off = (wm * 128 + l15) * RB;
  const int w_off = A_BYTES + (wn * 128 + l15) * RB;
  const uint32_t bytes_a = (uint32_t)((long long)p.M * p.lda * 2), bytes_w = (uint32_t)((long long)p.N * p.ldw * 2);
  int bm = 0, bn = 0;
  uint32_t vpa[JA], vpw[JW];
  f32x4 acc[8][8];

  // fragment I of a k-slice in the order the MFMAs first need them: A0, W0 .. W7, A1 .. A7   (k chunk of the lane: 4 KK + kg)
#define W16_FRAG(FA, FB, BASE, KK, I)                                                                                    \
  {                                                                                                                      \
    const int coff_ = ((((KK) * 4 + kg) ^ swz) << 4);                                                                    \
    if ((I) == 0) FA[0] = *reinterpret_cast<const bf16x8*>((BASE) + a_off + coff_);                                      \
    else if ((I) <= 8) FB[(I) - 1] = *reinterpret_cast<const bf16x8*>((BASE) + w_off + ((I) - 1) * 16 * RB + coff_);     \
    else FA[(I) - 8] = *reinterpret_cast<const bf16x8*>((BASE) + a_off + ((I) - 8) * 16 * RB + coff_);                   \
  }
  // one k-slice of 64 MFMAs: j = (mt, nt) = (j / 8, j % 8).  PIECES: 0 none, 1 = all sixteen pieces of K-tile kt + 2 (into the buffer the
  // barrier in front of this slice released), one per four MFMAs
#define W16_ROW(MT, FA, FB, GA, GB, NBASE, NKK, NEXT, PIECES)                                                             \
  _Pragma("unroll") for (int nt_ = 0; nt_ < 8; ++nt_) {                                                                  \
    constexpr int mt_c = (MT);                                                                                           \
    const int j_ = mt_c * 8 + nt_;                                                                                       \
    if ((PIECES) && W4_DO_DMA) {                                                                                         \
      _Pragma("unroll") for (int i_ = 0; i_ < JA + JW; ++i_)                                                             \
        if (j_ == i_ * NSLOT / (JA + JW)) {                                                                              \
          if (i_ < JA)                                                                                                   \
            gemm_dma_piece(smem + (kt & 1) * STAGE + (i_ * NW + w) * (RPI * RB), p.A, bytes_a, vpa[i_ < JA ? i_ : 0], (uint32_t)(kt + 2) * (uint32_t)RB); \
          else                                                                                                           \
            gemm_dma_piece(smem + (kt & 1) * STAGE + A_BYTES + ((i_ - JA) * NW + w) * (RPI * RB), p.W, bytes_w,          \
                           vpw[i_ >= JA ? i_ - JA : 0], (uint32_t)(kt + 2) * (uint32_t)RB);                              \
        }                                                                                                                \
    }                                                                                                                    \
    if (NEXT) {                                                                                                          \
      _Pragma("unroll") for (int f_ = 0; f_ < NFRAG; ++f_)                                                               \
        if (j_ == f_ * NSLOT / NFRAG + 1) W16_FRAG(GA, GB, NBASE, NKK, f_)                                               \
    }                                                                                                                    \
    acc[mt_c][nt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[nt_], FA[mt_c], acc[mt_c][nt_], 0, 0, 0);                \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
  // (eight explicit rows of eight MFMAs: a single 64-trip loop exceeds the unroller's budget and comes back as a RUN-TIME loop whose
  // slot tests and fragment indices are dynamic - 7000 basic blocks and the fragments in scratch memory)
#define W16_SLICE(FA, FB, GA, GB, NBASE, NKK, NEXT, PIECES)                                                              \
  W16_ROW(0, FA, FB, GA, GB, NBASE, NKK, NEXT, PIECES) W16_ROW(1, FA, FB, GA, GB, NBASE, NKK, NEXT, PIECES)              \
  W16_ROW(2, FA, FB, GA, GB, NBASE, NKK, NEXT, PIECES) W16_ROW(3, FA, FB, GA, GB, NBASE, NKK, NEXT, PIECES)              \
  W16_ROW(4, FA, FB, GA, GB, NBASE, NKK, NEXT, PIECES) W16_ROW(5, FA, FB, GA, GB, NBASE, NKK, NEXT, PIECES)              \
  W16_ROW(6, FA, FB, GA, GB, NBASE, NKK, NEXT, PIECES) W16_ROW(7, FA, FB, GA, GB, NBASE, NKK, NEXT, PIECES)
#define W16_KTILE(DMA, NEXT)                                                                                             \
  {                                                                                                                      \
    const char* base = smem + (kt & 1) * STAGE;                                                                          \
    const char* nbase = smem + ((kt + 1) & 1) * STAGE;                                                                   \
    W16_SLICE(fa0, fb0, fa1, fb1, base, 1, true, false)                                                                  \
    W4_SYNC                                                                                                              \
    W16_SLICE(fa1, fb1, fa0, fb0, nbase, 0, NEXT, DMA)                                                                   \
  }

  const int ntiles = p.tiles_m * p.tiles_n;
  const int my_xcd = blockIdx.x & 7;
  const int my_count = (ntiles - my_xcd + 7) >> 3;
  int cur = blockIdx.x >> 3;
#define W16_OPEN_TILE(T)                                                                                                  \
  {                                                                                                                      \
    int bid = (T) * 8 + my_xcd;                                                                                          \
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;                                            \
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;                                                 \
    constexpr int GROUP_M = W4_GROUP_M;                                                                                  \
    const int per_group = GROUP_M * p.tiles_n;                                                                           \
    const int gid = bid / per_group;                                                                                     \
    const int first_m = gid * GROUP_M;                                                                                   \
    const int gsize = min(p.tiles_m - first_m, GROUP_M);                                                                 \
    bm = first_m + (bid % per_group) % gsize;                                                                            \
    bn = (bid % per_group) / gsize;                                                                                      \
    _Pragma("unroll") for (int j = 0; j < JA; ++j) vpa[j] = vA + (uint32_t)((long long)(bm * BM + (j * NW + w) * RPI) * p.lda * 2); \
    _Pragma("unroll") for (int j = 0; j < JW; ++j) vpw[j] = vW + (uint32_t)((long long)(bn * BN + (j * NW + w) * RPI) * p.ldw * 2); \
    gemm_stage_dma<JA, JW, NW, RPI * RB>(smem, smem + A_BYTES, p.A, bytes_a, p.W, bytes_w, vpa, vpw, w, 0u);             \
  }
  if (cur < my_count) W16_OPEN_TILE(cur)
  while (cur < my_count) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nk > 1) {  // K-tile 1 goes out before the loop (inside it, K-tile kt + 2 is fetched in the second slice of K-tile kt)
      gemm_stage_dma_ext<JA, JW, NW, RPI * RB>(smem + STAGE, smem + STAGE + A_BYTES, p.A, bytes_a, p.W, bytes_w, vpa, vpw, w, (uint32_t)RB);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(JA + JW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) W16_FRAG(fa0, fb0, smem, 0, i)
    int kt = 0;
    for (; kt < nk - 2; ++kt) W16_KTILE(true, true)
    if (kt < nk - 1) {
      W16_KTILE(false, true)
      ++kt;
    }
    W16_KTILE(false, false)
#undef W16_KTILE
#undef W16_SLICE
#undef W16_ROW
#undef W16_FRAG

    // ---- epilogue (first form): straight from the accumulators, 4 consecutive columns per lane and tile; branch-free: rows / columns
    // outside the matrix get an out-of-range buffer offset, which the bounds-checked loads / stores drop
    __syncthreads();
    const int nxt = cur + (int)(gridDim.x >> 3);
    const int bm_e = bm, bn_e = bn;
    if (nxt < my_count) W16_OPEN_TILE(nxt)
    {
      const uint32_t esz = OUT_F32 ? 4u : 2u;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((long long)p.M * p.ldo * esz), 0x00020000);
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.residual : p.bias), 0,
                                                                          RES ? (int)((long long)p.M * p.ldr * 4) : 16, 0x00020000);
      const int m_base = bm_e * BM + wm * 128 + l15;
      const int n_base = bn_e * BN + wn * 128 + 4 * kg;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int n0 = n_base + nt * 16;
        const bool n_ok = n0 < p.N;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + (n_ok ? n0 : 0));
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          const int m = m_base + mt * 16;
          const bool ok = n_ok && m < p.M;
          float v0 = acc[mt][nt][0] + b4.x, v1 = acc[mt][nt][1] + b4.y, v2 = acc[mt][nt][2] + b4.z, v3 = acc[mt][nt][3] + b4.w;
          if (ACT == 1) {
            gelu_erf2(v0, v1);
            gelu_erf2(v2, v3);
          }
          if (RES) {
            const uint32_t roff = ok ? (uint32_t)(((long long)m * p.ldr + n0) * 4) : 0xfffffff0u;
            const mrb_u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(rr, roff, 0, 0);
            v0 += __uint_as_float(r4[0]); v1 += __uint_as_float(r4[1]); v2 += __uint_as_float(r4[2]); v3 += __uint_as_float(r4[3]);
          }
          const uint32_t ooff = ok ? (uint32_t)(((long long)m * p.ldo + n0) * esz) : 0xfffffff0u;
          if (OUT_F32) {
            mrb_u32x4 o4 = {__float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2), __float_as_uint(v3)};
            __builtin_amdgcn_raw_buffer_store_b128(o4, ro, ooff, 0, 0);
          } else {
            typedef uint32_t mrb_u32x2 __attribute__((ext_vector_type(2)));
            mrb_u32x2 o2 = {pack2bf(v0, v1), pack2bf(v2, v3)};
            __builtin_amdgcn_raw_buffer_store_b64(o2, ro, ooff, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    cur = nxt;
  }
#undef W16_OPEN_TILE
}

// ---- the same 256 x 256 x 64 four-wave tile with a DEFERRED epilogue (bf16 output, bias, optional exact-erf GELU: ViT qkv / fc1).
// tools/w4_stamps.py showed where gemm_w4_kernel's time goes: 6 us (bias) to 15.5 us (bias + GELU) of every 40-57 us tile are an
// epilogue during which the MFMA pipe idles — with one wave per SIMD nothing else can run.  Here a finished tile leaves the
// accumulators at once: bf16(acc) (the bias was folded into the accumulator start value) is parked in 128 VGPRs, and while the NEXT
// tile's K loop runs, one 32 x 32 block of it per K-tile is activated (GELU in fp32), re-packed and stored in the shadow of that
// K-tile's 64 MFMAs.  No LDS staging: a v_permlane32_swap pair gives every lane 8 consecutive output columns (lanes l / l + 32 hold
// the two 4-column halves of one row), i.e. 16-B stores.  The parked registers form a FIFO (block 0 is always regs 0..7; the rest
// shift down by 8 per K-tile: 120 v_mov in MFMA shadow), so the loop body needs no dynamic register indexing.
// Numerics: identical to gemm_w4_kernel for ACT = 0; with GELU the pre-activation is rounded to bf16 before the activation (as the
// reference's autocast path does with its fp16 fc1 output) instead of after it.
#ifndef W4D_NPARK
#define W4D_NPARK 16
#endif
// slot (0 .. 63 = MFMA slots of one K-tile; the stage hand-over with its vmcnt(0) sits in front of slot 48) of the lane exchange, of the
// two stores, of the first FIFO-shift step, and registers moved per shift step
#ifdef EXP_W4D_NOSTORE
#define W4D_STORE_GUARD && p.K == -7
#else
#define W4D_STORE_GUARD
#endif
#ifndef W4D_S_SWAP
#define W4D_S_SWAP 24
#define W4D_S_STORE 25
#define W4D_S_SHIFT 26
#define W4D_SHIFT_PER 4
#endif
template <int ACT>
__global__ __launch_bounds__(256, 1) void gemm_w4d_kernel(const GemmArgs p) {
  constexpr int TN = 4, BM = 256, BN = 256, WN = 128, RB = 128, NW = 4, RPI = 8;
  constexpr int A_BYTES = BM * RB, W_BYTES = BN * RB, STAGE = A_BYTES + W_BYTES;
  constexpr int JA = BM / RPI / NW, JW = BN / RPI / NW;
  constexpr int NSLOT = 4 * TN, NFRAG = 4 + TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int hi = lane >> 5, l31 = lane & 31;
  const int chunk = (lane & 7) ^ ((w * 4 + (lane >> 4)) & 7);
  const uint32_t vA = (uint32_t)((long long)(lane >> 3) * p.lda * 2) + chunk * 16;
  const uint32_t vW = (uint32_t)((long long)(lane >> 3) * p.ldw * 2) + chunk * 16;
  const int nk = p.K / 64;
  const int swz = (lane >> 1) & 7;
  const int a_off = (wm * 128 + l31) * RB;
  const int w_off = A_BYTES + (wn * WN + l31) * RB;
  const uint32_t bytes_a = (uint32_t)((long long)p.M * p.lda * 2), bytes_w = (uint32_t)((long long)p.N * p.ldw * 2);
  int bm = 0, bn = 0;
  uint32_t vpa0 = 0, vpw0 = 0;   // source offset of piece 0 of the tile's A / W rows (pieces are NW * RPI rows apart: a uniform stride)
  const uint32_t strA = (uint32_t)((long long)NW * RPI * p.lda * 2), strW = (uint32_t)((long long)NW * RPI * p.ldw * 2);
  f32x16 acc[4][TN];
  constexpr int NPARK = W4D_NPARK;  // blocks parked per tile; the other 16 - NPARK are finished at once (register budget: 8 VGPRs per block)
  uint32_t stash[8 * NPARK];    // FIFO of the previous tile's bf16 results: block b = regs 8b .. 8b+7 = (g, pair) of rows l31
  uint32_t imm[8];
  int blk_cur = 0;
  int st_left = 0;              // blocks of the parked tile not yet stored
  int st_row = 0, st_col = 0;   // parked tile: first row / column of this WAVE's 128 x 128 part
  // scratch of the block in flight (kept across the MFMA slots of one K-tile)
  mrb_f2 dx, dt, dt2, dp;

#define W4D_SYNC asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier();
#define W4D_FRAG(FA, FB, BASE, KK, I)                                                                                    \
  {                                                                                                                      \
    const int coff_ = ((((KK) * 2 + hi) ^ swz) << 4);                                                                    \
    if ((I) == 0) FA[0] = *reinterpret_cast<const bf16x8*>((BASE) + a_off + coff_);                                      \
    else if ((I) <= TN) FB[(I) - 1] = *reinterpret_cast<const bf16x8*>((BASE) + w_off + ((I) - 1) * 32 * RB + coff_);    \
    else FA[(I) - TN] = *reinterpret_cast<const bf16x8*>((BASE) + a_off + ((I) - TN) * 32 * RB + coff_);                 \
  }
  // deferred-epilogue step S (0 .. 63) of the block at the head of the FIFO.  Steps 3i / 3i+1 / 3i+2 (i < 8): value pair i
  // (unpack + clamp | Horner | finish + pack); 24: lane exchange; 25: the two 16-B stores; 26 .. 55: FIFO shift, 4 registers per step.
#define W4D_STEP_ON(ARR, S, NSH)                                                                                                    \
  {                                                                                                                      \
    if ((S) < 24) {                                                                                                      \
      const int i_ = (S) / 3, ph_ = (S) % 3;                                                                             \
      if (ACT == 1) {                                                                                                    \
        if (ph_ == 0) {                                                                                                  \
          dx[0] = __uint_as_float(ARR[i_] << 16); dx[1] = __uint_as_float(ARR[i_] & 0xffff0000u);                    \
          dt = dx * 0.70710678118654752440f;                                                                             \
          dt[0] = __builtin_amdgcn_fmed3f(dt[0], -MRB_ERF_L, MRB_ERF_L); dt[1] = __builtin_amdgcn_fmed3f(dt[1], -MRB_ERF_L, MRB_ERF_L); \
          dt2 = dt * dt;                                                                                                 \
        } else if (ph_ == 1) {                                                                                           \
          MRB_ERF_HORNER(dp, dt2)                                                                                        \
        } else {                                                                                                         \
          mrb_f2 r_ = dt * dp;                                                                                           \
          r_[0] = __builtin_amdgcn_fmed3f(r_[0], -1.0f, 1.0f); r_[1] = __builtin_amdgcn_fmed3f(r_[1], -1.0f, 1.0f);      \
          const mrb_f2 h_ = dx * 0.5f;                                                                                   \
          const mrb_f2 y_ = h_ * r_ + h_;                                                                                \
          ARR[i_] = pack2bf(y_[0], y_[1]);                                                                             \
        }                                                                                                                \
      }                                                                                                                  \
    } else if ((S) == W4D_S_SWAP) {                                                                                        \
      _Pragma("unroll") for (int gp_ = 0; gp_ < 2; ++gp_)                                                                \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                               \
          auto r_ = __builtin_amdgcn_permlane32_swap(ARR[(2 * gp_) * 2 + h_], ARR[(2 * gp_ + 1) * 2 + h_], false, false); \
          ARR[(2 * gp_) * 2 + h_] = r_[0];                                                                                \
          ARR[(2 * gp_ + 1) * 2 + h_] = r_[1];                                                                            \
        }                                                                                                                \
    } else if ((S) == W4D_S_STORE) {                                                                                        \
      const int blk_ = blk_cur;                           /* (mt, nt) = (blk / 4, blk % 4) */                            \
      const int row_ = st_row + (blk_ >> 2) * 32 + l31;                                                                  \
      const int col_ = st_col + (blk_ & 3) * 32 + 8 * hi;                                                                \
      if (row_ < p.M W4D_STORE_GUARD) {                                                                                  \
        bf16_t* o_ = reinterpret_cast<bf16_t*>(p.out) + (long long)row_ * p.ldo + col_;                                  \
        if (col_ < p.N) *reinterpret_cast<uint4*>(o_) = make_uint4(ARR[0], ARR[1], ARR[2], ARR[3]);                          \
        if (col_ + 16 < p.N) *reinterpret_cast<uint4*>(o_ + 16) = make_uint4(ARR[4], ARR[5], ARR[6], ARR[7]);                \
      }                                                                                                                  \
    } else if ((S) >= W4D_S_SHIFT && (NSH) > 0) {                                                                        \
      _Pragma("unroll") for (int m_ = 0; m_ < W4D_SHIFT_PER; ++m_) {                                                     \
        const int b_ = ((S) - W4D_S_SHIFT) * W4D_SHIFT_PER + m_;                                                         \
        if (b_ < 8 * (NPARK - 1)) stash[b_] = stash[b_ + 8];                                                             \
      }                                                                                                                  \
    }                                                                                                                    \
  }
#define W4D_STEP(S) W4D_STEP_ON(stash, S, 1)
#define W4D_SLICE(FA, FB, GA, GB, NBASE, NKK, NEXT, DMA_A, DMA_W, DEFER, SL)                                             \
  _Pragma("unroll") for (int j_ = 0; j_ < NSLOT; ++j_) {                                                                 \
    if ((DMA_A)) {                                                                                                       \
      _Pragma("unroll") for (int q_ = 0; q_ < JA; ++q_)                                                                  \
        if (j_ == q_ * NSLOT / JA)                                                                                       \
          gemm_dma_piece(smem + (kt & 1) * STAGE + (q_ * NW + w) * (RPI * RB), p.A, bytes_a, vpa0 + (uint32_t)q_ * strA, (uint32_t)(kt + 2) * (uint32_t)RB); \
    }                                                                                                                    \
    if ((DMA_W)) {                                                                                                       \
      _Pragma("unroll") for (int q_ = 0; q_ < JW; ++q_)                                                                  \
        if (j_ == q_ * NSLOT / JW + 1)                                                                                   \
          gemm_dma_piece(smem + ((kt + 1) & 1) * STAGE + A_BYTES + (q_ * NW + w) * (RPI * RB), p.W, bytes_w, vpw0 + (uint32_t)q_ * strW, \
                         (uint32_t)(kt + 1) * (uint32_t)RB);                                                             \
    }                                                                                                                    \
    if (NEXT) {                                                                                                          \
      _Pragma("unroll") for (int f_ = 0; f_ < NFRAG; ++f_)                                                               \
        if (j_ == f_ * NSLOT / NFRAG) W4D_FRAG(GA, GB, NBASE, NKK, f_)                                                   \
    }                                                                                                                    \
    acc[j_ / TN][j_ % TN] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[j_ % TN], FA[j_ / TN], acc[j_ / TN][j_ % TN], 0, 0, 0); \
    if (DEFER) {                                                                                                         \
      _Pragma("unroll") for (int s_ = 0; s_ < 64; ++s_)                                                                  \
        if (s_ == (SL) * NSLOT + j_) W4D_STEP(s_)                                                                        \
    }                                                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  }
#define W4D_KTILE(DMA, NEXT, DEFER)                                                                                      \
  {                                                                                                                      \
    const char* base = smem + (kt & 1) * STAGE;                                                                          \
    const char* nbase = smem + ((kt + 1) & 1) * STAGE;                                                                   \
    W4D_SLICE(fa0, fb0, fa1, fb1, base, 1, true, false, NEXT, DEFER, 0)                                                  \
    W4D_SLICE(fa1, fb1, fa0, fb0, base, 2, true, false, false, DEFER, 1)                                                 \
    W4D_SLICE(fa0, fb0, fa1, fb1, base, 3, true, false, false, DEFER, 2)                                                 \
    W4D_SYNC                                                                                                             \
    W4D_SLICE(fa1, fb1, fa0, fb0, nbase, 0, NEXT, DMA, false, DEFER, 3)                                                  \
  }
  // every step of the head block back to back (no MFMAs to hide behind): the tail of a launch, or a K too short for 16 K-tiles
#define W4D_FLUSH_ONE                                                                                                    \
  {                                                                                                                      \
    blk_cur = NPARK - st_left;                                                                                           \
    _Pragma("unroll") for (int s_ = 0; s_ < 64; ++s_) W4D_STEP(s_)                                                       \
    --st_left;                                                                                                           \
  }

  const int ntiles = p.tiles_m * p.tiles_n;
  const int my_xcd = blockIdx.x & 7;
  const int my_count = (ntiles - my_xcd + 7) >> 3;
  int cur = blockIdx.x >> 3;
#define W4D_OPEN_TILE(T)                                                                                                  \
  {                                                                                                                      \
    int bid = (T) * 8 + my_xcd;                                                                                          \
    const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;                                            \
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;                                                 \
    constexpr int GROUP_M = W4_GROUP_M;                                                                                  \
    const int per_group = GROUP_M * p.tiles_n;                                                                           \
    const int gid = bid / per_group;                                                                                     \
    const int first_m = gid * GROUP_M;                                                                                   \
    const int gsize = min(p.tiles_m - first_m, GROUP_M);                                                                 \
    bm = first_m + (bid % per_group) % gsize;                                                                            \
    bn = (bid % per_group) / gsize;                                                                                      \
    vpa0 = vA + (uint32_t)((long long)(bm * BM + w * RPI) * p.lda * 2);                                                  \
    vpw0 = vW + (uint32_t)((long long)(bn * BN + w * RPI) * p.ldw * 2);                                                  \
    _Pragma("unroll") for (int j = 0; j < JA; ++j) gemm_dma_piece(smem + (j * NW + w) * (RPI * RB), p.A, bytes_a, vpa0 + (uint32_t)j * strA, 0u); \
    _Pragma("unroll") for (int j = 0; j < JW; ++j) gemm_dma_piece(smem + A_BYTES + (j * NW + w) * (RPI * RB), p.W, bytes_w, vpw0 + (uint32_t)j * strW, 0u); \
  }
  if (cur < my_count) W4D_OPEN_TILE(cur)
  while (cur < my_count) {
    // accumulators start at the bias of their column (4 consecutive columns per accumulator group): the bias add costs nothing
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = bn * BN + wn * WN + nt * 32 + 8 * g + 4 * hi;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && c0 < p.N) b = *reinterpret_cast<const float4*>(p.bias + c0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          acc[mt][nt][4 * g] = b.x; acc[mt][nt][4 * g + 1] = b.y; acc[mt][nt][4 * g + 2] = b.z; acc[mt][nt][4 * g + 3] = b.w;
        }
      }
    if (nk > 1) {
#pragma unroll
      for (int j = 0; j < JA; ++j) gemm_dma_piece(smem + STAGE + (j * NW + w) * (RPI * RB), p.A, bytes_a, vpa0 + (uint32_t)j * strA, (uint32_t)RB);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(JA) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    bf16x8 fa0[4], fb0[TN], fa1[4], fb1[TN];
#pragma unroll
    for (int i = 0; i < NFRAG; ++i) W4D_FRAG(fa0, fb0, smem, 0, i)
    int kt = 0;
    for (; kt < nk - 2 && st_left > 0; ++kt) {   // K-tiles that also retire one parked block each
      blk_cur = NPARK - st_left;
      W4D_KTILE(true, true, true)
      --st_left;
    }
    for (; kt < nk - 2; ++kt) W4D_KTILE(true, true, false)
    if (kt < nk - 1) {
      W4D_KTILE(false, true, false)
      ++kt;
    }
    W4D_KTILE(false, false, false)
    while (st_left > 0) W4D_FLUSH_ONE   // (only when nk - 2 < 16)

    // ---- this tile: park bf16(acc) and move on
    __syncthreads();  // every wave is done with both stage buffers
    const int nxt = cur + (int)(gridDim.x >> 3);
    st_row = bm * BM + wm * 128;
    st_col = bn * BN + wn * WN;
    if (nxt < my_count) W4D_OPEN_TILE(nxt)  // the next tile's first K-tile flies while the accumulators are parked
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        if (mt * 4 + nt < NPARK) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            stash[((mt * 4 + nt) * 4 + g) * 2] = pack2bf(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1]);
            stash[((mt * 4 + nt) * 4 + g) * 2 + 1] = pack2bf(acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
          }
        } else {  // not parked: finished here (the part of the epilogue that stays exposed)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            imm[g * 2] = pack2bf(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1]);
            imm[g * 2 + 1] = pack2bf(acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]);
          }
          blk_cur = mt * 4 + nt;
#pragma unroll
          for (int s_ = 0; s_ < 64; ++s_) W4D_STEP_ON(imm, s_, 0)
        }
      }
    st_left = NPARK;
    cur = nxt;
  }
  while (st_left > 0) W4D_FLUSH_ONE
#undef W4D_OPEN_TILE
#undef W4D_FLUSH_ONE
#undef W4D_KTILE
#undef W4D_SLICE
#undef W4D_STEP
#undef W4D_STEP_ON
#undef W4D_FRAG
#undef W4D_SYNC
}

// ---- skinny-M kernel (decoder rows, M <= a few 32-row tiles): weight-streaming bound.  One block = 32 output
// columns x 32 rows; its 4 waves split K, each lane streams 64 contiguous bytes of one W row per 64-wide k block
// (the contraction order inside a k block is permuted identically for W and X so both load 16-B vectors from full
// 128-B lines), partial sums are reduced through LDS.
template <bool OUT_F32, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(const GemmArgs p) {
  __shared__ float red[NW - 1][16][64];
  constexpr int UNR = NW > 4 ? 2 : 4;  // k blocks in flight per wave (16 waves x 64 lanes leave 128 VGPRs per lane)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int n_row = blockIdx.x * 32 + l31;  // W row (MFMA A-operand row)
  const int m_row = blockIdx.y * p.m_rows_per_block + l31;  // X row (MFMA B-operand column)
  const bool n_ok = n_row < p.N, m_ok = m_row < p.M && l31 < p.m_rows_per_block;
  const int nkb = p.K >> 6;  // 64-wide k blocks of the main segment
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const bf16_t* wp = p.W + (long long)n_row * p.ldw + hi * 32;
  const bf16_t* xp = p.A + (long long)m_row * p.lda + hi * 32;
  // each of the NW waves owns a contiguous share of the k blocks; UNR blocks (8 independent 16-B loads per lane each) are in flight
  const int zs = p.k_splits > 1 ? (int)blockIdx.z : 0;
  const int per_z = p.k_splits > 1 ? (nkb + p.k_splits - 1) / p.k_splits : nkb;
  const int z_beg = zs * per_z, z_end = min(nkb, z_beg + per_z);
  const int per = (max(z_end - z_beg, 0) + NW - 1) / NW;
  const int kb_beg = z_beg + w * per, kb_end = min(z_end, kb_beg + per);
#pragma unroll 1
  for (int kb0 = kb_beg; kb0 < kb_end; kb0 += UNR) {
    bf16x8 wf[UNR][4], xf[UNR][4];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int kb = min(kb0 + u, kb_end - 1);
      const bool live = kb0 + u < kb_end;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        wf[u][s] = (n_ok && live) ? *reinterpret_cast<const bf16x8*>(wp + kb * 64 + s * 8) : zero;
        xf[u][s] = (m_ok && live) ? *reinterpret_cast<const bf16x8*>(xp + kb * 64 + s * 8) : zero;
      }
    }
    if (p.a_drop.seed_ptr) {  // dropout(x) for the LoRA "down" product: element (m_row, k) of the [M, K] input, pair-hashed
      const uint32_t seed = mrb_seed_load(p.a_drop.seed_ptr);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const uint32_t kbase = (uint32_t)m_row * (uint32_t)p.K + (uint32_t)(min(kb0 + u, kb_end - 1) * 64 + hi * 32);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          union { bf16x8 v; uint32_t d[4]; } f;
          f.v = xf[u][s];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            bool k0, k1;
            mrb_keep2(kbase + (uint32_t)(s * 8 + 2 * i), seed, p.a_drop.site, p.a_drop.thresh24, k0, k1);
            f.d[i] = (k0 ? f.d[i] & 0xffffu : 0u) | (k1 ? f.d[i] & 0xffff0000u : 0u);
          }
          xf[u][s] = f.v;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u][s], xf[u][s], acc, 0, 0, 0);
  }
  if (p.Aext && w == NW - 1 && zs == 0) {  // K-extension segment (one 64-wide block), taken by the last wave (of the first K split)
    const bf16_t* wpe = p.Wext + (long long)(n_ok ? n_row : 0) * p.ldwext + hi * 32;
    const bf16_t* xpe = p.Aext + (long long)(m_ok ? m_row : 0) * p.ldaext + hi * 32;
    f32x16 e;
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = 0.f;
    // (all eight loads first, from clamped rows, then the selects: a conditional load per MFMA was four round trips in series)
    bf16x8 wfe[4], xfe[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wfe[s] = *reinterpret_cast<const bf16x8*>(wpe + s * 8);
      xfe[s] = *reinterpret_cast<const bf16x8*>(xpe + s * 8);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(n_ok ? wfe[s] : zero, m_ok ? xfe[s] : zero, e, 0, 0, 0);
    if (p.ext_first && p.ext_drop.seed_ptr) {  // LoRA backward form: mask the extension product (see GemmArgs)
      const uint32_t seed = mrb_seed_load(p.ext_drop.seed_ptr);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const uint32_t n0 = (uint32_t)(blockIdx.x * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          bool k0, k1;
          mrb_keep2((uint32_t)m_row * (uint32_t)p.N + n0 + i, seed, p.ext_drop.site, p.ext_drop.thresh24, k0, k1);
          e[4 * g + i] = k0 ? e[4 * g + i] * p.ext_drop.inv_keep : 0.f;
          e[4 * g + i + 1] = k1 ? e[4 * g + i + 1] * p.ext_drop.inv_keep : 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += e[r];
  }
  if (w > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int j = 0; j < NW - 1; ++j) acc[r] += red[j][r][lane];
    }
    if (p.a_drop.seed_ptr) {  // the kept inputs' 1/(1-p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= p.a_drop.inv_keep;
    }
    if (m_ok && OUT_F32 && p.k_splits > 1) {
      // K-split form: this block's partial product (bias by the first split, the output dropout mask on every partial: it is linear)
      // is added to the pre-initialised fp32 output
      const uint32_t seed = p.drop.seed_ptr ? mrb_seed_load(p.drop.seed_ptr) : 0u;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = blockIdx.x * 32 + 8 * g + 4 * hi;
        if (n0 < p.N) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = acc[4 * g + i];
            if (p.bias && zs == 0) v += p.bias[n0 + i];
            if (p.drop.seed_ptr)
              v = mrb_keep((uint32_t)m_row * (uint32_t)p.N + (uint32_t)(n0 + i), seed, p.drop.site, p.drop.thresh24) ? v * p.drop.inv_keep : 0.f;
            atomicAdd(reinterpret_cast<float*>(p.out) + (long long)m_row * p.ldo + n0 + i, v);
          }
        }
      }
    } else if (m_ok) {
      // bias / residual of the lane's four column groups first (clamped columns), seed by scalar load: every load in flight together
      const uint32_t seed = p.drop.seed_ptr ? mrb_seed_load(p.drop.seed_ptr) : 0u;
      float4 bq[4], rq[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = blockIdx.x * 32 + 8 * g + 4 * hi, n0c = n0 < p.N ? n0 : 0;
        bq[g] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0c) : make_float4(0.f, 0.f, 0.f, 0.f);
        rq[g] = p.residual ? *reinterpret_cast<const float4*>(p.residual + (long long)m_row * p.ldr + n0c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      MRB_ALL_LOADS_DONE();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = blockIdx.x * 32 + 8 * g + 4 * hi;
        if (n0 < p.N) epilogue_store4v(p, seed, OUT_F32, m_row, n0, acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], p.N, bq[g], rq[g]);
      }
    }
  }
}

static bool pf_tail_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MRB_GEMM_PF_TAIL"); on = (e && e[0] == '1') ? 1 : 0; }
  return on == 1;
}
static bool thin_ticket_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MRB_GEMM_THIN_TICKET"); on = (e && e[0] == '1') ? 1 : 0; }
  return on == 1;
}
static bool roles_last_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MRB_GEMM_ROLES_LAST"); on = (e && e[0] == '0') ? 0 : 1; }
  return on == 1;
}

template <int BM, int BN, int WGM, int WGN, bool OUT_F32, bool GATED, int BK = 64, int NS = 2>
static int launch_tile(GemmArgs& a, hipStream_t st) {
  constexpr int BNO = GATED ? BN / 2 : BN;
  const int ncols = GATED ? a.N / 2 : a.N;
  a.tiles_m = (a.M + BM - 1) / BM;
  a.tiles_n = (ncols + BNO - 1) / BNO;
  {
    static int env_gm = -1;
    if (env_gm < 0) { const char* e = getenv("MRB_GROUP_M"); env_gm = e ? atoi(e) : 0; }
    a.group_m = env_gm > 0 ? env_gm : 8;
  }
  constexpr int TN_ = BN / WGN / 32, SLAB = 32 * (TN_ * 32 * 4 + 16) * WGM * WGN;   // epilogue staging (see the kernel)
  constexpr int LDS = NS * (BM + BN) * BK * 2 > SLAB ? NS * (BM + BN) * BK * 2 : SLAB;
  auto kern = gemm_tile_kernel<BM, BN, WGM, WGN, OUT_F32, GATED, BK, NS>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      mrblip_set_error("gemm: cannot raise dynamic LDS to %d", LDS);
      return MRBLIP_ELAUNCH;
    }
    attr_set = true;
  }
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { mrblip_set_error("gemm: cannot query device"); return MRBLIP_ELAUNCH; }
    num_cu = prop.multiProcessorCount;
  }
  // 256x256 (one block per CU): persistent, one block per CU walks the tiles.  128x128: one block per tile (the hardware's dynamic
  // dispatch of 2 blocks/CU measured slightly faster than a static persistent walk).
  const int ntiles = a.tiles_m * a.tiles_n;
  const int grid = (NS == 2 && LDS > 80 * 1024 && ntiles > num_cu) ? num_cu : ntiles;
  a.pf_blocks = (a.pf_n16 > 0 || a.pf_n16_2 > 0) ? (a.pf_blocks + 7) / 8 * 8 : 0;
  a.th_blocks = a.th_flags ? ((a.M + 15) / 16 + 7) / 8 * 8 : 0;
  // Role layout.  With a thin role the roles go by TICKET (see the kernel): thin | prefetch | tiles in ticket order, i.e. in the order
  // the workgroups start to run.  Without one (prefetch only: nothing waits for it) by block id: behind the tiles where those leave
  // workgroup slots idle, in a partly empty last round, or in front.
  constexpr int per_cu = gemm_min_blocks(LDS, WGM * WGN);
  const int slots = num_cu * per_cu;
  if (a.th_blocks > 0 && thin_ticket_enabled()) {
    a.th_base = 0; a.pf_base = a.th_blocks; a.tile_base = a.th_blocks + a.pf_blocks;
  } else if (a.th_blocks > 0) {
    // Default: roles by BLOCK ID with the thin role in FRONT of the tiles.  Every XCD's dispatcher hands out its share of the grid in
    // order and producers never wait, so on a GPU that this process has to itself a tile can only wait for producers that other,
    // FINITE kernels delay.  That is not a guarantee — see the ticket mode above, which is one — but it costs no atomics: the ticket
    // mode's 500-1400 same-address atomics per launch cost 2.3 ms per QVH step (72.0 vs 69.7 ms), more than the role saves.  A
    // starved hand-over fails loudly either way (error word -> skipped AdamW -> raise).  MRB_GEMM_THIN_TICKET=1 selects the ticket
    // mode (GPU shared between processes: bench.py's MRB_BENCH_SHARE_GPU test hook sets it).
    a.th_tick = nullptr;
    if (a.pf_blocks > 0 && grid + 32 <= slots && roles_last_enabled()) {   // thin | tiles | prefetch
      a.th_base = 0; a.tile_base = a.th_blocks; a.pf_base = a.th_blocks + grid;
    } else {                                                               // prefetch | thin | tiles
      a.pf_base = 0; a.th_base = a.pf_blocks; a.tile_base = a.pf_blocks + a.th_blocks;
    }
  } else {
    a.th_tick = nullptr;
    if (a.pf_blocks > 0 && grid + 32 <= slots && roles_last_enabled()) {   // tiles | prefetch
      a.th_base = 0; a.tile_base = 0; a.pf_base = grid;
    } else if (a.pf_blocks > 0 && grid == ntiles && grid > slots && pf_tail_enabled() && slots - grid % slots >= a.pf_blocks) {
      a.th_base = 0; a.tile_base = 0; a.pf_base = grid;
    } else {                                                               // prefetch | tiles
      a.pf_base = 0; a.th_base = a.pf_blocks; a.tile_base = a.pf_blocks;
    }
  }
  static_assert(LDS >= THIN_RED_BYTES(2), "the thin role's partial sums live in the tile's LDS");
  hipLaunchKernelGGL(kern, dim3(grid + a.pf_blocks + a.th_blocks), dim3(WGM * WGN * 64), LDS, st, a);
  return mrblip_check_launch("gemm_tile");
}

static void mk_drop_arg(DropoutArg& d, const uint32_t* seed_ptr, uint32_t site, float p) {
  d.seed_ptr = (p > 0.f) ? seed_ptr : nullptr;
  d.site = site;
  d.thresh24 = (uint32_t)(p * 65536.0f + 0.5f);
  d.inv_keep = 1.0f / (1.0f - p);
}

// CU reserve of the persistent GEMM kernels (tile configs 13 / 14): the number of CUs (a multiple of 8, one per XCD) a launch leaves to
// other streams.  It is a PER-CALL argument: bits 8..16 of tile_cfg (tile_cfg = cfg | reserve << 8).  The setter below only provides a
// per-THREAD default for callers that cannot pass it (thread_local: no process-global mutable state, so the forward thread and the
// autograd thread can enqueue GEMMs with different reserves concurrently); returns the calling thread's previous default.
static thread_local int g_cu_reserve = 0;

extern "C" int mrblip_gemm_set_cu_reserve(int n_cus) {
  const int prev = g_cu_reserve;
  g_cu_reserve = n_cus < 0 ? 0 : n_cus / 8 * 8;
  return prev;
}

// shared by the C entry points: validation, tile selection, launch.  ext_first / ext_* / a_* : see GemmArgs.
// one-shot extras of the calling thread's NEXT mrblip_gemm_bf16 / mrblip_gemm_lora_dx launch (round 4): see GemmArgs.tout / ext_group_n
struct GemmExtra { void* tout[3]; int t_inner, t_rows, t_spad; long long t_bs, t_hs, t_stride; int t_count, ext_group_n; bool set; };
static thread_local GemmExtra g_gemm_extra = {};
extern "C" int mrblip_gemm_set_extra(void* tout0, void* tout1, void* tout2, int t_inner, int t_rows, int t_spad, long long t_bs, long long t_hs,
                                     long long t_stride, int t_count, int ext_group_n) {
  const bool any_t = tout0 || tout1 || tout2;
  MRB_REQUIRE(!any_t || (t_inner > 0 && (t_inner % 64) == 0 && t_rows > 0 && t_spad == (t_rows + 31) / 32 * 32), "gemm_set_extra: head-transposed copies need a head layout (t_inner %% 64 == 0, t_spad == roundup32(t_rows))");
  MRB_REQUIRE(ext_group_n >= 0 && (ext_group_n % 256) == 0, "gemm_set_extra: ext_group_n must be a multiple of 256");
  MRB_REQUIRE(t_count >= 0 && (t_count == 0 || (tout0 && t_stride > 0 && (t_stride % 8) == 0)), "gemm_set_extra: t_count ranges need tout0 and a 16-B aligned t_stride");
  g_gemm_extra = GemmExtra{{tout0, tout1, tout2}, any_t ? t_inner : 0, t_rows, t_spad, t_bs, t_hs, t_stride, t_count, ext_group_n, true};
  return MRBLIP_OK;
}

// one-shot, like the extras: the calling thread's NEXT GEMM launch also streams [ptr, ptr + bytes) through the memory-side cache with
// n_blocks extra workgroups (generic tile kernels; the other forms ignore it — it is a hint and changes no result)
struct GemmPrefetch { const void* ptr; long long bytes; const void* ptr2; long long bytes2; int n_blocks; };
static thread_local GemmPrefetch g_gemm_prefetch = {};
extern "C" int mrblip_gemm_set_prefetch(const void* ptr, long long bytes, const void* ptr2, long long bytes2, int n_blocks) {
  MRB_REQUIRE(bytes >= 0 && bytes2 >= 0 && ((uintptr_t)ptr % 16) == 0 && ((uintptr_t)ptr2 % 16) == 0 && n_blocks >= 0 && n_blocks <= 1024,
              "gemm_set_prefetch: 16-byte aligned ranges, at most 1024 blocks");
  g_gemm_prefetch = GemmPrefetch{ptr, ptr ? bytes : 0, ptr2, ptr2 ? bytes2 : 0, n_blocks};
  return MRBLIP_OK;
}

// one-shot: the calling thread's NEXT GEMM launch computes its own K-extension operand Aext = dropout(A) acat^T (see GemmArgs.th_*).
// flags: >= ceil(M / 16) + 1 words the launches of ONE stream may share (the LAST word is an error word: a tile whose bounded wait ran out
// stores 0xffffffff there — a protocol failure is a wrong result the caller can detect, never a hung GPU); epoch: a value no earlier launch
// on that stream left in them.
// err (may be NULL = the last flag word): the word a timed-out tile sets.  Callers hand ONE word per device to every stream's launches and
// to mrblip_adamw_guarded, so that an optimizer step never applies gradients of a step in which a wait ran out.
struct GemmThin { const void* acat; long long lda; int R, K; uint32_t site; float p; uint32_t* flags; long long n_flags; uint32_t epoch; uint32_t* err; bool set; };
static thread_local GemmThin g_gemm_thin = {};
static thread_local int g_thin_stall = 0;
// TEST HOOK: while on, the thin-role workgroups of this thread's launches exit without computing or publishing anything, so every
// consumer tile runs into its bounded wait — the way tests/ prove that a protocol failure is loud (error word, skipped AdamW, raise).
extern "C" int mrblip_gemm_debug_stall_thin(int on) {
  const int prev = g_thin_stall;
  g_thin_stall = on ? 1 : 0;
  return prev;
}
extern "C" int mrblip_gemm_set_thin(const void* acat, long long lda, int R, int K, uint32_t site, float p_drop, uint32_t* flags, long long n_flags,
                                    uint32_t epoch, uint32_t* err) {
  MRB_REQUIRE(acat && flags && ((uintptr_t)acat % 16) == 0 && R > 0 && R <= 32 && (R % 8) == 0 && K > 0 && (K % 32) == 0 && (lda % 8) == 0 && lda >= K,
              "gemm_set_thin: acat [R <= 32, K %% 32 == 0] with 16-B rows and a flag buffer");
  g_gemm_thin = GemmThin{acat, lda, R, K, site, p_drop, flags, n_flags, epoch, err, true};
  return MRBLIP_OK;
}

// drop every pending one-shot (extras, prefetch range, thin role) of the calling thread: for a caller that fails between a setter and the
// launch it was meant for (mrblip/ops.py does this in its exception path), so that they cannot ride on an unrelated later GEMM
extern "C" int mrblip_gemm_clear_one_shots(void) {
  g_gemm_extra = GemmExtra{};
  g_gemm_prefetch = GemmPrefetch{};
  g_gemm_thin = GemmThin{};
  return MRBLIP_OK;
}

static int gemm_dispatch(const void* A, long long lda, const void* W, long long ldw, const void* Aext, long long ldaext,
                         const void* Wext, long long ldwext, int M, int N, int K, void* out, long long ldo, int out_f32,
                         void* out2, long long ldo2, const float* bias, const float* residual, long long ldr, int act,
                         int gated, const uint32_t* seed_ptr, uint32_t site, float p_drop, int tile_cfg, int ext_first,
                         uint32_t ext_site, float ext_p, uint32_t a_site, float a_p, hipStream_t stream, bool f16 = false) {
  // the one-shot extras belong to THIS call whatever happens to it (a call that fails below must not leave them to the next GEMM)
  const GemmExtra extra = g_gemm_extra;
  g_gemm_extra = GemmExtra{};
  const GemmPrefetch pf = g_gemm_prefetch;
  g_gemm_prefetch = GemmPrefetch{};
  const GemmThin th = g_gemm_thin;
  g_gemm_thin = GemmThin{};
  MRB_REQUIRE(M > 0 && N > 0 && K >= 0 && (K % 64) == 0, "gemm: need M,N>0 and K%%64==0 (M=%d N=%d K=%d)", M, N, K);
  MRB_REQUIRE(K > 0 || Aext, "gemm: empty contraction");
  MRB_REQUIRE((N % 8) == 0, "gemm: N %% 8 != 0 (N=%d)", N);
  MRB_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && (ldo % (out_f32 ? 4 : 8)) == 0 && (!out2 || (ldo2 % 8) == 0), "gemm: leading dims must keep 16-B alignment");
  MRB_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm: pointers must be 16-B aligned");
  MRB_REQUIRE((Aext == nullptr) == (Wext == nullptr), "gemm: Aext/Wext must come together");
  MRB_REQUIRE(((long long)(M + 256) * lda * 2 + 256) < (1ll << 32) && ((long long)(N + 256) * ldw * 2 + 256) < (1ll << 32),
              "gemm: operand exceeds the 4 GiB buffer-descriptor range");
  MRB_REQUIRE(!gated || (!out_f32 && !bias && !residual && act == 0 && (N % 16) == 0), "gemm: gated mode takes no bias/residual/act");
  MRB_REQUIRE(act != 2 || (out2 && !out_f32 && !gated && !bias && !residual && !Aext && !(p_drop > 0.f) && (N % 8) == 0 && (ldo2 % 8) == 0 && ((uintptr_t)out2 % 16) == 0),
              "gemm: act 2 (GELU backward) = bf16 out, out2 = the saved pre-activation (read), no bias / residual / dropout / K extension");
  GemmArgs a;
  a.ext_first = ext_first;
  a.m_rows_per_block = 32;
  a.k_splits = 1;
  a.part_stride = 0;
  a.tout[0] = (bf16_t*)extra.tout[0]; a.tout[1] = (bf16_t*)extra.tout[1]; a.tout[2] = (bf16_t*)extra.tout[2];
  a.t_inner = extra.set ? extra.t_inner : 0; a.t_rows = extra.t_rows; a.t_spad = extra.t_spad; a.t_bs = extra.t_bs; a.t_hs = extra.t_hs;
  a.t_stride = extra.t_stride; a.t_count = extra.set ? extra.t_count : 0;
  a.ext_group_n = extra.set ? extra.ext_group_n : 0;
  a.th_flags = nullptr; a.th_err = nullptr; a.th_tick = nullptr; a.th_blocks = 0; a.th_stall = 0; a.th_A = nullptr; a.th_lda = 0; a.th_R = a.th_K = 0; a.th_epoch = 0;
  mk_drop_arg(a.th_drop, seed_ptr, 0, 0.f);
  if (th.set) {
    MRB_REQUIRE(Aext && !ext_first && !f16 && M > 64 && th.K <= K && ldaext >= th.R && (ldaext % 4) == 0 && ((uintptr_t)Aext % 8) == 0 && th.n_flags >= (M + 15) / 16 + 2 &&
                    (long long)M * lda * 2 < (1ll << 31) && !(th.p > 0.f && !seed_ptr),
                "gemm: the thin role needs a K extension read last, more than 64 rows and one flag per 16 rows");
    a.th_A = (const bf16_t*)th.acat; a.th_lda = th.lda; a.th_R = th.R; a.th_K = th.K; a.th_flags = th.flags; a.th_err = th.err ? th.err : th.flags + (th.n_flags - 1); a.th_epoch = th.epoch; a.th_stall = g_thin_stall;
    a.th_tick = th.flags + (th.n_flags - 3);   // two words: ticket, finished-workgroup count (zero between launches: the kernel resets them)
    MRB_REQUIRE(th.n_flags >= (M + 15) / 16 + 4, "gemm: the thin role's flag buffer needs ceil(M / 16) + 4 words (flags, ticket, count, error)");
    mk_drop_arg(a.th_drop, seed_ptr, th.site, th.p);
  }
  a.pf_ptr = pf.ptr; a.pf_n16 = pf.bytes / 16; a.pf_ptr2 = pf.ptr2; a.pf_n16_2 = pf.bytes2 / 16; a.pf_blocks = pf.n_blocks;
  const bool has_extra = a.t_inner > 0 || a.ext_group_n > 0;
  MRB_REQUIRE(a.t_inner == 0 || (!out_f32 && !gated && act == 0 && !(p_drop > 0.f) && !residual && !out2 && (M == a.t_rows || (a.t_rows % 32) == 0) && (M % a.t_rows) == 0),
              "gemm: head-transposed copies need a bf16 output with a plain / bias epilogue and one clip or t_rows %% 32 == 0");
  MRB_REQUIRE(a.ext_group_n == 0 || (Aext && !gated && (N % a.ext_group_n) == 0 && ldaext >= (long long)(N / a.ext_group_n) * 64), "gemm: ext_group_n needs a K extension with one 64-column slot per group");
  mk_drop_arg(a.ext_drop, seed_ptr, ext_site, ext_p);
  mk_drop_arg(a.a_drop, seed_ptr, a_site, a_p);
  MRB_REQUIRE(!(ext_p > 0.f || a_p > 0.f) || seed_ptr, "gemm: dropout needs a device seed pointer");
  MRB_REQUIRE(!ext_first || (Aext && !gated), "gemm: ext_first needs a K-extension and no gating");
  a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.Aext = (const bf16_t*)Aext; a.Wext = (const bf16_t*)Wext;
  a.out = out; a.out2 = out2; a.bias = bias; a.residual = residual;
  a.lda = lda; a.ldw = ldw; a.ldaext = Aext ? ldaext : 0; a.ldwext = Wext ? ldwext : 0; a.ldo = ldo; a.ldo2 = ldo2; a.ldr = ldr;
  a.M = M; a.N = N; a.K = K; a.act = act; a.tiles_m = a.tiles_n = 0;
  a.drop.seed_ptr = (p_drop > 0.f) ? seed_ptr : nullptr;
  a.drop.site = site;
  a.drop.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f);
  a.drop.inv_keep = 1.0f / (1.0f - p_drop);
  MRB_REQUIRE(!(p_drop > 0.f) || seed_ptr, "gemm: dropout needs a device seed pointer");
  // tile_cfg: 0 auto, 1 = 256x256 (8 waves, persistent), 2 = 128x128, 3 = skinny, 4 = 64x128, 5 = 64x64, 8 = 256x256 with 16 waves,
  // 13 = 256x256 with 4 waves (plain epilogues); measured and not auto-selected:
  // 6 = 256x256 with 4 waves of 128x128 (1 wave/SIMD: 0.7x), 7 = 256x128 BK=32 3-stage (= 128x128), [128x128 BK=32 at 3-4 blocks/CU: 0.8x,
  // 256x128 / 128x256 BK=64 with one 4-wave block per CU: 0.6x; 256x128 / 128x256 BK=32 with 8 waves of 64x64, two blocks per CU: 0.7x]
  const int reserve_arg = (tile_cfg >> 8) & 0x1ff;   // per-call CU reserve (see mrblip_gemm_set_cu_reserve)
  const int k_splits = (tile_cfg >> 17) & 0xf;        // skinny kernel: K split with atomic accumulation into a pre-initialised fp32 output
  int cfg = tile_cfg & 0xff;
  if (f16) {   // IEEE fp16 operands: the frozen ViT's GEMMs, which all take the 4-wave 256x256 kernel with a plain epilogue
    MRB_REQUIRE(cfg == 0 || cfg == 13, "gemm_f16: only the 4-wave 256x256 kernel (tile_cfg 0 / 13) has an fp16 form");
    cfg = 13;
  }
  if (cfg == 0) {
    if (M <= 64 && !gated) cfg = 3;
    else {
      // measured on MI355X (tools/gemm_bench.py, tools/gemm_kfit.py): the 256x256 kernel (1 block/CU) only wins for long-K, many-tile
      // problems; 128x128 (2 blocks/CU hide each other's staging latency and epilogue) wins at the ViT shapes; outputs with fewer
      // 128x128 tiles than resident block slots (T5 [2012,2048], Q-Former [1920,768]) are tile-starved -> 64x128 / 64x64 tiles.
      const int ncols = gated ? N / 2 : N;
      const long long t256 = (long long)((M + 255) / 256) * ((ncols + (gated ? 127 : 255)) / (gated ? 128 : 256));
      const long long t128 = (long long)((M + 127) / 128) * ((ncols + (gated ? 63 : 127)) / (gated ? 64 : 128));
      const long long t64x128 = (long long)((M + 63) / 64) * ((ncols + (gated ? 63 : 127)) / (gated ? 64 : 128));
      // 256x256 with 16 waves of 64x64 (4 waves per SIMD hide each other's LDS-DMA issue; half the L2->LDS bytes per flop of 128x128)
      // is the fastest main loop, but it runs one block per CU: it pays only when its rounds are full (ViT fc1: 1464 tiles = 5.7
      // rounds of 256 CUs) or when one partial round covers most CUs (T5 qkv: 192 tiles)
      static int ncu = 0;
      if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
      }
      const long long rounds = (t256 + ncu - 1) / ncu;
      const double eff256 = (double)t256 / (double)(rounds * ncu);
      static int no8 = -1;
      if (no8 < 0) no8 = getenv("MRB_NO_CFG8") ? 1 : 0;
      if (!no8 && !gated && K >= 1024 && (eff256 >= 0.9 || (rounds == 1 && eff256 >= 0.7))) cfg = 8;
      // plain-epilogue GEMMs with many 256x256 tiles or a long K (the frozen ViT's qkv / fc1 / fc2): four waves of 128x128 with the
      // hand-pipelined K loop (cfg 13) - 10-20 % faster than every other form standalone, ~1 ms per step in the train step
      // (round 3: also the ViT proj, 366 tiles of K = 1408 — it ran on the generic 128x128 tile until then: 71.95 -> 71.43 ms per step)
      static int no13 = -1;
      if (no13 < 0) no13 = getenv("MRB_NO_CFG13") ? 1 : 0;
      if (!no13 && !gated && !Aext && !out2 && !(p_drop > 0.f) && (act == 0 || act == 1) && M >= 4096 && (t256 >= 2 * ncu || (t256 >= ncu && K >= 1024)))
        cfg = 13;
      // (256x128 with 16 waves, cfg 10, is 5-15 % faster than 128x128 at the ViT qkv shape standalone, but in the step it made things
      // worse: one more persistent 16-wave block per CU starves the small kernels of the clip that shares the GPU with the look-ahead)
      if (cfg != 0) {
      } else if (t128 >= 400) cfg = 2;
      else if (t64x128 >= 400 || gated) cfg = 4;
      else cfg = 5;
    }
  }
  if (has_extra) {   // the extras live in the generic tile kernel's staging / epilogue; the transposed copies need 64-column wave slabs
    if (a.t_inner > 0 && cfg == 5) cfg = 4;
    if (cfg == 13 || (cfg == 3 && (tile_cfg & 0xff) == 0)) cfg = (M >= 1024 && N >= 1024) ? 2 : 4;
    MRB_REQUIRE(cfg == 1 || cfg == 2 || cfg == 4 || cfg == 7 || cfg == 8 || cfg == 9 || cfg == 11 || (cfg >= 18 && cfg <= 21) || (a.t_inner == 0 && (cfg == 5 || cfg == 10 || cfg == 12)),
                "gemm: head-transposed copies / grouped K extension are not available in tile config %d", cfg);
  }
  MRB_REQUIRE(!a.th_flags || !(cfg == 3 || (cfg >= 13 && cfg <= 17)), "gemm: the thin role lives in the generic tile kernel (tile config %d has none)", cfg);
  if (act == 2) {   // the GELU-backward epilogue lives in the generic tile kernel's non-gated store path
    if (cfg == 3 || cfg == 8 || (cfg >= 13 && cfg <= 17)) cfg = (M >= 1024 && N >= 1024) ? 4 : 5;
    MRB_REQUIRE(cfg == 2 || cfg == 4 || cfg == 5 || (cfg >= 18 && cfg <= 23), "gemm: act 2 (GELU backward) is not available in tile config %d", cfg);
  }
  if (cfg == 3) {
    MRB_REQUIRE(!gated, "gemm: skinny kernel has no gated epilogue");
    // rows of the tall operand per block: 32, fewer when the grid would leave most CUs idle (LoRA down / g products at M = 2012)
    int rpb = 32;
    static int rpb32 = -1;
    if (rpb32 < 0) rpb32 = getenv("MRB_SKINNY_RPB32") ? 1 : 0;
    while (!rpb32 && rpb > 8 && (long long)((N + 31) / 32) * ((M + rpb - 1) / rpb) < 192 && M > rpb) rpb >>= 1;
    a.m_rows_per_block = rpb;
    dim3 grid((N + 31) / 32, (M + rpb - 1) / rpb);
    if (k_splits > 1) {
      MRB_REQUIRE(out_f32 && !residual && !out2 && act == 0, "gemm: the K-split form adds into a pre-initialised fp32 output (no residual / out2 / act)");
      a.k_splits = k_splits;
      grid.z = k_splits;
    }
    // (a 16-wave K-split changes nothing here: with one lane per operand row these launches are bound by the number of row-gather
    // load instructions one CU's address unit can retire, not by a wave's sequential load rounds)
    // (measured and dropped for the tall-and-thin LoRA products of the encoder, M = 2012, N <= 32: one block per 32 rows with 16 waves
    // splitting K - all lanes of every load and dropout hash valid, 4 waves per CU - 10.9 vs 12.4 us at K = 2048 but 38 vs 25 us at
    // K = 10240, +0.4 ms per step: profiles/r02_skinny_tall16.txt)
    if (out_f32) hipLaunchKernelGGL((gemm_skinny_kernel<true, 4>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gemm_skinny_kernel<false, 4>), grid, dim3(256), 0, stream, a);
    return mrblip_check_launch("gemm_skinny");
  }
  if (cfg == 1) {
    if (gated) return launch_tile<256, 256, 2, 4, false, true>(a, stream);
    return out_f32 ? launch_tile<256, 256, 2, 4, true, false>(a, stream) : launch_tile<256, 256, 2, 4, false, false>(a, stream);
  }
  if (cfg == 6) {
    MRB_REQUIRE(!gated, "gemm: the 4-wave 256x256 tile has no gated epilogue");
    return out_f32 ? launch_tile<256, 256, 2, 2, true, false>(a, stream) : launch_tile<256, 256, 2, 2, false, false>(a, stream);
  }
  if (cfg == 7) {  // 256x128 tile, 4 waves of 128x64, BK = 32, 3-stage ring (72 KB): two blocks per CU at 0.75 KB of LDS reads per MFMA
    if (gated) return launch_tile<256, 128, 2, 2, false, true, 32, 3>(a, stream);
    return out_f32 ? launch_tile<256, 128, 2, 2, true, false, 32, 3>(a, stream) : launch_tile<256, 128, 2, 2, false, false, 32, 3>(a, stream);
  }
  if (cfg == 8) {  // 256x256, 16 waves of 64x64 (4 waves per SIMD), one persistent block per CU
    if (gated) return launch_tile<256, 256, 4, 4, false, true>(a, stream);
    return out_f32 ? launch_tile<256, 256, 4, 4, true, false>(a, stream) : launch_tile<256, 256, 4, 4, false, false>(a, stream);
  }
  if (cfg == 9) {  // 128x128 with 8 waves of 32x64, two blocks per CU (4 waves per SIMD): measured = cfg 2, not auto-selected
    if (gated) return launch_tile<128, 128, 4, 2, false, true>(a, stream);
    return out_f32 ? launch_tile<128, 128, 4, 2, true, false>(a, stream) : launch_tile<128, 128, 4, 2, false, false>(a, stream);
  }
  if (cfg == 10) {  // 256x128, 16 waves of 64x32, one persistent block per CU
    MRB_REQUIRE(!gated, "gemm: cfg 10 has no gated epilogue");
    return out_f32 ? launch_tile<256, 128, 4, 4, true, false>(a, stream) : launch_tile<256, 128, 4, 4, false, false>(a, stream);
  }
  if (cfg == 11) {  // 128x256, 16 waves of 32x64 (measured <= cfg 10, not auto-selected)
    MRB_REQUIRE(!gated, "gemm: cfg 11 has no gated epilogue");
    return out_f32 ? launch_tile<128, 256, 4, 4, true, false>(a, stream) : launch_tile<128, 256, 4, 4, false, false>(a, stream);
  }
  if (cfg == 12) {  // 256x192, 8 waves of 64x96, one persistent block per CU: N = 1408 (ViT proj / fc2) is 7.3 x 192 -> 488 tiles = 1.9 rounds
    MRB_REQUIRE(!gated, "gemm: cfg 12 has no gated epilogue");
    return out_f32 ? launch_tile<256, 192, 4, 2, true, false>(a, stream) : launch_tile<256, 192, 4, 2, false, false>(a, stream);
  }
  if (cfg == 15) {  // cfg 13's tile with the deferred epilogue (bf16 out, bias, optional GELU, no residual): ViT qkv / fc1
    MRB_REQUIRE(!gated && !Aext && !out2 && !(p_drop > 0.f) && !out_f32 && !residual && (act == 0 || act == 1), "gemm: cfg 15 = bf16 out, bias, optional GELU only");
    a.tiles_m = (M + 255) / 256;
    a.tiles_n = (N + 255) / 256;
    const int LDS = 2 * (256 + 256) * 128;
    static int ncu15 = 0;
    if (ncu15 == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      ncu15 = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int nt15 = a.tiles_m * a.tiles_n;
    const int reserve = reserve_arg ? reserve_arg / 8 * 8 : g_cu_reserve;
    const int cus = ncu15 - reserve > 8 ? ncu15 - reserve : 8;
    const int grid = nt15 < cus ? (nt15 + 7) / 8 * 8 : cus;
    static bool attr15[2] = {};
    if (act == 1) {
      auto k = gemm_w4d_kernel<1>;
      if (!attr15[1]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) { mrblip_set_error("gemm: cannot raise dynamic LDS to %d", LDS); return MRBLIP_ELAUNCH; }
        attr15[1] = true;
      }
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, stream, a);
    } else {
      auto k = gemm_w4d_kernel<0>;
      if (!attr15[0]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) { mrblip_set_error("gemm: cannot raise dynamic LDS to %d", LDS); return MRBLIP_ELAUNCH; }
        attr15[0] = true;
      }
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, stream, a);
    }
    return mrblip_check_launch("gemm_w4d");
  }
  if (cfg == 16) {  // cfg 13's tile on 16x16x32 MFMAs
    MRB_REQUIRE(!gated && !Aext && !out2 && !(p_drop > 0.f) && (act == 0 || act == 1), "gemm: cfg 16 takes plain epilogues only");
    a.tiles_m = (M + 255) / 256;
    a.tiles_n = (N + 255) / 256;
    const int LDS = 2 * (256 + 256) * 128;
    static int ncu16 = 0;
    if (ncu16 == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      ncu16 = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int nt16 = a.tiles_m * a.tiles_n;
    const int reserve = reserve_arg ? reserve_arg / 8 * 8 : g_cu_reserve;
    const int cus = ncu16 - reserve > 8 ? ncu16 - reserve : 8;
    const int grid = nt16 < cus ? (nt16 + 7) / 8 * 8 : cus;
    const int variant = (out_f32 ? 4 : 0) | (act == 1 ? 2 : 0) | (residual ? 1 : 0);
    static bool attr_set16[8] = {};
#define MRB_W16_LAUNCH(V, F32, ACT_, RES_)                                                                                         \
  case V: {                                                                                                                        \
    auto k = gemm_w16_kernel<F32, ACT_, RES_>;                                                                                     \
    if (!attr_set16[V]) {                                                                                                          \
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {                    \
        mrblip_set_error("gemm: cannot raise dynamic LDS to %d", LDS);                                                             \
        return MRBLIP_ELAUNCH;                                                                                                     \
      }                                                                                                                            \
      attr_set16[V] = true;                                                                                                        \
    }                                                                                                                              \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, stream, a);                                                                  \
    break;                                                                                                                         \
  }
    switch (variant) {
      MRB_W16_LAUNCH(0, false, 0, false)
      MRB_W16_LAUNCH(1, false, 0, true)
      MRB_W16_LAUNCH(2, false, 1, false)
      MRB_W16_LAUNCH(3, false, 1, true)
      MRB_W16_LAUNCH(4, true, 0, false)
      MRB_W16_LAUNCH(5, true, 0, true)
      MRB_W16_LAUNCH(6, true, 1, false)
      MRB_W16_LAUNCH(7, true, 1, true)
    }
#undef MRB_W16_LAUNCH
    return mrblip_check_launch("gemm_w16");
  }
  // cfg 17 (round 4): cfg 13 with a third stage buffer for the W operand (gemm_w4_kernel W3).  Measured (profiles/r04_w3_bench.txt):
  // bit-identical; with the chip to itself equal within noise (8192^3 1426 vs 1426 TFLOP/s, fc1 252.5 vs 250.5 us, fc2 298 vs 303 us);
  // on the 192 CUs the look-ahead runs on: 8192^3 1217 vs 1140 TFLOP/s, fc2 280 vs 285 us, fc1 in the step 340 vs 348 us; step 70.48 vs
  // 70.45 ms.  So the late W pieces were NOT what the K loop waits for; the form is kept as the library's choice for an automatic cfg 13
  // because it is never slower where it runs (MRB_W4_W3=0: two stages; an explicit tile_cfg 13 always means two stages)
  bool w3 = cfg == 17;
  if (cfg == 13 && !f16 && (tile_cfg & 0xff) == 0) {
    static int w3env = -1;
    if (w3env < 0) { const char* e = getenv("MRB_W4_W3"); w3env = (e && e[0] == '0') ? 0 : 1; }
    w3 = w3env == 1;
  }
  if (cfg == 17) cfg = 13;
  if (cfg == 13 || cfg == 14) {  // four waves of 128 x 128 (cfg 13, 256x256 tile) / 128 x 96 (cfg 14, 256x192), hand-pipelined K loop
    MRB_REQUIRE(gated || (!Aext && !out2 && !(p_drop > 0.f)), "gemm: cfg 13 / 14 take plain epilogues only (or the gated-GELU form of cfg 13)");
    MRB_REQUIRE(!gated || (cfg == 13 && !Aext && !f16 && (N % 16) == 0 && (!out2 || (long long)M * ldo2 * 2 < (1ll << 31))),
                "gemm: the gated form of the 4-wave kernel is cfg 13 on bf16 operands without a K extension ([x | u] x [W | B]^T instead)");
    MRB_REQUIRE(act == 0 || act == 1, "gemm: cfg 13 / 14 know act 0 / 1");
    MRB_REQUIRE((long long)M * ldo * (out_f32 ? 4 : 2) < (1ll << 31) && (!residual || (long long)M * ldr * 4 < (1ll << 31)),
                "gemm: cfg 13 / 14 address output and residual through 2 GiB buffer resources");
    const int bn13 = cfg == 13 ? 256 : 192;
    a.tiles_m = (M + 255) / 256;
    a.tiles_n = gated ? (N / 2 + 127) / 128 : (N + bn13 - 1) / bn13;      // (gated: 128 output columns per tile)
    if (gated) w3 = false;
    const int stage13 = (256 + bn13) * 128, slab13 = 4 * 32 * (bn13 / 2 * 4 + 16);
    MRB_REQUIRE(!w3 || (cfg == 13 && !f16), "gemm: the three-W-stage form (cfg 17) exists for the bf16 256x256 tile");
    const int LDS = w3 ? 2 * stage13 + 256 * 128 : (2 * stage13 > stage13 + slab13 ? 2 * stage13 : stage13 + slab13);
    static int ncu13 = 0;
    if (ncu13 == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      ncu13 = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int nt13 = a.tiles_m * a.tiles_n;
    // g_cu_reserve: CUs this persistent kernel leaves alone (mrblip_gemm_set_cu_reserve).  Its blocks hold a whole CU each for the
    // whole launch (all 512 registers of every SIMD), so kernels of another stream can only start on CUs it does not occupy: the
    // frozen-ViT look-ahead of the train step gives the clip that is being trained a quarter of the chip this way.
    const int reserve = reserve_arg ? reserve_arg / 8 * 8 : g_cu_reserve;
    const int cus = ncu13 - reserve > 8 ? ncu13 - reserve : 8;
    // (a multiple of 8: the same number of blocks on every XCD.  Measured and dropped in round 3: shrinking the grid to the fewest CUs
    // that keep the round count — ceil(tiles / rounds): ViT fc1 1464 tiles = 8 rounds on 184 CUs as on 192 — so that the other stream
    // gets the difference: +0.3 ms per step; the CUs of a partly filled last round are not idle, they go to the other stream EARLIER.)
    const int grid = nt13 < cus ? (nt13 + 7) / 8 * 8 : cus;
    const int variant = gated ? 32 : (w3 ? 24 : 0) + ((f16 ? 16 : 0) | (cfg == 14 ? 8 : 0) | (out_f32 ? 4 : 0) | (act == 1 ? 2 : 0) | (residual ? 1 : 0));
    static bool attr_set13[33] = {};
#define MRB_W4_LAUNCH(V, F32, ACT_, RES_, TN_, ...)                                                                                \
  case V: {                                                                                                                        \
    auto k = gemm_w4_kernel<F32, ACT_, RES_, TN_, ##__VA_ARGS__>;                                                                                 \
    if (!attr_set13[V]) {                                                                                                          \
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {                    \
        mrblip_set_error("gemm: cannot raise dynamic LDS to %d", LDS);                                                             \
        return MRBLIP_ELAUNCH;                                                                                                     \
      }                                                                                                                            \
      attr_set13[V] = true;                                                                                                        \
    }                                                                                                                              \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, stream, a);                                                                  \
    break;                                                                                                                         \
  }
    switch (variant) {
      MRB_W4_LAUNCH(0, false, 0, false, 4)
      MRB_W4_LAUNCH(1, false, 0, true, 4)
      MRB_W4_LAUNCH(2, false, 1, false, 4)
      MRB_W4_LAUNCH(3, false, 1, true, 4)
      MRB_W4_LAUNCH(4, true, 0, false, 4)
      MRB_W4_LAUNCH(5, true, 0, true, 4)
      MRB_W4_LAUNCH(6, true, 1, false, 4)
      MRB_W4_LAUNCH(7, true, 1, true, 4)
      MRB_W4_LAUNCH(8, false, 0, false, 3)
      MRB_W4_LAUNCH(9, false, 0, true, 3)
      MRB_W4_LAUNCH(10, false, 1, false, 3)
      MRB_W4_LAUNCH(11, false, 1, true, 3)
      MRB_W4_LAUNCH(12, true, 0, false, 3)
      MRB_W4_LAUNCH(13, true, 0, true, 3)
      MRB_W4_LAUNCH(14, true, 1, false, 3)
      MRB_W4_LAUNCH(15, true, 1, true, 3)
      // fp16 operands (ViT): qkv (bf16-sized out + bias), fc1 (+ GELU), proj / fc2 (fp32 residual stream), patch embedding (fp32 out + bias)
      MRB_W4_LAUNCH(16, false, 0, false, 4, true)
      MRB_W4_LAUNCH(18, false, 1, false, 4, true)
      MRB_W4_LAUNCH(20, true, 0, false, 4, true)
      MRB_W4_LAUNCH(21, true, 0, true, 4, true)
      // three W stages (cfg 17): the frozen ViT's four bf16 forms
      MRB_W4_LAUNCH(24, false, 0, false, 4, false, true)
      MRB_W4_LAUNCH(26, false, 1, false, 4, false, true)
      MRB_W4_LAUNCH(28, true, 0, false, 4, false, true)
      MRB_W4_LAUNCH(29, true, 0, true, 4, false, true)
      // gated-GELU (T5 wi_0 / wi_1): bf16 y + bf16 pre-activations, dropout
      MRB_W4_LAUNCH(32, false, 0, false, 4, false, false, false, true)
      default:
        mrblip_set_error("gemm: no such 4-wave kernel variant (%d)%s", variant, f16 ? " - the fp16 form exists for bias / bias+GELU (16-bit out) and fp32 out with or without residual" : "");
        return MRBLIP_EINVAL;
    }
#undef MRB_W4_LAUNCH
    return mrblip_check_launch("gemm_w4");
  }
  // (round 3, measured at the T5 [2012 x 2048] outputs and removed again: 128x128 tiles — one per CU, 2/3 of the L2->LDS traffic of
  // 64x128 — with a 3- or 4-stage ring, or with 8 waves of 64x32: K = 10240: 115.9 / 114.6 / 112.5 us against 120.6 us for the plain
  // 128x128 and 105.1 us for 64x128 at two blocks per CU; profiles/r03_gemm_t5_tiles.txt)
  // deeper rings of the small tiles: with weights that come from HBM (every layer's own, not the re-used panel of a stand-alone loop) one
  // K-tile of prefetch distance (0.5 us) is shorter than the memory latency
  if (cfg == 18) {  // 64x128, 3 stages (72 KB: still two blocks per CU)
    if (gated) return launch_tile<64, 128, 2, 2, false, true, 64, 3>(a, stream);
    return out_f32 ? launch_tile<64, 128, 2, 2, true, false, 64, 3>(a, stream) : launch_tile<64, 128, 2, 2, false, false, 64, 3>(a, stream);
  }
  if (cfg == 19) {  // 64x128, 4 stages (96 KB: one block per CU)
    if (gated) return launch_tile<64, 128, 2, 2, false, true, 64, 4>(a, stream);
    return out_f32 ? launch_tile<64, 128, 2, 2, true, false, 64, 4>(a, stream) : launch_tile<64, 128, 2, 2, false, false, 64, 4>(a, stream);
  }
  if (cfg == 20) {  // 128x128, 3 stages (96 KB)
    if (gated) return launch_tile<128, 128, 2, 2, false, true, 64, 3>(a, stream);
    return out_f32 ? launch_tile<128, 128, 2, 2, true, false, 64, 3>(a, stream) : launch_tile<128, 128, 2, 2, false, false, 64, 3>(a, stream);
  }
  if (cfg == 21) {  // 128x128, 4 stages (128 KB)
    if (gated) return launch_tile<128, 128, 2, 2, false, true, 64, 4>(a, stream);
    return out_f32 ? launch_tile<128, 128, 2, 2, true, false, 64, 4>(a, stream) : launch_tile<128, 128, 2, 2, false, false, 64, 4>(a, stream);
  }
  // round 6: the same for the 64x64 tile (the Q-Former's M = 1920 products over K = 768: twelve K-tiles, each a full LDS-DMA latency with two stages)
  if (cfg == 22) {  // 64x64, 3 stages (48 KB: three blocks per CU)
    MRB_REQUIRE(!gated, "gemm: the 64x64 tile has no gated epilogue");
    return out_f32 ? launch_tile<64, 64, 2, 2, true, false, 64, 3>(a, stream) : launch_tile<64, 64, 2, 2, false, false, 64, 3>(a, stream);
  }
  if (cfg == 23) {  // 64x64, 4 stages (64 KB: two blocks per CU)
    MRB_REQUIRE(!gated, "gemm: the 64x64 tile has no gated epilogue");
    return out_f32 ? launch_tile<64, 64, 2, 2, true, false, 64, 4>(a, stream) : launch_tile<64, 64, 2, 2, false, false, 64, 4>(a, stream);
  }
  if (cfg == 4) {
    if (gated) return launch_tile<64, 128, 2, 2, false, true>(a, stream);
    return out_f32 ? launch_tile<64, 128, 2, 2, true, false>(a, stream) : launch_tile<64, 128, 2, 2, false, false>(a, stream);
  }
  if (cfg == 5) {
    MRB_REQUIRE(!gated, "gemm: the 64x64 tile has no gated epilogue");
    return out_f32 ? launch_tile<64, 64, 2, 2, true, false>(a, stream) : launch_tile<64, 64, 2, 2, false, false>(a, stream);
  }
  if (gated) return launch_tile<128, 128, 2, 2, false, true>(a, stream);
  return out_f32 ? launch_tile<128, 128, 2, 2, true, false>(a, stream) : launch_tile<128, 128, 2, 2, false, false>(a, stream);
}

extern "C" int mrblip_gemm_bf16(const void* A, long long lda, const void* W, long long ldw, const void* Aext, long long ldaext,
                                const void* Wext, long long ldwext, int M, int N, int K, void* out, long long ldo, int out_f32,
                                void* out2, long long ldo2, const float* bias, const float* residual, long long ldr, int act,
                                int gated, const uint32_t* seed_ptr, uint32_t site, float p_drop, int tile_cfg,
                                hipStream_t stream) {
  return gemm_dispatch(A, lda, W, ldw, Aext, ldaext, Wext, ldwext, M, N, K, out, ldo, out_f32, out2, ldo2, bias, residual, ldr, act, gated,
                       seed_ptr, site, p_drop, tile_cfg, 0, 0, 0.f, 0, 0.f, stream);
}

// The same entry with IEEE fp16 operands (and fp16 for a 16-bit output): the frozen ViT's GEMMs (plain epilogues, 4-wave 256x256 kernel).
extern "C" int mrblip_gemm_f16(const void* A, long long lda, const void* W, long long ldw, const void* Aext, long long ldaext,
                               const void* Wext, long long ldwext, int M, int N, int K, void* out, long long ldo, int out_f32,
                               void* out2, long long ldo2, const float* bias, const float* residual, long long ldr, int act,
                               int gated, const uint32_t* seed_ptr, uint32_t site, float p_drop, int tile_cfg,
                               hipStream_t stream) {
  return gemm_dispatch(A, lda, W, ldw, Aext, ldaext, Wext, ldwext, M, N, K, out, ldo, out_f32, out2, ldo2, bias, residual, ldr, act, gated,
                       seed_ptr, site, p_drop, tile_cfg, 0, 0, 0.f, 0, 0.f, stream, true);
}

// LoRA "down" product with the input dropout fused into the operand load:  U[M, N] = dropout(X)[M, K] Acat[N, K]^T, bf16 out.
// peft Linear.forward: lora_A(lora_dropout(x)).
extern "C" int mrblip_gemm_lora_down(const void* X, long long ldx, const void* Acat, long long lda_, int M, int N, int K, void* U, long long ldu,
                                     const uint32_t* seed_ptr, uint32_t site, float p_drop, hipStream_t stream) {
  return gemm_dispatch(X, ldx, Acat, lda_, nullptr, 0, nullptr, 0, M, N, K, U, ldu, 0, nullptr, 0, nullptr, nullptr, 0, 0, 0, seed_ptr, 0, 0.f, 3,
                       0, 0, 0.f, site, p_drop, stream);
}

// LoRA backward input gradient in one launch:  dX[M, N] = dY[M, K] Wt[N, K]^T (+ residual) + mask(site, p) * (G[M, 64] AcatT[N, 64]^T)
// where mask is the lora_dropout keep mask of the forward input (scaled by 1/(1-p)), N = in_features, K = padded out_features.
extern "C" int mrblip_gemm_lora_dx(const void* dY, long long lddy, const void* Wt, long long ldwt, const void* G, long long ldg,
                                   const void* AcatT, long long ldat, int M, int N, int K, void* dX, long long lddx, int out_f32,
                                   const float* residual, long long ldr, const uint32_t* seed_ptr, uint32_t site, float p_drop,
                                   int tile_cfg, hipStream_t stream) {
  return gemm_dispatch(dY, lddy, Wt, ldwt, G, ldg, AcatT, ldat, M, N, K, dX, lddx, out_f32, nullptr, 0, nullptr, residual, ldr, 0, 0, seed_ptr, 0,
                       0.f, tile_cfg, 1, site, p_drop, 0, 0.f, stream);
}

// K-split form of the 4-wave kernel (round 5; gemm_w4_kernel<..., SPLIT>):  out[s] = A[:, s-th K range] W[:, s-th K range]^T for s < k_splits,
// and with a K extension  out[k_splits] = Aext Wext^T  — PARTIAL products, parts part_stride elements apart, fp32 or bf16.  The consumer
// adds them (mrblip_rmsnorm_bwd_parts / mrblip_gated_gelu_bwd_parts: in part order, the extension part under the lora_dropout mask).
// For the LoRA input gradients of the T5 encoder:  dX = dY W (+) mask (.) (g A): lora.py Linear.forward differentiated, blip2_mr.py:182-200.
// tile_cfg 13 (256x256 tiles) or 14 (256x192).  K %% (64 k_splits) == 0.
extern "C" int mrblip_gemm_ksplit(const void* A, long long lda, const void* W, long long ldw, const void* Aext, long long ldaext, const void* Wext,
                                  long long ldwext, int M, int N, int K, void* out, long long ldo, long long part_stride, int out_f32, int k_splits,
                                  int tile_cfg, hipStream_t stream) {
  const int cfg = tile_cfg & 0xff;
  MRB_REQUIRE(cfg == 13 || cfg == 14 || cfg == 22, "gemm_ksplit: tile_cfg 13 (256x256), 14 (256x192) or 22 (256x128)");
  MRB_REQUIRE(M > 0 && N > 0 && (N % 8) == 0 && k_splits >= 1 && k_splits <= 16 && K > 0 && (K % (64 * k_splits)) == 0,
              "gemm_ksplit: need M, N > 0, N %% 8 == 0 and K %% (64 k_splits) == 0 (M=%d N=%d K=%d k_splits=%d)", M, N, K, k_splits);
  MRB_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && (ldo % (out_f32 ? 4 : 8)) == 0 && (part_stride % 8) == 0, "gemm_ksplit: leading dims must keep 16-B alignment");
  MRB_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_ksplit: pointers must be 16-B aligned");
  MRB_REQUIRE((Aext == nullptr) == (Wext == nullptr) && (!Aext || (((uintptr_t)Aext % 16) == 0 && ((uintptr_t)Wext % 16) == 0 && (ldaext % 8) == 0 && (ldwext % 8) == 0 && ldaext >= 64 && ldwext >= 64)),
              "gemm_ksplit: the K extension is a pair of 64-column operands with 16-B rows");
  MRB_REQUIRE(((long long)(M + 256) * lda * 2 + 256) < (1ll << 32) && ((long long)(N + 256) * ldw * 2 + 256) < (1ll << 32),
              "gemm_ksplit: operand exceeds the 4 GiB buffer-descriptor range");
  const int parts = k_splits + (Aext ? 1 : 0);
  MRB_REQUIRE(part_stride >= (long long)(M - 1) * ldo + N && ((long long)(parts - 1) * part_stride + (long long)M * ldo) * (out_f32 ? 4 : 2) < (1ll << 31),
              "gemm_ksplit: the parts must not overlap and together stay below 2 GiB");
  // the one-shot extras of the generic tile kernel are not for this launch: drop them loudly
  MRB_REQUIRE(!g_gemm_extra.set && !g_gemm_thin.set && !g_gemm_prefetch.ptr, "gemm_ksplit: head-transposed copies / thin role / prefetch role belong to the generic tile kernel");
  GemmArgs a = {};
  a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.Aext = (const bf16_t*)Aext; a.Wext = (const bf16_t*)Wext;
  a.out = out; a.lda = lda; a.ldw = ldw; a.ldaext = ldaext; a.ldwext = ldwext; a.ldo = ldo;
  a.M = M; a.N = N; a.K = K; a.k_splits = k_splits; a.part_stride = part_stride;
  a.drop.inv_keep = a.ext_drop.inv_keep = a.a_drop.inv_keep = a.th_drop.inv_keep = 1.0f;
  const int bn = cfg == 13 ? 256 : cfg == 14 ? 192 : 128;
  a.tiles_m = (M + 255) / 256;
  a.tiles_n = (N + bn - 1) / bn;
  const int stage = (256 + bn) * 128, slab = 4 * 32 * (bn / 2 * 4 + 16);
  static int w3s = -1;    // MRB_KSPLIT_W3=1: the 256x256 form with a third W stage (experiment; see cfg 17)
  if (w3s < 0) { const char* e = getenv("MRB_KSPLIT_W3"); w3s = (e && e[0] == '1') ? 1 : 0; }
  const bool w3 = w3s == 1 && cfg == 13;
  const int LDS = w3 ? 2 * stage + 256 * 128 : (2 * stage > stage + slab ? 2 * stage : stage + slab);
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int units = a.tiles_m * a.tiles_n * parts;
  const int reserve = ((tile_cfg >> 8) & 0x1ff) ? ((tile_cfg >> 8) & 0x1ff) / 8 * 8 : g_cu_reserve;
  const int cus = ncu - reserve > 8 ? ncu - reserve : 8;
  const int grid = units < cus ? (units + 7) / 8 * 8 : cus;
  const int variant = w3 ? 6 + (out_f32 ? 1 : 0) : ((cfg == 14 ? 2 : cfg == 22 ? 4 : 0) | (out_f32 ? 1 : 0));
  static bool attr_set[8] = {};
#define MRB_W4S_LAUNCH(V, F32, TN_, ...)                                                                                           \
  case V: {                                                                                                                        \
    auto k = gemm_w4_kernel<F32, 0, false, TN_, false, ##__VA_ARGS__>;                                                             \
    if (!attr_set[V]) {                                                                                                            \
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {                    \
        mrblip_set_error("gemm_ksplit: cannot raise dynamic LDS to %d", LDS);                                                      \
        return MRBLIP_ELAUNCH;                                                                                                     \
      }                                                                                                                            \
      attr_set[V] = true;                                                                                                          \
    }                                                                                                                              \
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS, stream, a);                                                                  \
    break;                                                                                                                         \
  }
  switch (variant) {
    MRB_W4S_LAUNCH(0, false, 4, false, true)
    MRB_W4S_LAUNCH(1, true, 4, false, true)
    MRB_W4S_LAUNCH(2, false, 3, false, true)
    MRB_W4S_LAUNCH(3, true, 3, false, true)
    MRB_W4S_LAUNCH(4, false, 2, false, true)
    MRB_W4S_LAUNCH(5, true, 2, false, true)
    MRB_W4S_LAUNCH(6, false, 4, true, true)
    MRB_W4S_LAUNCH(7, true, 4, true, true)
  }
#undef MRB_W4S_LAUNCH
  return mrblip_check_launch("gemm_ksplit");
}

// ---- thin "TN" product for the LoRA weight gradients:  D[r, c] += sum_m U[m, r] * drop(Y)[m, c],  r < 32, contraction over the
// ROWS of two row-major matrices (dB^T = u^T dy, dA = g^T dropout(x)).  No transposed copies: the MFMA fragments are gathered with
// 2-byte loads that are 64-B coalesced across lanes (lane = column), which is cheap for this memory-light product (Y is read once).
// One block = 32 columns of Y, its 8 waves split M and reduce through LDS; every (r, c) has one owner block, so the += needs no
// atomics.  Up to 4 adapters share one launch: rows [8j, 8j+8) go to seg[j] for the columns seg[j] owns (block-diagonal for dB^T).
struct TnSeg { float* out; int col0, ncols; long long ld; };
struct TnArgs {
  const bf16_t* Y; const bf16_t* U;
  long long ldy, ldu;
  int M, C, R;
  TnSeg seg[4];
  DropoutArg drop;
};

struct TnArgs2 { TnArgs a, b; int blocks_a; int has_b; };  // two independent problems in one launch (blocks [0, blocks_a) -> a, the rest -> b)
// Round 5: up to TN_MAX_PROBLEMS independent problems in one launch — the weight-gradient pairs of ALL LoRA groups of a T5 layer (round
// 4: one launch per group, ~305 launches and ~5 ms of side-stream kernel time per step for ~20 GFLOP).  blk_end[i] = first block id
// behind problem i.
#define TN_MAX_PROBLEMS 16
struct TnMulti { TnArgs p[TN_MAX_PROBLEMS]; int blk_end[TN_MAX_PROBLEMS]; int n; };

// the block body shared by both kernels: the problem's scalars arrive in registers (see the note on kernarg re-fetches below), its four
// output segments through a pointer into the kernel arguments (read once, in the epilogue).
// CT (round 6): 32-column tiles of Y per block.  With CT = 1 a block reads 64 B of every Y row — half a cache line, the other half belongs
// to the neighbour block, and every block re-reads all of U: the encoder layer's launch moved 128 MB of Y and 128 MB of U through the
// L2s in 66-77 us (1.65 TB/s of algorithmic bytes).  CT = 2: whole 128-B lines per row, half the blocks, half the U traffic; the U
// fragment of a slice serves both tiles.  Same per-wave row ranges, same MFMA accumulation order, same wave-order reduction: the same bits.
template <int CT>
__device__ __forceinline__ void lora_tn_body(const bf16_t* __restrict__ const Y, const bf16_t* __restrict__ const U, const long long ldy,
                                             const long long ldu, const int M, const int C, const int R, const uint32_t* const seed_ptr,
                                             const uint32_t site, const uint32_t thresh24, const float inv_keep, const TnSeg* const segs,
                                             const int bx) {
  const bool has_drop = seed_ptr != nullptr;
  // Each wave stages its own 16-row slices of Y (16 x 32 CT columns) and U (16 x 32) with 16-B global loads (CT per lane for Y, one for
  // U), parks them in a wave-private LDS slot and gathers the k-major MFMA fragments from there with the LDS transpose read.
  // UNROLL 16-row slices are in flight per wave and go through the wave's LDS slots in pairs.  (The 2 / 4 / 8 / 16 sweep of
  // profiles/r03_lora_tn_unroll.txt — no gain from more slices in flight — was taken BEFORE the kernarg re-fetch above was found; see
  // profiles/r03_lora_tn_v2.txt for the sweep after it.)
#ifndef LORA_TN_UNROLL
#define LORA_TN_UNROLL 2
#endif
  constexpr int UNROLL = LORA_TN_UNROLL;  // slices per iteration = LDS slots per wave
  constexpr int YP = 64 * CT;             // row pitch of a Y slice image (bytes)
  constexpr int YIMG = 16 * YP, UIMG = 16 * 64, SLOT = YIMG + UIMG;
  // LDS: the wave-private staging slots and, after the M loop, the cross-wave reduction buffer (32 KB) share one allocation
  constexpr int SLOT_BYTES = 8 * UNROLL * SLOT, RED_BYTES = 8 * 16 * 64 * 4;
  __shared__ __attribute__((aligned(16))) char tn_lds[SLOT_BYTES > RED_BYTES ? SLOT_BYTES : RED_BYTES];
  typedef float red_t[16][64];
  red_t* const red = reinterpret_cast<red_t*>(tn_lds);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hi = lane >> 5, l31 = lane & 31;
  const int c0 = bx * 32 * CT;
  const uint32_t seed = has_drop ? *seed_ptr : 0u;
  f32x16 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  const int steps = (M + 15) / 16;                 // 16 rows of M per MFMA
  const int per = (steps + 7) / 8;
  const int s_beg = w * per, s_end = min(steps, s_beg + per);
  // staging roles.  U: row lane >> 2 of the 16-row slice, 8-column chunk lane & 3.  Y: 4 CT lanes per row, 16 / CT rows per instruction
  const int srow = lane >> 2, schunk = lane & 3;
  constexpr int YLPR = 4 * CT, YRPI = 64 / YLPR;
  const int yrow = lane / YLPR, ychunk = lane % YLPR;
  // transpose-read addresses inside the slice images ([16][32 CT] for Y, [16][32] for U)
  const int trr = 8 * hi + ((lane & 15) >> 2), trc = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int tr_y = trr * YP + trc, tr_u = trr * 64 + trc;
  const bool ychunk_ok = c0 + 8 * ychunk < C, uchunk_ok = 8 * schunk < R;
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const bf16_t* const ysrc = Y + c0 + 8 * ychunk;
  const bf16_t* const usrc = U + 8 * schunk;
  char* const myslot = tn_lds + w * (UNROLL * SLOT);
  bf16x8 gy[UNROLL][CT], gu[UNROLL];
  // software pipeline: the global loads of iteration i+1 are issued as soon as iteration i's registers have been parked in LDS, and
  // fly while iteration i's fragments are gathered and multiplied
  auto fetch = [&](int s0) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const bool s_ok = s0 + u < s_end;
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const int m = (s0 + u) * 16 + j * YRPI + yrow;
        const bool ok = s_ok && m < M;
        gy[u][j] = (ok && ychunk_ok) ? *reinterpret_cast<const bf16x8*>(ysrc + (long long)(ok ? m : 0) * ldy) : zero;
      }
      const int m = (s0 + u) * 16 + srow;
      const bool ok = s_ok && m < M;
      gu[u] = (ok && uchunk_ok) ? *reinterpret_cast<const bf16x8*>(usrc + (long long)(ok ? m : 0) * ldu) : zero;
    }
  };
  if (s_beg < s_end) fetch(s_beg);
#pragma unroll 1
  for (int s0 = s_beg; s0 < s_end; s0 += UNROLL) {
#pragma unroll
    for (int v = 0; v < UNROLL; ++v) {
#pragma unroll
      for (int j = 0; j < CT; ++j) *reinterpret_cast<bf16x8*>(myslot + v * SLOT + (j * YRPI + yrow) * YP + ychunk * 16) = gy[v][j];
      *reinterpret_cast<bf16x8*>(myslot + v * SLOT + YIMG + srow * 64 + schunk * 16) = gu[v];
    }
    if (s0 + UNROLL < s_end) fetch(s0 + UNROLL);   // wave-uniform
#pragma unroll
    for (int v = 0; v < UNROLL; ++v) {
      // k-major MFMA fragments out of the row-major slices: lane (column l31, hi) needs rows 8 hi .. 8 hi + 7 of its column.  gfx950's
      // LDS transpose read hands a 16-lane group the columns of a [4 rows][16 columns] block (lane i supplies the address of 4
      // contiguous columns of row i / 4, lane c receives the 4 rows of column c): two reads per operand instead of eight 2-byte reads
      // (the same gather as the ViT attention's row-major V, attention.hip; lane map probed in tools/probes/tr_probe.hip)
      typedef short tn_v4s __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(3))) tn_v4s* tn_tr_ptr;
      const char* su = myslot + v * SLOT + YIMG + tr_u;
      const tn_v4s u0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_tr_ptr)(su)), u1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_tr_ptr)(su + 4 * 64));
      const bf16x8 uf = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const char* sy = myslot + v * SLOT + tr_y + ct * 64;
        const tn_v4s y0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_tr_ptr)(sy)), y1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_tr_ptr)(sy + 4 * YP));
        bf16x8 yf = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
        if (has_drop) {  // wave-uniform
          // One 32-bit hash serves the element pair (c even, c + 1) of a row (mrb_keep) and the pair sits in neighbouring lanes: each lane
          // hashes FOUR of its eight rows and takes the other four from its partner (DPP quad_perm [1,0,3,2]) — the hashes were ~1/3 of
          // this kernel's time (o group 14.7 us with the mask, 9.6 us without).  (C % 8 == 0 and c0 % 32 == 0: the pair never straddles rows.)
          const int q = lane & 1;
          const int c = c0 + 32 * ct + l31;
          uint32_t mine[4], theirs[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int m = (s0 + v) * 16 + 8 * hi + 4 * q + jj;
            mine[jj] = mrb_hash(((uint32_t)m * (uint32_t)C + (uint32_t)c) >> 1, seed, site);
          }
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) theirs[jj] = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine[jj], 0xB1, 0xf, 0xf, true);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t h = ((j >> 2) == q) ? mine[j & 3] : theirs[j & 3];
            const bool keep = (q ? (h >> 16) : (h & 0xffffu)) >= thresh24;
            yf[j] = keep ? (short)f2bf(bf2f((bf16_t)yf[j]) * inv_keep) : (short)0;
          }
        }
        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uf, yf, acc[ct], 0, 0, 0);
      }
    }
  }
  // epilogue spread over the 8 waves, one column tile at a time: wave w owns accumulator rows r = 2w, 2w+1 (both in adapter segment
  // w >> 1), sums the 8 partials in wave order (same order as ever: bit-identical) and does its two read-modify-writes with both loads in
  // flight.  (Before: wave 0 did all 16 as a chain of load -> wait -> store round trips.)
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    __syncthreads();  // every wave is done with its staging slots (ct > 0: with the previous tile's partials): the reduction buffer may overwrite them
#pragma unroll
    for (int r = 0; r < 16; ++r) red[w][r][lane] = acc[ct][r];
    __syncthreads();
    const int c = c0 + 32 * ct + l31;
    const bool c_ok = c < C;
    const int sgi = w >> 1;
    const TnSeg sg = segs[sgi];
    float v[2], old[2];
    float* ptr[2];
    bool ok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = 2 * w + t;
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;   // D row (= LoRA rank index over the stacked adapters), column c = lane
      v[t] = red[0][r][lane];
#pragma unroll
      for (int j = 1; j < 8; ++j) v[t] += red[j][r][lane];
      ok[t] = c_ok && row < R && sg.out != nullptr && c >= sg.col0 && c < sg.col0 + sg.ncols;
      ptr[t] = sg.out + ((long long)(row & 7) * sg.ld + (c - sg.col0));
      old[t] = ok[t] ? *ptr[t] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (ok[t]) *ptr[t] = old[t] + v[t];
  }
}

template <int CT>
__global__ __launch_bounds__(512) void lora_tn_kernel(const TnArgs2 q) {
  const bool second = (int)blockIdx.x >= q.blocks_a;  // block-uniform
  const int bx = second ? blockIdx.x - q.blocks_a : blockIdx.x;
  // The problem's scalars are copied out of the kernel arguments ONCE.  (Round 3, ISA reading: with `const TnArgs& p = second ? q.b : q.a`
  // the compiler kept a run-time pointer into the kernarg segment and re-fetched p.Y / p.ldy / p.drop.* with s_load + s_waitcnt lgkmcnt(0)
  // at every use — 40 scalar round trips per loop iteration, ~1 us per pair of slices.)
#define TN_SEL(f) (second ? q.b.f : q.a.f)
  lora_tn_body<CT>(TN_SEL(Y), TN_SEL(U), TN_SEL(ldy), TN_SEL(ldu), TN_SEL(M), TN_SEL(C), TN_SEL(R), TN_SEL(drop.seed_ptr), TN_SEL(drop.site),
               TN_SEL(drop.thresh24), TN_SEL(drop.inv_keep), second ? q.b.seg : q.a.seg, bx);
#undef TN_SEL
}

template <int CT>
__global__ __launch_bounds__(512) void lora_tn_multi_kernel(const TnMulti q) {
  int pi = 0, b0 = 0;   // block-uniform: the problem this block belongs to (scalar compares over <= 16 cumulative counts)
#pragma unroll 1
  for (int i = 0; i + 1 < q.n; ++i)
    if ((int)blockIdx.x >= q.blk_end[i]) { pi = i + 1; b0 = q.blk_end[i]; }
  const TnArgs& a = q.p[pi];
  // (copied into registers once, as above — the references below are each evaluated exactly here)
  const bf16_t* const Y = a.Y; const bf16_t* const U = a.U;
  const long long ldy = a.ldy, ldu = a.ldu;
  const int M = a.M, C = a.C, R = a.R;
  const uint32_t* const seed_ptr = a.drop.seed_ptr;
  const uint32_t site = a.drop.site, thresh24 = a.drop.thresh24;
  const float inv_keep = a.drop.inv_keep;
  lora_tn_body<CT>(Y, U, ldy, ldu, M, C, R, seed_ptr, site, thresh24, inv_keep, a.seg, (int)blockIdx.x - b0);
}

// (A VALU form of these products - one lane per Y column, the row's U values as wave-uniform scalar loads, M split over 8 waves and
// merged through LDS - was built and measured: 130-270 us per launch against 23-30 us here.  With 2-byte loads per lane and 4 waves per
// CU it keeps ~0.5 MB in flight where the HBM pipe needs ~16 MB; the MFMA form's 16-B row loads win by an order of magnitude.)
// column tiles per block (CT).  Measured (tools/lora_grads_bench.py, profiles/r06_lora_tn_ct.txt): the encoder layer's batched launch (16
// problems, 992 blocks of 32 columns) 45.7 -> 36.2 us with CT = 2 (25.9 us without the lora_dropout hashes); a single group's launch
// (64-256 blocks) gets SLOWER, 12-14 -> 20 us — half the blocks on a chip it did not fill, and every wave's chain of slices is latency
// bound.  So: CT = 2 for batched launches whose every problem has a column count that is a multiple of 64, CT = 1 otherwise.
// MRB_LORA_TN_CT = 1 / 2 forces one form for A/B (2 still needs the multiples of 64).
static int tn_ct(const TnArgs* const* ps, int n, bool batched) {
  static int env = -1;
  if (env < 0) { const char* e = getenv("MRB_LORA_TN_CT"); env = e ? atoi(e) : 0; }
  if (env == 1 || (env == 0 && !batched)) return 1;
  for (int i = 0; i < n; ++i)
    if (ps[i]->C % 64) return 1;
  return 2;
}
static int launch_tn(TnArgs2& q, hipStream_t stream) {   // blocks of a, then blocks of b
  const TnArgs* ps[2] = {&q.a, &q.b};
  const int ct = tn_ct(ps, 2, false);
  q.blocks_a = (q.a.C + 32 * ct - 1) / (32 * ct);
  const int blocks = q.blocks_a + (q.has_b ? (q.b.C + 32 * ct - 1) / (32 * ct) : 0);
  if (ct == 2) hipLaunchKernelGGL(lora_tn_kernel<2>, dim3(blocks), dim3(512), 0, stream, q);
  else hipLaunchKernelGGL(lora_tn_kernel<1>, dim3(blocks), dim3(512), 0, stream, q);
  return mrblip_check_launch("lora_tn");
}

static int tn_fill(TnArgs& a, const void* Y, long long ldy, const void* U, long long ldu, int M, int C, int R, float* const* outs, const int* col0,
                   const int* ncols, const long long* lds, const uint32_t* seed_ptr, uint32_t site, float p_drop) {
  MRB_REQUIRE(M > 0 && C > 0 && R > 0 && R <= 32 && (R % 8) == 0, "lora_tn: bad shape (M=%d C=%d R=%d)", M, C, R);
  MRB_REQUIRE((C % 8) == 0 && (ldy % 8) == 0 && (ldu % 8) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)U % 16) == 0, "lora_tn: 16-B alignment");
  a.Y = (const bf16_t*)Y; a.U = (const bf16_t*)U; a.ldy = ldy; a.ldu = ldu; a.M = M; a.C = C; a.R = R;
  for (int j = 0; j < 4; ++j) {
    if (j < R / 8) a.seg[j] = TnSeg{outs[j], col0 ? col0[j] : 0, ncols ? ncols[j] : C, lds[j]};
    else a.seg[j] = TnSeg{nullptr, 0, 0, 0};
  }
  mk_drop_arg(a.drop, seed_ptr, site, p_drop);
  return MRBLIP_OK;
}

extern "C" int mrblip_lora_tn(const void* Y, long long ldy, const void* U, long long ldu, int M, int C, int R, float* const* outs,
                              const int* col0, const int* ncols, const long long* lds, const uint32_t* seed_ptr, uint32_t site, float p_drop,
                              hipStream_t stream) {
  TnArgs2 q = {};
  if (int e = tn_fill(q.a, Y, ldy, U, ldu, M, C, R, outs, col0, ncols, lds, seed_ptr, site, p_drop)) return e;
  q.b = q.a;
  q.has_b = 0;
  return launch_tn(q, stream);
}

// The weight-gradient pairs of SEVERAL fused groups in ONE launch (round 5): jobs[i] is what one mrblip_lora_grads call took (same
// arithmetic, same block -> output ownership, so the same bits); at most 8 jobs (16 problems).
struct MrblipLoraGradsJob {   // (= the typedef of include/mrblip_hip.h)
  const void* dY; long long lddy; const void* U; long long ldu; const void* X; long long ldx; const void* G; long long ldg;
  int M, N, K, R;
  float* dBt[4]; int b_col0[4]; int b_ncols[4]; long long b_lds[4];
  float* dA[4]; long long a_lds[4];
  uint32_t site; float p_drop;
};
extern "C" int mrblip_lora_grads_batched(int n_jobs, const MrblipLoraGradsJob* jobs, const uint32_t* seed_ptr, hipStream_t stream) {
  MRB_REQUIRE(n_jobs > 0 && 2 * n_jobs <= TN_MAX_PROBLEMS && jobs, "lora_grads_batched: 1..%d jobs", TN_MAX_PROBLEMS / 2);
  TnMulti q = {};
  int blocks = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const MrblipLoraGradsJob& j = jobs[i];
    if (int e = tn_fill(q.p[2 * i], j.dY, j.lddy, j.U, j.ldu, j.M, j.N, j.R, j.dBt, j.b_col0, j.b_ncols, j.b_lds, nullptr, 0, 0.f)) return e;
    if (int e = tn_fill(q.p[2 * i + 1], j.X, j.ldx, j.G, j.ldg, j.M, j.K, j.R, j.dA, nullptr, nullptr, j.a_lds, seed_ptr, j.site, j.p_drop)) return e;
  }
  q.n = 2 * n_jobs;
  const TnArgs* ps[TN_MAX_PROBLEMS];
  for (int i = 0; i < q.n; ++i) ps[i] = &q.p[i];
  const int ct = tn_ct(ps, q.n, true);
  for (int i = 0; i < q.n; ++i) {
    blocks += (q.p[i].C + 32 * ct - 1) / (32 * ct);
    q.blk_end[i] = blocks;
  }
  if (ct == 2) hipLaunchKernelGGL(lora_tn_multi_kernel<2>, dim3(blocks), dim3(512), 0, stream, q);
  else hipLaunchKernelGGL(lora_tn_multi_kernel<1>, dim3(blocks), dim3(512), 0, stream, q);
  return mrblip_check_launch("lora_tn_multi");
}

// Both LoRA weight gradients of a fused group in ONE launch:
//   dBt_j[r, c - col0_j] += sum_m U[m, 8j + r] dY[m, c]   (c in adapter j's output columns)      peft lora_B.weight.grad^T
//   dA_j[r, k]           += sum_m G[m, 8j + r] dropout(X)[m, k]                                   peft lora_A.weight.grad
extern "C" int mrblip_lora_grads(const void* dY, long long lddy, const void* U, long long ldu, const void* X, long long ldx, const void* G,
                                 long long ldg, int M, int N, int K, int R, float* const* dBt, const int* b_col0, const int* b_ncols,
                                 const long long* b_lds, float* const* dA, const long long* a_lds, const uint32_t* seed_ptr, uint32_t site,
                                 float p_drop, hipStream_t stream) {
  TnArgs2 q = {};
  if (int e = tn_fill(q.a, dY, lddy, U, ldu, M, N, R, dBt, b_col0, b_ncols, b_lds, nullptr, 0, 0.f)) return e;
  if (int e = tn_fill(q.b, X, ldx, G, ldg, M, K, R, dA, nullptr, nullptr, a_lds, seed_ptr, site, p_drop)) return e;
  q.has_b = 1;
  return launch_tn(q, stream);
}
