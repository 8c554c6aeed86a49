// The LoRA "down" product of TALL inputs on the matrix cores, as a body two kernels share (round 4):
//   * lora_thin_kernel (lora.hip): a launch of its own, 8 waves per 16 rows;
//   * the thin ROLE of the generic tile GEMM (gemm.hip): the first workgroups of the GEMM that consumes u compute it while the tiles
//     already run their K loops, and publish it row block by row block (write-through stores + a flag) — the launch of its own, 8-15 us
//     of the T5 encoder's forward per projection with the chip to itself, disappears.
// A block owns ROWS (16, or 8) rows; K is cut into EIGHT shares whatever the number of waves: with 8 or more waves wave s takes share s,
// with 4 waves wave w takes shares w and w + 4 one after the other.  The eight partial accumulators meet in LDS and are added in share
// order, so both users produce the same bits.
//   u[m, 0:R] = (sum_k mask(m, k) x[m, k] A[r, k]) / (1 - p)        x, A bf16; fp32 accumulation; u bf16
// Every wave issues a whole batch of k-steps' operands as 16-B loads straight from global memory into v_mfma_f32_16x16x32_bf16 fragments
// (x: lane (m = l & 15, kg = l >> 4) holds 8 consecutive k of row m; A: the same lane map over r — A is 32..96 KB and L2 resident);
// nothing is staged in LDS, every x byte is fetched once.
#pragma once
#include "common.h"

typedef uint32_t thin_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t thin_u32x2 __attribute__((ext_vector_type(2)));

struct ThinArgs {
  const bf16_t* X; long long ldx;   // [M, K]
  const bf16_t* A; long long lda;   // [R, K]
  bf16_t* U; long long ldu;         // [M, >= R]
  int M, K, R;
  const uint32_t* seed_ptr; uint32_t site, thresh16; float inv_keep;   // seed_ptr == nullptr: no mask
};

#define THIN_RED_BYTES(NT) (8 * (NT) * 64 * 16)

// NT = 16-wide r tiles (R <= 16 NT), UB = k-steps (of 32) per batch, NWAVES = waves of the calling block (all of them must call),
// PUBLISH: the u rows leave with write-through (sc1) stores that the storing wave drains — the caller may then raise a flag that blocks on
// OTHER XCDs poll.  side_job(): runs between the main loop and the reduction (the stand-alone kernel's fp32 init rows).
template <int NT, int UB, int ROWS, int NWAVES, bool PUBLISH, class F>
__device__ __forceinline__ void lora_thin_body(const ThinArgs& p, int m0, f32x4 (*red)[NT][64], int w, int lane, F side_job) {
  const int l15 = lane & 15, kg = lane >> 4;
  const int row = m0 + l15;
  const bool row_ok = l15 < ROWS && row < p.M;
  const bool has_drop = p.seed_ptr != nullptr;
  const uint32_t seed = has_drop ? mrb_seed_load(p.seed_ptr) : 0u;
  // bounds-checked operands: rows >= M of X and rows >= R of A lie beyond the last byte of their resource and read as zero
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X), 0, (int)((((long long)p.M - 1) * p.ldx + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)((((long long)p.R - 1) * p.lda + p.K) * 2), 0x00020000);
  const uint32_t xoff = (uint32_t)(((long long)row * p.ldx + kg * 8) * 2);
  uint32_t aoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) aoff[t] = (uint32_t)(((long long)(t * 16 + l15) * p.lda + kg * 8) * 2);
  const int nks = p.K >> 5;                       // k-steps of 32 (K % 32 == 0)
  const int per = (nks + 7) >> 3;
#pragma unroll 1
  for (int sh = w; sh < 8; sh += NWAVES) {
    const int ks0 = sh * per, ks1 = min(nks, ks0 + per);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    thin_u32x4 xf[2][UB], af[2][UB][NT];
    auto fetch = [&](int buf, int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const bool k_ok = ks + u < ks1;               // past the share: an out-of-range offset -> zeros, no memory traffic
        const uint32_t kb = (uint32_t)(ks + u) * 64u;
        xf[buf][u] = __builtin_amdgcn_raw_buffer_load_b128(rx, (k_ok && row_ok) ? xoff + kb : 0xfffffff0u, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) af[buf][u][t] = __builtin_amdgcn_raw_buffer_load_b128(ra, k_ok ? aoff[t] + kb : 0xfffffff0u, 0, 0);
      }
    };
    auto consume = [&](int buf, int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        thin_u32x4 x = xf[buf][u];
        if (has_drop) {  // wave-uniform
          const uint32_t e = (uint32_t)row * (uint32_t)p.K + (uint32_t)((ks + u) * 32 + kg * 8);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bool k0, k1;
            mrb_keep2(e + 2 * q, seed, p.site, p.thresh16, k0, k1);
            x[q] = (k0 ? x[q] & 0xffffu : 0u) | (k1 ? x[q] & 0xffff0000u : 0u);
          }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[buf][u][t]), __builtin_bit_cast(bf16x8, x), acc[t], 0, 0, 0);
      }
    };
    // two batches of operands in flight: batch b+1 is requested before batch b is multiplied
    // (the fetches are UNCONDITIONAL — past the share they read out of range, which costs no memory traffic: a conditional fetch makes the
    // compiler merge "loaded" and "not loaded" register sets with copies, i.e. wait for every load right after issuing it)
    fetch(0, ks0);
#pragma unroll 1
    for (int ks = ks0; ks < ks1; ks += 2 * UB) {
      fetch(1, ks + UB);
      consume(0, ks);
      fetch(0, ks + 2 * UB);
      if (ks + UB < ks1) consume(1, ks + UB);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) red[sh][t][lane] = acc[t];
  }
  side_job();
  __syncthreads();
  if (w < NT) {  // wave t finishes r tile t: lane (m = l15, kg) holds r = 16 t + 4 kg .. + 3 of row m
    f32x4 v = red[0][w][lane];
#pragma unroll
    for (int j = 1; j < 8; ++j) v += red[j][w][lane];
    const float post = has_drop ? p.inv_keep : 1.0f;
    const int r0 = w * 16 + 4 * kg;
    if (row_ok && r0 < p.R) {
      const thin_u32x2 o = {pack2bf(v[0] * post, v[1] * post), pack2bf(v[2] * post, v[3] * post)};
      if constexpr (PUBLISH) {
        const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(p.U, 0, (int)(((long long)p.M * p.ldu) * 2), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b64(o, ru, (uint32_t)(((long long)row * p.ldu + r0) * 2), 0, 16 /* sc1: write-through */);
      } else {
        *reinterpret_cast<thin_u32x2*>(p.U + (long long)row * p.ldu + r0) = o;
      }
    }
    if constexpr (PUBLISH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the storing wave drains its stores before the caller's flag
  }
}
