// Q-Former query branch, ONE launch per layer (gfx950): a workgroup owns one frame — its 32 query tokens — for the whole layer.
//
// Replaces the launch chain of one BertLayer's query branch (Qformer.py:111-289 self-attention + BertSelfOutput, 402-474 the layer with
// its cross-attention every second layer, 349-375 intermediate_query / output_query): qkv GEMM, 32 x 32 self-attention, output dense +
// dropout + residual + LayerNorm, [query GEMM, cross-attention over the frame's 257 image tokens, output dense + LayerNorm], FFN 768 ->
// 3072 -> 768 with erf-GELU, dropout, residual, LayerNorm — 7 (11 with cross-attention) launches of M = 60 x 32 = 1920-row GEMMs at 0.10
// of the MFMA roof (profiles/r05_layer_timeline.txt: 121 launches, 2.7 ms of kernels in 3.5 ms of wall per step).
//
// Why one workgroup per frame works: every query-side operation is row-wise (dense, LayerNorm) or confined to one frame's tokens
// (attention), so a frame never needs another frame's data — no grid-wide hand-over between the sub-layers, only workgroup barriers.
// The frame's state is tiny (32 x 768), the weights are what moves: 14.2 MB per layer (16.6 MB with cross-attention), streamed by every
// workgroup from L2 (frames on one XCD walk the same weight rows at the same pace) into a wave-private LDS ring by LDS-DMA.  The launch
// occupies F CUs (60 at QVH) and leaves the rest of the chip to the look-ahead ViT that runs beside it.
//
// MEASURED (profiles/r06_qformer_fused.txt): correct on its first run, and SLOWER than the chain — 4.8 ms per 12-layer forward against 2.0 ms,
// for 8, 60 or 120 frames alike: one workgroup pulls all of a layer's weights through ONE CU, and one CU draws ~43 GB/s through LDS-DMA with
// the 48 KB this ring keeps in flight (27 GB/s with fragment-shaped loads straight into registers, QF_W_DMA=0).  The engine therefore keeps
// the launch chain (MRB_QF_FUSED=1 selects this kernel); the file stays as the measured alternative, tested against the chain.
//
// Layout.  All GEMMs run "transposed": Y^T[n][m] = W[n][:] . X^T[:][m] with the WEIGHT rows as the MFMA A operand and the 32 tokens as
// the B operand (32x32x16 bf16), so a lane owns ONE token m = lane & 31 and its 16 accumulator registers of a 32-feature tile hold
// features 8 (r >> 2) + 4 hi + (r & 3): LayerNorm statistics, biases, residual adds and softmax are in-lane (+ one lane ^ 32 exchange),
// and the fp32 residual stream is re-read by every epilogue in exactly the accumulator layout of the N = 768 products (wave w owns
// features [192 w, 192 w + 192); a lane reads and writes only its own elements).  The token operand X lives in LDS as bf16 [32][768] (row pitch 1552 B: conflict-free
// 16-B fragment reads).  Self-attention never leaves the registers: with the projection's accumulators packed to bf16 in register
// order, Q and K are both "lane = token, 8 packed head dims" fragments with the SAME dim permutation — all the MFMA contraction needs —
// and V is produced by the operand-swapped MFMA (lane = head dim, registers = tokens), which is the A operand of O^T = V^T P^T with the
// key permutation the packed probabilities carry.  Cross-attention reads K rows and V^T rows of the frame from HBM with that permutation.
//
// Numerics: the same rounding points as the launch chain (bf16 q / k / v / probabilities / attention output / GELU output, fp32
// accumulation, fp32 two-pass LayerNorm, erf-GELU of common.h) and the same dropout draws (element dropout: mrb_keep4 on row * 768 + n;
// probability dropout: draws v3, attention.hip) — outputs agree with the chain to fp32 summation order (tests/test_qformer_fused_gpu.py).
#include "common.h"

struct QfLayerArgs {   // kernel-side form of mrblip_qformer_layer (include/mrblip_hip.h)
  const bf16_t *qkv_w, *so_w, *cq_w, *co_w, *i_w, *o_w;            // bf16 [N, K] row-major (ld = K): 2304x768, 768x768, 768x768, 768x768, 3072x768, 768x3072
  const float *qkv_b, *so_b, *s_lnw, *s_lnb, *cq_b, *co_b, *c_lnw, *c_lnb, *i_b, *o_b, *o_lnw, *o_lnb;
  const float* x_in;      // fp32 [F * 32, 768]: the layer's input (LayerNorm output of the layer below)
  float* x_out;           // fp32 [F * 32, 768]
  bf16_t* xb_out;         // bf16 [F * 32, ldxb]  (optional)
  long long ldxb;
  bf16_t* qkv;            // saved for the backward: bf16 [F * 32, 2304]
  bf16_t* o;              //                        bf16 [F * 32, ldo] attention output (pre output-dense)
  long long ldo;
  float* lse;             //                        fp32 [F, 12, 32]
  float* y;               //                        fp32 [F * 32, 768] pre-LayerNorm sum of the self-attention block
  bf16_t* qc;             // cross-attention (has_cross): bf16 [F * 32, 768]
  bf16_t* oc;             //                        bf16 [F * 32, ldo]
  float* lsec;            //                        fp32 [F, 12, 32]
  float* y2;              //                        fp32 [F * 32, 768]
  const bf16_t* kv;       // bf16 [F * Tv, 1536]: K | V projections of the frame's image tokens
  const bf16_t* vt;       // bf16 [F, 12, 64, Tvp]: V^T, zero padded to Tvp (a multiple of 32)
  bf16_t* hpre;           // bf16 [F * 32, 3072] pre-GELU (for gelu_bwd)
  float* y3;              // fp32 [F * 32, 768]
  int F, Tv, Tvp, has_cross;
  DropoutArg d_sattn, d_so, d_cattn, d_co, d_ffn;
  float eps;
};

namespace qf {
constexpr int D = 768, H = 12, NQ = 32, DI = 3072;
constexpr int XSTR = D * 2 + 16;           // LDS row pitch of the token operand (bytes): 388 dwords = 4 mod 32 -> 8 rows cover all banks with 16-B reads
constexpr int HC = 512;                    // FFN chunk (intermediate features per pass)
constexpr int HSTR = HC * 2 + 16;
constexpr int XB_BYTES = NQ * XSTR;        // 49664
constexpr int HB_BYTES = NQ * HSTR;        // 33280
constexpr int RED_BYTES = 2 * 4 * 32 * 4;
constexpr int NSTG = 4, TILE_BYTES = 4096; // W ring of the LDS-DMA form: 4 tiles of [32 rows][64 k] per wave
constexpr int W_BYTES = 4 * NSTG * TILE_BYTES;
constexpr int LDS_BYTES = XB_BYTES + HB_BYTES + RED_BYTES + W_BYTES;   // 149504
}  // namespace qf

typedef __attribute__((address_space(3))) void* qf_lds_ptr_t;

__device__ __forceinline__ bf16x8 qf_pack8(const f32x16& a, int r0) {
  union { bf16x8 v8; uint32_t u[4]; } r;
  r.u[0] = pack2bf(a[r0], a[r0 + 1]); r.u[1] = pack2bf(a[r0 + 2], a[r0 + 3]);
  r.u[2] = pack2bf(a[r0 + 4], a[r0 + 5]); r.u[3] = pack2bf(a[r0 + 6], a[r0 + 7]);
  return r.v8;
}
__device__ __forceinline__ float qf_sum_x32(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
}
__device__ __forceinline__ float qf_max_x32(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}
__device__ __forceinline__ uint32_t qf_rot8(uint32_t h) { return __builtin_amdgcn_alignbit(h, h, 8); }
__device__ __forceinline__ uint32_t qf_draw(uint32_t hash, int j) {   // attention.hip draws v3
  const uint32_t w = (j & 2) ? qf_rot8(hash) : hash;
  return (j & 1) ? (w >> 16) : (w & 0xffffu);
}

// One wave's product  acc[t] (+)= W[rows of tile t][k0 .. k0 + 64 nkc) . X^T  for NT 32-feature tiles, K walked in 64-wide chunks.
// The weights go from L2 / HBM STRAIGHT into registers: lane (n, hi) of a tile reads the 64 contiguous bytes W[n][64 kc + 32 hi .. + 31]
// as four 16-B loads — the A operands of the chunk's four MFMA steps under the k permutation "step s of lane half hi = k 32 hi + 8 s .. + 7",
// which the token operand's LDS reads follow (any permutation serves a contraction as long as both operands use it).  A ring of
// NT * KCB tiles (16 registers each) is kept in flight per wave: 12 tiles = 48 KB per wave, 192 KB per CU — what hides the ~1 us of an L2
// miss (the first version staged the tiles through a 4-deep LDS-DMA ring per wave, 48 KB in flight per CU: 380 us per layer, latency
// bound; profiles/r06_qformer_fused.txt).  The compiler counts the loads (they return in order) and waits with vmcnt(N) per tile.
// row0(t): first weight row of tile t.  SWAPMASK bit t: operand-swapped MFMA for tile t (lane = feature, registers = tokens).
template <int NT, int SWAPMASK, int KCB, typename RowFn>
__device__ __forceinline__ void qf_wave_gemm(f32x16 (&acc)[NT], const char* xl, int xstr, int nkc, const bf16_t* W, long long ldw, long long k0,
                                             uint32_t w_bytes, RowFn row0, int lane) {
  typedef uint32_t qf_u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, (int)w_bytes, 0x00020000);
  const int l31 = lane & 31, hi = lane >> 5;
  uint32_t voff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) voff[t] = (uint32_t)((((long long)row0(t) + l31) * ldw + k0 + 32 * hi) * 2);
  constexpr int DEPTH = NT * KCB;
  bf16x8 ring[DEPTH][4];
  auto load = [&](bf16x8 (&f)[4], int kc, int t) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
      f[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[t] + 16 * s, (uint32_t)(128 * kc), 0));
  };
  const char* xrow = xl + l31 * xstr + 64 * hi;
  auto chunk = [&](int kc, int j, bool more) __attribute__((always_inline)) {
    bf16x8 xf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xrow + 128 * kc + 16 * s);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if ((SWAPMASK >> t) & 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[s], ring[j * NT + t][s], acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[j * NT + t][s], xf[s], acc[t], 0, 0, 0);
      }
      if (more) load(ring[j * NT + t], kc + KCB, t);
    }
  };
#pragma unroll
  for (int j = 0; j < KCB; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t) load(ring[j * NT + t], j, t);
  int kb = 0;
#pragma unroll 1
  for (; kb + KCB < nkc; kb += KCB) {
#pragma unroll
    for (int j = 0; j < KCB; ++j) chunk(kb + j, j, true);
  }
#pragma unroll
  for (int j = 0; j < KCB; ++j) chunk(kb + j, j, false);
}

// The LDS-DMA form of the same product (QF_W_DMA, the default — measured 4.9 ms per 12-layer forward against 7.3 ms for the register ring
// above: the fragment-shaped register loads take the texture-address path lane by lane, profiles/r06_qformer_fused.txt).  The wave streams
// ITS weight rows through its private ring of NSTG [32][64] tiles: 8 rows x 128 B per DMA instruction (whole cache lines), 16-B pieces
// XOR-swizzled by the row through the SOURCE address; three tiles in flight, counted vmcnt waits (LDS-DMA loads retire in order), the
// fragments of tile i + 1 are read while the MFMAs of tile i run.  Plain k order (step s of lane half hi = k 16 s + 8 hi .. + 7).
template <int NT, int SWAPMASK, typename RowFn>
__device__ __forceinline__ void qf_wave_gemm_dma(f32x16 (&acc)[NT], const char* xl, int xstr, int nkc, const bf16_t* W, long long ldw, long long k0,
                                                 uint32_t w_bytes, RowFn row0, char* ring, int lane) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(W), 0, (int)w_bytes, 0x00020000);
  const int r8 = lane >> 3, c8 = lane & 7, l31 = lane & 31, hi = lane >> 5;
  uint32_t voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) voff[j] = (uint32_t)((long long)(8 * j + r8) * ldw * 2) + (uint32_t)((c8 ^ r8) * 16);
  const int T = nkc * NT;
  auto issue = [&](int idx) __attribute__((always_inline)) {
    const int kc = idx / NT, t = idx - kc * NT;
    const uint32_t soff = (uint32_t)(((long long)row0(t) * ldw + k0 + 64 * kc) * 2);
    char* buf = ring + (idx & (qf::NSTG - 1)) * qf::TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (qf_lds_ptr_t)(buf + j * 1024), 16, voff[j], soff, 0, 0);
  };
  int foff[4];   // fragment read offsets inside a tile: row l31, piece 2 s + hi at slot piece ^ (row & 7)
#pragma unroll
  for (int s = 0; s < 4; ++s) foff[s] = l31 * 128 + (((2 * s + hi) ^ (l31 & 7)) << 4);
  auto rdw = [&](bf16x8 (&f)[4], int idx) __attribute__((always_inline)) {
    const char* buf = ring + (idx & (qf::NSTG - 1)) * qf::TILE_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const bf16x8*>(buf + foff[s]);
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing but this product's DMA may be counted below
  issue(0);
  issue(1);
  issue(2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  bf16x8 wc[4], wn[4], xf[4];
  rdw(wc, 0);
  const char* xrow = xl + l31 * xstr + 16 * hi;
  for (int kc = 0; kc < nkc; ++kc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) xf[s] = *reinterpret_cast<const bf16x8*>(xrow + (64 * kc + 16 * s) * 2);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int idx = kc * NT + t;
      if (idx + 3 < T) issue(idx + 3);   // into the slot of tile idx - 1, whose fragments the previous trip's MFMAs consumed
      if (idx + 1 < T) {                 // tile idx + 1 must have landed: the tiles behind it may still fly
        const int behind = min(idx + 3, T - 1) - (idx + 1);
        if (behind >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rdw(wn, idx + 1);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if ((SWAPMASK >> t) & 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[s], wc[s], acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[s], xf[s], acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) wc[s] = wn[s];
    }
  }
}

#ifndef QF_W_DMA
#define QF_W_DMA 1
#endif
template <int NT, int SWAPMASK, int KCB, typename RowFn>
__device__ __forceinline__ void qf_gemm(f32x16 (&acc)[NT], const char* xl, int xstr, int nkc, const bf16_t* W, long long ldw, long long k0, uint32_t w_bytes,
                                        RowFn row0, char* ring, int lane) {
#if QF_W_DMA
  qf_wave_gemm_dma<NT, SWAPMASK>(acc, xl, xstr, nkc, W, ldw, k0, w_bytes, row0, ring, lane);
#else
  qf_wave_gemm<NT, SWAPMASK, KCB>(acc, xl, xstr, nkc, W, ldw, k0, w_bytes, row0, lane);
#endif
}

__device__ __forceinline__ void qf_zero(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// acc (+ bias) -> dropout -> + residual (fp32, global) -> y (stored: the backward's LayerNorm input) -> LayerNorm -> fp32 to xdst, bf16 to
// the LDS token operand (after a barrier: every wave has finished reading the operand this overwrites) and optionally to xb_out.
// Lane (m, hi), tile t, register r = 4 j + i  <->  feature nb + 32 t + 8 j + 4 hi + i.  xres / xdst may be the same buffer: a lane reads
// and writes only its own elements.
__device__ __forceinline__ void qf_block_epilogue(f32x16 (&acc)[6], int nb, int row, int hi, int l31, int w, const float* bias, const DropoutArg& dr,
                                                  const float* xres, float* ysave, const float* lnw, const float* lnb, float eps, float* red, char* xl,
                                                  float* xdst, bf16_t* xb_out, long long ldxb) {
  const uint32_t seed = dr.seed_ptr ? mrb_seed_load(dr.seed_ptr) : 0u;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nb + 32 * t + 8 * j + 4 * hi;
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      const float4 xr = *reinterpret_cast<const float4*>(xres + (long long)row * qf::D + n);
      float v0 = acc[t][4 * j] + b.x, v1 = acc[t][4 * j + 1] + b.y, v2 = acc[t][4 * j + 2] + b.z, v3 = acc[t][4 * j + 3] + b.w;
      if (dr.seed_ptr) {
        bool k0, k1, k2, k3;
        mrb_keep4((uint32_t)row * (uint32_t)qf::D + (uint32_t)n, seed, dr.site, dr.thresh24, k0, k1, k2, k3);
        v0 = k0 ? v0 * dr.inv_keep : 0.f; v1 = k1 ? v1 * dr.inv_keep : 0.f;
        v2 = k2 ? v2 * dr.inv_keep : 0.f; v3 = k3 ? v3 * dr.inv_keep : 0.f;
      }
      v0 += xr.x; v1 += xr.y; v2 += xr.z; v3 += xr.w;
      acc[t][4 * j] = v0; acc[t][4 * j + 1] = v1; acc[t][4 * j + 2] = v2; acc[t][4 * j + 3] = v3;
      *reinterpret_cast<float4*>(ysave + (long long)row * qf::D + n) = make_float4(v0, v1, v2, v3);
      s += (v0 + v1) + (v2 + v3);
    }
  // two-pass LayerNorm statistics over the row's 768 features: in-lane, lane ^ 32, then the four waves through LDS
  s = qf_sum_x32(s);
  if (hi == 0) red[w * 32 + l31] = s;
  __syncthreads();
  const float mean = (red[l31] + red[32 + l31] + red[64 + l31] + red[96 + l31]) * (1.0f / qf::D);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = acc[t][r] - mean;
      q += d * d;
    }
  q = qf_sum_x32(q);
  if (hi == 0) red[128 + w * 32 + l31] = q;
  __syncthreads();
  const float var = (red[128 + l31] + red[160 + l31] + red[192 + l31] + red[224 + l31]) * (1.0f / qf::D);
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nb + 32 * t + 8 * j + 4 * hi;
      const float4 g = *reinterpret_cast<const float4*>(lnw + n), b = *reinterpret_cast<const float4*>(lnb + n);
      const float v0 = (acc[t][4 * j] - mean) * rstd * g.x + b.x, v1 = (acc[t][4 * j + 1] - mean) * rstd * g.y + b.y;
      const float v2 = (acc[t][4 * j + 2] - mean) * rstd * g.z + b.z, v3 = (acc[t][4 * j + 3] - mean) * rstd * g.w + b.w;
      const uint2 pk = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
      *reinterpret_cast<uint2*>(xl + l31 * qf::XSTR + n * 2) = pk;
      *reinterpret_cast<float4*>(xdst + (long long)row * qf::D + n) = make_float4(v0, v1, v2, v3);
      if (xb_out) *reinterpret_cast<uint2*>(xb_out + (long long)row * ldxb + n) = pk;
    }
  __syncthreads();   // the new operand is complete
}

// One head's attention for the wave's 32 queries (lane = query): flash-style walk over nkt 32-key tiles with the running maximum in the
// log2 domain, lazy rescale, dropout draws v3 on the probabilities (attention.hip) and the row sum of the UNdropped probabilities.
// S^T tile: lane (query, hi), register r <-> key 32 kt + 8 (r >> 2) + 4 hi + (r & 3); the packed probabilities (registers 8 s .. 8 s + 7)
// are the B operand of O^T += V^T P^T.  kfrag(kt, s): A operand of the score product (32 keys x the 16 head dims of step s, in the dim
// permutation the packed q carries); vfrag(kt, t, s): A operand of the output product (head dims 32 t .. + 31 x the 16 keys of step s).
template <typename KF, typename VF>
__device__ __forceinline__ void qf_attend(int nkt, int nkeys, const bf16x8 (&qp)[4], KF&& kfrag, VF&& vfrag, float scale2, int hi, uint32_t row_id,
                                          const DropoutArg& dr, f32x16 (&oa)[2], float& m_out, float& l_out) {
  const uint32_t seed = dr.seed_ptr ? mrb_seed_load(dr.seed_ptr) : 0u;
  const uint32_t skq = (uint32_t)((nkeys + 3) >> 2);
  const uint32_t tl = dr.seed_ptr ? (row_id * skq + (uint32_t)hi) * MRB_H1 + mrb_lin_base(seed, dr.site) : 0u;
  float m_run = -1.0e30f, l_run = 0.f;
  qf_zero(oa[0]);
  qf_zero(oa[1]);
  bf16x8 kc[4], kn[4], vc[2][2], vn[2][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) kc[s] = kn[s] = kfrag(0, s);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) vc[t][s] = vn[t][s] = vfrag(0, t, s);
#pragma unroll 1
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) {   // the next tile's operands fly while this one is computed
#pragma unroll
      for (int s = 0; s < 4; ++s) kn[s] = kfrag(kt + 1, s);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) vn[t][s] = vfrag(kt + 1, t, s);
    }
    f32x16 sc;
    qf_zero(sc);
#pragma unroll
    for (int s = 0; s < 4; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc[s], qp[s], sc, 0, 0, 0);
    const bool edge = 32 * kt + 32 > nkeys;
    float mx = -1.0e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = sc[r] * scale2;
      if (edge) v = (32 * kt + 8 * (r >> 2) + 4 * hi + (r & 3)) < nkeys ? v : -__builtin_inff();
      sc[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = qf_max_x32(mx);
    if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oa[t][r] *= alpha;
    }
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc[r] = __builtin_amdgcn_exp2f(sc[r] - m_run);
      ps += sc[r];
    }
    l_run += ps;
    if (dr.seed_ptr) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // keys 32 kt + 8 j + 4 hi + i: quad index 8 kt + 2 j + hi, draw i
        const uint32_t hsh = mrb_lin_fin24(tl + (uint32_t)(8 * kt + 2 * j) * MRB_H1);
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[4 * j + i] = qf_draw(hsh, i) >= dr.thresh24 ? sc[4 * j + i] : 0.f;
      }
    }
    const bf16x8 p0 = qf_pack8(sc, 0), p1 = qf_pack8(sc, 8);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      oa[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc[t][0], p0, oa[t], 0, 0, 0);
      oa[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc[t][1], p1, oa[t], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) kc[s] = kn[s];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) vc[t][s] = vn[t][s];
  }
  const float l_tot = qf_sum_x32(l_run);
  const float inv = (dr.seed_ptr ? dr.inv_keep : 1.0f) / l_tot;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oa[t][r] *= inv;
  m_out = m_run;
  l_out = l_tot;
}

// the packed output of head hh goes to opk[hh] with STATIC indices (hh is a rolled loop counter: a dynamic index would send the array to scratch)
__device__ __forceinline__ void qf_keep_head(bf16x8 (&opk)[3][4], int hh, const f32x16 (&oa)[2]) {
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (hh == k) {
      opk[k][0] = qf_pack8(oa[0], 0); opk[k][1] = qf_pack8(oa[0], 8);
      opk[k][2] = qf_pack8(oa[1], 0); opk[k][3] = qf_pack8(oa[1], 8);
    }
}
// attention output of the wave's three heads -> the LDS token operand and the saved copy
__device__ __forceinline__ void qf_store_heads(const bf16x8 (&opk)[3][4], char* xl, bf16_t* og, long long ldo, int row, int l31, int hi, int w) {
#pragma unroll
  for (int hh = 0; hh < 3; ++hh)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = 64 * (3 * w + hh) + 32 * t + 8 * j + 4 * hi;
        union { bf16x8 v8; uint2 u2[2]; } u;
        u.v8 = opk[hh][2 * t + (j >> 1)];
        *reinterpret_cast<uint2*>(xl + l31 * qf::XSTR + n * 2) = u.u2[j & 1];
        *reinterpret_cast<uint2*>(og + (long long)row * ldo + n) = u.u2[j & 1];
      }
}

__global__ __launch_bounds__(256, 1) void qformer_layer_fwd_kernel(const QfLayerArgs p) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* xl = sm;                                   // bf16 [32][768] token operand (LayerNorm output / attention output)
  char* hl = sm + qf::XB_BYTES;                    // bf16 [32][512] FFN chunk
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ring = sm + qf::XB_BYTES + qf::HB_BYTES + qf::RED_BYTES + w * (qf::NSTG * qf::TILE_BYTES);   // (QF_W_DMA)
  float* red = reinterpret_cast<float*>(sm + qf::XB_BYTES + qf::HB_BYTES);
  const int f = blockIdx.x;
  const int row = f * qf::NQ + l31;                // this lane's token (global row)
  const int nb = 192 * w;                          // this wave's feature slice of the N = 768 products
  const float scale2 = 0.125f * 1.4426950408889634f;

  // ---- the frame's hidden state: bf16 token operand in LDS (the fp32 residual is re-read from memory by the epilogues: a lane's own elements)
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nb + 32 * t + 8 * j + 4 * hi;
      const float4 v = *reinterpret_cast<const float4*>(p.x_in + (long long)row * qf::D + n);
      *reinterpret_cast<uint2*>(xl + l31 * qf::XSTR + n * 2) = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
    }
  __syncthreads();

  // ---- self-attention: wave w owns heads 3 w .. 3 w + 2
  bf16x8 opk[3][4];   // packed attention outputs of the wave's heads (lane = token; [t][jp]: head dims 32 t + 16 jp + 8 (e >> 2) + 4 hi + (e & 3))
#pragma unroll 1
  for (int hh = 0; hh < 3; ++hh) {
    const int h = 3 * w + hh;
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) qf_zero(acc[t]);
    // tiles 0, 1: q_h   2, 3: k_h   4, 5: v_h (operand-swapped: lane = head dim, registers = tokens)
    qf_gemm<6, 0x30, 2>(acc, xl, qf::XSTR, qf::D / 64, p.qkv_w, qf::D, 0, 3u * qf::D * qf::D * 2u,
                             [&](int t) { return (t >> 1) * qf::D + 64 * h + 32 * (t & 1); }, ring, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = (t >> 1) * qf::D + 64 * h + 32 * (t & 1) + 8 * j + 4 * hi;
        const float4 b = *reinterpret_cast<const float4*>(p.qkv_b + n);
        acc[t][4 * j] += b.x; acc[t][4 * j + 1] += b.y; acc[t][4 * j + 2] += b.z; acc[t][4 * j + 3] += b.w;
        *reinterpret_cast<uint2*>(p.qkv + (long long)row * (3 * qf::D) + n) =
            make_uint2(pack2bf(acc[t][4 * j], acc[t][4 * j + 1]), pack2bf(acc[t][4 * j + 2], acc[t][4 * j + 3]));
      }
#pragma unroll
    for (int t = 4; t < 6; ++t) {
      const int n = 2 * qf::D + 64 * h + 32 * (t & 1) + l31;
      const float b = p.qkv_b[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[t][r] += b;
        const int m = 8 * (r >> 2) + 4 * hi + (r & 3);
        p.qkv[(long long)(f * qf::NQ + m) * (3 * qf::D) + n] = f2bf(acc[t][r]);
      }
    }
    // S^T[key][query] = K Q^T: four MFMA steps over the 64 head dims, step s = registers 8 (s & 1) .. + 7 of tile s >> 1 on both sides;
    // V^T (tiles 4, 5: lane = head dim, registers = tokens) is the A operand of O^T = V^T P^T as it stands
    bf16x8 qp[4], kp[4], vp[2][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qp[s] = qf_pack8(acc[s >> 1], 8 * (s & 1));
      kp[s] = qf_pack8(acc[2 + (s >> 1)], 8 * (s & 1));
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) vp[t][s] = qf_pack8(acc[4 + t], 8 * s);
    f32x16 oa[2];
    float mrun, ltot;
    qf_attend(1, qf::NQ, qp, [&](int, int s) { return kp[s]; }, [&](int, int t, int s) { return vp[t][s]; }, scale2, hi,
              (uint32_t)(f * qf::H + h) * qf::NQ + (uint32_t)l31, p.d_sattn, oa, mrun, ltot);
    qf_keep_head(opk, hh, oa);
    if (hi == 0) p.lse[(long long)(f * qf::H + h) * qf::NQ + l31] = mrun * 0.6931471805599453f + __logf(ltot);
  }
  __syncthreads();   // every wave is done with the LayerNorm'd operand: the attention output takes its place
  qf_store_heads(opk, xl, p.o, p.ldo, row, l31, hi, w);
  __syncthreads();
  {
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) qf_zero(acc[t]);
    qf_gemm<6, 0, 2>(acc, xl, qf::XSTR, qf::D / 64, p.so_w, qf::D, 0, (uint32_t)(qf::D * qf::D * 2), [&](int t) { return nb + 32 * t; }, ring, lane);
    qf_block_epilogue(acc, nb, row, hi, l31, w, p.so_b, p.d_so, p.x_in, p.y, p.s_lnw, p.s_lnb, p.eps, red, xl, p.x_out, nullptr, 0);
  }

  // ---- cross-attention over the frame's image tokens (every second layer)
  if (p.has_cross) {
    f32x16 qa[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) qf_zero(qa[t]);
    qf_gemm<6, 0, 2>(qa, xl, qf::XSTR, qf::D / 64, p.cq_w, qf::D, 0, (uint32_t)(qf::D * qf::D * 2), [&](int t) { return nb + 32 * t; }, ring, lane);
    bf16x8 qp[3][4];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = nb + 32 * t + 8 * j + 4 * hi;
        const float4 b = *reinterpret_cast<const float4*>(p.cq_b + n);
        qa[t][4 * j] += b.x; qa[t][4 * j + 1] += b.y; qa[t][4 * j + 2] += b.z; qa[t][4 * j + 3] += b.w;
        *reinterpret_cast<uint2*>(p.qc + (long long)row * qf::D + n) = make_uint2(pack2bf(qa[t][4 * j], qa[t][4 * j + 1]), pack2bf(qa[t][4 * j + 2], qa[t][4 * j + 3]));
      }
      qp[t >> 1][2 * (t & 1)] = qf_pack8(qa[t], 0);
      qp[t >> 1][2 * (t & 1) + 1] = qf_pack8(qa[t], 8);
    }
    const int nkt = (p.Tv + 31) >> 5;
#pragma unroll 1
    for (int hh = 0; hh < 3; ++hh) {
      const int h = 3 * w + hh;
      bf16x8 qh[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) qh[s] = hh == 0 ? qp[0][s] : hh == 1 ? qp[1][s] : qp[2][s];
      const bf16_t* kb = p.kv + (long long)f * p.Tv * (2 * qf::D) + 64 * h + 4 * hi;
      const bf16_t* vb = p.vt + ((long long)(f * qf::H + h) * 64 + l31) * p.Tvp + 4 * hi;
      const int tv1 = p.Tv - 1;
      const long long tvp = p.Tvp;
      f32x16 oa[2];
      float mrun, ltot;
      qf_attend(nkt, p.Tv, qh,
                [&](int kt, int s) {   // key row min(32 kt + lane, Tv - 1); head dims 16 s + 8 (e >> 2) + 4 hi + (e & 3): the permutation the packed q carries
                  const bf16_t* kr = kb + (long long)min(32 * kt + l31, tv1) * (2 * qf::D) + 16 * s;
                  union { bf16x8 v8; uint2 u2[2]; } kf;
                  kf.u2[0] = *reinterpret_cast<const uint2*>(kr);
                  kf.u2[1] = *reinterpret_cast<const uint2*>(kr + 8);
                  return kf.v8;
                },
                [&](int kt, int t, int s) {   // V^T row 32 t + lane; keys 32 kt + 16 s + 8 (e >> 2) + 4 hi + (e & 3)
                  const bf16_t* vr = vb + 32 * t * tvp + 32 * kt + 16 * s;
                  union { bf16x8 v8; uint2 u2[2]; } vf;
                  vf.u2[0] = *reinterpret_cast<const uint2*>(vr);
                  vf.u2[1] = *reinterpret_cast<const uint2*>(vr + 8);
                  return vf.v8;
                },
                scale2, hi, (uint32_t)(f * qf::H + h) * qf::NQ + (uint32_t)l31, p.d_cattn, oa, mrun, ltot);
      qf_keep_head(opk, hh, oa);
      if (hi == 0) p.lsec[(long long)(f * qf::H + h) * qf::NQ + l31] = mrun * 0.6931471805599453f + __logf(ltot);
    }
    __syncthreads();
    qf_store_heads(opk, xl, p.oc, p.ldo, row, l31, hi, w);
    __syncthreads();
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) qf_zero(acc[t]);
    qf_gemm<6, 0, 2>(acc, xl, qf::XSTR, qf::D / 64, p.co_w, qf::D, 0, (uint32_t)(qf::D * qf::D * 2), [&](int t) { return nb + 32 * t; }, ring, lane);
    qf_block_epilogue(acc, nb, row, hi, l31, w, p.co_b, p.d_co, p.x_out, p.y2, p.c_lnw, p.c_lnb, p.eps, red, xl, p.x_out, nullptr, 0);
  }

  // ---- FFN: six passes over 512 intermediate features; the GELU output of a pass is the token operand of the second product
  {
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) qf_zero(acc[t]);
#pragma unroll 1
    for (int c = 0; c < qf::DI / qf::HC; ++c) {
      f32x16 ha[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) qf_zero(ha[t]);
      const int hb = qf::HC * c + 128 * w;   // this wave's 128 intermediate features of the pass
      qf_gemm<4, 0, 2>(ha, xl, qf::XSTR, qf::D / 64, p.i_w, qf::D, 0, (uint32_t)((long long)qf::DI * qf::D * 2), [&](int t) { return hb + 32 * t; }, ring, lane);
      uint2 hp[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = hb + 32 * t + 8 * j + 4 * hi;
          const float4 b = *reinterpret_cast<const float4*>(p.i_b + n);
          float v0 = ha[t][4 * j] + b.x, v1 = ha[t][4 * j + 1] + b.y, v2 = ha[t][4 * j + 2] + b.z, v3 = ha[t][4 * j + 3] + b.w;
          *reinterpret_cast<uint2*>(p.hpre + (long long)row * qf::DI + n) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
          gelu_erf2(v0, v1);
          gelu_erf2(v2, v3);
          hp[t][j] = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
        }
      if (c > 0) __syncthreads();   // the previous pass's second product has read the chunk
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<uint2*>(hl + l31 * qf::HSTR + (128 * w + 32 * t + 8 * j + 4 * hi) * 2) = hp[t][j];
      __syncthreads();
      qf_gemm<6, 0, 2>(acc, hl, qf::HSTR, qf::HC / 64, p.o_w, qf::DI, (long long)qf::HC * c, (uint32_t)((long long)qf::D * qf::DI * 2),
                            [&](int t) { return nb + 32 * t; }, ring, lane);
    }
    qf_block_epilogue(acc, nb, row, hi, l31, w, p.o_b, p.d_ffn, p.x_out, p.y3, p.o_lnw, p.o_lnb, p.eps, red, xl, p.x_out, p.xb_out, p.ldxb);
  }
}

// the public argument block (include/mrblip_hip.h: mrblip_qformer_layer) — plain pointers and sizes
struct mrblip_qformer_layer {
  const void *qkv_w, *so_w, *cq_w, *co_w, *i_w, *o_w;
  const float *qkv_b, *so_b, *s_lnw, *s_lnb, *cq_b, *co_b, *c_lnw, *c_lnb, *i_b, *o_b, *o_lnw, *o_lnb;
  const float* x_in;
  float* x_out;
  void* xb_out;
  long long ldxb;
  void *qkv, *o;
  long long ldo;
  float *lse, *y;
  void *qc, *oc;
  float *lsec, *y2;
  const void *kv, *vt;
  void* hpre;
  float* y3;
  int F, Tv, Tvp, has_cross;
  const uint32_t* seed_ptr;
  float p_drop;
  uint32_t site_sattn, site_so, site_cattn, site_co, site_ffn;
  float eps;
};

static DropoutArg qf_drop(const uint32_t* seed_ptr, uint32_t site, float p) {
  DropoutArg d;
  const bool on = seed_ptr != nullptr && p > 0.f;
  d.seed_ptr = on ? seed_ptr : nullptr;
  d.site = site;
  d.thresh24 = (uint32_t)(p * 65536.0f + 0.5f);
  if (on && d.thresh24 < 1u) d.thresh24 = 1u;
  d.inv_keep = on ? 1.0f / (1.0f - p) : 1.0f;
  return d;
}

extern "C" int mrblip_qformer_layer_fwd(const mrblip_qformer_layer* a, hipStream_t stream) {
  MRB_REQUIRE(a != nullptr && a->F > 0, "qformer_layer_fwd: no frames");
  MRB_REQUIRE(a->qkv_w && a->so_w && a->i_w && a->o_w && a->qkv_b && a->so_b && a->s_lnw && a->s_lnb && a->i_b && a->o_b && a->o_lnw && a->o_lnb,
              "qformer_layer_fwd: missing weight");
  MRB_REQUIRE(a->x_in && a->x_out && a->qkv && a->o && a->lse && a->y && a->hpre && a->y3, "qformer_layer_fwd: missing buffer");
  MRB_REQUIRE(a->ldo >= qf::D && a->ldo % 4 == 0, "qformer_layer_fwd: ldo %lld", a->ldo);
  MRB_REQUIRE(!a->xb_out || (a->ldxb >= qf::D && a->ldxb % 4 == 0), "qformer_layer_fwd: ldxb %lld", a->ldxb);
  MRB_REQUIRE(a->p_drop >= 0.f && a->p_drop < 1.f, "qformer_layer_fwd: p_drop %f", a->p_drop);
  if (a->has_cross) {
    MRB_REQUIRE(a->qc && a->oc && a->lsec && a->y2 && a->kv && a->vt && a->cq_w && a->co_w && a->cq_b && a->co_b && a->c_lnw && a->c_lnb,
                "qformer_layer_fwd: missing cross-attention buffer");
    MRB_REQUIRE(a->Tv >= 1 && a->Tvp >= a->Tv && a->Tvp % 32 == 0, "qformer_layer_fwd: Tv %d / Tvp %d", a->Tv, a->Tvp);
  }
  QfLayerArgs k;
  k.qkv_w = (const bf16_t*)a->qkv_w; k.so_w = (const bf16_t*)a->so_w; k.cq_w = (const bf16_t*)a->cq_w; k.co_w = (const bf16_t*)a->co_w;
  k.i_w = (const bf16_t*)a->i_w; k.o_w = (const bf16_t*)a->o_w;
  k.qkv_b = a->qkv_b; k.so_b = a->so_b; k.s_lnw = a->s_lnw; k.s_lnb = a->s_lnb; k.cq_b = a->cq_b; k.co_b = a->co_b; k.c_lnw = a->c_lnw; k.c_lnb = a->c_lnb;
  k.i_b = a->i_b; k.o_b = a->o_b; k.o_lnw = a->o_lnw; k.o_lnb = a->o_lnb;
  k.x_in = a->x_in; k.x_out = a->x_out; k.xb_out = (bf16_t*)a->xb_out; k.ldxb = a->ldxb;
  k.qkv = (bf16_t*)a->qkv; k.o = (bf16_t*)a->o; k.ldo = a->ldo; k.lse = a->lse; k.y = a->y;
  k.qc = (bf16_t*)a->qc; k.oc = (bf16_t*)a->oc; k.lsec = a->lsec; k.y2 = a->y2; k.kv = (const bf16_t*)a->kv; k.vt = (const bf16_t*)a->vt;
  k.hpre = (bf16_t*)a->hpre; k.y3 = a->y3;
  k.F = a->F; k.Tv = a->Tv; k.Tvp = a->Tvp; k.has_cross = a->has_cross;
  k.d_sattn = qf_drop(a->seed_ptr, a->site_sattn, a->p_drop); k.d_so = qf_drop(a->seed_ptr, a->site_so, a->p_drop);
  k.d_cattn = qf_drop(a->seed_ptr, a->site_cattn, a->p_drop); k.d_co = qf_drop(a->seed_ptr, a->site_co, a->p_drop);
  k.d_ffn = qf_drop(a->seed_ptr, a->site_ffn, a->p_drop);
  k.eps = a->eps;
  // (idempotent; a thread-local "done" flag only skips the repeated driver call)
  static thread_local bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(qformer_layer_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, qf::LDS_BYTES) != hipSuccess) {
      mrblip_set_error("qformer_layer_fwd: cannot reserve %d bytes of LDS", qf::LDS_BYTES);
      return MRBLIP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(qformer_layer_fwd_kernel, dim3(a->F), dim3(256), qf::LDS_BYTES, stream, k);
  return mrblip_check_launch("qformer_layer_fwd");
}
