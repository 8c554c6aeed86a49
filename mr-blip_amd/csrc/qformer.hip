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
// Layout.  All GEMMs run "transposed": Y^T[n][m] = W[n][:] . X^T[:][m] with the WEIGHT rows as the MFMA A operand and the 32 tokens as
// the B operand (32x32x16 bf16), so a lane owns ONE token m = lane & 31 and its 16 accumulator registers of a 32-feature tile hold
// features 8 (r >> 2) + 4 hi + (r & 3): LayerNorm statistics, biases, residual adds and softmax are in-lane (+ one lane ^ 32 exchange),
// and the fp32 residual stream of the frame stays in registers in exactly the accumulator layout of the N = 768 products (wave w owns
// features [192 w, 192 w + 192): 96 registers).  The token operand X lives in LDS as bf16 [32][768] (row pitch 1552 B: conflict-free
// 16-B fragment reads).  Self-attention never leaves the registers: with the projection's accumulators packed to bf16 in register
// order, Q and K are both "lane = token, 8 packed head dims" fragments with the SAME dim permutation — all the MFMA contraction needs —
// and V is produced by the operand-swapped MFMA (lane = head dim, registers = tokens), which is the A operand of O^T = V^T P^T with the
// key permutation the packed probabilities carry.  Cross-attention reads K rows and V^T rows of the frame from HBM with that permutation.
//
// Numerics: the same rounding points as the launch chain (bf16 q / k / v / probabilities / attention output / GELU output, fp32
// accumulation, fp32 two-pass LayerNorm, erf-GELU of common.h) and the same dropout draws (element dropout: mrb_keep4 on row * 768 + n;
// probability dropout: draws v3, attention.hip) — outputs agree with the chain to fp32 summation order (tests/test_qformer_fused_gpu.py).
#include "common.h"

struct QfLayerArgs {   // mirrors mrblip_qformer_layer (include/mrblip_hip.h)
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
constexpr int NSTG = 4, TILE_BYTES = 4096; // W ring: 4 tiles of [32 rows][64 k] per wave
constexpr int W_BYTES = 4 * NSTG * TILE_BYTES;
constexpr int RED_BYTES = 2 * 4 * 32 * 4;
constexpr int LDS_BYTES = XB_BYTES + HB_BYTES + W_BYTES + RED_BYTES;   // 149504
constexpr int MAXKT = 9;                   // cross-attention: key tiles of 32 (Tv <= 288)
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
// The wave streams ITS weight rows through its private ring of NSTG [32][64] tiles: LDS-DMA, 8 rows x 128 B per instruction (whole
// cache lines), 16-B pieces XOR-swizzled by the row through the SOURCE address; three tiles in flight, counted vmcnt waits (LDS-DMA
// loads retire in order), fragments of tile i + 1 are read while the MFMAs of tile i run.  No workgroup barrier inside.
// row0(t): first weight row of tile t.  SWAPMASK bit t: operand-swapped MFMA for tile t (lane = feature, registers = tokens).
template <int NT, int SWAPMASK, typename RowFn>
__device__ __forceinline__ void qf_wave_gemm(f32x16 (&acc)[NT], const char* xl, int xstr, int nkc, const bf16_t* W, long long ldw, long long k0,
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

__device__ __forceinline__ void qf_zero(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// v[t][r] (+bias) -> dropout -> + residual xr -> y (stored) -> LayerNorm -> xr; then the bf16 copy goes to the LDS operand (after a
// barrier: every wave has finished reading the operand this overwrites) and optionally to x_out / xb_out.
// Lane (m, hi), tile t, register r = 4 j + i  <->  feature nb + 32 t + 8 j + 4 hi + i.
__device__ __forceinline__ void qf_block_epilogue(f32x16 (&acc)[6], float (&xr)[6][16], int nb, int row, int hi, int l31, int w, const float* bias,
                                                  const DropoutArg& dr, float* ysave, const float* lnw, const float* lnb, float eps, float* red, char* xl,
                                                  float* x_out, bf16_t* xb_out, long long ldxb) {
  const uint32_t seed = dr.seed_ptr ? mrb_seed_load(dr.seed_ptr) : 0u;
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nb + 32 * t + 8 * j + 4 * hi;
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      float v0 = acc[t][4 * j] + b.x, v1 = acc[t][4 * j + 1] + b.y, v2 = acc[t][4 * j + 2] + b.z, v3 = acc[t][4 * j + 3] + b.w;
      if (dr.seed_ptr) {
        bool k0, k1, k2, k3;
        mrb_keep4((uint32_t)row * (uint32_t)qf::D + (uint32_t)n, seed, dr.site, dr.thresh24, k0, k1, k2, k3);
        v0 = k0 ? v0 * dr.inv_keep : 0.f; v1 = k1 ? v1 * dr.inv_keep : 0.f;
        v2 = k2 ? v2 * dr.inv_keep : 0.f; v3 = k3 ? v3 * dr.inv_keep : 0.f;
      }
      v0 += xr[t][4 * j]; v1 += xr[t][4 * j + 1]; v2 += xr[t][4 * j + 2]; v3 += xr[t][4 * j + 3];
      xr[t][4 * j] = v0; xr[t][4 * j + 1] = v1; xr[t][4 * j + 2] = v2; xr[t][4 * j + 3] = v3;
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
      const float d = xr[t][r] - mean;
      q += d * d;
    }
  q = qf_sum_x32(q);
  if (hi == 0) red[128 + w * 32 + l31] = q;
  __syncthreads();   // (also: every wave is past its reads of the LDS operand the bf16 copy below overwrites)
  const float var = (red[128 + l31] + red[160 + l31] + red[192 + l31] + red[224 + l31]) * (1.0f / qf::D);
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nb + 32 * t + 8 * j + 4 * hi;
      const float4 g = *reinterpret_cast<const float4*>(lnw + n), b = *reinterpret_cast<const float4*>(lnb + n);
      const float v0 = (xr[t][4 * j] - mean) * rstd * g.x + b.x, v1 = (xr[t][4 * j + 1] - mean) * rstd * g.y + b.y;
      const float v2 = (xr[t][4 * j + 2] - mean) * rstd * g.z + b.z, v3 = (xr[t][4 * j + 3] - mean) * rstd * g.w + b.w;
      xr[t][4 * j] = v0; xr[t][4 * j + 1] = v1; xr[t][4 * j + 2] = v2; xr[t][4 * j + 3] = v3;
      const uint2 pk = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
      *reinterpret_cast<uint2*>(xl + l31 * qf::XSTR + n * 2) = pk;
      if (x_out) *reinterpret_cast<float4*>(x_out + (long long)row * qf::D + n) = make_float4(v0, v1, v2, v3);
      if (xb_out) *reinterpret_cast<uint2*>(xb_out + (long long)row * ldxb + n) = pk;
    }
  __syncthreads();   // the new operand is complete
}

// softmax + dropout of NKT score tiles held in registers (lane = query, register r of tile kt <-> key 32 kt + 8 (r >> 2) + 4 hi + (r & 3)),
// packed probabilities out (bf16, unnormalised, dropped), row sum of the UNdropped probabilities and the running maximum (log2 domain).
template <int NKT>
__device__ __forceinline__ void qf_softmax(f32x16 (&sc)[NKT], bf16x8 (&pp)[NKT][2], float scale2, int nkeys, int hi, uint32_t row_id, const DropoutArg& dr,
                                           float& m_out, float& l_out) {
  float mx = -1.0e30f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * kt + 8 * (r >> 2) + 4 * hi + (r & 3);
      float v = sc[kt][r] * scale2;
      if (32 * kt + 32 > nkeys) v = key < nkeys ? v : -__builtin_inff();
      sc[kt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = qf_max_x32(mx);
  float l = 0.f;
  const uint32_t seed = dr.seed_ptr ? mrb_seed_load(dr.seed_ptr) : 0u;
  const uint32_t skq = (uint32_t)((nkeys + 3) >> 2);
  const uint32_t tl = dr.seed_ptr ? (row_id * skq + (uint32_t)hi) * MRB_H1 + mrb_lin_base(seed, dr.site) : 0u;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(sc[kt][r] - mx);
      l += p;
      sc[kt][r] = p;
    }
    if (dr.seed_ptr) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // keys 32 kt + 8 j + 4 hi + i: quad index 8 kt + 2 j + hi, draw i
        const uint32_t hsh = mrb_lin_fin24(tl + (uint32_t)(8 * kt + 2 * j) * MRB_H1);
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[kt][4 * j + i] = qf_draw(hsh, i) >= dr.thresh24 ? sc[kt][4 * j + i] : 0.f;
      }
    }
    pp[kt][0] = qf_pack8(sc[kt], 0);
    pp[kt][1] = qf_pack8(sc[kt], 8);
  }
  m_out = mx;
  l_out = qf_sum_x32(l);
}

__global__ __launch_bounds__(256, 1) void qformer_layer_fwd_kernel(const QfLayerArgs p) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* xl = sm;                                   // bf16 [32][768] token operand (LayerNorm output / attention output)
  char* hl = sm + qf::XB_BYTES;                    // bf16 [32][512] FFN chunk
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ring = sm + qf::XB_BYTES + qf::HB_BYTES + w * (qf::NSTG * qf::TILE_BYTES);
  float* red = reinterpret_cast<float*>(sm + qf::XB_BYTES + qf::HB_BYTES + qf::W_BYTES);
  const int f = blockIdx.x;
  const int row = f * qf::NQ + l31;                // this lane's token (global row)
  const int nb = 192 * w;                          // this wave's feature slice of the N = 768 products
  const float scale2 = 0.125f * 1.4426950408889634f;

  // ---- the frame's residual stream: fp32 registers in accumulator layout + bf16 operand in LDS
  float xr[6][16];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = nb + 32 * t + 8 * j + 4 * hi;
      const float4 v = *reinterpret_cast<const float4*>(p.x_in + (long long)row * qf::D + n);
      xr[t][4 * j] = v.x; xr[t][4 * j + 1] = v.y; xr[t][4 * j + 2] = v.z; xr[t][4 * j + 3] = v.w;
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
    qf_wave_gemm<6, 0x30>(acc, xl, qf::XSTR, qf::D / 64, p.qkv_w, qf::D, 0, 3u * qf::D * qf::D * 2u,
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
    // S^T[key][query] = K Q^T: four MFMA steps over the 64 head dims, step s = registers 8 (s & 1) .. + 7 of tile s >> 1 on both sides
    f32x16 sc[1];
    qf_zero(sc[0]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
      sc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf_pack8(acc[2 + (s >> 1)], 8 * (s & 1)), qf_pack8(acc[s >> 1], 8 * (s & 1)), sc[0], 0, 0, 0);
    bf16x8 pp[1][2];
    float mrun, ltot;
    qf_softmax<1>(sc, pp, scale2, qf::NQ, hi, (uint32_t)(f * qf::H + h) * qf::NQ + (uint32_t)l31, p.d_sattn, mrun, ltot);
    f32x16 oa[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      qf_zero(oa[t]);
#pragma unroll
      for (int s = 0; s < 2; ++s) oa[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf_pack8(acc[4 + t], 8 * s), pp[0][s], oa[t], 0, 0, 0);
    }
    const float inv = (p.d_sattn.seed_ptr ? p.d_sattn.inv_keep : 1.0f) / ltot;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oa[t][r] *= inv;
      opk[hh][2 * t] = qf_pack8(oa[t], 0);
      opk[hh][2 * t + 1] = qf_pack8(oa[t], 8);
    }
    if (hi == 0) p.lse[(long long)(f * qf::H + h) * qf::NQ + l31] = mrun * 0.6931471805599453f + __logf(ltot);
  }
  __syncthreads();   // every wave is done with the LayerNorm'd operand: the attention output takes its place
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
        *reinterpret_cast<uint2*>(p.o + (long long)row * p.ldo + n) = u.u2[j & 1];
      }
  __syncthreads();
  {
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) qf_zero(acc[t]);
    qf_wave_gemm<6, 0>(acc, xl, qf::XSTR, qf::D / 64, p.so_w, qf::D, 0, (uint32_t)(qf::D * qf::D * 2), [&](int t) { return nb + 32 * t; }, ring, lane);
    qf_block_epilogue(acc, xr, nb, row, hi, l31, w, p.so_b, p.d_so, p.y, p.s_lnw, p.s_lnb, p.eps, red, xl, nullptr, nullptr, 0);
  }

  // ---- cross-attention over the frame's image tokens (every second layer)
  if (p.has_cross) {
    f32x16 qa[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) qf_zero(qa[t]);
    qf_wave_gemm<6, 0>(qa, xl, qf::XSTR, qf::D / 64, p.cq_w, qf::D, 0, (uint32_t)(qf::D * qf::D * 2), [&](int t) { return nb + 32 * t; }, ring, lane);
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
      f32x16 sc[qf::MAXKT];
      const bf16_t* kb = p.kv + (long long)f * p.Tv * (2 * qf::D) + 64 * h;
#pragma unroll
      for (int kt = 0; kt < qf::MAXKT; ++kt) {
        qf_zero(sc[kt]);
        if (kt < nkt) {
          const int key = min(32 * kt + l31, p.Tv - 1);
          const bf16_t* kr = kb + (long long)key * (2 * qf::D);
#pragma unroll
          for (int s = 0; s < 4; ++s) {   // head dims 32 (s >> 1) + 16 (s & 1) + 8 (e >> 2) + 4 hi + (e & 3): the permutation the packed q carries
            union { bf16x8 v8; uint2 u2[2]; } kf;
            kf.u2[0] = *reinterpret_cast<const uint2*>(kr + 16 * s + 4 * hi);
            kf.u2[1] = *reinterpret_cast<const uint2*>(kr + 16 * s + 8 + 4 * hi);
            sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf.v8, qp[hh][s], sc[kt], 0, 0, 0);
          }
        }
      }
      bf16x8 pp[qf::MAXKT][2];
      float mrun, ltot;
      qf_softmax<qf::MAXKT>(sc, pp, scale2, p.Tv, hi, (uint32_t)(f * qf::H + h) * qf::NQ + (uint32_t)l31, p.d_cattn, mrun, ltot);
      f32x16 oa[2];
      qf_zero(oa[0]);
      qf_zero(oa[1]);
      const bf16_t* vb = p.vt + ((long long)(f * qf::H + h) * 64) * p.Tvp;
#pragma unroll
      for (int kt = 0; kt < qf::MAXKT; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const bf16_t* vr = vb + (long long)(32 * t + l31) * p.Tvp + 32 * kt;
#pragma unroll
            for (int s = 0; s < 2; ++s) {   // keys 32 kt + 16 s + 8 (e >> 2) + 4 hi + (e & 3)
              union { bf16x8 v8; uint2 u2[2]; } vf;
              vf.u2[0] = *reinterpret_cast<const uint2*>(vr + 16 * s + 4 * hi);
              vf.u2[1] = *reinterpret_cast<const uint2*>(vr + 16 * s + 8 + 4 * hi);
              oa[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v8, pp[kt][s], oa[t], 0, 0, 0);
            }
          }
        }
      }
      const float inv = (p.d_cattn.seed_ptr ? p.d_cattn.inv_keep : 1.0f) / ltot;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) oa[t][r] *= inv;
        opk[hh][2 * t] = qf_pack8(oa[t], 0);
        opk[hh][2 * t + 1] = qf_pack8(oa[t], 8);
      }
      if (hi == 0) p.lsec[(long long)(f * qf::H + h) * qf::NQ + l31] = mrun * 0.6931471805599453f + __logf(ltot);
    }
    __syncthreads();
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
          *reinterpret_cast<uint2*>(p.oc + (long long)row * p.ldo + n) = u.u2[j & 1];
        }
    __syncthreads();
    f32x16 acc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) qf_zero(acc[t]);
    qf_wave_gemm<6, 0>(acc, xl, qf::XSTR, qf::D / 64, p.co_w, qf::D, 0, (uint32_t)(qf::D * qf::D * 2), [&](int t) { return nb + 32 * t; }, ring, lane);
    qf_block_epilogue(acc, xr, nb, row, hi, l31, w, p.co_b, p.d_co, p.y2, p.c_lnw, p.c_lnb, p.eps, red, xl, nullptr, nullptr, 0);
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
      qf_wave_gemm<4, 0>(ha, xl, qf::XSTR, qf::D / 64, p.i_w, qf::D, 0, (uint32_t)((long long)qf::DI * qf::D * 2), [&](int t) { return hb + 32 * t; }, ring, lane);
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
      qf_wave_gemm<6, 0>(acc, hl, qf::HSTR, qf::HC / 64, p.o_w, qf::DI, (long long)qf::HC * c, (uint32_t)((long long)qf::D * qf::DI * 2),
                         [&](int t) { return nb + 32 * t; }, ring, lane);
    }
    qf_block_epilogue(acc, xr, nb, row, hi, l31, w, p.o_b, p.d_ffn, p.y3, p.o_lnw, p.o_lnb, p.eps, red, xl, p.x_out, p.xb_out, p.ldxb);
  }
}

extern "C" int mrblip_qformer_layer_fwd(const QfLayerArgs* a, hipStream_t stream) {
  MRB_REQUIRE(a != nullptr && a->F > 0, "qformer_layer_fwd: no frames");
  MRB_REQUIRE(a->x_in && a->x_out && a->qkv && a->o && a->lse && a->y && a->hpre && a->y3, "qformer_layer_fwd: missing buffer");
  MRB_REQUIRE(a->ldo >= qf::D && a->ldo % 4 == 0, "qformer_layer_fwd: ldo");
  if (a->has_cross) {
    MRB_REQUIRE(a->qc && a->oc && a->lsec && a->y2 && a->kv && a->vt && a->cq_w && a->co_w, "qformer_layer_fwd: missing cross-attention buffer");
    MRB_REQUIRE(a->Tv >= 1 && a->Tv <= 32 * qf::MAXKT && a->Tvp >= a->Tv && a->Tvp % 32 == 0, "qformer_layer_fwd: Tv %d / Tvp %d", a->Tv, a->Tvp);
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(qformer_layer_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, qf::LDS_BYTES) != hipSuccess) {
      mrblip_set_error("qformer_layer_fwd: cannot reserve %d bytes of LDS", qf::LDS_BYTES);
      return MRBLIP_ELAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(qformer_layer_fwd_kernel, dim3(a->F), dim3(256), qf::LDS_BYTES, stream, *a);
  return mrblip_check_launch("qformer_layer_fwd");
}
